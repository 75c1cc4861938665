// Library-level glue of libtacotron2_amd.so: ABI version, last-error text, struct sizes.
#include "common.h"
#include <string.h>
#include <vector>

static thread_local char g_err[512] = "";

extern "C" void t2amd_set_error_(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* t2amd_last_error(void) { return g_err; }
static int g_validate_only = 0;
extern "C" int t2amd_validate_only_flag_(void) { return g_validate_only; }
extern "C" int t2amd_set_validate_only(int on) { g_validate_only = on ? 1 : 0; return T2AMD_OK; }
extern "C" int t2amd_abi_version(void) { return T2AMD_ABI_VERSION; }

extern "C" int t2amd_struct_sizes(int* out, int max_n) {
    const int sizes[] = {
        (int)sizeof(t2amd_gemm_desc), (int)sizeof(t2amd_seg),       (int)sizeof(t2amd_lstm_step),
        (int)sizeof(t2amd_skinny_gemm), (int)sizeof(t2amd_addend),  (int)sizeof(t2amd_lstm_bwd),
        (int)sizeof(t2amd_attn_fwd),  (int)sizeof(t2amd_attn_bwd),  (int)sizeof(t2amd_dec_train),
        (int)sizeof(t2amd_dec_train_bwd), (int)sizeof(t2amd_lstm_seq), (int)sizeof(t2amd_dec_infer),
        (int)sizeof(t2amd_small_linear), (int)sizeof(t2amd_tensor_list), (int)sizeof(t2amd_adam_hyper),
        (int)sizeof(t2amd_dec_persist), (int)sizeof(t2amd_gemm16_desc),
    };
    const int n = (int)(sizeof(sizes) / sizeof(sizes[0]));
    for (int i = 0; i < n && i < max_n; ++i) out[i] = sizes[i];
    return n;
}

// ---- live kernel timing (bench.py roofline) ---------------------------------------------------
static int g_prof_tag = -1, g_prof_max = 0, g_prof_n = 0;
static std::vector<hipEvent_t> g_prof_ev;
// cost of an EMPTY event bracket on the launch stream (two hipEventRecords back to back), measured once per
// profiling session just before the first real bracket: the part of a bracket that is not the kernel
static hipEvent_t g_cal_ev[8];
static int g_cal_have = 0, g_cal_done = 0;

extern "C" void t2amd_profile_mark_(int tag, int end, hipStream_t s) {
    if (g_prof_tag < 0 || tag != g_prof_tag || g_validate_only) return;
    if (!end) {
        if (g_prof_n >= g_prof_max) return;
        if (!g_cal_done) {
            if (!g_cal_have) {
                for (int i = 0; i < 8; ++i) (void)hipEventCreate(&g_cal_ev[i]);
                g_cal_have = 1;
            }
            for (int i = 0; i < 8; ++i) (void)hipEventRecord(g_cal_ev[i], s);
            g_cal_done = 1;
        }
        (void)hipEventRecord(g_prof_ev[2 * g_prof_n], s);
    } else {
        if (g_prof_n >= g_prof_max) return;
        (void)hipEventRecord(g_prof_ev[2 * g_prof_n + 1], s);
        ++g_prof_n;
    }
}

// Event pair for the next profiled launch of role `tag`, or false when that role is not being profiled.  The
// launch site passes the pair to hipExtLaunchKernelGGL, which stamps them from the dispatch's own start / end
// timestamps (the clock rocprofv3 --kernel-trace reads): no bracket overhead to calibrate away.
extern "C" bool t2amd_profile_pair_(int tag, hipEvent_t* e0, hipEvent_t* e1) {
    if (g_prof_tag < 0 || tag != g_prof_tag || g_validate_only || g_prof_n >= g_prof_max) return false;
    *e0 = g_prof_ev[2 * g_prof_n];
    *e1 = g_prof_ev[2 * g_prof_n + 1];
    ++g_prof_n;
    return true;
}

extern "C" int t2amd_profile_enable(int tag, int max_launches) {
    g_prof_n = 0;
    g_cal_done = 0;
    if (tag < 0 || max_launches <= 0) {
        g_prof_tag = -1;
        return T2AMD_OK;
    }
    while ((int)g_prof_ev.size() < 2 * max_launches) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) T2_FAIL("profile_enable: hipEventCreate failed");
        g_prof_ev.push_back(e);
    }
    g_prof_max = max_launches;
    g_prof_tag = tag;
    return T2AMD_OK;
}

// smallest elapsed time between two back-to-back event records of the calibration burst (ms), 0 if none
extern "C" int t2amd_profile_event_overhead(float* ms) {
    T2_REQUIRE(ms != nullptr, "profile_event_overhead: null arg");
    *ms = 0.f;
    if (!g_cal_done) return T2AMD_OK;
    float best = -1.f;
    if (hipEventSynchronize(g_cal_ev[7]) != hipSuccess) T2_FAIL("profile_event_overhead: event sync failed");
    for (int i = 0; i + 1 < 8; ++i) {
        float e = 0.f;
        if (hipEventElapsedTime(&e, g_cal_ev[i], g_cal_ev[i + 1]) == hipSuccess && (best < 0.f || e < best)) best = e;
    }
    *ms = best < 0.f ? 0.f : best;
    return T2AMD_OK;
}

extern "C" int t2amd_profile_read(float* total_ms, int* count) {
    T2_REQUIRE(total_ms && count, "profile_read: null args");
    float tot = 0.f;
    for (int i = 0; i < g_prof_n; ++i) {
        if (hipEventSynchronize(g_prof_ev[2 * i + 1]) != hipSuccess) T2_FAIL("profile_read: event sync failed");
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof_ev[2 * i], g_prof_ev[2 * i + 1]) != hipSuccess)
            T2_FAIL("profile_read: elapsed time failed");
        tot += ms;
    }
    *total_ms = tot;
    *count = g_prof_n;
    g_prof_tag = -1;
    return T2AMD_OK;
}

// tools only: n dependent launches of a trivial kernel on `stream` (per-launch floor measurement)
__global__ void t2_nop_kernel(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0 && p) p[0] += 1.0f; }
extern "C" int t2amd_debug_launch_chain_(float* p, int n, int blocks, void* stream) {
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(t2_nop_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    return 0;
}

// tests only (tests/test_zz9_dp_gpu.py: co-residency stress): `ncus` workgroups that each take a whole CU's LDS
// (160 KB dynamic: nothing else that needs LDS can be placed beside them) and sleep until `*stop` becomes non-zero or
// `ms` milliseconds of the 100 MHz wall clock have passed -- the stand-in for a co-resident RCCL kernel that takes CUs
// away from the engine's hand-off kernels.  `arrived` counts the workgroups that have started.
__global__ void t2_hold_cu_kernel(unsigned long long ticks, const int* stop, int* arrived) {
    extern __shared__ char hold_smem[];
    if (threadIdx.x == 0) {
        hold_smem[0] = 1;
        atomicAdd(arrived, 1);
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < ticks) {
            if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            __builtin_amdgcn_s_sleep(64);
        }
    }
}
extern "C" int t2amd_debug_hold_cus_(int ncus, float ms, const int* stop, int* arrived, void* stream) {
    T2_REQUIRE(ncus > 0 && ncus <= 256 && ms > 0.f && ms <= 5000.f && stop && arrived, "debug_hold_cus: bad args");
    const int lds = 160 * 1024;
    if (hipFuncSetAttribute((const void*)t2_hold_cu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        T2_FAIL("debug_hold_cus: cannot raise the LDS limit");
    hipLaunchKernelGGL(t2_hold_cu_kernel, dim3(ncus), dim3(64), lds, (hipStream_t)stream,
                       (unsigned long long)(ms * 1e5f), stop, arrived);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// tools/microbench_launch.py: the same chain captured once into a hipGraph and replayed `reps` times; returns the
// average milliseconds per replay (HIP events on `stream`), or a negative HIP error code.
extern "C" float t2amd_debug_graph_chain_(float* p, int n, int blocks, int reps, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) return -(float)e;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(t2_nop_kernel, dim3(blocks < 0 ? -blocks : blocks), dim3(256), 0, s, p);
    e = hipStreamEndCapture(s, &g);
    if (e != hipSuccess) return -(float)e;
    e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    if (e != hipSuccess) return -(float)e;
    // replay on the LEGACY DEFAULT stream when `blocks` is negative (capture is not allowed there, launching is)
    hipStream_t ls = blocks < 0 ? nullptr : s;
    e = hipGraphLaunch(ge, ls);
    if (e != hipSuccess) return -(float)e;
    (void)hipStreamSynchronize(ls);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, ls);
    for (int r = 0; r < reps; ++r) (void)hipGraphLaunch(ge, ls);
    (void)hipEventRecord(e1, ls);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    return ms / reps;
}

// tools only: capture whatever the caller launches on `stream` between begin and end into a hipGraph, then replay it
// `reps` times and return the average milliseconds per replay (negative HIP error code on failure).
extern "C" int t2amd_debug_capture_begin_(void* stream) {
    return (int)hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed);
}
extern "C" float t2amd_debug_capture_end_(void* stream, int reps) {
    hipStream_t s = (hipStream_t)stream;
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipStreamEndCapture(s, &g);
    if (e != hipSuccess) return -(float)e;
    e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    if (e != hipSuccess) return -(float)e;
    (void)hipGraphLaunch(ge, s);
    (void)hipStreamSynchronize(s);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) (void)hipGraphLaunch(ge, s);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    return ms / reps;
}

// tools only: device buffer of 128 wall-clock stamps (T2AMD_ATTN_TS=1), written by thread 0 of workgroup 0 of the
// instrumented kernels at their phase boundaries (attention: slots 0-63, wide LSTM step kernel: 64-79, its plain
// variant: 80-95); nullptr when the switch is off, so the kernels skip the stamps.
static unsigned long long* g_dbg_ts = nullptr;
extern "C" unsigned long long* t2amd_debug_ts_() {
    static int init = 0;
    if (!init) {
        init = 1;
        const char* e = getenv("T2AMD_ATTN_TS");
        if (e && e[0] == '1' && hipMalloc((void**)&g_dbg_ts, 128 * sizeof(unsigned long long)) != hipSuccess) g_dbg_ts = nullptr;
        if (g_dbg_ts) {
            (void)hipMemset(g_dbg_ts, 0, 128 * sizeof(unsigned long long));
            const char* pk = getenv("T2AMD_ATTN_TS_PICK");
            const unsigned long long pick = pk ? strtoull(pk, nullptr, 10) : 1200ull;
            (void)hipMemcpy(g_dbg_ts + 127, &pick, sizeof(pick), hipMemcpyHostToDevice);
        }
    }
    return g_dbg_ts;
}
extern "C" int t2amd_debug_attn_ts_(unsigned long long* out128) {
    if (!g_dbg_ts) return -1;
    return (int)hipMemcpy(out128, g_dbg_ts, 128 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}


// tools/microbench_barrier.py: cost of a device-scope barrier across `blocks` co-resident workgroups — the number that
// decides whether a weight-stationary persistent decode step ("next") pays.  One counter per round
// (no reset race); thread 0 of every workgroup arrives with a release add and polls with acquire loads at agent
// scope.  Every spin is BOUNDED: past 2^22 polls the workgroup raises status[0] and leaves, so a mistake cannot hang
// the GPU.  `lds_bytes` of dynamic LDS pin the residency (e.g. 145000 -> one workgroup per CU).  clk[0] = wall-clock
// ticks (100 MHz) workgroup 0 spent in `rounds` barriers.
__global__ void __launch_bounds__(256) t2_grid_barrier_kernel(unsigned* counters, int rounds, unsigned long long* clk,
                                                              int* status) {
    extern __shared__ float t2_barrier_lds[];
    if (threadIdx.x == 0) t2_barrier_lds[0] = 0.0f;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    bool bad = false;
    for (int r = 0; r < rounds && !bad; ++r) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&counters[r], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(&counters[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
                if (++spins > (1 << 22)) {
                    atomicExch(status, 1);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            t2_barrier_lds[0] = spins > (1 << 22) ? 1.0f : 0.0f;
        }
        __syncthreads();
        bad = t2_barrier_lds[0] != 0.0f;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = wall_clock64() - t0;
}

// counters: >= rounds zeroed uint32; clk: 1 uint64; status: 1 zeroed int32 (all device memory owned by the caller).
extern "C" int t2amd_debug_grid_barrier_(unsigned* counters, int rounds, int blocks, int lds_bytes,
                                         unsigned long long* clk, int* status, void* stream) {
    if (!counters || !clk || !status || rounds < 1 || blocks < 1 || blocks > 2048 || lds_bytes < 4) return -1;
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute((const void*)t2_grid_barrier_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            lds_bytes) != hipSuccess)
        return -2;
    hipLaunchKernelGGL(t2_grid_barrier_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, counters, rounds,
                       clk, status);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
