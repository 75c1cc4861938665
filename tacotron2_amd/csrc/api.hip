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
// SHA-1 of every kernel source + the header this binary was built from (tacotron2_amd/build.py passes it in); "" for a
// build made by hand without it.  native.load() refuses a library whose hash is not that of the sources beside it.
#ifndef T2AMD_SOURCE_SHA1
#define T2AMD_SOURCE_SHA1 ""
#endif
extern "C" const char* t2amd_source_sha1(void) { return T2AMD_SOURCE_SHA1; }

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

// tools/microbench_edge.py: what ONE all-to-all edge of a decoder time step costs when it is kept inside a persistent
// launch instead of being a kernel boundary (MI355X_MICROARCH.md price list rows barrier-xcd,
// publish-large, handoff-payload).  Geometry of the training chain: `gridDim.x` co-resident 512-thread workgroups (one per
// CU: `lds_bytes` pins that), every workgroup PUBLISHES `pub_bytes` of a shared buffer (its slice of h / ctx / gate
// gradients: plain 16-byte stores, every wave drains them, lane 0 release-fences), arrives at an XCD-hierarchical barrier
// (per-XCC arrival counter -> top counter -> per-XCC generation word, the guide's barrier-xcd), acquire-fences and
// CONSUMES `con_bytes` of the buffer (what the next phase reads: the whole 128 KB bf16 h for an LSTM tile, 4 KB for an
// attention workgroup) with 16-byte loads, `work_ns` of sleep standing in for the phase's own work.  All spins are
// bounded (status != 0: a workgroup gave up).  clk[0] = wall-clock ticks (100 MHz) of workgroup 0 over `rounds` edges.
struct EdgeParams {
    float4* buf; long long buf_f4; unsigned* xcnt; unsigned* top; unsigned* gen; unsigned* census; int rounds;
    int pub_f4, con_f4, work_sleeps; unsigned long long* clk; int* status; float* sink;
};
__global__ void __launch_bounds__(512) t2_edge_kernel(EdgeParams p) {
    extern __shared__ float t2_edge_lds[];
    __shared__ int s_flag;
    const int tid = threadIdx.x;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    // census: how many workgroups live on each XCC (placement is observed, never assumed), then one flat barrier
    if (tid == 0) {
        atomicAdd(&p.census[xcc], 1u);
        __hip_atomic_fetch_add(&p.census[8], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(&p.census[8], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            if (++spins > (1 << 22)) { atomicExch(p.status, 1); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        s_flag = spins > (1 << 22);
    }
    __syncthreads();
    if (s_flag) return;
    const unsigned mine = __hip_atomic_load(&p.census[xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned nx = 0;
    for (int i = 0; i < 8; ++i) nx += __hip_atomic_load(&p.census[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
    float acc = 0.f;
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < p.rounds; ++r) {
        // the phase's own work
        for (int i = 0; i < p.work_sleeps; ++i) __builtin_amdgcn_s_sleep(16);
        // publish this workgroup's slice
        float4* mineb = p.buf + ((long long)blockIdx.x * p.pub_f4) % p.buf_f4;
        for (int i = tid; i < p.pub_f4; i += 512) mineb[i] = make_float4((float)r, acc, 1.f, 2.f);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned target = (unsigned)(r + 1);
            const unsigned got = __hip_atomic_fetch_add(&p.xcnt[xcc * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            if (got + 1 == mine * target) {              // last arriver of this XCC: cross-XCC stage
                __hip_atomic_fetch_add(p.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(p.top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nx * target) {
                    if (++spins > (1 << 22)) { atomicExch(p.status, 2); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                __hip_atomic_store(&p.gen[xcc * 32], target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_load(&p.gen[xcc * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (++spins > (1 << 22)) { atomicExch(p.status, 3); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            s_flag = spins > (1 << 22);
        }
        __syncthreads();
        if (s_flag) return;
        // consume
        const float4* src = p.buf + ((long long)blockIdx.x * 64) % p.buf_f4;
        for (int i = tid; i < p.con_f4; i += 512) {
            const float4 v = src[(i) % p.buf_f4];
            acc += v.x + v.w;
        }
        if (tid < 64) t2_edge_lds[tid] = acc;
    }
    if (tid == 0 && blockIdx.x == 0) p.clk[0] = wall_clock64() - t0;
    if (acc == 123.456f) p.sink[0] = acc;
}
// counters: 8*32+8*32+1+9 zeroed uint32 laid out as [xcnt 256][gen 256][top 1][pad 7][census 9]; buf: >= buf_bytes device
// bytes; clk 1 uint64; status 1 zeroed int32; sink 1 float
extern "C" int t2amd_debug_edge_(void* buf, long long buf_bytes, unsigned* counters, int rounds, int blocks, int lds_bytes,
                                 int pub_bytes, int con_bytes, int work_sleeps, unsigned long long* clk, int* status,
                                 float* sink, void* stream) {
    if (!buf || !counters || !clk || !status || !sink || rounds < 1 || blocks < 1 || blocks > 512 || buf_bytes < 4096) return -1;
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute((const void*)t2_edge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
        return -2;
    EdgeParams p;
    p.buf = (float4*)buf; p.buf_f4 = buf_bytes / 16; p.xcnt = counters; p.gen = counters + 256; p.top = counters + 512;
    p.census = counters + 520; p.rounds = rounds; p.pub_f4 = pub_bytes / 16; p.con_f4 = con_bytes / 16;
    p.work_sleeps = work_sleeps; p.clk = clk; p.status = status; p.sink = sink;
    hipLaunchKernelGGL(t2_edge_kernel, dim3(blocks), dim3(512), lds_bytes < 256 ? 256 : lds_bytes, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---- tools/microbench_edge_flagdata.py (round 4): the two all-to-all edges of a FORWARD decoder time step as
// FLAG + DATA hand-offs (Guideline 16 R1: 16-byte write-through payload stores, every storing wave drains, ONE relaxed flag store
// per workgroup; consumers poll the 256 flags with one wave, then read the payload with sc1 loads straight from L2) -- not a
// barrier (tools/microbench_edge.py measured that: 6.1-8.6 us) and not granules.  256 co-resident 512-thread workgroups
// alternate two roles per round, in the chain's own geometry:
//   role L ("LSTM pair", workgroup j of 256): waits for the 256 T flags of the previous round, reads `con_l` bytes of what the
//          attention step left (ctx / h: the activation matrix every LSTM workgroup streams), publishes its 64 rows x 16 B
//          h-slice (4 hidden units, f32) and ONE flag;
//   role T ("attention", workgroup (b, s) = (j / 4, j % 4)): waits for the 256 L flags, reads its utterance's whole h row
//          (4 KB = 256 x 16 B from 256 different producers), publishes its 128-channel context slice (512 B) and ONE flag.
// Every payload word carries the round number: a consumer that reads a stale word counts it (`stale`), so the number is for a
// hand-off that is also CORRECT.  mode 0: both roles in one persistent launch; mode 1 / 2: one role's body of one round as a
// launch of its own (no flags: the kernel boundary orders them) -- the host alternates them for the launch-chain comparison.
struct EdgeFDParams {
    f32x4* h;            // [64][256] float4: row b, producer j
    f32x4* ctx;          // [64][4][32] float4: utterance b, slice s
    unsigned* flagL;      // [256]
    unsigned* flagT;      // [256]
    int rounds, mode, round0, con_l_f4, delay, work_l, work_t;
    int hier;             // 1: two-level wait -- workgroups 0..7 poll the 256 flags (wave 1) and republish one word each (go[8], own 128-byte
                          // lines behind the flags); every workgroup's wave 0 polls go[j & 7] only
    unsigned long long* clk;   // [4]: total ticks, ticks spent waiting in role T, in role L
    int* status;
    unsigned* stale;
};
typedef unsigned u32x4_efd __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ void efd_store_sc1(f32x4* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
static __device__ __forceinline__ f32x4 efd_load_sc1(const f32x4* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
static __device__ __forceinline__ bool efd_wait(const unsigned* flags, unsigned target, int delay, int* status, int lane) {
    // one wave polls 256 flags: 16 bytes (4 flags) per lane, relaxed agent-scope loads, after a short pause
    for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(1);
    const unsigned long long t0 = wall_clock64();
    unsigned spins = 0;
    for (;;) {
        u32x4_efd f;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(f) : "v"(flags + 4 * lane) : "memory");
        const bool ok = f.x >= target && f.y >= target && f.z >= target && f.w >= target;
        if (__all(ok)) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 63u) == 0 && wall_clock64() - t0 > 3000000ull) { if (lane == 0) atomicExch(status, 1); return false; }
    }
}
// two-level wait: `go` = 8 words on 8 separate 128-byte lines.  Collector workgroups (blockIdx < 8): wave 1 sweeps the 256 flags and
// stores go[blockIdx] = target; every workgroup: wave 0 polls go[blockIdx & 7].
static __device__ __forceinline__ bool efd_wait_hier(const unsigned* flags, unsigned* go, unsigned target, int delay, int* status, int lane,
                                                     int wave, int j) {
    if (wave == 1 && j < 8) {
        if (!efd_wait(flags, target, delay, status, lane)) return false;
        if (lane == 0) __hip_atomic_store(go + 32 * j, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    if (wave != 0) return true;
    const unsigned long long t0 = wall_clock64();
    unsigned spins = 0;
    for (;;) {
        const unsigned v = __hip_atomic_load(go + 32 * (j & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= target) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 63u) == 0 && wall_clock64() - t0 > 3000000ull) { if (lane == 0) atomicExch(status, 2); return false; }
    }
}
__global__ __launch_bounds__(512) void t2_edge_flagdata_kernel(EdgeFDParams p) {
    __shared__ int s_fail;
    __shared__ unsigned long long s_wait[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = blockIdx.x, b = j >> 2, sl = j & 3;
    if (tid == 0) { s_fail = 0; s_wait[0] = s_wait[1] = 0; }
    __syncthreads();
    float acc = 0.f;
    unsigned stale = 0;
    const unsigned long long t_begin = wall_clock64();
    const int r_lo = p.mode == 0 ? 0 : p.round0, r_hi = p.mode == 0 ? p.rounds : p.round0 + 1;
    for (int r = r_lo; r < r_hi; ++r) {
        const float tag = (float)(r + 1);
        if (p.mode != 2) {
            // ---------------- role L ----------------
            if (r > 0) {
                if (p.mode == 0) {
                    if (p.hier) {
                        if (!efd_wait_hier(p.flagT, p.flagT + 512, (unsigned)r, p.delay, p.status, lane, wave, j) && lane == 0) s_fail = 1;
                    } else if (wave == 0) {
                        const unsigned long long w0 = wall_clock64();
                        if (!efd_wait(p.flagT, (unsigned)r, p.delay, p.status, lane) && lane == 0) s_fail = 1;
                        if (lane == 0) s_wait[1] += wall_clock64() - w0;
                    }
                    __syncthreads();
                    if (s_fail) return;
                }
                // what the LSTM workgroup streams: ctx of every utterance (and as much more as con_l asks for, wrapping)
                for (int i = tid; i < p.con_l_f4; i += 512) {
                    const f32x4 v = p.mode == 0 ? efd_load_sc1(p.ctx + (i & 8191)) : p.ctx[i & 8191];
                    if (v.x != (float)r) ++stale;
                    acc += v.y;
                }
            }
            for (int i = 0; i < p.work_l; ++i) __builtin_amdgcn_s_sleep(16);
            if (tid < 64) {
                const f32x4 v = f32x4{tag, acc, (float)j, 1.f};
                if (p.mode == 0) efd_store_sc1(p.h + tid * 256 + j, v); else p.h[tid * 256 + j] = v;
            }
            if (p.mode == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains (only wave 0 stored)
                __syncthreads();
                if (tid == 0) __hip_atomic_store(p.flagL + j, (unsigned)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (p.mode != 1) {
            // ---------------- role T ----------------
            if (p.mode == 0) {
                if (p.hier) {
                    if (!efd_wait_hier(p.flagL, p.flagL + 1024, (unsigned)(r + 1), p.delay, p.status, lane, wave, j) && lane == 0) s_fail = 1;
                } else if (wave == 0) {
                    const unsigned long long w0 = wall_clock64();
                    if (!efd_wait(p.flagL, (unsigned)(r + 1), p.delay, p.status, lane) && lane == 0) s_fail = 1;
                    if (lane == 0) s_wait[0] += wall_clock64() - w0;
                }
                __syncthreads();
                if (s_fail) return;
            }
            if (tid < 256) {                                              // the utterance's h: 256 x 16 B from 256 producers
                const f32x4 v = p.mode == 0 ? efd_load_sc1(p.h + b * 256 + tid) : p.h[b * 256 + tid];
                if (v.x != tag) ++stale;
                acc += v.z;
            }
            for (int i = 0; i < p.work_t; ++i) __builtin_amdgcn_s_sleep(16);
            if (tid < 32) {
                const f32x4 v = f32x4{tag, acc, 2.f, 3.f};
                if (p.mode == 0) efd_store_sc1(p.ctx + (b * 4 + sl) * 32 + tid, v); else p.ctx[(b * 4 + sl) * 32 + tid] = v;
            }
            if (p.mode == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(p.flagT + j, (unsigned)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (stale) atomicAdd(p.stale, stale);
    if (tid == 0 && blockIdx.x == 0 && p.mode == 0) {
        p.clk[0] = wall_clock64() - t_begin;
        p.clk[1] = s_wait[0];
        p.clk[2] = s_wait[1];
    }
    if (acc == 123.456f) p.status[0] = 7;
}
// h: 64*256 float4 (256 KB), ctx: 8192 float4 (128 KB), flags: 1280 zeroed uint32 (256 + 256 step counters, then the 2 x 8 "go"
// words of the two-level wait on their own 128-byte lines), clk: 4 uint64, status / stale: zeroed.
// mode 0: one persistent launch of `rounds` rounds (mode 4: the same with the two-level wait); mode 3: the launch chain --
// `rounds` x (L launch, T launch).
extern "C" int t2amd_debug_edge_flagdata_(void* h, void* ctx, unsigned* flags, int rounds, int mode, int con_l_bytes, int delay,
                                          int work_l, int work_t, int lds_bytes, unsigned long long* clk, int* status,
                                          unsigned* stale, void* stream) {
    if (!h || !ctx || !flags || !clk || !status || !stale || rounds < 1) return -1;
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute((const void*)t2_edge_flagdata_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
        return -2;
    EdgeFDParams p;
    p.h = (f32x4*)h; p.ctx = (f32x4*)ctx; p.flagL = flags; p.flagT = flags + 256; p.rounds = rounds; p.round0 = 0;
    p.con_l_f4 = con_l_bytes / 16; p.delay = delay; p.work_l = work_l; p.work_t = work_t; p.clk = clk; p.status = status; p.stale = stale;
    const int lds = lds_bytes < 256 ? 256 : lds_bytes;
    hipStream_t s = (hipStream_t)stream;
    p.hier = mode == 4 ? 1 : 0;
    if (mode == 0 || mode == 4) {
        p.mode = 0;
        hipLaunchKernelGGL(t2_edge_flagdata_kernel, dim3(256), dim3(512), lds, s, p);
    } else {
        for (int r = 0; r < rounds; ++r) {
            p.round0 = r;
            p.mode = 1;
            hipLaunchKernelGGL(t2_edge_flagdata_kernel, dim3(256), dim3(512), lds, s, p);
            p.mode = 2;
            hipLaunchKernelGGL(t2_edge_flagdata_kernel, dim3(256), dim3(512), lds, s, p);
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
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

// A HIP stream whose kernels run on CUs [first, first + count) only (hipExtStreamCreateWithCUMask): lets a latency-bound
// chain of small launches and a throughput product run side by side without competing for the same CUs.
extern "C" void* t2amd_debug_stream_cu_range_(int first, int count) {
    if (first < 0 || count <= 0 || first + count > 1024) return nullptr;
    uint32_t mask[32] = {0};
    for (int i = first; i < first + count; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, 32, mask) != hipSuccess) return nullptr;
    return (void*)s;
}
extern "C" int t2amd_debug_stream_destroy_(void* stream) {
    return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? T2AMD_OK : T2AMD_ERR_LAUNCH;
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
