// Shared device/host helpers for the MI355X (gfx950 / CDNA4) Tacotron 2 mel engine.
// Wave = 64 lanes everywhere; MFMA f32 forms are exact-f32 (k-ordered fmaf chain).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/tacotron2_amd.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define T2_WAVE 64

// Last error text, readable through t2amd_last_error().
extern "C" void t2amd_set_error_(const char* msg);

#define T2_FAIL(msg)                                                                   \
    do {                                                                               \
        t2amd_set_error_(msg);                                                         \
        return T2AMD_ERR_ARG;                                                          \
    } while (0)

#define T2_REQUIRE(cond, msg)                                                          \
    do {                                                                               \
        if (!(cond)) T2_FAIL(msg);                                                     \
    } while (0)

// Validate-only mode (t2amd_set_validate_only): every host-side check and loop runs, no kernel is
// launched.  Used by the CPU test-suite to exercise argument plumbing without a GPU; outputs are
// left untouched, so it is NOT a compute path.
extern "C" int t2amd_validate_only_flag_(void);

extern "C" void t2amd_profile_mark_(int tag, int end, hipStream_t s);

#define T2_LAUNCH(kern, grid, block, lds, stream, ...)                                 \
    do {                                                                               \
        if (!t2amd_validate_only_flag_())                                              \
            hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);           \
    } while (0)

// The same launch carrying an event pair when bench.py's roofline leg profiles role `role` (t2amd_profile_enable): the
// pair is stamped by the dispatch itself (hipExtLaunchKernelGGL start/stop events = the kernel begin/end timestamps
// rocprofv3 --kernel-trace reports), so there is no bracket overhead to calibrate away.  Roles: 3 fused LSTM pair of a
// decoder time step, 4 attention backward (+ folded cells), 5 attention forward (one-launch form), 6 BPTT dgrad pair,
// 7 the persistent forward loop (all time steps' LSTM pairs and attention steps in ONE launch).
extern "C" bool t2amd_profile_pair_(int tag, hipEvent_t* e0, hipEvent_t* e1);
#define T2_LAUNCH_ROLE(role, kern, grid, block, lds, stream, ...)                      \
    do {                                                                               \
        if (t2amd_validate_only_flag_()) break;                                        \
        hipEvent_t pe0_ = nullptr, pe1_ = nullptr;                                     \
        if (t2amd_profile_pair_(role, &pe0_, &pe1_))                                   \
            hipExtLaunchKernelGGL(kern, grid, block, lds, stream, pe0_, pe1_, 0, __VA_ARGS__); \
        else                                                                           \
            hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);           \
    } while (0)

#define T2_LAUNCH_CHECK()                                                              \
    do {                                                                               \
        if (t2amd_validate_only_flag_()) break;                                        \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) {                                                        \
            t2amd_set_error_(hipGetErrorString(e_));                                   \
            return T2AMD_ERR_LAUNCH;                                                   \
        }                                                                              \
    } while (0)

#define T2_PROPAGATE(call)                                                             \
    do {                                                                               \
        int rc_ = (call);                                                              \
        if (rc_ != T2AMD_OK) return rc_;                                               \
    } while (0)

// f32 -> bf16 bits, round to nearest even
__device__ __forceinline__ unsigned short t2_f32_to_bf16(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// two f32 -> packed bf16 pair (a in the low half), round to nearest even, one VALU instruction
__device__ __forceinline__ unsigned t2_cvt_pk_bf16(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---- split-bf16 operand images (round 6: the 'bf16x3' mode of the LSTM tiles, include/tacotron2_amd.h t2amd_split_bf16x3_f32) ----
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (x - hi is exact in f32): 16 mantissa bits, |x - hi - lo| <= 2^-17 |x|.
__device__ __forceinline__ void t2_split_bf16(float x, unsigned short& hi, unsigned short& lo) {
    hi = t2_f32_to_bf16(x);
    lo = t2_f32_to_bf16(x - __uint_as_float((unsigned)hi << 16));
}
// position (in bf16 units) of k's hi value inside a row of the split image: groups of 16 k = 16 hi then 16 lo; lo sits 16 further
__device__ __forceinline__ long long t2_x3_pos(long long k) { return ((k >> 4) << 5) + (k & 15); }

__device__ __forceinline__ float t2_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
// sigmoid on v_exp_f32 / v_rcp_f32 (|err| ~1e-7): the bf16-mode kernels' cell; the parity-mode kernels keep libm's
__device__ __forceinline__ float t2_sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
// tanh without divergent paths (libm's tanhf is two exec-masked branches, both taken by a mixed wave).
// |x| < 0.625: x + x*x2*P(x2), the minimax polynomial libm uses on that range; otherwise 1 - 2/(exp(2|x|) + 1) on
// v_exp_f32 / v_rcp_f32 (absolute error < 5e-7).  Both are evaluated and one is selected.
__device__ __forceinline__ float t2_tanh(float x) {
    const float ax = fabsf(x);
    const float x2 = ax * ax;
    float p = fmaf(x2, __uint_as_float(0xbbbac73du), __uint_as_float(0x3ca908c9u));
    p = fmaf(x2, p, __uint_as_float(0xbd5c1c4eu));
    p = fmaf(x2, p, __uint_as_float(0x3e088382u));
    p = fmaf(x2, p, __uint_as_float(0xbeaaaa99u));
    const float small = fmaf(x2, ax * p, ax);
    const float e = __builtin_amdgcn_exp2f(ax * 2.8853900817779268f);      // exp(2|x|); inf for large |x| -> big = 1
    const float big = fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
    return copysignf(ax < 0.625f ? small : big, x);
}

// One-range form for the bf16 compute mode of the TRAINING loops (round 5): tanh(x) = 1 - 2 / (exp(2x) + 1) straight on v_exp_f32 /
// v_rcp_f32 -- 5 VALU operations instead of ~17 (no |x|, no polynomial, no select, no copysign); exp -> inf gives 1, exp -> 0 gives -1.
// Absolute error < 3e-7 like the two-range form's large branch, but NOT relatively accurate near 0 (|x| < 1e-3: the subtraction loses
// it): fine beside bf16 products (the attention energies sum v * tanh), not for the fp32 parity mode, which keeps t2_tanh.  The
// attention backward recomputes the forward's tanh with the same selection, so the pair stays consistent.  -DT2AMD_TANH_EXACT: off.
__device__ __forceinline__ float t2_tanh_1r(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}
template <bool FAST>
__device__ __forceinline__ float t2_tanh_sel(float x) {
#ifdef T2AMD_TANH_EXACT
    return t2_tanh(x);
#else
    if constexpr (FAST) return t2_tanh_1r(x);
    else return t2_tanh(x);
#endif
}

// DPP lane exchanges (VALU, no LDS crossbar: a __shfl_xor is a ds_bpermute the compiler waits for one at a time).
// The four steps pair exactly the lanes the xor-1/2/4/8 butterfly pairs, so sums are bitwise the same.
template <int CTRL>
__device__ __forceinline__ float t2_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over each aligned group of 16 lanes, result in every lane of the group
__device__ __forceinline__ float row16_sum(float v) {
    v += t2_dpp<0xB1>(v);      // quad_perm [1,0,3,2]  (lane ^ 1)
    v += t2_dpp<0x4E>(v);      // quad_perm [2,3,0,1]  (lane ^ 2)
    v += t2_dpp<0x141>(v);     // row_half_mirror      (the other quad of the 8-lane half: same partner sums as lane ^ 4)
    v += t2_dpp<0x140>(v);     // row_mirror           (the other half of the row: as lane ^ 8)
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, t2_dpp<0xB1>(v));
    v = fmaxf(v, t2_dpp<0x4E>(v));
    v = fmaxf(v, t2_dpp<0x141>(v));
    v = fmaxf(v, t2_dpp<0x140>(v));
    return v;
}
__device__ __forceinline__ float wave_reduce_sum(float v) {
    v = row16_sum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
    v = row16_max(v);
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}


// tools only: phase stamps of ONE mid-loop launch per instrumented kernel.  Compiled in only with
// -DT2AMD_PHASE_STAMPS (python -m tacotron2_amd.build --stamps -> lib/libtacotron2_amd_stamps.so, selected with
// T2AMD_LIB) -- even a never-taken stamp branch in the chain kernels costs ~2 % of a training step.  With
// T2AMD_ATTN_TS=1, thread 0 of workgroup 0 counts the launches of its kernel in slot base+15 and stamps only the launch
// whose number is in slot 127 (T2AMD_ATTN_TS_PICK, default 1200 = second training step, mid sequence), so that
// tools/phase_stamps_step.py sees a warm, regular step rather than the last (special) one.
__device__ __forceinline__ bool t2_ts_begin(unsigned long long* ts, int base) {
    if (!ts || blockIdx.x != 0 || blockIdx.y != 0 || threadIdx.x != 0) return false;
    const unsigned long long n = atomicAdd(&ts[base + 15], 1ull);
    if (n != ts[127]) return false;
    ts[base] = wall_clock64();
    return true;
}
__device__ __forceinline__ void t2_ts_mark(bool on, unsigned long long* ts, int slot) {
    if (on) ts[slot] = wall_clock64();
}

// internal entry points shared between translation units (not part of the C ABI)
extern "C" unsigned long long* t2amd_debug_ts_();
int t2amd_proj_finish_small_(const t2amd_small_linear* a, int* out_lengths, uint8_t* active, int* done_count, int t,
                             int max_steps, float thr, int gate_row, void* stream);

int t2amd_check_lstm_bwd_(const t2amd_lstm_bwd* a);      // rnn.hip: argument checks of one cell-backward descriptor

static inline bool t2_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int t2_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
