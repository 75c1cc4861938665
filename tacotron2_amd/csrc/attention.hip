// Location-sensitive attention step kernels for gfx950 — four workgroups per utterance.
//
// Reference arithmetic (model.py:43-86, 22-26, 358-365):
//   q      = W_q h_att                                   (128)
//   loc    = Dense(Conv1d([w_prev ; w_cum]))             (Ti x 128)   2->32 ch, k=31, then 32->128
//   e[ti]  = v . tanh(q + loc[ti] + processed_memory[ti])
//   w      = softmax(mask(e));  ctx = w @ memory;  w_cum += w
// The conv and the dense layer are folded once per optimiser step into one 62-tap filter per
// attention dim, U[d][c*31+k] = sum_f Wd[d][f] Wc[f][c][k] (t2amd_fold_location_f32).
//
// Decomposition (B = 64 utterances must fill 256 CUs, and nothing may need a grid-wide barrier):
//   forward   K_e  grid (4 dim-slices, B): q for its 32 dims (32 rows of W_q), loc^T = U . im2col^T on the
//                  exact-f32 MFMA (16x16x4; U fragments live in VGPRs, the Toeplitz operand is read
//                  straight from the two LDS windows), tanh, v-weighted PARTIAL energy over its dims.
//             K_c  grid (4 channel-slices, B): sums the 4 partial energies (fixed order), softmax, the
//                  slice's 128 context channels; slice 0 also writes the weights and updates w_cum.
//   backward  K_b1 grid (4 position-slices, B): dctx, dw = dctx . memory + carries, partial sum(w dw).
//             K_b2 grid (4 dim-slices, B): recompute loc/tanh, dpre, dv, dq, d_pm, dU (MFMA), the
//                  location-input gradient dcol^T = U^T dpre (MFMA) + col2im as PARTIAL carries,
//                  and the slice's part of dh = W_q^T dq.
// Every cross-workgroup reduction is a small slab of partials summed in a fixed order by the next
// kernel: results are bit-reproducible run to run.
#include "common.h"
#include "cell_bwd.h"
#include "skinny_wide.h"
#include <vector>
#include <cstring>
#include <stdlib.h>
#include <stddef.h>

// Abandoned in-launch hand-offs (bounded spins that ran out: a workgroup of an utterance never became resident within
// 50 ms).  The kernels poison their outputs with NaN so that nothing can use a half-exchanged step; this counter says WHY
// a step went non-finite: the host reads it (t2amd_attn_handoff_timeouts) when the gradient norm is not finite and
// switches to the separate-launch forms, which make no co-residency assumption.  Cold path only.
__device__ unsigned int t2_attn_timeouts = 0u;
__device__ __forceinline__ void t2_attn_gave_up() { atomicAdd(&t2_attn_timeouts, 1u); }
extern "C" int t2amd_attn_handoff_timeouts(int reset) {
    if (t2amd_validate_only_flag_()) return 0;
    unsigned int v = 0u;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(t2_attn_timeouts), sizeof(v)) != hipSuccess) return -1;
    if (reset && v != 0u) {
        const unsigned int z = 0u;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(t2_attn_timeouts), &z, sizeof(z));
    }
    return (int)(v > 0x7fffffffu ? 0x7fffffffu : v);
}
// tests only: what a timed-out hand-off does to the counter
__global__ void t2_attn_bump_kernel() { t2_attn_gave_up(); }
extern "C" int t2amd_debug_attn_timeout_(void* stream) {
    hipLaunchKernelGGL(t2_attn_bump_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

typedef __bf16 at_bf16x8 __attribute__((ext_vector_type(8)));
#define AD T2AMD_ATT_DIM       // 128
#define NTAP T2AMD_LOC_TAPS    // 62
#define LK T2AMD_LOC_KERNEL    // 31
#define HALO 15
#define NSL T2AMD_ATT_SLICES   // 4 slices per utterance in every kernel
#define DSL (AD / NSL)         // 32 attention dims per K_e / K_b2 workgroup
#define NCS 4                  // context-channel slices (K_c workgroups per utterance)
#define NTS 4                  // position slices (K_b1 workgroups per utterance)
#define DCL 68                 // dcol_s row stride (floats) of the position-major image (A/B builds: -DT2AMD_DCOL_ROWMAJOR)
// Round 4: dcol lives TAP-MAJOR, dcolT_s[64 taps][NP + 4]: the product is formed transposed (dpre as the A operand, U^T as
// the B operand: the very same registers, same products summed over the same k order -> the same bits), so a lane holds FOUR
// POSITIONS of one tap and stores them as one float4, and col2im's 31 reads per output walk consecutive addresses --
// conflict-free, where the position-major image made 64 consecutive rows of a float4-aligned stride share 16 banks (4-way).
// (NP + 4) / 4 is odd for NP a multiple of 8, so the 16 tap rows of a float4 store cover all 64 banks as well.
#ifdef T2AMD_DCOL_ROWMAJOR
#define DCOL_FLOATS(np) ((size_t)(np) * DCL)
#else
#define DCOL_FLOATS(np) ((size_t)64 * ((np) + 4))
#endif
#define DPL 48                 // dpre_s row stride (floats): 48*lg mod 64 = {0,48,32,16}: conflict-free A reads

static inline int attn_tip(int Ti) { return (((Ti + 15) / 16) * 16 + 2 * HALO + 2 + 3) / 4 * 4; }

// LDS offset (without the position) of location tap `tap` = c*31+k in the two-window image
// win_s[2][TIP] (halo coordinates: win_s[c][ti + k] = w_c[ti + k - 15]).  Taps 62, 63 are padding.
__device__ __forceinline__ int tap_offset(int tap, int TIP) {
    if (tap >= NTAP) return 0;
    return tap < LK ? tap : TIP + tap - LK;
}

// The two halo windows go to LDS in two halves: the loads of the first pass (one element of each window per
// thread) are issued with the rest of the prologue loads and only written after them (a conditional load inside the
// staging loop is waited for on the spot); utterances longer than the block take the remaining passes in a loop.
struct WinRegs { float wp, cm; };
// SC1 (the persistent training-forward kernel): values another workgroup of the SAME launch wrote (write-through) are read
// with device-scope loads that bypass this CU's L1 -- a plain load may hit a line cached before the producer's store.
template <bool SC1> __device__ __forceinline__ float ld_xwg(const float* p) {
    if constexpr (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool SC1> __device__ __forceinline__ void st_xwg(float* p, float v) {
    if constexpr (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool SC1 = false>
__device__ __forceinline__ WinRegs stage_windows_issue(int TIP, int Ti, const float* wprev, const float* cum, int tid) {
    int ti = tid - HALO;
    ti = ti < 0 ? 0 : (ti > Ti - 1 ? Ti - 1 : ti);       // clamped: always a valid element, selected below
    WinRegs r;
    r.cm = ld_xwg<SC1>(cum + ti);
    r.wp = wprev ? ld_xwg<SC1>(wprev + ti) : 0.f;
    return r;
}
template <bool SC1 = false>
__device__ __forceinline__ void stage_windows_finish(const WinRegs& r, float* win_s, int TIP, int Ti, const float* wprev,
                                                     const float* cum, int tid, int nthreads) {
    if (tid < TIP) {
        const int ti = tid - HALO;
        const bool in = (ti >= 0 && ti < Ti);
        win_s[tid] = in ? r.wp : 0.f;
        win_s[TIP + tid] = in ? r.cm : 0.f;
    }
    for (int i = tid + nthreads; i < TIP; i += nthreads) {
        const int ti = i - HALO;
        const bool in = (ti >= 0 && ti < Ti);
        win_s[i] = (in && wprev) ? ld_xwg<SC1>(wprev + ti) : 0.f;
        win_s[TIP + i] = in ? ld_xwg<SC1>(cum + ti) : 0.f;
    }
}
// the slice's 32 rows of U (1984 contiguous floats, 16-byte aligned): one float4 per thread
__device__ __forceinline__ float4 stage_u_issue(const float* __restrict__ Uslice, int tid) {
    const int i = tid < DSL * NTAP / 4 ? tid : 0;
    return reinterpret_cast<const float4*>(Uslice)[i];
}
__device__ __forceinline__ void stage_u_finish(const float4& r, float* u_s, int tid) {
    if (tid < DSL * NTAP / 4) reinterpret_cast<float4*>(u_s)[tid] = r;
}

// loc^T tile: acc[dt][r] = sum_tap U[dt*16 + 4*lg + r][tap] * win[c(tap)][pos + k(tap)],  pos = lane&15
__device__ __forceinline__ void loc_tile(const float (&ua)[2][16], const float* __restrict__ win_s, int TIP,
                                         int pos, int lg, f32x4& acc0, f32x4& acc1) {
    acc0 = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    // all 16 window operands first (left to itself the compiler pairs every LDS read with its MFMA and waits for it)
    float bw[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) bw[kk] = win_s[tap_offset(kk * 4 + lg, TIP) + pos];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[0][kk], bw[kk], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[1][kk], bw[kk], acc1, 0, 0, 0);
    }
}

__device__ __forceinline__ void load_u_frag(float (&ua)[2][16], const float* __restrict__ U, int dbase, int l15, int lg) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int tap = kk * 4 + lg;
            ua[dt][kk] = tap < NTAP ? U[(long long)(dbase + dt * 16 + l15) * NTAP + tap] : 0.f;
        }
}

// ---- split-bf16 form of the location product (bf16 compute mode) ------------------------------------------
// One v_mfma_f32_16x16x32_bf16 per (dim tile, window channel c): K = 32 taps = the channel's 31 + one zero column.
// A = U[d][c*31 + k] (lane: d = l15, k = 8*lg + e), B = win[c][pos + k] (lane: pos = l15, k = 8*lg + e: eight
// consecutive floats of the halo window).  Both operands are split x = hi + lo (bf16 each) and the product is
// Uh.Wh + Uh.Wl + Ul.Wh: ~2^-17 relative per product instead of bf16's 2^-9, at 12 MFMAs of 16 cycles per tile
// instead of 32 of 32.
struct UFrag16 { uint4 hi[2][2], lo[2][2]; };      // [dim tile][channel]
__device__ __forceinline__ void split8(const float (&f)[8], uint4& hi, uint4& lo) {
    hi.x = t2_cvt_pk_bf16(f[0], f[1]); hi.y = t2_cvt_pk_bf16(f[2], f[3]);
    hi.z = t2_cvt_pk_bf16(f[4], f[5]); hi.w = t2_cvt_pk_bf16(f[6], f[7]);
    lo.x = t2_cvt_pk_bf16(f[0] - __uint_as_float(hi.x << 16), f[1] - __uint_as_float(hi.x & 0xffff0000u));
    lo.y = t2_cvt_pk_bf16(f[2] - __uint_as_float(hi.y << 16), f[3] - __uint_as_float(hi.y & 0xffff0000u));
    lo.z = t2_cvt_pk_bf16(f[4] - __uint_as_float(hi.z << 16), f[5] - __uint_as_float(hi.z & 0xffff0000u));
    lo.w = t2_cvt_pk_bf16(f[6] - __uint_as_float(hi.w << 16), f[7] - __uint_as_float(hi.w & 0xffff0000u));
}
__device__ __forceinline__ void load_u_frag16(UFrag16& uf, const float* __restrict__ u_s, int l15, int lg) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 8 * lg + e;
                f[e] = k < LK ? u_s[(dt * 16 + l15) * NTAP + c * LK + k] : 0.f;
            }
            split8(f, uf.hi[dt][c], uf.lo[dt][c]);
        }
}
__device__ __forceinline__ void loc_tile16(const UFrag16& uf, const float* __restrict__ win_s, int TIP, int pos, int lg,
                                           f32x4& acc0, f32x4& acc1) {
    acc0 = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    float f0[8], f1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        f0[e] = win_s[pos + 8 * lg + e];
        f1[e] = win_s[TIP + pos + 8 * lg + e];
    }
    uint4 h0, l0, h1, l1;
    split8(f0, h0, l0);
    split8(f1, h1, l1);
#define T2_M16(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(at_bf16x8, (A)), __builtin_bit_cast(at_bf16x8, (B)), C, 0, 0, 0)
    T2_M16(uf.lo[0][0], h0, acc0); T2_M16(uf.lo[1][0], h0, acc1);
    T2_M16(uf.hi[0][0], l0, acc0); T2_M16(uf.hi[1][0], l0, acc1);
    T2_M16(uf.lo[0][1], h1, acc0); T2_M16(uf.lo[1][1], h1, acc1);
    T2_M16(uf.hi[0][1], l1, acc0); T2_M16(uf.hi[1][1], l1, acc1);
    T2_M16(uf.hi[0][0], h0, acc0); T2_M16(uf.hi[1][0], h0, acc1);
    T2_M16(uf.hi[0][1], h1, acc0); T2_M16(uf.hi[1][1], h1, acc1);
#undef T2_M16
}

// acc + sum of 8 bf16 values (packed in the 16 bytes of m, ascending channel order) times 8 floats (g0, g1)
__device__ __forceinline__ float dot8_bf16(const float4& m, const float4& g0, const float4& g1, float acc) {
    const unsigned u0 = __float_as_uint(m.x), u1 = __float_as_uint(m.y), u2 = __float_as_uint(m.z), u3 = __float_as_uint(m.w);
    acc = fmaf(__uint_as_float(u0 << 16), g0.x, acc);
    acc = fmaf(__uint_as_float(u0 & 0xffff0000u), g0.y, acc);
    acc = fmaf(__uint_as_float(u1 << 16), g0.z, acc);
    acc = fmaf(__uint_as_float(u1 & 0xffff0000u), g0.w, acc);
    acc = fmaf(__uint_as_float(u2 << 16), g1.x, acc);
    acc = fmaf(__uint_as_float(u2 & 0xffff0000u), g1.y, acc);
    acc = fmaf(__uint_as_float(u3 << 16), g1.z, acc);
    acc = fmaf(__uint_as_float(u3 & 0xffff0000u), g1.w, acc);
    return acc;
}

struct AttnFwdParams {
    t2amd_attn_fwd a; int tip; int dbg; unsigned long long* ts;
    // one-launch form (attn_fwd_fused_kernel): launch token, float offset of the granule block in a.ws, K_c's LDS offset
    unsigned token; long long gran_off; int kc_smem_off; int delay;
};
typedef unsigned long long at_u64;
__device__ __forceinline__ void gran_publish(at_u64* g, unsigned tag, float v) {
    __hip_atomic_store(g, ((at_u64)tag << 32) | (at_u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// timing experiments only (tools/microbench_attn.py): T2AMD_ATTN_STAGE=n makes the kernels return after stage n
// timing experiments only (tools/microbench_attn.py --phases): with T2AMD_ATTN_TS=1 the first thread of workgroup
// (0,0) of every attention kernel stamps the 100 MHz wall clock at its phase boundaries into a 64-entry device buffer
// (slots 0-15 K_e, 16-31 K_c, 32-47 K_b1, 48-63 K_b2; api.hip owns the buffer), read back with t2amd_debug_attn_ts_.
static unsigned long long* attn_ts_buffer() { return t2amd_debug_ts_(); }
// T2AMD_ATTN_STAGE=n (tools/microbench_attn.py stage ablation) makes the kernels return after stage n: like the phase
// stamps, compiled in only in the instrumented build -- an early return is a barrier to the compiler's code motion.
#ifdef T2AMD_PHASE_STAMPS
#define T2_STAGE_RETURN(n) do { if (p.dbg == (n)) return; } while (0)
#else
#define T2_STAGE_RETURN(n) do { } while (0)
#endif
#ifdef T2AMD_PHASE_STAMPS
#define T2_TS(slot)                                                                        \
    do {                                                                                   \
        if (((slot) & 15) == 0) ts_on = t2_ts_begin(p.ts, (slot));                         \
        else t2_ts_mark(ts_on, p.ts, (slot));                                              \
    } while (0)
#else
#define T2_TS(slot) do { (void)ts_on; } while (0)
#endif

static int attn_dbg_stage() {
    static int v = -2;
    if (v == -2) { const char* e = getenv("T2AMD_ATTN_STAGE"); v = e ? atoi(e) : 0; }
    return v;
}

// ---------------------------------------------------------------------------------------
// K_e: partial energies over 32 attention dims
// ---------------------------------------------------------------------------------------
#define KE_NT 512
// MINW = waves per SIMD the register allocation must leave room for: 2 (one workgroup per CU, every load of the prologue
// in flight at once -- the latency-bound training / small-batch shape, <= 256 workgroups in a launch) or 4 (two resident
// workgroups per CU for launches of several rounds of workgroups: batched inference at B = 256 is 1024 of them).
// The body is a device function so that the one-launch forward (attn_fwd_fused_kernel below) can run it as its first
// phase.  GRAN: the partial energies leave as 8-byte {launch token, f32} granules, [B][Ti][4 slices], instead of floats.
// `after_prologue` runs once every prologue load has been issued and h has been staged (i.e. landed): what it issues
// flies behind the q phase and the tiles without holding up anything of this phase (loads complete in order).
// PERSIST (the persistent training-forward kernel): h, the previous weights and the cumulative weights were written by other
// workgroups of this very launch -- device-scope (sc1) loads.
// `before_h` runs right before the load of h is issued, with every other prologue load already in flight: the persistent
// kernel waits for the LSTM tiles' flags THERE, so that the W_q rows, the processed-memory rows, U, v and the windows -- none of
// which depends on this step's h -- travel during that wait instead of behind it.
// FASTT: the bf16 compute mode's one-range tanh (common.h t2_tanh_1r) -- chosen by the MODE, never by the launch form: every form of a
// mode (one launch, two launches, persistent loop) and the backward's recompute (attn_bwd_main_body: M16) use the same function.
template <bool GRAN, bool EARLYP, bool PERSIST, bool FASTT, class Pre, class Hook>
__device__ __forceinline__ void ke_phase(const AttnFwdParams& p, float* smem, const int ds, const int b, bool& ts_on,
                                         Pre&& before_h, Hook&& after_prologue) {
    const t2amd_attn_fwd& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int Ti = a.Ti, Hq = a.Hq, TIP = p.tip;
    T2_TS(0);
    float* win_s = smem;             // [2][TIP]
    float* q_s = win_s + 2 * TIP;    // [32]
    float* u_s = q_s + DSL;          // [32][62] the slice's rows of U (contiguous in HBM)
    float* h_s = u_s + DSL * NTAP;   // [Hq] this utterance's query input
    // Every load of the prologue is issued before the first one is consumed, in consumption order (the wait
    // counter is in-order): utterance length, processed-memory rows of this wave's first two position tiles,
    // U slice, window elements, v, then the W_q / h stream of the q phase.  Nothing is selected or compared on
    // a loaded value before the q phase has issued its loads (a select on a fresh load is waited for on the spot).
    const int len_raw = a.lens ? a.lens[b] : Ti;
    const float* __restrict__ pmb = a.pm + (long long)b * Ti * AD + ds * DSL + 4 * lg;
    float4 pmA[2], pmB[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        int pos = (wv + rr * (KE_NT / 64)) * 16 + l15;
        pos = pos < Ti ? pos : Ti - 1;                  // clamped; rows past the utterance are never used
        pmA[rr] = *reinterpret_cast<const float4*>(pmb + (long long)pos * AD);
        pmB[rr] = *reinterpret_cast<const float4*>(pmb + (long long)pos * AD + 16);
    }
    const float* wprev_b = a.w_prev ? a.w_prev + (long long)b * a.ld_wprev : nullptr;
    const float* cum_b = a.cum + (long long)b * Ti;
    const float4 ureg = stage_u_issue(a.U + (long long)ds * DSL * NTAP, tid);
    const WinRegs wreg = stage_windows_issue<PERSIST>(TIP, Ti, wprev_b, cum_b, tid);
    float vv[2][4];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) vv[dt][r] = a.v[ds * DSL + dt * 16 + 4 * lg + r];
    // q[d] = W_q[d][:] . h for the slice's 32 dims: 16 threads per row, 256 contiguous bytes per group and
    // instruction.  h goes through LDS (one float4 per thread), so a thread keeps only its 16 W_q float4 in flight
    // and the whole 1024-wide row is one round trip.
#ifdef T2AMD_ATTN_FWD_LATE                     // A/B builds only: the round-2 order
    constexpr bool EARLY = false;
#else
    constexpr bool EARLY = GRAN && EARLYP;     // one-launch form of the bf16 mode: location product ahead of the q product (see
                                               // below; the fp32-mode instantiation has no registers left for it: 25 spills)
#endif
    const bool split16 = a.loc_split_bf16 != 0;
    float ua[2][16];
    UFrag16 uf;
    f32x4 loc0[2], loc1[2];
    float qacc = 0.f;
    {
        const int d = tid >> 4, part = tid & 15;
        const float4* __restrict__ W4 = reinterpret_cast<const float4*>(a.Wq + (long long)(ds * DSL + d) * Hq);
        const float4* __restrict__ h4 = reinterpret_cast<const float4*>(a.h + (long long)b * a.ld_h);
        float4* h_s4 = reinterpret_cast<float4*>(h_s);
        const int n4 = Hq >> 2;
        // first 1024 columns straight-line (a loop header makes the compiler drain every pending load).
        // bf16 mode (a.Wq16): the row is 128 sixteen-byte units of 8 bf16 -- half the bytes of the stream that bounds
        // this prologue (128 KB of W_q per workgroup in f32); h and the sums stay f32.
        const bool wq16 = a.Wq16 != nullptr;
        const float4* __restrict__ W16 = reinterpret_cast<const float4*>(a.Wq16) + (long long)(ds * DSL + d) * (Hq >> 3);
        const int n8 = Hq >> 3;
        float4 w[16];
        if (wq16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = part + 16 * u;
                w[u] = W16[i < n8 ? i : part];
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = part + 16 * u;
                w[u] = W4[i < n4 ? i : part];
            }
        }
        {
            // h staged behind the W_q loads: its store is the first consumer of the whole prologue
            before_h();
            float4 hv;
            if constexpr (PERSIST) {
                const float* hq = a.h + (long long)b * a.ld_h + 4 * (tid < n4 ? tid : 0);
                hv.x = ld_xwg<true>(hq); hv.y = ld_xwg<true>(hq + 1); hv.z = ld_xwg<true>(hq + 2); hv.w = ld_xwg<true>(hq + 3);
            } else {
                hv = h4[tid < n4 ? tid : 0];
            }
            if (tid < n4) h_s4[tid] = hv;
            for (int j = tid + KE_NT; j < n4; j += KE_NT) h_s4[j] = h4[j];
        }
        after_prologue();
        if constexpr (EARLY) {
            // (round 3) the windows and the U slice go to LDS ahead of the q product -- their loads were issued before
            // the W_q stream, so they have landed -- and the location product of this lane's first two position tiles,
            // which needs nothing else, runs while the W_q rows are still on their way (same MFMAs on the same operands
            // as in the tile loop below: bit-identical)
            stage_windows_finish<PERSIST>(wreg, win_s, TIP, Ti, wprev_b, cum_b, tid, KE_NT);
            stage_u_finish(ureg, u_s, tid);
        }
        __syncthreads();
        if constexpr (EARLY) {
            if (split16) load_u_frag16(uf, u_s, l15, lg);
            else load_u_frag(ua, u_s, 0, l15, lg);
            const int nmt_e = (len_raw + 15) >> 4;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int mt = wv + rr * (KE_NT / 64);
                if (mt < nmt_e) {
                    if (split16) loc_tile16(uf, win_s, TIP, mt * 16 + l15, lg, loc0[rr], loc1[rr]);
                    else loc_tile(ua, win_s, TIP, mt * 16 + l15, lg, loc0[rr], loc1[rr]);
                }
            }
        }
        if (wq16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = part + 16 * u;
                if (i < n8) qacc = dot8_bf16(w[u], h_s4[2 * i], h_s4[2 * i + 1], qacc);
            }
            for (int i = part + 128; i < n8; i += 16) qacc = dot8_bf16(W16[i], h_s4[2 * i], h_s4[2 * i + 1], qacc);   // Hq > 1024
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = part + 16 * u;
                if (i < n4) {
                    const float4 x = h_s4[i];
                    qacc = fmaf(w[u].x, x.x, qacc);
                    qacc = fmaf(w[u].y, x.y, qacc);
                    qacc = fmaf(w[u].z, x.z, qacc);
                    qacc = fmaf(w[u].w, x.w, qacc);
                }
            }
        }
        for (int i = part + 256; i < (wq16 ? 0 : n4); i += 16) {          // Hq > 1024
            const float4 ww = W4[i], x = h_s4[i];
            qacc = fmaf(ww.x, x.x, qacc);
            qacc = fmaf(ww.y, x.y, qacc);
            qacc = fmaf(ww.z, x.z, qacc);
            qacc = fmaf(ww.w, x.w, qacc);
        }
    }
    if constexpr (!EARLY) {
        stage_windows_finish<PERSIST>(wreg, win_s, TIP, Ti, wprev_b, cum_b, tid, KE_NT);
        stage_u_finish(ureg, u_s, tid);
    }
    {
        const int d = tid >> 4, part = tid & 15;
        qacc = row16_sum(qacc);
        if (part == 0) {
            q_s[d] = qacc;
            if (a.q_out) a.q_out[(long long)b * a.ld_q + ds * DSL + d] = qacc;
        }
    }
    const int len = len_raw;
    const int nmt = (len + 15) >> 4;
    __syncthreads();
    T2_TS(1);
    T2_STAGE_RETURN(1);
    if constexpr (!EARLY) {
        if (split16) load_u_frag16(uf, u_s, l15, lg);
        else load_u_frag(ua, u_s, 0, l15, lg);
    }
    float qv[2][4];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) qv[dt][r] = q_s[dt * 16 + 4 * lg + r];

    float* __restrict__ eout = a.ws + ((long long)ds * a.B + b) * Ti;
    at_u64* __restrict__ gout = reinterpret_cast<at_u64*>(a.ws + p.gran_off) + (long long)b * Ti * NSL + ds;
    int round = 0;
    for (int mt = wv; mt < nmt; mt += KE_NT / 64, ++round) {
        const int pos = mt * 16 + l15;
        float4 pm0, pm1;
        if (round == 0) { pm0 = pmA[0]; pm1 = pmB[0]; }
        else if (round == 1) { pm0 = pmA[1]; pm1 = pmB[1]; }
        else {
            pm0 = pm1 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pos < Ti) {
                pm0 = *reinterpret_cast<const float4*>(pmb + (long long)pos * AD);
                pm1 = *reinterpret_cast<const float4*>(pmb + (long long)pos * AD + 16);
            }
        }
        f32x4 acc0, acc1;
        if (EARLY && round < 2) {
            acc0 = round == 0 ? loc0[0] : loc0[1];
            acc1 = round == 0 ? loc1[0] : loc1[1];
        } else if (split16) loc_tile16(uf, win_s, TIP, pos, lg, acc0, acc1);
        else loc_tile(ua, win_s, TIP, pos, lg, acc0, acc1);
        float e = vv[0][0] * t2_tanh_sel<FASTT>(acc0[0] + qv[0][0] + pm0.x);
        e = fmaf(vv[0][1], t2_tanh_sel<FASTT>(acc0[1] + qv[0][1] + pm0.y), e);
        e = fmaf(vv[0][2], t2_tanh_sel<FASTT>(acc0[2] + qv[0][2] + pm0.z), e);
        e = fmaf(vv[0][3], t2_tanh_sel<FASTT>(acc0[3] + qv[0][3] + pm0.w), e);
        e = fmaf(vv[1][0], t2_tanh_sel<FASTT>(acc1[0] + qv[1][0] + pm1.x), e);
        e = fmaf(vv[1][1], t2_tanh_sel<FASTT>(acc1[1] + qv[1][1] + pm1.y), e);
        e = fmaf(vv[1][2], t2_tanh_sel<FASTT>(acc1[2] + qv[1][2] + pm1.z), e);
        e = fmaf(vv[1][3], t2_tanh_sel<FASTT>(acc1[3] + qv[1][3] + pm1.w), e);
        e += __shfl_xor(e, 16, 64);
        e += __shfl_xor(e, 32, 64);
        if (lg == 0 && pos < Ti) {
            if constexpr (GRAN) gran_publish(gout + (long long)pos * NSL, p.token, e);
            else eout[pos] = e;
        }
    }
    T2_TS(2);
}

template <int MINW, bool FASTT>
__global__ __launch_bounds__(KE_NT, MINW) void attn_energy_kernel(AttnFwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    bool ts_on = false;
    if (p.a.active && !p.a.active[blockIdx.y]) return;
    ke_phase<false, false, false, FASTT>(p, smem, blockIdx.x, blockIdx.y, ts_on, [] {}, [] {});
}

// ---------------------------------------------------------------------------------------
// K_e for decode batches of more than 128 utterances (bf16 mode): FOUR utterances per workgroup, side by side.
//
// With one utterance per workgroup a launch at B = 256 is four workgroups per CU, each fetching the slice's 64 KB of
// bf16 W_q rows again (64 MB of the ~100 MB a step moves L2 -> CU) and each paying its own load -> dot -> barrier ->
// tiles latency chain.  Here the W_q rows are fetched once per four utterances and stay in registers for four dot
// products, the four query vectors / window pairs / lengths sit side by side in LDS, and the 8 waves share the
// 4 x ceil(Ti/16) position tiles: one latency chain, one round of B/4 x 4 workgroups.  (Walking the four utterances one
// after the other inside a workgroup measured slower than the one-utterance form: four chains in sequence.)
// Per-thread arithmetic of q and of every energy is that of attn_energy_kernel: results are bit-identical.
// ---------------------------------------------------------------------------------------
#define KE4_U 4
// PF = tiles per wave whose processed-memory rows are prefetched with the prologue, ahead of the W_q / h stream; the rest
// are fetched on demand.  Measured at B = 256, Ti = 187 (6 tiles per wave): PF = 2 15.4 us, PF = 2 issued after the W_q / h
// stream 15.8 us, PF = 6 17.9 us -- more rows in flight only delay the q phase (the wait counter is in-order).
#define KE4_PF 2
__global__ __launch_bounds__(KE_NT) void attn_energy4_kernel(AttnFwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const t2amd_attn_fwd& a = p.a;
    const int ds = blockIdx.x, b0 = blockIdx.y * KE4_U;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int Ti = a.Ti, Hq = a.Hq, TIP = p.tip, B = a.B;
    float* win_s = smem;                         // [U][2][TIP]
    float* q_s = win_s + KE4_U * 2 * TIP;        // [U][32]
    float* u_s = q_s + KE4_U * DSL;              // [32][62]
    float* h_s = u_s + DSL * NTAP;               // [U][Hq]
    int* len_s = reinterpret_cast<int*>(h_s + KE4_U * Hq);   // [U] (0: absent or finished utterance)
    const int NMT = (Ti + 15) >> 4;              // position tiles per utterance (padded length)

    // ---- prologue: every load issued before the first one is consumed; nothing is compared on a loaded value before
    // the last load has been issued (a comparison on a fresh load is waited for on the spot) ----
    int bc[KE4_U], len_raw[KE4_U], act_raw[KE4_U];
#pragma unroll
    for (int u = 0; u < KE4_U; ++u) {
        const int bu = b0 + u;
        bc[u] = bu < B ? bu : B - 1;             // clamped: loads stay in bounds, results of absent rows are dropped
        act_raw[u] = a.active ? a.active[bc[u]] : 1;
        len_raw[u] = a.lens ? a.lens[bc[u]] : Ti;
    }
    constexpr int PF = KE4_PF;
    float4 pmA[PF], pmB[PF];                     // processed-memory rows of this wave's first PF tiles
    auto issue_pm = [&]() {
#pragma unroll
        for (int rr = 0; rr < PF; ++rr) {
            int g = wv + rr * (KE_NT / 64);
            g = g < KE4_U * NMT ? g : KE4_U * NMT - 1;
            const int u = g / NMT, mt = g - u * NMT;
            int pos = mt * 16 + l15;
            pos = pos < Ti ? pos : Ti - 1;
            const int bb = b0 + u < B ? b0 + u : B - 1;
            const float* pr = a.pm + ((long long)bb * Ti + pos) * AD + ds * DSL + 4 * lg;
            pmA[rr] = *reinterpret_cast<const float4*>(pr);
            pmB[rr] = *reinterpret_cast<const float4*>(pr + 16);
        }
    };
    issue_pm();
    const float4 ureg = stage_u_issue(a.U + (long long)ds * DSL * NTAP, tid);
    WinRegs wreg[KE4_U];
#pragma unroll
    for (int u = 0; u < KE4_U; ++u)
        wreg[u] = stage_windows_issue(TIP, Ti, a.w_prev ? a.w_prev + (long long)bc[u] * a.ld_wprev : nullptr,
                                      a.cum + (long long)bc[u] * Ti, tid);
    float vv[2][4];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) vv[dt][r] = a.v[ds * DSL + dt * 16 + 4 * lg + r];
    // q[u][d] = W_q[d][:] . h_u: 16 threads per row, the row's bf16 units stay in registers for the four utterances
    const int d = tid >> 4, part = tid & 15;
    const float4* __restrict__ W16 = reinterpret_cast<const float4*>(a.Wq16) + (long long)(ds * DSL + d) * (Hq >> 3);
    const int n8 = Hq >> 3, n4 = Hq >> 2;
    float4 w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = part + 16 * k;
        w[k] = W16[i < n8 ? i : part];
    }
    float4* h_s4 = reinterpret_cast<float4*>(h_s);
    auto h_src = [&](int j) {
        const int u = j / n4, k = j - u * n4;
        const int bb = b0 + u < B ? b0 + u : B - 1;
        return reinterpret_cast<const float4*>(a.h + (long long)bb * a.ld_h) + k;
    };
    // the four query inputs, staged behind the W_q loads; the first two float4 per thread straight-line (a loop header
    // makes the compiler drain every pending load), which is all of them at Hq = 1024
    const int tot = KE4_U * n4;
    const float4 hv0 = *h_src(tid < tot ? tid : 0);
    const float4 hv1 = *h_src(tid + KE_NT < tot ? tid + KE_NT : 0);
    if (tid < tot) h_s4[tid] = hv0;
    if (tid + KE_NT < tot) h_s4[tid + KE_NT] = hv1;
    for (int j = tid + 2 * KE_NT; j < tot; j += KE_NT) h_s4[j] = *h_src(j);
    bool live[KE4_U];
#pragma unroll
    for (int u = 0; u < KE4_U; ++u) live[u] = b0 + u < B && act_raw[u] != 0;
    if (tid < KE4_U) {
        int lv = 0;
#pragma unroll
        for (int u = 0; u < KE4_U; ++u) lv = tid == u ? (live[u] ? len_raw[u] : 0) : lv;
        len_s[tid] = lv;
    }
    __syncthreads();
    float qacc[KE4_U];
#pragma unroll
    for (int u = 0; u < KE4_U; ++u) {
        const float4* hu = reinterpret_cast<const float4*>(h_s + u * Hq);
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = part + 16 * k;
            if (i < n8) acc = dot8_bf16(w[k], hu[2 * i], hu[2 * i + 1], acc);
        }
        for (int i = part + 128; i < n8; i += 16) acc = dot8_bf16(W16[i], hu[2 * i], hu[2 * i + 1], acc);   // Hq > 1024
        qacc[u] = acc;
    }
#pragma unroll
    for (int u = 0; u < KE4_U; ++u)
        stage_windows_finish(wreg[u], win_s + u * 2 * TIP, TIP, Ti, a.w_prev ? a.w_prev + (long long)bc[u] * a.ld_wprev : nullptr,
                             a.cum + (long long)bc[u] * Ti, tid, KE_NT);
    stage_u_finish(ureg, u_s, tid);
#pragma unroll
    for (int u = 0; u < KE4_U; ++u) {
        const float qs = row16_sum(qacc[u]);
        if (part == 0) {
            q_s[u * DSL + d] = qs;
            if (a.q_out && live[u]) a.q_out[(long long)(b0 + u) * a.ld_q + ds * DSL + d] = qs;
        }
    }
    __syncthreads();

    // ---- energies: the 8 waves share the U x NMT position tiles ----
    UFrag16 uf;
    load_u_frag16(uf, u_s, l15, lg);
    auto tile = [&](int g, float4 pm0, float4 pm1, bool fetched) {
        const int u = g / NMT, mt = g - u * NMT;
        const int len = len_s[u];
        if (mt * 16 >= len) return;                    // past the utterance (or the utterance is absent / finished)
        const int pos = mt * 16 + l15;
        if (!fetched) {
            pm0 = pm1 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pos < Ti) {
                const float* pr = a.pm + ((long long)(b0 + u) * Ti + pos) * AD + ds * DSL + 4 * lg;
                pm0 = *reinterpret_cast<const float4*>(pr);
                pm1 = *reinterpret_cast<const float4*>(pr + 16);
            }
        }
        float qv[2][4];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) qv[dt][r] = q_s[u * DSL + dt * 16 + 4 * lg + r];
        f32x4 acc0, acc1;
        loc_tile16(uf, win_s + u * 2 * TIP, TIP, pos, lg, acc0, acc1);
        float e = vv[0][0] * t2_tanh_sel<true>(acc0[0] + qv[0][0] + pm0.x);
        e = fmaf(vv[0][1], t2_tanh_sel<true>(acc0[1] + qv[0][1] + pm0.y), e);
        e = fmaf(vv[0][2], t2_tanh_sel<true>(acc0[2] + qv[0][2] + pm0.z), e);
        e = fmaf(vv[0][3], t2_tanh_sel<true>(acc0[3] + qv[0][3] + pm0.w), e);
        e = fmaf(vv[1][0], t2_tanh_sel<true>(acc1[0] + qv[1][0] + pm1.x), e);
        e = fmaf(vv[1][1], t2_tanh_sel<true>(acc1[1] + qv[1][1] + pm1.y), e);
        e = fmaf(vv[1][2], t2_tanh_sel<true>(acc1[2] + qv[1][2] + pm1.z), e);
        e = fmaf(vv[1][3], t2_tanh_sel<true>(acc1[3] + qv[1][3] + pm1.w), e);
        e += __shfl_xor(e, 16, 64);
        e += __shfl_xor(e, 32, 64);
        if (lg == 0 && pos < Ti) a.ws[((long long)ds * B + (b0 + u)) * Ti + pos] = e;
    };
#pragma unroll
    for (int rr = 0; rr < PF; ++rr) {
        const int g = wv + rr * (KE_NT / 64);
        if (g < KE4_U * NMT) tile(g, pmA[rr], pmB[rr], true);
    }
    for (int g = wv + PF * (KE_NT / 64); g < KE4_U * NMT; g += KE_NT / 64)          // rows fetched on demand
        tile(g, make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), false);
}

// ---------------------------------------------------------------------------------------
// K_c: softmax over the utterance + one quarter of the context channels
// ---------------------------------------------------------------------------------------
#define KC_NT 512
// M16: the context rows come from a bf16 copy of the encoder memory (t2amd_attn_fwd.memory16, the engine's bf16
// compute mode): a thread then owns 8 channels (one 16-byte load per row) of twice as many row groups, i.e. half
// the bytes of the f32 stream this kernel is bound by; weights, accumulation and the context stay f32.
// The body is split so that the one-launch forward (attn_fwd_fused_kernel) can issue the context rows at its very
// start (KcPre: everything that does not depend on the energies) and run the rest behind the energy hand-off.
template <bool M16>
struct KcPre {
    static constexpr int MAXR = M16 ? 6 : 12;     // memory rows a thread keeps in registers; longer utterances take extra passes
    float4 mrow[MAXR];
    float c_old0;
    int len;
};
template <bool M16, bool LOAD_E, bool PERSIST = false> __device__ __forceinline__ void kc_issue(const AttnFwdParams& p, int cs, int b, KcPre<M16>& r, float (&e_first)[4]);
template <bool M16, bool FUSED, bool PERSIST = false> __device__ __forceinline__ void kc_finish(const AttnFwdParams& p, float* smem, int cs, int b, bool& ts_on,
                                                                                          const KcPre<M16>& r, const float (&e_first)[4]);

template <bool M16>
__global__ __launch_bounds__(KC_NT) void attn_context_kernel(AttnFwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    bool ts_on = false;
    const t2amd_attn_fwd& a = p.a;
    const int cs = blockIdx.x, b = blockIdx.y;
    if (a.active && !a.active[b]) return;
    T2_TS(16);
    float e_first[4];
    KcPre<M16> r;
    kc_issue<M16, true>(p, cs, b, r, e_first);
    kc_finish<M16, false>(p, smem, cs, b, ts_on, r, e_first);
}

// LOAD_E: the four partial energies of this thread's first position are loaded here (two-launch form), first
template <bool M16, bool LOAD_E, bool PERSIST>
__device__ __forceinline__ void kc_issue(const AttnFwdParams& p, const int cs, const int b, KcPre<M16>& r, float (&e_first)[4]) {
    constexpr int CPT = M16 ? 8 : 4;          // channels per thread
    constexpr int KC_MAXR = KcPre<M16>::MAXR;
    const t2amd_attn_fwd& a = p.a;
    const int tid = threadIdx.x;
    const int Ti = a.Ti, E = a.E;
    // Length through the scalar path (s_load from the constant address space): a vector load would put it in the same
    // in-order queue as the streams below.
    const int len = a.lens ? *reinterpret_cast<const __attribute__((address_space(4))) int*>(
                                 reinterpret_cast<uintptr_t>(a.lens + b)) : Ti;

    // Row offsets of this thread's share of memory[b], computed before any load is issued (address arithmetic placed
    // between loads can falsely depend on a pending destination register and drain the queue).
    // (EC4 / E4 count 16-byte units: 4 floats, or 8 bf16 in the M16 form)
    const int EC = E / NCS, EC4 = EC / CPT, E4 = E / CPT;
    int parts = KC_NT / EC4;
    if (parts > 32) parts = 32;
    const int c4 = tid % EC4, part = tid / EC4;
    const float4* __restrict__ M4 = reinterpret_cast<const float4*>(M16 ? a.memory16 : (const void*)a.memory) +
                                    (long long)b * Ti * E4 + cs * EC4 + c4;
    long long roff[KC_MAXR];
#pragma unroll
    for (int i = 0; i < KC_MAXR; ++i) roff[i] = (long long)(part + i * parts) * E4;
    const int tc0 = tid < Ti ? tid : Ti - 1;
    const float* const cum_b = a.cum + (long long)b * Ti;
    const float* __restrict__ e0 = a.ws + (long long)b * Ti;
    const long long es = (long long)a.B * Ti;
    __builtin_amdgcn_sched_barrier(0);

    // Partial energies of this thread's first position first: loads complete in order, so the softmax below waits
    // for these four only, not for the 1 KB-per-row context stream issued behind them.
    if constexpr (LOAD_E) {
#pragma unroll
        for (int k = 0; k < 4; ++k) e_first[k] = e0[k * es + tc0];
    }
    // slice 0 also carries the cumulative weights forward: its read-modify-write operand is fetched now
    r.c_old0 = 0.f;
    if (cs == 0) r.c_old0 = ld_xwg<PERSIST>(cum_b + tc0);
    // The context rows do not depend on the softmax.  Rows past the utterance are clamped to row 0 (an L1 hit) and
    // get weight 0 below: the kernel is bound by the rows it moves, skipping the padding is worth the scalar wait.
#pragma unroll
    for (int i = 0; i < KC_MAXR; ++i) {
        const int ti = part + i * parts;
        r.mrow[i] = M4[ti < len ? roff[i] : 0ll];
    }
    r.len = len;
    __builtin_amdgcn_sched_barrier(0);
}

// FUSED: Ti <= KC_NT (one position per thread; the host checks), e_first came through the granules
// PERSIST: the weights row and the cumulative weights (read by the utterance's four workgroups at the next time step of the
// same launch) and the bf16 context (read by every LSTM tile) leave as write-through stores; the caller drains them.
template <bool M16, bool FUSED, bool PERSIST>
__device__ __forceinline__ void kc_finish(const AttnFwdParams& p, float* smem, const int cs, const int b, bool& ts_on,
                                          const KcPre<M16>& r, const float (&e_first)[4]) {
    constexpr int CPT = M16 ? 8 : 4;
    constexpr int KC_MAXR = KcPre<M16>::MAXR;
    const t2amd_attn_fwd& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int Ti = a.Ti, E = a.E, B = a.B;
    float* w_s = smem;                    // [Ti rounded to 4]
    float* red_s = w_s + ((Ti + 3) & ~3); // [16]
    float* part_s = red_s + 16;           // [parts][EC]
    const int len = r.len;
    const int EC = E / NCS, EC4 = EC / CPT, E4 = E / CPT;
    int parts = KC_NT / EC4;
    if (parts > 32) parts = 32;
    const int c4 = tid % EC4, part = tid / EC4;
    const bool worker = part < parts;
    const float4* __restrict__ M4 = reinterpret_cast<const float4*>(M16 ? a.memory16 : (const void*)a.memory) +
                                    (long long)b * Ti * E4 + cs * EC4 + c4;
    const float* __restrict__ e0 = a.ws + (long long)b * Ti;
    const long long es = (long long)B * Ti;
    float* const cum_b = a.cum + (long long)b * Ti;
    const float c_old0 = r.c_old0;
    const float4 (&mrow)[KC_MAXR] = r.mrow;

    float lmax = -INFINITY;
    if (tid < Ti) {
        const float e = tid < len ? ((e_first[0] + e_first[1]) + e_first[2]) + e_first[3] : -INFINITY;
        w_s[tid] = e;
        lmax = e;
    }
    for (int ti = tid + KC_NT; ti < Ti; ti += KC_NT) {
        float e = -INFINITY;
        if (ti < len) e = ((e0[ti] + e0[es + ti]) + e0[2 * es + ti]) + e0[3 * es + ti];
        w_s[ti] = e;
        lmax = fmaxf(lmax, e);
    }
    lmax = wave_reduce_max(lmax);
    if (lane == 0) red_s[wv] = lmax;
    __syncthreads();
    T2_TS(17);
    float gmax = red_s[0];
#pragma unroll
    for (int i = 1; i < KC_NT / 64; ++i) gmax = fmaxf(gmax, red_s[i]);
    float lsum = 0.f;
    for (int ti = tid; ti < Ti; ti += KC_NT) {
        const float ex = (ti < len) ? expf(w_s[ti] - gmax) : 0.f;
        w_s[ti] = ex;
        lsum += ex;
    }
    lsum = wave_reduce_sum(lsum);
    if (lane == 0) red_s[8 + wv] = lsum;
    __syncthreads();
    T2_TS(18);
    float gsum = red_s[8];
#pragma unroll
    for (int i = 1; i < KC_NT / 64; ++i) gsum += red_s[8 + i];
    const float inv = 1.0f / gsum;
    {
        float* wout = a.w_out + (long long)b * a.ld_wout;
        float* cum = cum_b;
        float* csave = a.cum_save ? a.cum_save + (long long)b * Ti : nullptr;
        for (int ti = tid; ti < Ti; ti += KC_NT) {
            const float w = w_s[ti] * inv;
            w_s[ti] = w;
            if (cs == 0) {
                st_xwg<PERSIST>(wout + ti, w);
                const float c_old = (ti == tid) ? c_old0 : ld_xwg<PERSIST>(cum + ti);
                if (csave) csave[ti] = c_old;
                st_xwg<PERSIST>(cum + ti, c_old + w);
            }
        }
    }
    __syncthreads();
    T2_TS(19);
    // context channels [cs*EC, (cs+1)*EC)
    if (worker) {
        float acc[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) acc[j] = 0.f;
        auto fma_row = [&](const float4& m, float w) {
            if constexpr (M16) {
                const unsigned u[4] = {__float_as_uint(m.x), __float_as_uint(m.y), __float_as_uint(m.z), __float_as_uint(m.w)};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[2 * j] = fmaf(w, __uint_as_float(u[j] << 16), acc[2 * j]);
                    acc[2 * j + 1] = fmaf(w, __uint_as_float(u[j] & 0xffff0000u), acc[2 * j + 1]);
                }
            } else {
                acc[0] = fmaf(w, m.x, acc[0]); acc[1] = fmaf(w, m.y, acc[1]);
                acc[2] = fmaf(w, m.z, acc[2]); acc[3] = fmaf(w, m.w, acc[3]);
            }
        };
#pragma unroll
        for (int i = 0; i < KC_MAXR; ++i) {
            const int ti = part + i * parts;
            fma_row(mrow[i], ti < len ? w_s[ti] : 0.f);
        }
        for (int ti = part + KC_MAXR * parts; ti < len; ti += parts) {      // utterances longer than MAXR*parts
            const float4 m = M4[(long long)ti * E4];
            fma_row(m, w_s[ti]);
        }
#pragma unroll
        for (int j = 0; j < CPT; j += 4)
            *reinterpret_cast<float4*>(&part_s[part * EC + c4 * CPT + j]) = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
    }
    __syncthreads();
    T2_TS(20);
#ifdef T2AMD_SW_VECTOR_WT
    if constexpr (PERSIST) {
        // (round 5, measured and NOT adopted: see the epilogue of skinny_wide.h) the context leaves as 16-BYTE write-through stores -- eight consecutive lanes hold eight consecutive channels:
        // lane 0 of a group writes the bf16 copy as one store (bf16 mode) / lanes 0 and 4 the f32 values as two (fp32 mode, where the
        // LSTM tiles of this launch read the f32 context itself) -- instead of one fabric write per channel (skinny_wide.h, epilogue).
        // EC is a multiple of 8 (host check), so a group never straddles the slice.
        for (int c0 = 0; c0 < EC; c0 += KC_NT) {
            const int c = c0 + tid;
            float sv = 0.f;
            if (c < EC)
                for (int q = 0; q < parts; ++q) sv += part_s[q * EC + c];
            float v8[8];
            v8[0] = sv;
#pragma unroll
            for (int k = 1; k < 8; ++k) v8[k] = __shfl_down(sv, k, 64);
            if (c < EC) {
                float* const cp = &a.ctx_out[(long long)b * a.ld_ctx + cs * EC + c];
                if (a.ctx16_out) {
                    *cp = sv;                                   // only the backward reads the f32 context in the bf16 mode
                    if ((tid & 7) == 0) {
                        const sk_u32x4 pk = {(unsigned)t2_f32_to_bf16(v8[0]) | ((unsigned)t2_f32_to_bf16(v8[1]) << 16),
                                             (unsigned)t2_f32_to_bf16(v8[2]) | ((unsigned)t2_f32_to_bf16(v8[3]) << 16),
                                             (unsigned)t2_f32_to_bf16(v8[4]) | ((unsigned)t2_f32_to_bf16(v8[5]) << 16),
                                             (unsigned)t2_f32_to_bf16(v8[6]) | ((unsigned)t2_f32_to_bf16(v8[7]) << 16)};
                        unsigned short* const c16 = reinterpret_cast<unsigned short*>(a.ctx16_out) + (long long)b * a.ld_ctx16 + cs * EC + c;
                        asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(c16), "v"(pk) : "memory");
                    }
                } else if ((tid & 3) == 0) {
                    const f32x4 v = {v8[0], v8[1], v8[2], v8[3]};
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(cp), "v"(v) : "memory");
                }
            }
        }
        T2_TS(21);
        return;
    }
#endif
    for (int c = tid; c < EC; c += KC_NT) {
        float s = 0.f;
        for (int q = 0; q < parts; ++q) s += part_s[q * EC + c];
        // (PERSIST, fp32 mode: the LSTM tiles of this launch read the f32 context itself -- there is no bf16 copy: write-through)
        if (PERSIST && !a.ctx16_out) st_xwg<PERSIST>(&a.ctx_out[(long long)b * a.ld_ctx + cs * EC + c], s);
        else a.ctx_out[(long long)b * a.ld_ctx + cs * EC + c] = s;
        if (a.ctx16_out && a.ctx16_x3) {
            // 'bf16x3' mode (round 6): the LSTM tiles' operand is the split hi/lo image of the context (t2amd_split_bf16x3_f32 layout)
            unsigned short hi_, lo_;
            t2_split_bf16(s, hi_, lo_);
            unsigned short* c16 = reinterpret_cast<unsigned short*>(a.ctx16_out) + (long long)b * a.ld_ctx16 * 2 + t2_x3_pos(cs * EC + c);
            if constexpr (PERSIST) {
                __hip_atomic_store(c16, hi_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(c16 + 16, lo_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                c16[0] = hi_; c16[16] = lo_;
            }
        } else if (a.ctx16_out) {
            unsigned short* c16 = reinterpret_cast<unsigned short*>(a.ctx16_out) + (long long)b * a.ld_ctx16 + cs * EC + c;
            if constexpr (PERSIST) __hip_atomic_store(c16, t2_f32_to_bf16(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *c16 = t2_f32_to_bf16(s);
        }
    }
    T2_TS(21);
}

// ---------------------------------------------------------------------------------------
// One-launch forward: K_e's phase, energy hand-off, K_c's phase (t2amd_set_attn_fwd_fused).
//
// K_e -> K_c is the one boundary of a decoder time step that lies inside an utterance: workgroup (s, b) computes the
// partial energies of dim slice s, the four workgroups of utterance b exchange them as 8-byte {launch token, f32}
// granules ([B][Ti][4], one write-through store per position and slice, polled by thread ti of every consumer: the data
// is the flag), and workgroup (s, b) carries on as K_c for context-channel slice s.  K_c's context rows -- its only
// long-latency operand -- are issued behind K_e's prologue and land during the q phase and the tiles.  Same per-thread
// arithmetic and summation order as the two launches: bit-identical weights, context and cumulative weights.
// ---------------------------------------------------------------------------------------
// The energy hand-off of the one-launch forms: thread ti < len polls the four granules of position ti (32 contiguous bytes)
// until they carry this step's token.  Bounded like every spin here: 50 ms of the 100 MHz wall clock, then NaN energies
// (-> NaN weights and context) instead of a hung GPU.
__device__ __forceinline__ void fwd_energy_granules(const AttnFwdParams& p, const int b, const int len, float (&e_first)[4]) {
    const t2amd_attn_fwd& a = p.a;
    const int tid = threadIdx.x;
    const int Ti = a.Ti;
    const at_u64* g = reinterpret_cast<const at_u64*>(a.ws + p.gran_off) + ((long long)b * Ti + (tid < Ti ? tid : Ti - 1)) * NSL;
    const bool need = tid < len;
    for (int d_ = 0; d_ < p.delay; ++d_) __builtin_amdgcn_s_sleep(1);
    at_u64 x[NSL];
#pragma unroll
    for (int k = 0; k < NSL; ++k) x[k] = __hip_atomic_load(g + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0_ = wall_clock64();
    unsigned spins_ = 0;
    bool bad = false;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NSL; ++k) ok = ok && (unsigned)(x[k] >> 32) == p.token;
        if (__all(ok || !need)) break;
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int k = 0; k < NSL; ++k) x[k] = __hip_atomic_load(g + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((++spins_ & 255u) == 0 && wall_clock64() - t0_ > 5000000ll) { bad = true; t2_attn_gave_up(); break; }
    }
#pragma unroll
    for (int k = 0; k < NSL; ++k) e_first[k] = bad ? __builtin_nanf("") : __uint_as_float((unsigned)x[k]);
}

template <bool M16>
__global__ __launch_bounds__(KE_NT, 2) void attn_fwd_fused_kernel(AttnFwdParams p) {
    static_assert(KE_NT == KC_NT, "one thread mapping for both phases");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    bool ts_on = false;
    const t2amd_attn_fwd& a = p.a;
    const int sl = blockIdx.x, b = blockIdx.y;
    if (a.active && !a.active[b]) return;
    const int tid = threadIdx.x;
    KcPre<M16> r;
    float e_first[4] = {0.f, 0.f, 0.f, 0.f};
    ke_phase<true, M16, false, M16>(p, smem, sl, b, ts_on, [] {}, [&] { kc_issue<M16, false>(p, sl, b, r, e_first); });
    T2_TS(16);
    fwd_energy_granules(p, b, r.len, e_first);
    kc_finish<M16, true>(p, smem + p.kc_smem_off, sl, b, ts_on, r, e_first);
}

static int g_attn_fwd_fused = -1;          // -1: environment / default; 0 / 1: t2amd_set_attn_fwd_fused
static unsigned g_attn_fwd_token = 0;
#define T2_ATTN_FWD_FUSED_DEFAULT 1      // measured on MI355X: 63.0 vs 64.0 ms per training step (profiles/r02_aa_ab_fwd_fused.json)
extern "C" int t2amd_set_attn_fwd_fused(int on) {
    T2_REQUIRE(on == 0 || on == 1 || on == -1, "set_attn_fwd_fused: -1 (default), 0 or 1");
    g_attn_fwd_fused = on;
    return T2AMD_OK;
}
extern "C" long long t2amd_attn_fwd_ws_floats(int B, int Ti) {
    // [4][B][Ti] partial energies (two-launch form), then the granule block [B][Ti][4] x 8 bytes; a multiple of 4 floats
    const long long e = ((long long)NSL * B * Ti + 3) / 4 * 4;
    return e + 2ll * NSL * B * Ti;
}

extern "C" int t2amd_attention_step_fwd_f32(const t2amd_attn_fwd* a, void* stream) {
    T2_REQUIRE(a && a->h && a->Wq && a->U && a->v && a->pm && a->memory && a->cum && a->w_out && a->ctx_out && a->ws,
               "attn_fwd: null pointer");
    T2_REQUIRE(a->B > 0 && a->Ti > 0 && a->Ti <= 8192, "attn_fwd: Ti out of range");
    T2_REQUIRE(a->E % (4 * NCS) == 0 && a->E >= 4 * NCS && a->E <= 4096, "attn_fwd: E must be a multiple of 32, <= 4096");
    T2_REQUIRE(a->Hq % 32 == 0 && a->Hq > 0, "attn_fwd: Hq must be a multiple of 32");
    T2_REQUIRE(t2_aligned16(a->Wq) && t2_aligned16(a->memory) && t2_aligned16(a->pm) && t2_aligned16(a->h) &&
                   a->ld_h % 4 == 0,
               "attn_fwd: Wq/memory/pm/h must be 16-byte aligned");
    AttnFwdParams p;
    p.a = *a;
    p.tip = attn_tip(a->Ti);
    p.dbg = attn_dbg_stage();
    p.ts = attn_ts_buffer();
    p.token = 0; p.gran_off = 0; p.kc_smem_off = 0; p.delay = 0;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds_e = sizeof(float) * (2 * (size_t)p.tip + DSL + DSL * NTAP + (size_t)a->Hq);
    const int EC = a->E / NCS;
    T2_REQUIRE(!a->memory16 || (t2_aligned16(a->memory16) && EC % 8 == 0), "attn_fwd: memory16 must be 16-byte aligned, E a multiple of 32");
    T2_REQUIRE(!a->Wq16 || (t2_aligned16(a->Wq16) && a->Hq % 128 == 0), "attn_fwd: Wq16 must be 16-byte aligned, Hq a multiple of 128");
    int parts = KC_NT / (EC / (a->memory16 ? 8 : 4));
    if (parts > 32) parts = 32;
    T2_REQUIRE(parts >= 1, "attn_fwd: E too large");
    const size_t lds_c = sizeof(float) * ((size_t)((a->Ti + 3) & ~3) + 16 + (size_t)parts * EC);
    T2_REQUIRE(lds_e <= 64 * 1024 && lds_c <= 64 * 1024, "attn_fwd: Ti too large for the LDS windows");
    // T2AMD_KE_FORM (A/B runs only): 0 one utterance per workgroup, 1 the two-workgroups-per-CU register allocation of
    // it, 2 four utterances side by side in a workgroup.  Default: 2 for B > 128 (more than two rounds of one-utterance
    // workgroups) in the bf16 mode, 1 for B > 128 otherwise, else 0.
    static const int ke_form = [] { const char* e = getenv("T2AMD_KE_FORM"); return e ? atoi(e) : -1; }();
    int form = ke_form >= 0 ? ke_form : ((long long)NSL * a->B > 512 ? 2 : 0);
    const size_t lds_e4 = sizeof(float) * (KE4_U * (2 * (size_t)p.tip + DSL + (size_t)a->Hq) + DSL * NTAP + KE4_U);
    // the four-utterance form is bf16-mode only.  It hard-codes the one-range tanh, which every other form -- and the backward's
    // recompute -- selects on memory16: it is taken only when memory16 is there too, so that a caller who sets Wq16 without
    // memory16 (the public struct allows it) cannot get a forward and a backward that disagree about tanh
    if (form == 2 && !(a->Wq16 && a->memory16 && a->loc_split_bf16 && lds_e4 <= 64 * 1024)) form = 1;
    // one launch (T2AMD_ATTN_FWD_FUSED=0/1, t2amd_set_attn_fwd_fused): the one-utterance form of K_e only, one position
    // per thread, and the granule block of ws present and 8-byte aligned
    static const bool fused_env = [] { const char* e = getenv("T2AMD_ATTN_FWD_FUSED"); return e ? e[0] != '0' : T2_ATTN_FWD_FUSED_DEFAULT != 0; }();
    static const int fwd_delay = [] { const char* e = getenv("T2AMD_ATTN_FWD_DELAY"); const int v = e ? atoi(e) : 8; return v < 0 ? 0 : (v > 100 ? 100 : v); }();
    p.gran_off = ((long long)NSL * a->B * a->Ti + 3) / 4 * 4;
    const size_t lds_ea = (lds_e + 15) / 16 * 16;
    if ((g_attn_fwd_fused < 0 ? fused_env : g_attn_fwd_fused != 0) && form == 0 && a->Ti <= KC_NT &&
        a->ws_floats >= p.gran_off + 2ll * NSL * a->B * a->Ti && (reinterpret_cast<uintptr_t>(a->ws + p.gran_off) & 7u) == 0 &&
        lds_ea + lds_c <= 64 * 1024) {
        if (++g_attn_fwd_token == 0) ++g_attn_fwd_token;
        p.token = g_attn_fwd_token;
        p.kc_smem_off = (int)(lds_ea / sizeof(float));
        p.delay = fwd_delay;
        if (a->memory16) T2_LAUNCH_ROLE(5, attn_fwd_fused_kernel<true>, dim3(NSL, a->B), dim3(KE_NT), lds_ea + lds_c, s, p);
        else T2_LAUNCH_ROLE(5, attn_fwd_fused_kernel<false>, dim3(NSL, a->B), dim3(KE_NT), lds_ea + lds_c, s, p);
        T2_LAUNCH_CHECK();
        return T2AMD_OK;
    }
    if (form == 2) T2_LAUNCH(attn_energy4_kernel, dim3(NSL, t2_cdiv(a->B, KE4_U)), dim3(KE_NT), lds_e4, s, p);
    else if (form == 1) {
        if (a->memory16) T2_LAUNCH((attn_energy_kernel<4, true>), dim3(NSL, a->B), dim3(KE_NT), lds_e, s, p);
        else T2_LAUNCH((attn_energy_kernel<4, false>), dim3(NSL, a->B), dim3(KE_NT), lds_e, s, p);
    } else {
        if (a->memory16) T2_LAUNCH((attn_energy_kernel<2, true>), dim3(NSL, a->B), dim3(KE_NT), lds_e, s, p);
        else T2_LAUNCH((attn_energy_kernel<2, false>), dim3(NSL, a->B), dim3(KE_NT), lds_e, s, p);
    }
    if (a->memory16) T2_LAUNCH(attn_context_kernel<true>, dim3(NCS, a->B), dim3(KC_NT), lds_c, s, p);
    else T2_LAUNCH(attn_context_kernel<false>, dim3(NCS, a->B), dim3(KC_NT), lds_c, s, p);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// =========================================================================================
// The teacher-forced decoder loop, forward, as ONE persistent launch (round 4; BASELINE north_star: "the decode loop is fused
// into a persistent wavefront-resident kernel"; reference model.py:405-411 around Decoder.decode :340-379).
//
// The launch chain runs a time step as two dependent launches -- the fused LSTM pair (skinny_wide_kernel: LSTM_a(t) beside
// LSTM_d(t-1)) and the one-launch attention step (attn_fwd_fused_kernel) -- 2 x To kernel boundaries of ~2.95 us each on an
// MI355X (trivial 256-workgroup kernels, tools/microbench_edge_flagdata.py).  Here the same two bodies, unchanged in their
// arithmetic (skinny_wide_body, ke_phase / kc_finish: device functions shared with those kernels), alternate inside one
// launch of max(Ha/8 + Hd/8, 4 B) co-resident 512-thread workgroups, one per CU, and the two all-to-all edges of a step are
// FLAG + DATA hand-offs (Guideline 16 R1; measured 2.1 us per edge in this geometry, profiles/r04_microbench_edge_flagdata.json):
//   workgroup j < Ha/8       LSTM_a tile j;  Ha/8 <= j < Ha/8 + Hd/8   LSTM_d tile;   j < 4 B   attention workgroup (b, s) = (j/4, j%4)
//   L(t):  LSTM_a(t) | LSTM_d(t-2): activations by sc1 LDS-DMA.  The wait for the attention of step t-1 (every workgroup's
//          flagT >= t) sits INSIDE the tile (DtpGate): weights and the ungated activation segments stream first, only the
//          segments that need step t-1 -- ctx(t-1) for LSTM_a, h_dec(t-3) for LSTM_d -- are issued behind it.  h and its bf16
//          copy out as write-through stores -> every wave drains -> LSTM_a tiles: flagA[j] = t+1
//   T(t):  K_e prologue; right before its one load of h: wait until every LSTM_a tile has finished step t (flagA >= t+1) ->
//          energy granules among the utterance's four workgroups, K_c: weights / cumulative weights / bf16 context out
//          write-through -> drain -> flagT[j] = t+1.  A workgroup WITHOUT an attention role (B < 64) observes flagA the same
//          way and raises its flagT[j] too: flagT has one counter per workgroup, so "all flagT >= t+1" says that every tile of
//          L(t) -- LSTM_d's included, which publish nothing of their own -- is complete, and every workgroup has observed it.
// Every exchanged tensor is written once per step with sc1 stores and read with sc1 loads / sc1 DMA; nothing else in the
// arithmetic differs from the chain: outputs are BIT-IDENTICAL to it.
// Spins are bounded (wall clock); a give-up sets *status and every workgroup leaves at its next wait.
// =========================================================================================
#ifndef T2AMD_DTP_LAG
#define T2AMD_DTP_LAG 1          // steps the decoder LSTM trails the attention LSTM by inside the persistent launch (1 or 2; A/B builds)
#endif
constexpr int DTP_LAG = T2AMD_DTP_LAG;
struct DecTrainPersist {
    t2amd_dec_train d;
    int tip, kc_smem_off, delay_a, delay_t, fail_off;
    int att_off;               // byte offset of the attention phase's LDS region: 0 (aliases the tile ring) or behind the ring
    int prefetch;              // 1: the next step's first four k-tiles are fetched from inside the attention phase (needs att_off > 0)
    int timing_no_d;           // tools only: skip the decoder-LSTM tiles (timing ceiling, results are garbage)
    unsigned token0;
    long long gran_off, ws_floats;
    unsigned* flagA;           // [Ha/8]           LSTM_a tile j has finished step t: t + 1
    unsigned* flagT;           // [workgroups]     workgroup j has finished L(t) and, if it has one, its attention step t: t + 1
    int* status;
    long long timeout_ticks, census_ticks;
    unsigned long long* ts;
    unsigned long long* prof;   // tools only (t2amd_debug_dtp_prof_): [2 workgroups][4] accumulated wall-clock ticks -- wait for
                                // the attention flags, LSTM tile, wait for the LSTM flags, attention step -- of workgroup 0
                                // (LSTM_a tile 0 + attention (0, 0)) and of workgroup Ha/8 (LSTM_d tile 0)
};

typedef unsigned dtp_u32x4 __attribute__((ext_vector_type(4)));
// one wave: all n flags >= target?  (16 bytes per lane when the count allows, else one word per lane and pass)
__device__ __forceinline__ bool dtp_wait(const unsigned* flags, const int n, const unsigned target, const int delay, int* status,
                                         const long long ticks, const int lane) {
    for (int d_ = 0; d_ < delay; ++d_) __builtin_amdgcn_s_sleep(1);
    const long long t0 = wall_clock64();
    unsigned spins = 0;
    const bool vec = (n & 3) == 0 && n <= 256;
    for (;;) {
        bool ok = true;
        if (vec) {
            if (4 * lane < n) {
                dtp_u32x4 f;
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(f) : "v"(flags + 4 * lane) : "memory");
                ok = f.x >= target && f.y >= target && f.z >= target && f.w >= target;
            }
        } else {
            for (int i = lane; i < n; i += 64) ok = ok && __hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
        }
        if (__all(ok)) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 63u) == 0) {
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;     // somebody gave up
            if (wall_clock64() - t0 > ticks) { if (lane == 0) atomicExch(status, 1); return false; }
        }
    }
}

// The tile's gate (skinny_wide.h): "every attention workgroup has finished step t-1", polled once at the tile's entry and, only if
// that poll did not see every flag yet, again (blocking) where the tile first needs ctx.
struct DtpGate {
    static constexpr bool on = true;
    const unsigned* flags; int n; unsigned target; int delay; int* status; long long ticks; int* fail_s;
    unsigned f0, f1, f2, f3;
    bool early_ok;
    __device__ __forceinline__ void early_issue(const int wave, const int lane) {
        f0 = f1 = f2 = f3 = 0xffffffffu;
        early_ok = false;
        if (wave == 0 && n <= 256) {
            const int i = 4 * lane;
            if (i < n) f0 = __hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (i + 1 < n) f1 = __hip_atomic_load(flags + i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (i + 2 < n) f2 = __hip_atomic_load(flags + i + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (i + 3 < n) f3 = __hip_atomic_load(flags + i + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __device__ __forceinline__ void early_check(const int wave, const int) {
        if (wave == 0 && n <= 256) early_ok = __all(f0 >= target && f1 >= target && f2 >= target && f3 >= target);
    }
    __device__ __forceinline__ void wait(const int wave, const int lane) {
        if (wave == 0 && !early_ok && !dtp_wait(flags, n, target, delay, status, ticks, lane) && lane == 0) fail_s[0] = 1;
        __syncthreads();
    }
};

// OM = operand mode of the LSTM tiles (t2amd_dec_train.bf16): 0 the f32 slabs themselves, 1 bf16 copies, 3 split-bf16 images
// (round 6, 'bf16x3': HA16 / HD16 / CTX16 / W*16 hold 4 bytes per k, strides in k as for f32)
template <int OM>
__device__ __forceinline__ void dtp_fill_a(const t2amd_dec_train& d, const int t, SkinnyParams& a) {
    constexpr bool BF = OM == 1;
    constexpr bool X3 = OM == 3;
    constexpr int US = X3 ? 2 : 1;                 // bf16 units per k in the operand copies
    // attention LSTM of step t: gates = GA[t] + [ctx_{t-1} | h_att_{t-1}] . Wa_rec^T   (loops.hip fill_a; BF: bf16 operand copies,
    // else the f32 slabs themselves)
    const long long sHa = (long long)d.B * d.Ha, sE = (long long)d.B * d.E;
    const unsigned short* c16 = (const unsigned short*)d.CTX16;
    const unsigned short* h16 = (const unsigned short*)d.HA16;
    a = SkinnyParams{};
    a.nseg = 2;
    // visited [h_att | ctx] over Wa_rec = [ctx columns | h_att columns] (explicit weight columns; the chain asks the per-step kernel
    // for the same order, loops.hip): h_att(t-1) has been complete since this workgroup's own attention phase of step t-1, so its
    // eight k-tiles run while the slowest attention workgroup is still producing ctx(t-1) -- the gate sits in front of segment 1.
    if constexpr (BF || X3) {
        a.x[0].p = t ? (const float*)(h16 + (t - 1) * sHa * US) : nullptr;
        a.x[1].p = t ? (const float*)(c16 + (t - 1) * sE * US) : nullptr;
        a.W = (const float*)d.Wa_rec16;
    } else {
        a.x[0].p = t ? d.HA + (t - 1) * sHa : nullptr;
        a.x[1].p = t ? d.CTX + (t - 1) * sE : nullptr;
        a.W = d.Wa_rec;
    }
    a.x[0].ld = d.Ha; a.x[0].width = d.Ha;
    a.x[1].ld = d.E; a.x[1].width = d.E;
    a.x[2].p = nullptr; a.x[2].ld = 0; a.x[2].width = 0;
    a.wcol[0] = d.E; a.wcol[1] = 0; a.wcol[2] = d.E + d.Ha;
    a.gate_seg = 1;
    a.Ktot = d.E + d.Ha; a.H = d.Ha; a.B = d.B; a.N = 4 * d.Ha;
    a.gin = d.GA + (long long)t * d.B * 4 * d.Ha; a.ld_gin = 4 * d.Ha;
    a.c_prev = t ? d.CA + (t - 1) * sHa : nullptr; a.ld_cprev = d.Ha;
    a.gates_out = d.GA + (long long)t * d.B * 4 * d.Ha; a.ld_gates = 4 * d.Ha;
    a.c_out = d.CA + t * sHa; a.ld_c = d.Ha;
    a.h_out = d.HA + t * sHa; a.ld_h = d.Ha;
    if constexpr (BF || X3) { a.h16_out = (unsigned short*)d.HA16 + t * sHa * US; a.ld_h16 = d.Ha; }
    a.keep = d.keep_att ? d.keep_att + t * sHa : nullptr; a.ld_keep = d.Ha; a.keep_scale = d.scale_att;
    a.gx = d.Ha / 8; a.gy = 1; a.gz = 1;
}
template <int OM>
__device__ __forceinline__ void dtp_fill_d(const t2amd_dec_train& d, const int u, SkinnyParams& a) {
    constexpr bool BF = OM == 1;
    constexpr bool X3 = OM == 3;
    constexpr int US = X3 ? 2 : 1;
    // decoder LSTM of step u: gates = bias_d + [h_att_u | ctx_u | h_dec_{u-1}] . Wd_cat^T   (loops.hip fill_d, bf16 operands)
    const long long sHa = (long long)d.B * d.Ha, sHd = (long long)d.B * d.Hd, sE = (long long)d.B * d.E;
    const unsigned short* c16 = (const unsigned short*)d.CTX16;
    const unsigned short* ha16 = (const unsigned short*)d.HA16;
    unsigned short* hd16 = (unsigned short*)d.HD16;
    a = SkinnyParams{};
    a.nseg = 3;
    if constexpr (BF || X3) {
        a.x[0].p = (const float*)(ha16 + u * sHa * US);
        a.x[1].p = (const float*)(c16 + u * sE * US);
        a.x[2].p = u ? (const float*)(hd16 + (u - 1) * sHd * US) : nullptr;
        a.W = (const float*)d.Wd_cat16;
    } else {
        a.x[0].p = d.HA + u * sHa;
        a.x[1].p = d.CTX + u * sE;
        a.x[2].p = u ? d.HD + (u - 1) * sHd : nullptr;
        a.W = d.Wd_cat;
    }
    a.x[0].ld = d.Ha; a.x[0].width = d.Ha;
    a.x[1].ld = d.E; a.x[1].width = d.E;
    a.x[2].ld = d.Hd; a.x[2].width = d.Hd;
    a.Ktot = d.Ha + d.E + d.Hd; a.H = d.Hd; a.B = d.B; a.N = 4 * d.Hd;
    a.bias = d.bias_d;
    a.c_prev = u ? d.CD + (u - 1) * sHd : nullptr; a.ld_cprev = d.Hd;
    a.gates_out = d.GD + (long long)u * d.B * 4 * d.Hd; a.ld_gates = 4 * d.Hd;
    a.c_out = d.CD + u * sHd; a.ld_c = d.Hd;
    a.h_out = d.HD + u * sHd; a.ld_h = d.Hd;
    if constexpr (BF || X3) { a.h16_out = hd16 + u * sHd * US; a.ld_h16 = d.Hd; }
    a.keep = d.keep_dec ? d.keep_dec + u * sHd : nullptr; a.ld_keep = d.Hd; a.keep_scale = d.scale_dec;
    a.gx = d.Hd / 8; a.gy = 1; a.gz = 1;
    a.wcol[0] = -1;               // stored order [h_att | ctx | h_dec]: h_att(u) first, what step u's attention made behind the gate
    a.gate_seg = DTP_LAG == 2 ? 2 : 1;
}

// The loop description is read from the kernel-argument segment INSIDE every iteration, through a pointer the compiler cannot
// see through: as an ordinary by-value argument every field (and every address derived from one: LICM) is loaded / computed
// once at entry and carried across the whole loop -- 256 VGPRs, 263 SGPR + 119 VGPR spills, 480 B of scratch per lane in the
// first build.  Scalar loads from the constant cache once per time step cost nothing next to that.
// Census of a persistent launch: every workgroup must be RESIDENT at once (they wait for each other).  A workgroup that is not --
// CUs held by another process or kernel -- would only be dispatched when a resident one exits, which never happens while
// they spin: found out HERE, within 2 ms and before anything is written, instead of at the first hand-off after 50 ms.
// All threads call; ends with a barrier; fail_s[0] (zeroed and published by the caller) != 0 afterwards = leave.
__device__ __forceinline__ void persist_census(unsigned* const census, int* const status, const long long census_ticks, int* const fail_s,
                                               const int wave, const int lane) {
    if (wave == 0) {
        if (lane == 0) atomicAdd(census, 1u);
        const long long t0 = wall_clock64();
        unsigned spins = 0;
        bool bad = false;
        if (census_ticks <= 0) {                    // tests: a forced give-up (T2AMD_DTP_TIMEOUT_TICKS=0), whatever the timing
            if (lane == 0) atomicCAS(status, 0, 3);
            bad = true;
        }
        while (!bad && __hip_atomic_load(census, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            __builtin_amdgcn_s_sleep(4);
            if ((++spins & 15u) == 0) {
                if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { bad = true; break; }
                if (wall_clock64() - t0 > census_ticks) { if (lane == 0) atomicCAS(status, 0, 3); bad = true; break; }
            }
        }
        if (bad && lane == 0) fail_s[0] = 1;
    }
    __syncthreads();
}

__device__ __forceinline__ const DecTrainPersist& dtp_args_late() {
    typedef const char __attribute__((address_space(4))) * kptr_t;
    kptr_t k = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return *reinterpret_cast<const DecTrainPersist*>((const char*)k);
}

// BF: the bf16 compute mode (bf16 copies of the recurrent operands and weights on the bf16 MFMA, bf16 attention streams); else the
// fp32 parity mode (round 5): the same loop over the f32 slabs themselves -- tiles on the exact-f32 MFMA (skinny_wide.h, F32), the
// attention phase in its f32 instantiation -- bit-identical to the fp32 launch chain, which runs the same tile.
// OM (round 6; was `bool BF`): 1 = bf16 mode, 0 = fp32 parity mode, 3 = 'bf16x3': the fp32 mode's loop -- f32 slabs, the attention phase in
// its exact-f32 instantiation -- with the LSTM tiles on SPLIT-bf16 operand images (skinny_wide.h SW_X3: hi.hi + lo.hi + hi.lo on the bf16
// MFMA, f32-class products at 3/16 of the exact-f32 matrix time); the tile epilogue and K_c write h / ctx as split images next to the
// f32 values.  Bit-identical to ITS launch chain (loops.hip, bf16 == 3), like the other two.
template <int OM>
__global__ __launch_bounds__(512) void dec_train_fwd_persistent_kernel(DecTrainPersist P_entry) {
    constexpr bool BF = OM == 1;
    constexpr bool X3 = OM == 3;
    constexpr int SWM = BF ? SW_BF16 : (X3 ? SW_X3 : SW_F32);
    extern __shared__ __attribute__((aligned(16))) char psmem_[];
    const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int To = P_entry.d.To;
    bool ts_on = false;
    if (tid == 0) reinterpret_cast<int*>(psmem_ + P_entry.fail_off)[0] = 0;
    __syncthreads();
    persist_census(P_entry.flagT + gridDim.x, P_entry.status, P_entry.census_ticks, reinterpret_cast<int*>(psmem_ + P_entry.fail_off), wave, lane);
    if (reinterpret_cast<int*>(psmem_ + P_entry.fail_off)[0]) return;
    // Step t runs  L(t): LSTM_a(t) [t < To]  |  LSTM_d(t - DTP_LAG) [t >= DTP_LAG]   then   T(t): the attention step [t < To].
    // DTP_LAG = 1 is the chain's pairing.  DTP_LAG = 2 (A/B build) lets the decoder LSTM trail by two steps: h_att(t-2) and ctx(t-2)
    // are then complete before L(t) begins and only its own recurrence h_dec(t-3), eight of twenty k-tiles, sits behind the tile's
    // gate instead of twelve.  Measured on one box it is the SLOWER form (23.32 vs 23.02 ms of forward, profiles/r04_j_lag_ab.txt):
    // the tiles were not waiting on the gate for long in the first place, and the older operands are colder.
    // (round 5) ring slots 0..3 of this workgroup hold the first four k-tiles of its NEXT tile: fetched by skinny_wide_prefetch4 from
    // inside the attention phase of the step before, drained and published by that phase's closing wait + barrier
    bool pref = false;
    for (int t = 0; t <= To + DTP_LAG - 1; ++t) {
        const DecTrainPersist& P = dtp_args_late();
        const t2amd_dec_train& d = P.d;
        int zero = 0;
        asm volatile("" : "+s"(zero));                   // LDS addresses are formed per iteration too
        char* const psmem = psmem_ + zero;
        float* const smem = reinterpret_cast<float*>(psmem + P.att_off);   // the attention phase's region
        int* const fail_s = reinterpret_cast<int*>(psmem + P.fail_off);      // behind both phases' regions
        const int nA = d.Ha / 8, nL = nA + d.Hd / 8, nT = NSL * d.B, nG = (int)gridDim.x;
        const bool isA = j < nA, isD = j >= nA && j < nL, isT = j < nT;
        const int b = j / NSL, sl = j % NSL;
        const bool prof_on = P.prof != nullptr && tid == 0 && (j == 0 || j == nA);
        unsigned long long* const prof = P.prof + (j == 0 ? 0 : 4);
        unsigned long long c0 = prof_on ? wall_clock64() : 0ull, c1;
#define DTP_PROF(slot) do { if (prof_on) { c1 = wall_clock64(); prof[slot] += c1 - c0; c0 = c1; } } while (0)
        // ---------------- L(t) ----------------
        // (P.timing_no_d -- tools only, T2AMD_DTP_TIMING_NO_D=1, NOT legal: h_dec is never produced -- runs the L phase with the
        // attention LSTM alone: the ceiling of "take the decoder LSTM off the loop's critical path", VERDICT r04 item 2.  A run-time
        // flag: as a compile-time variant the simplified role selection crashes hipcc's SimplifyCFG.)
        const bool tile = (isA && t < To) || (isD && t >= DTP_LAG && !P.timing_no_d);
        const bool tail = t >= To;                       // the two trailing iterations: no attention phase around them any more
        if (DTP_LAG > 1 && tile && tail) {
            // (nothing ran between the previous tiles and these that would have observed their inputs: wait in front of the tile)
            if (wave == 0 && !dtp_wait(P.flagT, nG, (unsigned)t, P.delay_a, P.status, P.timeout_ticks, lane) && lane == 0) fail_s[0] = 1;
            __syncthreads();
            if (fail_s[0]) return;
        }
        DTP_PROF(0);
        if (tile) {
            // ONE call site for both roles (two inlined copies of the tile body in sibling branches crash hipcc's SimplifyCFG)
            SkinnyParams sp;
            if (isA) dtp_fill_a<OM>(d, t, sp);
            else dtp_fill_d<OM>(d, t - DTP_LAG, sp);
            DtpGate gate;
            gate.flags = P.flagT; gate.n = nG; gate.target = (unsigned)t; gate.delay = P.delay_a; gate.status = P.status;
            gate.ticks = P.timeout_ticks; gate.fail_s = fail_s;
            if (t == 0 || (DTP_LAG > 1 && tail)) sp.gate_seg = 0;    // nothing to wait for at the first step; already waited in the tail
            skinny_wide_body<true, true, DtpGate, false, SWM>(sp, isA ? j : j - nA, psmem, P.ts, gate, pref);
            if (fail_s[0]) return;
        }
        pref = false;
        // the next step's tile of this workgroup, if it has one and its first segment can be fetched ahead (h_att(t): complete
        // once the LSTM flags of step t have been seen, i.e. from the middle of the attention phase below)
        const bool tileA_next = isA && t + 1 < To, tileD_next = isD && t + 1 >= DTP_LAG && DTP_LAG == 1 && !P.timing_no_d;
        const bool pf = P.prefetch != 0 && (tileA_next || tileD_next) &&
                        skinny_wide_prefetch_ok(d.Ha, (tileA_next ? d.E + d.Ha : d.Ha + d.E + d.Hd) / (BF ? 128 : 64), BF ? 128 : 64);
        auto prefetch_next = [&] {
            // (the description is read from the kernel-argument segment HERE: nothing of it is carried through the attention phase)
            const t2amd_dec_train& d2 = dtp_args_late().d;
            const bool isA2 = j < d2.Ha / 8;
            const void* const x0 = (BF || X3) ? (const void*)((const unsigned short*)d2.HA16 + (long long)t * d2.B * d2.Ha * (X3 ? 2 : 1))
                                              : (const void*)(d2.HA + (long long)t * d2.B * d2.Ha);
            const void* const W0 = (BF || X3) ? (isA2 ? d2.Wa_rec16 : d2.Wd_cat16) : (const void*)(isA2 ? d2.Wa_rec : d2.Wd_cat);
            skinny_wide_prefetch4<!BF>(x0, d2.Ha, W0, isA2 ? d2.E + d2.Ha : d2.Ha + d2.E + d2.Hd, isA2 ? d2.Ha : d2.Hd,
                                       isA2 ? d2.E : 0, d2.B, isA2 ? j : j - d2.Ha / 8, psmem);
        };
        // every storing wave drains its write-through stores (R1), then ONE flag per LSTM_a tile
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (isA && t < To && tid == 0) __hip_atomic_store(P.flagA + j, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        DTP_PROF(1);
        if (tail) {
            // no attention step left: the counter the last decoder tiles wait for is raised right here
            if (t < To + DTP_LAG - 1 && tid == 0) __hip_atomic_store(P.flagT + j, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        // ---------------- T(t) ----------------
        if (isT) {
            DTP_PROF(2);
            const long long sHa = (long long)d.B * d.Ha, sE = (long long)d.B * d.E;
            AttnFwdParams ap;
            ap.a = t2amd_attn_fwd{};
            ap.a.B = d.B; ap.a.Ti = d.Ti; ap.a.E = d.E; ap.a.Hq = d.Ha;
            ap.a.h = d.HA + t * sHa; ap.a.ld_h = d.Ha;
            ap.a.Wq = d.Wq; ap.a.U = d.U; ap.a.v = d.v; ap.a.pm = d.pm; ap.a.memory = d.memory; ap.a.lens = d.lens;
            ap.a.ws = d.attn_ws; ap.a.ws_floats = P.ws_floats;
            ap.a.w_prev = t ? d.ALIGN + (long long)(t - 1) * d.Ti : nullptr; ap.a.ld_wprev = (long long)To * d.Ti;
            ap.a.cum = d.cum_work;
            ap.a.cum_save = d.CUM + (long long)t * d.B * d.Ti;
            ap.a.w_out = d.ALIGN + (long long)t * d.Ti; ap.a.ld_wout = (long long)To * d.Ti;
            ap.a.ctx_out = d.CTX + t * sE; ap.a.ld_ctx = d.E;
            ap.a.q_out = d.Q + (long long)t * d.B * AD; ap.a.ld_q = AD;
            if constexpr (BF) {
                ap.a.ctx16_out = (void*)((unsigned short*)d.CTX16 + t * sE); ap.a.ld_ctx16 = d.E;
                ap.a.loc_split_bf16 = 1; ap.a.memory16 = d.memory16; ap.a.Wq16 = d.Wq16;
            }
            if constexpr (X3) {       // the split image of the context for the tiles; the step itself stays f32-class: exact f32
                                      // everywhere but the location conv, which runs as the split-bf16 product (~2^-17 per term)
                ap.a.ctx16_out = (void*)((unsigned short*)d.CTX16 + t * sE * 2); ap.a.ld_ctx16 = d.E; ap.a.ctx16_x3 = 1;
                ap.a.loc_split_bf16 = 1;
            }
            ap.tip = P.tip; ap.dbg = 0; ap.ts = P.ts;
            ap.token = P.token0 + (unsigned)t; ap.gran_off = P.gran_off; ap.kc_smem_off = P.kc_smem_off; ap.delay = 0;
            KcPre<BF> r;
            float e_first[4] = {0.f, 0.f, 0.f, 0.f};
            // the wait for the LSTM_a tiles of this step sits INSIDE the prologue, right before the one load that needs them (h):
            // W_q, processed memory, U, v and the windows are on their way while the flags are polled.  (A give-up lets the phase
            // run on with whatever h holds -- status is set, the step is poisoned behind the launch -- and leaves right after it.)
            ke_phase<true, BF, true, BF>(ap, smem, sl, b, ts_on,
                                       [&] {
                                           if (wave == 0 && !dtp_wait(P.flagA, nA, (unsigned)(t + 1), P.delay_t, P.status, P.timeout_ticks, lane) && lane == 0) fail_s[0] = 1;
                                           __syncthreads();
                                       },
                                       [&] { kc_issue<BF, false, true>(ap, sl, b, r, e_first); });
            if (fail_s[0]) return;
            // Issued HERE: this workgroup's partial energies are on their way and it is about to wait for its partners' -- a wait
            // that ends in s_waitcnt vmcnt(0) anyway (loads return in order, and any LDS read the compiler can see is ordered
            // behind every pending LDS-DMA), so the twelve DMA instructions per wave delay nothing the phase was not waiting for.
            if (pf && P.prefetch == 1) { prefetch_next(); pref = true; }
            fwd_energy_granules(ap, b, r.len, e_first);
            kc_finish<BF, true, true>(ap, smem + P.kc_smem_off, sl, b, ts_on, r, e_first);
            // (T2AMD_DTP_PREFETCH=2, A/B runs: issued here instead, in front of the phase's closing drain)
            if (pf && P.prefetch == 2) { prefetch_next(); pref = true; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(P.flagT + j, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            DTP_PROF(3);
        } else {
            // No attention role (fewer utterances than tiles).  The tiles of the next step wait for EVERY workgroup's counter -- a
            // tile written by a workgroup without one would be consumed unsynchronised -- and every workgroup must have observed
            // what an attention workgroup observes (the LSTM_a tiles of this step) before its next tile reads their output.
            if (wave == 0 && !dtp_wait(P.flagA, nA, (unsigned)(t + 1), P.delay_t, P.status, P.timeout_ticks, lane) && lane == 0) fail_s[0] = 1;
            __syncthreads();
            if (fail_s[0]) return;
            if (tid == 0) __hip_atomic_store(P.flagT + j, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (pf) {
                prefetch_next();
                pref = true;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (nothing else to do until the next tile: land and publish them here)
                __syncthreads();
            }
        }
#undef DTP_PROF
    }
}

// A give-up of the launch above turns one word of the step's data into NaN (the launch's own status is not read back by
// the training step: no host sync in the loop) and is counted, like an abandoned attention hand-off.
__global__ void dec_train_persist_poison_kernel(const int* status, float* poison) {
    if (*status != 0) {
        t2_attn_gave_up();
        *poison = __builtin_nanf("");
    }
}

// tools only: 8 zeroed device uint64 that the next launches accumulate their phase clocks into (NULL switches it off)
static unsigned long long* g_dtp_prof = nullptr;
extern "C" int t2amd_debug_dtp_prof_(unsigned long long* buf) { g_dtp_prof = buf; return T2AMD_OK; }

extern "C" long long t2amd_decoder_train_fwd_persistent_flag_bytes(int B, int Ha) {
    // LSTM_a tile counters, one counter per workgroup of the launch (at most 1024: the grid must fit the device), the arrival census
    (void)B;
    return 4ll * (Ha / 8 + 1024 + 1);
}

static int dtp_geometry(const t2amd_dec_train* p, size_t* lds_out, int* tip_out, int* kc_off_out, int* fail_off_out, int* att_off_out, int* pf_where_out) {
    T2_REQUIRE(p != nullptr, "dec_train_fwd_persistent: null args");
    T2_REQUIRE(p->bf16 == 0 || p->bf16 == 1 || p->bf16 == 3, "dec_train_fwd_persistent: bf16 must be 0 (fp32), 1 (bf16) or 3 (split-bf16 x3)");
    T2_REQUIRE(p->bf16 != 1 || (p->Wa_rec16 && p->Wd_cat16 && p->HA16 && p->HD16 && p->CTX16 && p->memory16 && p->Wq16),
               "dec_train_fwd_persistent: the bf16 operand mode needs the bf16 copies of the weights, the recurrent slabs, the memory and W_q");
    T2_REQUIRE(p->bf16 != 3 || (p->Wa_rec16 && p->Wd_cat16 && p->HA16 && p->HD16 && p->CTX16),
               "dec_train_fwd_persistent: the split-bf16 x3 mode needs the split images of the weights and of the recurrent slabs");
    {
        static const bool f32_env = [] { const char* e = getenv("T2AMD_TRAIN_FWD_PERSISTENT_FP32"); return !(e && e[0] == '0'); }();
        T2_REQUIRE(p->bf16 == 1 || f32_env, "dec_train_fwd_persistent: the fp32 / bf16x3 forms are switched off (T2AMD_TRAIN_FWD_PERSISTENT_FP32=0)");
    }
    T2_REQUIRE(p->B > 0 && p->B <= SK_ROWS, "dec_train_fwd_persistent: one 64-row tile (B <= 64)");
    T2_REQUIRE(p->Ti > 0 && p->Ti <= KC_NT && p->To > 0, "dec_train_fwd_persistent: one position per thread (Ti <= 512)");
    T2_REQUIRE(p->E % 128 == 0 && p->Ha % 128 == 0 && p->Hd % 128 == 0 && p->Ha <= 2048, "dec_train_fwd_persistent: E, Ha, Hd multiples of 128, Ha <= 2048");
    const int tip = attn_tip(p->Ti);
    const size_t lds_e = sizeof(float) * (2 * (size_t)tip + DSL + DSL * NTAP + (size_t)p->Ha);
    const int EC = p->E / NCS;
    T2_REQUIRE(EC % 8 == 0, "dec_train_fwd_persistent: E a multiple of 32");
    int parts = KC_NT / (EC / (p->bf16 == 1 ? 8 : 4));
    if (parts > 32) parts = 32;
    T2_REQUIRE(parts >= 1, "dec_train_fwd_persistent: E too large");
    const size_t lds_c = sizeof(float) * ((size_t)((p->Ti + 3) & ~3) + 16 + (size_t)parts * EC);
    const size_t lds_ea = (lds_e + 15) / 16 * 16;
    T2_REQUIRE(lds_ea + lds_c <= 64 * 1024, "dec_train_fwd_persistent: Ti too large for the LDS windows");
    const size_t ring = (size_t)SW_NBUF * (SW_XB + SW_WB);
    size_t lds = ring;
    if (lds_ea + lds_c > lds) lds = lds_ea + lds_c;
    lds = (lds + 15) / 16 * 16;
    *att_off_out = 0;
    // (round 5) when both fit, the attention phase gets its OWN region behind the ring, so that the ring can take the next step's
    // first tiles while the attention phase runs (skinny_wide_prefetch4): 96 KB + ~31 KB at Ti = 177.  T2AMD_DTP_PREFETCH=0: A/B runs.
    const char* const pf_e = getenv("T2AMD_DTP_PREFETCH");      // (read per call: tools A/B it within one process)
    const bool pf_env = !(pf_e && pf_e[0] == '0');
    const int pf_where = (pf_e && pf_e[0] == '2') ? 2 : 1;
    *pf_where_out = pf_where;
    const size_t both = ring + (lds_ea + lds_c + 15) / 16 * 16;
    if (pf_env && DTP_LAG == 1 && both + 16 <= 160 * 1024) {
        *att_off_out = (int)ring;
        lds = both;
    }
    *fail_off_out = (int)lds;
    *lds_out = lds + 16;
    *tip_out = tip;
    *kc_off_out = (int)(lds_ea / sizeof(float));
    return T2AMD_OK;
}

// 0 = this loop can run as one persistent launch on a device with `cus` compute units; else T2AMD_ERR_ARG + reason
extern "C" int t2amd_decoder_train_fwd_persistent_supported(const t2amd_dec_train* p, int cus) {
    size_t lds; int tip, kc, fo, ao, pw;
    T2_PROPAGATE(dtp_geometry(p, &lds, &tip, &kc, &fo, &ao, &pw));
    const int nL = p->Ha / 8 + p->Hd / 8, nT = NSL * p->B;
    const int grid = nL > nT ? nL : nT;
    // one workgroup per CU (96 KB of LDS each): every one of them must be resident at once
    T2_REQUIRE(grid <= cus && grid <= 1024, "dec_train_fwd_persistent: more workgroups than compute units (they must all be co-resident)");
    return T2AMD_OK;
}

extern "C" int t2amd_decoder_train_fwd_persistent_f32(const t2amd_dec_train* p, unsigned* flags, int* status, float* poison,
                                                      void* stream) {
    size_t lds; int tip, kc, fo, ao, pw;
    T2_PROPAGATE(dtp_geometry(p, &lds, &tip, &kc, &fo, &ao, &pw));
    T2_REQUIRE(flags && status, "dec_train_fwd_persistent: null flags / status");
    T2_REQUIRE(p->Wa_rec && p->Wd_cat && p->bias_d && p->Wq && p->U && p->v && p->GA && p->memory && p->pm && p->lens &&
                   p->HA && p->CA && p->GD && p->HD && p->CD && p->CTX && p->Q && p->ALIGN && p->CUM && p->cum_work && p->attn_ws,
               "dec_train_fwd_persistent: null pointer");
    const int B = p->B, Ti = p->Ti;
    const long long fwd_ws_floats = t2amd_attn_fwd_ws_floats(B, Ti);
    DecTrainPersist P;
    P.d = *p;
    P.tip = tip; P.kc_smem_off = kc; P.fail_off = fo;
    P.att_off = ao; P.prefetch = ao > 0 ? pw : 0;
    { const char* e = getenv("T2AMD_DTP_TIMING_NO_D"); P.timing_no_d = (e && e[0] == '1') ? 1 : 0; }
    // pre-poll pauses in s_sleep units (tuning knobs, read per call so that a tool can sweep them in one process)
    { const char* e = getenv("T2AMD_DTP_DELAY_L"); const int v = e ? atoi(e) : 4; P.delay_a = v < 0 ? 0 : (v > 400 ? 400 : v); }
    // (the wait for the LSTM flags sits inside the attention prologue, behind ~70 KB of loads: they ARE its pause -- flat from 0 to 16
    // units, 23.05 / 23.10 / 23.12 ms of forward; 24.1 at 64)
    { const char* e = getenv("T2AMD_DTP_DELAY_T"); const int v = e ? atoi(e) : 8; P.delay_t = v < 0 ? 0 : (v > 400 ? 400 : v); }
    P.gran_off = ((long long)NSL * B * Ti + 3) / 4 * 4;
    P.ws_floats = fwd_ws_floats;
    T2_REQUIRE((reinterpret_cast<uintptr_t>(p->attn_ws + P.gran_off) & 7u) == 0, "dec_train_fwd_persistent: attn_ws must be 8-byte aligned");
    // one token per time step, never zero, never one the granule block (zeroed below) has seen
    if (g_attn_fwd_token > 0xf0000000u - (unsigned)p->To) g_attn_fwd_token = 0;
    P.token0 = g_attn_fwd_token + 1;
    g_attn_fwd_token += (unsigned)p->To;
    P.flagA = flags; P.flagT = flags + p->Ha / 8;
    P.status = status;
    const char* te = getenv("T2AMD_DTP_TIMEOUT_TICKS");
    P.timeout_ticks = te ? atoll(te) : 5000000ll;                   // 50 ms of the 100 MHz wall clock
    P.census_ticks = P.timeout_ticks < 200000ll ? P.timeout_ticks : 200000ll;      // 2 ms; 0 = give up at once (tests)
    if (P.timeout_ticks < 1) P.timeout_ticks = 1;
    P.ts = attn_ts_buffer();
    P.prof = g_dtp_prof;
    if (t2amd_validate_only_flag_()) return T2AMD_OK;
    hipStream_t s = (hipStream_t)stream;
    T2_PROPAGATE(t2amd_fill_f32(p->cum_work, (long long)B * Ti, 0.f, stream));
    T2_PROPAGATE(t2amd_fill_f32(p->attn_ws + P.gran_off, fwd_ws_floats - P.gran_off, 0.f, stream));
    if (hipMemsetAsync(flags, 0, (size_t)t2amd_decoder_train_fwd_persistent_flag_bytes(B, p->Ha), s) != hipSuccess ||
        hipMemsetAsync(status, 0, sizeof(int), s) != hipSuccess)
        T2_FAIL("dec_train_fwd_persistent: memset failed");
    static size_t lds_set[3] = {0, 0, 0};
    const int om = p->bf16 == 1 ? 1 : (p->bf16 == 3 ? 2 : 0);
    if (lds > lds_set[om]) {
        const void* fn = om == 1 ? (const void*)dec_train_fwd_persistent_kernel<1>
                                 : (om == 2 ? (const void*)dec_train_fwd_persistent_kernel<3> : (const void*)dec_train_fwd_persistent_kernel<0>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            T2_FAIL("dec_train_fwd_persistent: cannot raise the dynamic LDS limit");
        lds_set[om] = lds;
    }
    const int nL = p->Ha / 8 + p->Hd / 8, nT = NSL * B;
    // role 7 of bench.py's roofline leg: the whole forward loop of the step is this one launch
    if (om == 1) T2_LAUNCH_ROLE(7, dec_train_fwd_persistent_kernel<1>, dim3(nL > nT ? nL : nT), dim3(512), lds, s, P);
    else if (om == 2) T2_LAUNCH_ROLE(7, dec_train_fwd_persistent_kernel<3>, dim3(nL > nT ? nL : nT), dim3(512), lds, s, P);
    else T2_LAUNCH_ROLE(7, dec_train_fwd_persistent_kernel<0>, dim3(nL > nT ? nL : nT), dim3(512), lds, s, P);
    T2_LAUNCH_CHECK();
    if (poison) {
        hipLaunchKernelGGL(dec_train_persist_poison_kernel, dim3(1), dim3(1), 0, s, status, poison);
        T2_LAUNCH_CHECK();
    }
    return T2AMD_OK;
}

// =========================================================================================
// Backward of one attention step.
// =========================================================================================
struct AttnBwdParams {
    t2amd_attn_bwd a; int tip; int np; int dbg; unsigned long long* ts; unsigned token; int kb1_smem_off; int fused_delay;
    t2amd_lstm_bwd cq, cx;      // CELL form: the folded cells by value (cq: takes W_q^T dq; cx: the independent one)
    int cx_q4;                  // cx.H / 16 = float4 unit groups of cx per workgroup; 0 = no cx
    long long gran_off;         // GRAN: float offset of the granule block in a.ws (8-byte aligned address)
};

// K_b1: dctx, dw[ti] = dctx . memory[ti] + carries, partial sum_ti w dw over a quarter of the positions.
// Half a wave (32 lanes) per memory row, 8 rows per pass.  One L2 round trip: the memory rows of the first 64
// positions of the slice (all four column groups: 32 float4 per lane), the gradient slabs and the carry partials
// are all issued before anything is consumed; nothing is compared or selected on a loaded value before that.
#define KB1_NT 512      // threads: 16 row groups of 32 lanes
#define KB1_NG (KB1_NT / 32)
#define KB1_MAXP 4      // passes kept in registers: 16 rows x 4 passes = 64 positions per slice (Ti <= 256)
// M16: rows come from the bf16 copy of the encoder memory (t2amd_attn_bwd.memory16): one 16-byte load is 8 channels,
// so two column groups cover E <= 512 and the kernel moves half the bytes it is bound by; dctx and the sums stay f32.
// The body is a device function so that the fused backward kernel (below) can run it as its first phase; `ts_on` is
// the caller's phase-stamp switch.
// GRAN (one-launch form only): dw and the partial sum leave as 8-byte {launch token, f32} granules -- one write-through
// store each, the data is the flag (Guideline 16 R2, as in csrc/decode_persist.hip) -- into the granule block of ws:
// one row of Ti + NTS granules per utterance, dw[0 .. Ti) then the NTS partial sums (so that a consumer thread polls
// exactly one granule: thread i < Ti + NTS the i-th of its utterance's row).
// PERSIST (the persistent backward loop below): the gradient slabs and the carry partials were written by other workgroups of
// the SAME launch one time step ago -- device-scope (sc1) loads -- and `before_slabs` (the wait for that step's dgrad tiles) runs
// right in front of them, BEHIND the memory-row stream, which does not depend on it.
// `after_rows` (round 5): called behind the issue of every load this phase waits for (memory rows, gradient slabs, carries) -- the
// one-launch backward issues the d_pm read-modify-write operands of ITS tile loop there (64 KB per workgroup that nobody touches for
// the next ~7 us) instead of in front of this phase: loads return in order, and this phase's dw -- what the utterance's three other
// workgroups are waiting for -- used to queue behind them.
struct KbNoHook { __device__ __forceinline__ void operator()() const {} };
// a * b + c as ONE v_fma_f32 that the vectoriser cannot pair into a v_pk_fma_f32 (round 5).  The f32 form of
// K_b1 accumulated its rows with packed FMAs whose DESTINATION pair was also a source pair read with a cross-half op_sel
// (`v_pk_fma_f32 v[34:35], v[10:11], v[34:35], v[66:67] op_sel:[0,1,0]`: both results read v35, one of them overwrites it): correct
// alone on the GPU in every run of four rounds, WRONG in lanes 48..63 of a wave (one row in two, 16 lanes of its dot product) in
// 60 % of the launches as soon as another kernel's MFMA waves share the SIMD -- a second process, or a GEMM on a side stream
// (tools/stress_attn_bwd.py; this is what tests/test_zz9_dp_gpu.py saw once in round 2 and again in round 5).  The library is now
// built without packed-f32 instructions altogether (build.py); this helper keeps the one place where the fault was PROVEN
// unpacked even in a build that turns them back on.
__device__ __forceinline__ float t2_fma_unpacked(float a, float b, float c) {
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// What K_b1 has in flight between the issue of its loads and their first use (round 5: the two halves are separate functions, so
// that the one-launch backward can issue this phase's loads -- memory rows, gradient slabs, carries: what its dw, the thing the
// utterance's three other workgroups wait for, is made of -- as the FIRST loads of the launch, ahead of its own prologue).
template <bool M16>
struct Kb1Regs {
    float4 pm[M16 ? 2 : 4][KB1_MAXP];
    float gsl[2][3][4];
    float cwv[4], ccv[4], dcv, exv, wv0;
    int len_raw;
};
template <bool M16, bool PERSIST = false, class Hook = KbNoHook, class Hook2 = KbNoHook>
__device__ __forceinline__ void kb1_issue(const AttnBwdParams& p, const int ts, const int b, bool& ts_on, Kb1Regs<M16>& R,
                                          Hook before_slabs = Hook(), Hook2 after_rows = Hook2()) {
    constexpr int CPT = M16 ? 8 : 4;           // channels per 16-byte load
    constexpr int KB1_MAXC = M16 ? 2 : 4;      // column groups kept in registers: KB1_MAXC x 32 loads = E <= 512
    const t2amd_attn_bwd& a = p.a;
    const int tid = PERSIST ? t2_tid_opaque() : (int)threadIdx.x;
    const int Ti = a.Ti, E = a.E, B = a.B;
    const int tsz = (Ti + NTS - 1) / NTS;
    T2_TS(32);
    const int len_raw = a.lens ? a.lens[b] : Ti;
    R.len_raw = len_raw;
    const int t0 = ts * tsz;
    int t1 = t0 + tsz;
    if (t1 > Ti) t1 = Ti;

    const int E4 = E / CPT;                        // 16-byte units per row
    const float4* __restrict__ M4 = reinterpret_cast<const float4*>(M16 ? a.memory16 : (const void*)a.memory) +
                                    (long long)b * Ti * E4;
    const int grp = tid >> 5, l32 = tid & 31;      // KB1_NG row groups of 32 lanes
    const int npass = (tsz + KB1_NG - 1) / KB1_NG; // passes that hold positions of this slice
#pragma unroll
    for (int i = 0; i < KB1_MAXP; ++i) {
        const int ti = t0 + grp + KB1_NG * i;
        const long long tc = (i < npass && ti < t1 && ti < len_raw) ? ti : t0;      // clamped: loaded, never used
#pragma unroll
        for (int g = 0; g < KB1_MAXC; ++g) {
            const int c = l32 + 32 * g;
            R.pm[g][i] = M4[tc * E4 + (c < E4 ? c : l32)];
        }
    }
    before_slabs();
    // gradient of the context: up to three addends of up to four slabs each, two channels per thread at most
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = tid + KB1_NT * u;
        const int cc = c < E ? c : 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const t2amd_addend& ad = a.dctx[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) R.gsl[u][i][k] = 0.f;
            if (ad.p) {
                const int n = ad.nsplit;
                const float* q0 = ad.p + (long long)b * ad.ld;       // wave-uniform slab bases (scalar adds only)
                const float* q1 = q0 + ad.split_stride;
                const float* q2 = q1 + ad.split_stride;
                const float* q3 = q2 + ad.split_stride;
                R.gsl[u][i][0] = ld_xwg<PERSIST>(q0 + cc);
                if (n > 1) R.gsl[u][i][1] = ld_xwg<PERSIST>(q1 + cc);
                if (n > 2) R.gsl[u][i][2] = ld_xwg<PERSIST>(q2 + cc);
                if (n > 3) R.gsl[u][i][3] = ld_xwg<PERSIST>(q3 + cc);
            }
        }
    }
    // carries: one thread per position of the slice (first pass in registers)
    const long long ps = (long long)B * 2 * Ti;                 // stride between dim-slice partials
    const float* __restrict__ cw = a.dwin_part + ((long long)b * 2 + 0) * Ti;
    const float* __restrict__ cc_ = a.dwin_part + ((long long)b * 2 + 1) * Ti;
    const float* __restrict__ dcum = a.dcum_acc + (long long)b * Ti;
    R.exv = 0.f;
    {
        const int ti = (t0 + tid < t1) ? t0 + tid : t0;         // clamped
#pragma unroll
        for (int k = 0; k < 4; ++k) { R.cwv[k] = ld_xwg<PERSIST>(cw + k * ps + ti); R.ccv[k] = ld_xwg<PERSIST>(cc_ + k * ps + ti); }
        R.dcv = dcum[ti];
        if (a.d_w_extra) R.exv = a.d_w_extra[(long long)b * a.ld_dwextra + ti];
        R.wv0 = a.w[(long long)b * a.ld_w + ti];
    }
    after_rows();        // (behind EVERY load this phase waits for: rows, gradient slabs, carries)
}

template <bool M16, bool GRAN, bool PERSIST = false>
__device__ __forceinline__ void kb1_consume(const AttnBwdParams& p, float* smem, const int ts, const int b, bool& ts_on,
                                            const Kb1Regs<M16>& R) {
    constexpr int CPT = M16 ? 8 : 4;
    constexpr int KB1_MAXC = M16 ? 2 : 4;
    const t2amd_attn_bwd& a = p.a;
    const int tid = PERSIST ? t2_tid_opaque() : (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int Ti = a.Ti, E = a.E, B = a.B;
    const int tsz = (Ti + NTS - 1) / NTS;
    float* dctx_s = smem;            // [E]
    float* base_s = dctx_s + E;      // [tsz] carries + running dcum (+ extra) per position of the slice
    float* wl_s = base_s + tsz;      // [tsz] this step's weights
    float* red_s = wl_s + tsz;       // [KB1_NT / 64]
    const int len_raw = R.len_raw;
    const int t0 = ts * tsz;
    int t1 = t0 + tsz;
    if (t1 > Ti) t1 = Ti;
    const int E4 = E / CPT;
    const float4* __restrict__ M4 = reinterpret_cast<const float4*>(M16 ? a.memory16 : (const void*)a.memory) +
                                    (long long)b * Ti * E4;
    const int grp = tid >> 5, l32 = tid & 31;
    const float4 (&pm)[KB1_MAXC][KB1_MAXP] = R.pm;
    const float (&gsl)[2][3][4] = R.gsl;
    const long long ps = (long long)B * 2 * Ti;
    const float* __restrict__ cw = a.dwin_part + ((long long)b * 2 + 0) * Ti;
    const float* __restrict__ cc_ = a.dwin_part + ((long long)b * 2 + 1) * Ti;
    float* __restrict__ dcum = a.dcum_acc + (long long)b * Ti;
    const float (&cwv)[4] = R.cwv;
    const float (&ccv)[4] = R.ccv;
    const float dcv = R.dcv, exv = R.exv, wv0 = R.wv0;
    // ---- consume --------------------------------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = tid + KB1_NT * u;
        if (c < E) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const t2amd_addend& ad = a.dctx[i];
                if (ad.p) {
                    const int n = ad.nsplit;
                    s += gsl[u][i][0];
                    if (n > 1) s += gsl[u][i][1];
                    if (n > 2) s += gsl[u][i][2];
                    if (n > 3) s += gsl[u][i][3];
                    if (n > 4) {
                        const float* q = ad.p + (long long)b * ad.ld + c;
                        for (int k = 4; k < n; ++k) s += ld_xwg<PERSIST>(q + (long long)k * ad.split_stride);
                    }
                }
            }
            dctx_s[c] = s;
            if (ts == 0) a.dctx_total[(long long)b * a.ld_dctx_total + c] = s;
        }
    }
    for (int c = tid + 2 * KB1_NT; c < E; c += KB1_NT) {       // E > 1024: remaining channels the plain way
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const t2amd_addend& ad = a.dctx[i];
            if (ad.p) {
                const float* q = ad.p + (long long)b * ad.ld + c;
                for (int k = 0; k < ad.nsplit; ++k) s += ld_xwg<PERSIST>(q + (long long)k * ad.split_stride);
            }
        }
        dctx_s[c] = s;
        if (ts == 0) a.dctx_total[(long long)b * a.ld_dctx_total + c] = s;
    }
    for (int i = tid; i < tsz; i += KB1_NT) {
        const int ti = t0 + i;
        float base = 0.f, w = 0.f;
        if (ti < t1) {
            float carry_w, carry_c, dc0, ex = 0.f;
            if (i == tid) {
                carry_w = ((cwv[0] + cwv[1]) + cwv[2]) + cwv[3];
                carry_c = ((ccv[0] + ccv[1]) + ccv[2]) + ccv[3];
                dc0 = dcv; ex = exv; w = wv0;
            } else {
                carry_w = ((ld_xwg<PERSIST>(cw + ti) + ld_xwg<PERSIST>(cw + ps + ti)) + ld_xwg<PERSIST>(cw + 2 * ps + ti)) + ld_xwg<PERSIST>(cw + 3 * ps + ti);
                carry_c = ((ld_xwg<PERSIST>(cc_ + ti) + ld_xwg<PERSIST>(cc_ + ps + ti)) + ld_xwg<PERSIST>(cc_ + 2 * ps + ti)) + ld_xwg<PERSIST>(cc_ + 3 * ps + ti);
                dc0 = dcum[ti];
                if (a.d_w_extra) ex = a.d_w_extra[(long long)b * a.ld_dwextra + ti];
                w = a.w[(long long)b * a.ld_w + ti];
            }
            const float dc = dc0 + carry_c;
            dcum[ti] = dc;
            base = carry_w + dc;
            if (a.d_w_extra) base += ex;
        }
        base_s[i] = base;
        wl_s[i] = w;
    }
    const int len = len_raw;
    __syncthreads();
    float* __restrict__ dwo = a.ws + (long long)b * Ti;
    at_u64* __restrict__ gdw = reinterpret_cast<at_u64*>(a.ws + p.gran_off) + (long long)b * (Ti + NTS);
    T2_TS(33);
    float psum = 0.f;
    for (int r0 = 0; r0 < tsz; r0 += KB1_NG * KB1_MAXP) {
        float acc[KB1_MAXP];
#pragma unroll
        for (int i = 0; i < KB1_MAXP; ++i) acc[i] = 0.f;
        if (r0 == 0) {
#pragma unroll
            for (int g = 0; g < KB1_MAXC; ++g) {
                const int c = l32 + 32 * g;
                if (c < E4) {
                    if constexpr (M16) {
                        const float4 g0 = *reinterpret_cast<const float4*>(&dctx_s[c * 8]);
                        const float4 g1 = *reinterpret_cast<const float4*>(&dctx_s[c * 8 + 4]);
#pragma unroll
                        for (int i = 0; i < KB1_MAXP; ++i) acc[i] = dot8_bf16(pm[g][i], g0, g1, acc[i]);
                    } else {
                        const float4 gq = *reinterpret_cast<const float4*>(&dctx_s[c * 4]);
#pragma unroll
                        for (int i = 0; i < KB1_MAXP; ++i) {
                            acc[i] = t2_fma_unpacked(pm[g][i].x, gq.x, acc[i]);
                            acc[i] = t2_fma_unpacked(pm[g][i].y, gq.y, acc[i]);
                            acc[i] = t2_fma_unpacked(pm[g][i].z, gq.z, acc[i]);
                            acc[i] = t2_fma_unpacked(pm[g][i].w, gq.w, acc[i]);
                        }
                    }
                }
            }
        }
        // column groups / positions beyond the register-resident block (E > 512 or Ti > 256)
        for (int c = l32 + (r0 == 0 ? 32 * KB1_MAXC : 0); c < E4; c += 32) {
            const float4 gq = *reinterpret_cast<const float4*>(&dctx_s[c * CPT]);
            float4 gq1 = gq;
            if constexpr (M16) gq1 = *reinterpret_cast<const float4*>(&dctx_s[c * CPT + 4]);
#pragma unroll
            for (int i = 0; i < KB1_MAXP; ++i) {
                const int ti = t0 + r0 + grp + KB1_NG * i;
                const float4 m = M4[(long long)(ti < t1 ? ti : t0) * E4 + c];
                if constexpr (M16) {
                    acc[i] = dot8_bf16(m, gq, gq1, acc[i]);
                } else {
                    acc[i] = t2_fma_unpacked(m.x, gq.x, acc[i]);
                    acc[i] = t2_fma_unpacked(m.y, gq.y, acc[i]);
                    acc[i] = t2_fma_unpacked(m.z, gq.z, acc[i]);
                    acc[i] = t2_fma_unpacked(m.w, gq.w, acc[i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < KB1_MAXP; ++i) {
            float s = row16_sum(acc[i]);
            s += __shfl_xor(s, 16, 64);
            const int li = r0 + grp + KB1_NG * i;
            const int ti = t0 + li;
            if (l32 == 0 && ti < t1) {
                const float dw = ((ti < len) ? s : 0.f) + base_s[li];
                // device-scope (write-through) stores: what the fused kernel's hand-off publishes
                if constexpr (GRAN) gran_publish(gdw + ti, p.token, dw);
                else __hip_atomic_store(&dwo[ti], dw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                psum = fmaf(wl_s[li], dw, psum);
            }
        }
    }
    T2_TS(34);
    // psum lives in lanes 0 and 32 of every wave
    psum += __shfl_xor(psum, 32, 64);
    if (lane == 0) red_s[wv] = psum;
    __syncthreads();
    if (tid == 0) {
        float s = red_s[0];
#pragma unroll
        for (int w = 1; w < KB1_NT / 64; ++w) s += red_s[w];
        if constexpr (GRAN) gran_publish(reinterpret_cast<at_u64*>(a.ws + p.gran_off) + (long long)b * (Ti + NTS) + Ti + ts, p.token, s);
        else __hip_atomic_store(&a.ws[(long long)B * Ti + (long long)ts * B + b], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    T2_TS(35);
}

// the whole phase in one piece (the separate-launch kernel, the token-form one-launch kernel, the persistent backward loop)
template <bool M16, bool GRAN, bool PERSIST = false, class Hook = KbNoHook, class Hook2 = KbNoHook>
__device__ __forceinline__ void kb1_phase(const AttnBwdParams& p, float* smem, const int ts, const int b, bool& ts_on,
                                          Hook before_slabs = Hook(), Hook2 after_rows = Hook2()) {
    Kb1Regs<M16> R;
    kb1_issue<M16, PERSIST>(p, ts, b, ts_on, R, before_slabs, after_rows);
    kb1_consume<M16, GRAN, PERSIST>(p, smem, ts, b, ts_on, R);
}

template <bool M16>
__global__ __launch_bounds__(KB1_NT) void attn_bwd_dw_kernel(AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    bool ts_on = false;
    kb1_phase<M16, false>(p, smem, blockIdx.x, blockIdx.y, ts_on);
}

// The folded cells' descriptors (AttnBwdParams.cq / .cx, ~100 scalar fields) are read from the kernel-argument segment
// where they are used: as ordinary by-value arguments the compiler loads every field at kernel entry and carries it in
// SGPRs (spilled to VGPR lanes, then to scratch) through the whole kernel.  The empty asm makes the segment pointer
// opaque at the point of the call, so the scalar loads cannot be hoisted above it.
__device__ __forceinline__ const t2amd_lstm_bwd& kernarg_cell_late(size_t offset) {
    typedef const char __attribute__((address_space(4))) * kptr_t;
    kptr_t k = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return *reinterpret_cast<const t2amd_lstm_bwd*>((const char*)(k + offset));
}

// K_b2: everything that lives in attention-dim space, for 32 dims (8 waves)
#define KB2_NT 512
#define KB2_NW (KB2_NT / 64)
// FUSED: the same workgroup first runs K_b1's phase for position slice ds (its own LDS region behind K_b2's), then
// hands its dw slice to the three other workgroups of the utterance through memory -- release store of a per-launch
// token into ws, acquire spin on the four tokens -- and carries on as K_b2.  One launch instead of two, and K_b2's
// prologue loads (issued before the K_b1 phase) land behind it.  The four workgroups of an utterance are consecutive
// in dispatch order, so a waiting workgroup's partners are always resident or next to be dispatched.
// CELL (needs FUSED): the closing phase also runs the LSTM cell backwards of the BPTT step (t2amd_attn_bwd.cell_q /
// cell_x) -- one launch less per time step.  The cell of the attention LSTM needs the full W_q^T dq of its utterance,
// which the four dim-slice workgroups hold as four partial sums; instead of exchanging those (4 x Hq floats each) they
// exchange dq (32 floats each, through dq_out, behind a second token published before col2im and polled after it), and
// workgroup ds forms the product for columns [ds Hq/4, (ds+1) Hq/4) over all 128 dims: eight 64-thread groups take 16
// dims each -- exactly the 16-dim partial sums the dh_out slabs are made of -- and the closing reduction adds them in
// the order a separate cell launch adds the slabs, so every bit of the result is the same.
// The body is a device function: attn_bwd_main_kernel runs it once per launch (the cells' descriptors read late from the
// kernel-argument segment); the persistent backward loop (dec_train_bwd_persistent_kernel) runs it once per time step inside
// ONE launch, with PERSIST: what other workgroups of the same launch wrote one step ago (gradient slabs, carries: kb1_phase,
// addend_issue4) is read with device-scope loads behind `before_slabs`, what they read next (col2im carries, the bf16 gate
// gradients) leaves as write-through stores.  Same arithmetic in the same order: bit-identical results.
struct KernargCells {
    __device__ __forceinline__ const t2amd_lstm_bwd& cq() const { return kernarg_cell_late(offsetof(AttnBwdParams, cq)); }
    __device__ __forceinline__ const t2amd_lstm_bwd& cx() const { return kernarg_cell_late(offsetof(AttnBwdParams, cx)); }
};
template <bool FUSED, bool M16, bool CELL, bool GRAN, bool PERSIST, class Cells, class Hook>
__device__ __forceinline__ void attn_bwd_main_body(const AttnBwdParams& p, float* const smem, const int ds, const int b, bool& ts_on,
                                                   const Cells cells, Hook before_slabs) {
    static_assert(FUSED || !CELL, "the folded cells need the one-launch form");
    static_assert(FUSED || !GRAN, "granules are the one-launch form's first hand-off");
    static_assert(!PERSIST || (FUSED && CELL && GRAN), "the persistent loop runs the one-launch form with the folded cells");
    const t2amd_attn_bwd& a = p.a;
    const int tid = PERSIST ? t2_tid_opaque() : (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int Ti = a.Ti, Hq = a.Hq, B = a.B, TIP = p.tip, NP = p.np;
    float* win_s = smem;                      // [2][TIP]
    float* de_s = win_s + 2 * TIP;            // [NP]
    float* dcol_s = de_s + NP;                // [NP][DCL]
    float* dpre_s = dcol_s + DCOL_FLOATS(NP);   // [NP][DPL]
#ifndef T2AMD_DCOL_ROWMAJOR
    const int NPP = NP + 4;                     // row stride of the tap-major dcol image
#endif
    float* red_s = dpre_s + (size_t)NP * DPL;   // [NW][2][32]
    float* dq_s = red_s + KB2_NW * 2 * DSL;   // [32]
    float* u_s = dq_s + DSL;                  // [32][62]
    float* dh_s = u_s + DSL * NTAP;           // [Hq] second-half partial of dh; CELL: [8][Hq/4] 16-dim partials
    float* dqall_s = dh_s + 2 * (size_t)Hq;   // CELL only: [128] the utterance's dq
    T2_TS(48);
    // (round 5, measured and NOT adopted: -DT2AMD_BWD_KB1_FIRST builds it) the loads of the K_b1 phase -- memory rows, gradient slabs,
    // carries: what its dw, the thing the utterance's three other workgroups wait for, is made of -- as the FIRST loads of the launch
    // (kb1_issue here, kb1_consume where the phase used to run).  Phase stamps say they are issued ~2 us in (behind this prologue's
    // issue, the wait of the window / U staging for ITS loads and four serialised scalar round trips at the top of the phase), yet
    // four alternating pairs of builds read 57.08-57.33 vs 56.73-56.91 ms per step: slower -- everything the staging and the early
    // tanh work touch then queues behind 106 KB of rows and slabs.
#ifdef T2AMD_BWD_KB1_FIRST
    constexpr bool KB1_FIRST = FUSED && GRAN && !PERSIST;
#else
    constexpr bool KB1_FIRST = false;
#endif
    Kb1Regs<M16> kb1r;
    if constexpr (KB1_FIRST) kb1_issue<M16, false>(p, ds, b, ts_on, kb1r);
    // Prologue loads: all issued before the first is consumed, nothing selected on a fresh load (see K_e).
    const int len_raw = a.lens ? a.lens[b] : Ti;
    const int dbase = ds * DSL;

    const float* __restrict__ pmb = a.pm + (long long)b * Ti * AD + dbase + 4 * lg;
    float* __restrict__ dpmb = a.d_pm + (long long)b * Ti * AD + dbase + 4 * lg;
    const float dv_old = a.dv_acc[(long long)b * AD + dbase + (tid < DSL ? tid : 0)];   // read-modify-write operand, fetched early
    float4 pmA[2], pmB[2], opA[2], opB[2];      // consumed in the tile loop
    // One-launch granule form: the d_pm operands are issued from INSIDE the K_b1 phase, behind its memory rows (kb1_phase,
    // after_rows) -- nothing reads them before the tile loop, and in front of the phase they delayed its dw by the time the CU's
    // memory pipe needs for 64 KB.  The processed-memory rows stay here: the early tanh work between the hand-off's publication
    // and its poll consumes them.  (-DT2AMD_BWD_OPS_FIRST: everything in front, the round-4 order, A/B builds.)
#ifdef T2AMD_BWD_OPS_FIRST
    constexpr bool LATE_OPS = false;
#else
    constexpr bool LATE_OPS = FUSED && GRAN && !KB1_FIRST;      // (KB1_FIRST: everything here is behind K_b1's loads anyway)
#endif
    auto issue_dpm_operands = [&] {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            int pos = (wv + rr * KB2_NW) * 16 + l15;
            pos = pos < Ti ? pos : Ti - 1;          // clamped; rows past the utterance are never used / stored
            opA[rr] = *reinterpret_cast<const float4*>(dpmb + (long long)pos * AD);
            opB[rr] = *reinterpret_cast<const float4*>(dpmb + (long long)pos * AD + 16);
        }
    };
    auto issue_pm_rows = [&] {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            int pos = (wv + rr * KB2_NW) * 16 + l15;
            pos = pos < Ti ? pos : Ti - 1;
            pmA[rr] = *reinterpret_cast<const float4*>(pmb + (long long)pos * AD);
            pmB[rr] = *reinterpret_cast<const float4*>(pmb + (long long)pos * AD + 16);
        }
    };
#ifdef T2AMD_BWD_PM_LATE
    constexpr bool PM_LATE = LATE_OPS;       // A/B builds: the processed-memory rows behind K_b1's loads too
#else
    constexpr bool PM_LATE = false;
#endif
    if constexpr (!PM_LATE) issue_pm_rows();
    if constexpr (!LATE_OPS) issue_dpm_operands();
    float sdv[NTS], w_r, dw_r;
    {
        const int tc = tid < Ti ? tid : Ti - 1;
        w_r = a.w[(long long)b * a.ld_w + tc];
        if constexpr (!FUSED) {          // K_b1's outputs: produced inside this launch when FUSED (loaded after the hand-off)
            const float* sd = a.ws + (long long)B * Ti;
#pragma unroll
            for (int k = 0; k < NTS; ++k) sdv[k] = sd[k * B + b];
            dw_r = a.ws[(long long)b * Ti + tc];
        }
    }
    const float* wprev_b = a.w_prev ? a.w_prev + (long long)b * a.ld_wprev : nullptr;
    const float* cumb_b = a.cum_before + (long long)b * Ti;
    const WinRegs wreg = stage_windows_issue(TIP, Ti, wprev_b, cumb_b, tid);
    const float4 ureg = stage_u_issue(a.U + (long long)dbase * NTAP, tid);
    float vv[2][4], qv[2][4], dva[2][4], dqa[2][4];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = dbase + dt * 16 + 4 * lg + r;
            vv[dt][r] = a.v[d];
            qv[dt][r] = a.q[(long long)b * a.ld_q + d];
            dva[dt][r] = 0.f;
            dqa[dt][r] = 0.f;
        }
    T2_TS(62);                                 // (every prologue load of the main body has been issued)
#ifdef T2AMD_ATTN_BWD_LATE                     // A/B builds only (python -m tacotron2_amd.build --variant ...): the round-2 order
    constexpr bool EARLY = false;
#else
    constexpr bool EARLY = FUSED && GRAN;      // staging and tanh ahead of the first hand-off (see below)
#endif
    float ua[2][16];
    UFrag16 uf;
    float th_pre[2][2][4];
    bool poison = false;
    if constexpr (FUSED && GRAN) {
        // First hand-off, granule form: no drain, no token, no second round trip -- every thread polls the granule of its
        // own position (threads Ti .. Ti+3 the four partial sums) until it carries this launch's token.  The poll sits
        // behind the wave's own K_b1 stores in the memory queue (operations complete in order), which is about when the
        // partners' granules land as well.  The slice sums travel through dq_s[0..3], the give-up flag through dq_s[4]
        // (dq_s proper is written after the tile loop).
        int* const gflag_s = reinterpret_cast<int*>(dq_s + 4);
        if (tid == 0) gflag_s[0] = 0;                  // published by the barriers inside kb1_phase
        // (round 3) the windows and the U slice go to LDS BEFORE K_b1 (their loads were the first ones issued: they have
        // landed by the time K_b1's own operands are touched; K_b1's barriers publish them) ...
#ifdef T2AMD_BWD_STAGE_LATE
        constexpr bool STAGE_LATE = LATE_OPS && EARLY;    // A/B builds: the window / U staging behind the issue of K_b1's loads
#else
        constexpr bool STAGE_LATE = false;
#endif
        if constexpr (EARLY && !STAGE_LATE) {
            stage_windows_finish(wreg, win_s, TIP, Ti, wprev_b, cumb_b, tid, KB2_NT);
            stage_u_finish(ureg, u_s, tid);
        }
        if constexpr (KB1_FIRST) kb1_consume<M16, true, false>(p, smem + p.kb1_smem_off, ds, b, ts_on, kb1r);
        else if constexpr (LATE_OPS) kb1_phase<M16, true, PERSIST>(p, smem + p.kb1_smem_off, ds, b, ts_on, before_slabs, [&] {
            if constexpr (PM_LATE) issue_pm_rows();
            issue_dpm_operands();
            if constexpr (STAGE_LATE) {
                stage_windows_finish(wreg, win_s, TIP, Ti, wprev_b, cumb_b, tid, KB2_NT);
                stage_u_finish(ureg, u_s, tid);
            }
        });
        else kb1_phase<M16, true, PERSIST>(p, smem + p.kb1_smem_off, ds, b, ts_on, before_slabs);
#ifndef T2AMD_BWD_LATE_POLL
        // (round 4) the first poll is issued BEFORE the independent work below: its round trip overlaps that work (59.48 vs
        // 59.63 ms per step over three alternating pairs of builds, same loss bits; -DT2AMD_BWD_LATE_POLL restores the old order)
        const at_u64* grow_e = reinterpret_cast<const at_u64*>(a.ws + p.gran_off) + (long long)b * (Ti + NTS);
        const int gi_e = tid < Ti + NTS ? tid : Ti + NTS - 1;
        at_u64 xd_e = __hip_atomic_load(grow_e + gi_e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        // ... so that the part of the tile loop that does NOT depend on K_b1 -- location product, + q + processed memory,
        // tanh -- runs HERE, while the partners' granules are on their way (it replaces the pre-poll pause), instead of
        // behind the hand-off: th of this lane's first two position tiles stays in registers.  Same operations on the same
        // values in the same order: bit-identical.
        if constexpr (!EARLY) {
            for (int d_ = 0; d_ < p.fused_delay; ++d_) __builtin_amdgcn_s_sleep(1);
        } else {
            if (a.bf16) load_u_frag16(uf, u_s, l15, lg);
            else load_u_frag(ua, u_s, 0, l15, lg);
            const int nmt_e = (len_raw + 15) >> 4;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int mt = wv + rr * KB2_NW;
                if (mt < nmt_e) {
                    f32x4 acc0, acc1;
                    if (a.bf16) loc_tile16(uf, win_s, TIP, mt * 16 + l15, lg, acc0, acc1);
                    else loc_tile(ua, win_s, TIP, mt * 16 + l15, lg, acc0, acc1);
                    const float pmv[2][4] = {{pmA[rr].x, pmA[rr].y, pmA[rr].z, pmA[rr].w}, {pmB[rr].x, pmB[rr].y, pmB[rr].z, pmB[rr].w}};
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) th_pre[rr][dt][r] = t2_tanh_sel<M16>((dt ? acc1[r] : acc0[r]) + qv[dt][r] + pmv[dt][r]);
                }
            }
        }
        T2_TS(59);                     // (the independent work between the hand-off's publication and its poll is done)
        const at_u64* grow = reinterpret_cast<const at_u64*>(a.ws + p.gran_off) + (long long)b * (Ti + NTS);   // (uniform)
        const int gi = tid < Ti + NTS ? tid : Ti + NTS - 1;          // Ti + NTS <= 512 (host check)
#ifndef T2AMD_BWD_LATE_POLL
        at_u64 xd = xd_e;
#else
        at_u64 xd = __hip_atomic_load(grow + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        {
            // bounded like every spin here: 50 ms of the 100 MHz wall clock, then NaN instead of a hung GPU
            const long long t0_ = wall_clock64();
            unsigned spins_ = 0;
            bool bad = false;
            for (;;) {
                if (__all((unsigned)(xd >> 32) == p.token)) break;
                __builtin_amdgcn_s_sleep(1);
                xd = __hip_atomic_load(grow + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((++spins_ & 255u) == 0 && wall_clock64() - t0_ > 5000000ll) { bad = true; t2_attn_gave_up(); break; }
            }
            if (bad && lane == 0) gflag_s[0] = 1;
        }
        T2_TS(60);                     // (every granule of this thread's wave carries the token)
        dw_r = __uint_as_float((unsigned)xd);
        if (tid >= Ti && tid < Ti + NTS) dq_s[tid - Ti] = dw_r;
        __syncthreads();
        T2_TS(61);
#pragma unroll
        for (int k = 0; k < NTS; ++k) sdv[k] = dq_s[k];
        poison = gflag_s[0] != 0;
    } else if constexpr (FUSED) {
        kb1_phase<M16, false>(p, smem + p.kb1_smem_off, ds, b, ts_on);
        // hand-off (write-through form): K_b1's outputs were stored with device-scope (sc1, write-through) stores; every
        // wave waits for their acknowledgement (explicit s_waitcnt vmcnt(0) below), then thread 0 publishes the
        // launch token; consumers poll it and read the payload with device-scope loads -- no L2 write-back /
        // invalidate (an acquire/release pair at agent scope cost ~7 us per hand-off here)
        // EVERY storing wave drains its write-through stores before the barrier (cdna_hip_programming.md Guideline 16 R1,
        // pitfall 14): __syncthreads() on gfx950 waits for LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier), so without
        // this the token below could overtake another wave's dw stores -- found in round 2 as a rare run-to-run difference
        // of the B = 64 gradients (tests/test_parity_gpu.py::test_full_size_properties), never at the B = 5 of the unit test
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* flags = reinterpret_cast<unsigned*>(a.ws + (long long)B * Ti + (long long)NTS * B) + b * NTS;
        if (tid == 0) __hip_atomic_store(flags + ds, p.token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < NTS) {
            // a short pause before the first poll: polls issued while the partners' write-through stores are still on
            // their way only delay them (round-2 finding on the persistent decode kernel, csrc/decode_persist.hip)
            for (int d_ = 0; d_ < p.fused_delay; ++d_) __builtin_amdgcn_s_sleep(1);
            // Bounded: the four workgroups of an utterance are consecutive in dispatch order, so a partner is always
            // resident or next in line -- but HIP promises no dispatch order.  After 50 ms of the 100 MHz wall clock the
            // wait is abandoned and this workgroup's dq is poisoned with NaN: every gradient of the step turns non-finite,
            // which the training loop's finite-norm check reports and skips (train.py), instead of a hung GPU.
            const long long t0_ = wall_clock64();
            unsigned spins_ = 0;
            while (__hip_atomic_load(flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.token) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins_ & 255u) == 0 && wall_clock64() - t0_ > 5000000ll) { poison = true; t2_attn_gave_up(); break; }
            }
        }
        __syncthreads();
        const float* sd = a.ws + (long long)B * Ti;
#pragma unroll
        for (int k = 0; k < NTS; ++k) sdv[k] = __hip_atomic_load(sd + k * B + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dw_r = __hip_atomic_load(a.ws + (long long)b * Ti + (tid < Ti ? tid : Ti - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- consume ----
    const int len = len_raw;
    const int nmt = (len + 15) >> 4;
    const int npos = nmt * 16;                // positions covered by the MFMA tiles
    {
        float sdot = 0.f;
#pragma unroll
        for (int k = 0; k < NTS; ++k) sdot += sdv[k];
        const float* __restrict__ wrow = a.w + (long long)b * a.ld_w;
        const float* __restrict__ dwi = a.ws + (long long)b * Ti;
        if (tid < NP) de_s[tid] = (tid < len) ? w_r * (dw_r - sdot) : 0.f;
        for (int ti = tid + KB2_NT; ti < NP; ti += KB2_NT) {
            float dwv = 0.f;
            if (ti < len) dwv = FUSED ? __hip_atomic_load(dwi + ti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : dwi[ti];
            de_s[ti] = (ti < len) ? wrow[ti] * (dwv - sdot) : 0.f;
        }
    }
    if constexpr (!EARLY) {
        stage_windows_finish(wreg, win_s, TIP, Ti, wprev_b, cumb_b, tid, KB2_NT);
        stage_u_finish(ureg, u_s, tid);
    }
    __syncthreads();
    T2_TS(49);
    T2_STAGE_RETURN(1);
    if constexpr (!EARLY) {
        if (a.bf16) load_u_frag16(uf, u_s, l15, lg);
        else load_u_frag(ua, u_s, 0, l15, lg);
    }
    // U^T as the A operand of dcol^T = U^T dpre: A[i = tap][k = lg], k-step (dt, r) <-> dim dt*16 + 4*lg + r
    float ut[4][2][4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tap = tt * 16 + l15;
                ut[tt][dt][r] = tap < NTAP ? u_s[(dt * 16 + 4 * lg + r) * NTAP + tap] : 0.f;
            }
    // bf16 mode: the same operand as one v_mfma_f32_16x16x32_bf16 fragment per tap tile.  MFMA k index 8*lg + e stands
    // for dim (e < 4 ? 4*lg + e : 16 + 4*lg + e - 4), so that the B fragment is exactly this lane's dpre registers.
    // (a.bf16 == 2, round 6, the 'bf16x3' mode: ONLY the recompute of the location conv above runs in its split-bf16 form -- an
    // f32-class product, ~2^-17 per term, what that mode's forward uses -- the two gradient products here stay exact f32)
    const bool use16 = a.bf16 == 1;
    // Row stride of dpre_s.  f32 products: 48 (the dU A-fragment reads of the four lane groups, rows lg + 4j, land on four disjoint
    // 16-bank windows).  bf16 products read rows 8 lg + e instead (8 x 48 = 0 mod 64: all four lane groups on the SAME 16 banks,
    // 4-way) and the tile loop's float4 stores of 16 consecutive rows hit 4 distinct bank groups (4-way): a stride of 36 makes the
    // stores conflict-free (36 l mod 64 covers all sixteen multiples of 4) and the reads 2-way.  (-DT2AMD_DPL48 keeps 48 for both.)
#ifdef T2AMD_DPL48
    const int dpl = DPL;
#else
    const int dpl = use16 ? 36 : DPL;
#endif
    uint4 utb[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        utb[tt].x = t2_cvt_pk_bf16(ut[tt][0][0], ut[tt][0][1]);
        utb[tt].y = t2_cvt_pk_bf16(ut[tt][0][2], ut[tt][0][3]);
        utb[tt].z = t2_cvt_pk_bf16(ut[tt][1][0], ut[tt][1][1]);
        utb[tt].w = t2_cvt_pk_bf16(ut[tt][1][2], ut[tt][1][3]);
    }

    int round = 0;
    for (int mt = wv; mt < nmt; mt += KB2_NW, ++round) {
        const int pos = mt * 16 + l15;
        float4 pm0, pm1, o0, o1;
        if (round == 0) { pm0 = pmA[0]; pm1 = pmB[0]; o0 = opA[0]; o1 = opB[0]; }
        else if (round == 1) { pm0 = pmA[1]; pm1 = pmB[1]; o0 = opA[1]; o1 = opB[1]; }
        else {
            pm0 = pm1 = o0 = o1 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pos < Ti) {
                pm0 = *reinterpret_cast<const float4*>(pmb + (long long)pos * AD);
                pm1 = *reinterpret_cast<const float4*>(pmb + (long long)pos * AD + 16);
            }
            if (pos < len) {      // d_pm read-modify-write: fetch now, add and store after the tile math
                o0 = *reinterpret_cast<float4*>(dpmb + (long long)pos * AD);
                o1 = *reinterpret_cast<float4*>(dpmb + (long long)pos * AD + 16);
            }
        }
        const bool pre = EARLY && round < 2;          // th of this tile was formed ahead of the hand-off
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (!pre) {
            if (a.bf16) loc_tile16(uf, win_s, TIP, pos, lg, acc0, acc1);
            else loc_tile(ua, win_s, TIP, pos, lg, acc0, acc1);
        }
        const float de = de_s[pos];
        float dp[2][4];
        const float pmv[2][4] = {{pm0.x, pm0.y, pm0.z, pm0.w}, {pm1.x, pm1.y, pm1.z, pm1.w}};
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float th;
                if (pre) th = round == 0 ? th_pre[0][dt][r] : th_pre[1][dt][r];
                else th = t2_tanh_sel<M16>((dt ? acc1[r] : acc0[r]) + qv[dt][r] + pmv[dt][r]);
                const float g = de * vv[dt][r] * (1.f - th * th);
                dva[dt][r] = fmaf(de, th, dva[dt][r]);
                dqa[dt][r] += g;
                dp[dt][r] = g;
            }
        // dcol^T[tap][pos] = sum_d U[d][tap] dpre[d][pos]: B operand = this lane's own dpre registers
        if (use16) {
            uint4 dpb;
            dpb.x = t2_cvt_pk_bf16(dp[0][0], dp[0][1]);
            dpb.y = t2_cvt_pk_bf16(dp[0][2], dp[0][3]);
            dpb.z = t2_cvt_pk_bf16(dp[1][0], dp[1][1]);
            dpb.w = t2_cvt_pk_bf16(dp[1][2], dp[1][3]);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                f32x4 c = {0.f, 0.f, 0.f, 0.f};
#ifdef T2AMD_DCOL_ROWMAJOR
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(at_bf16x8, utb[tt]),
                                                           __builtin_bit_cast(at_bf16x8, dpb), c, 0, 0, 0);
                *reinterpret_cast<float4*>(&dcol_s[(size_t)pos * DCL + tt * 16 + 4 * lg]) = make_float4(c[0], c[1], c[2], c[3]);
#else
                // transposed: this lane's column is tap tt*16 + l15, its four rows are positions mt*16 + 4*lg + r
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(at_bf16x8, dpb),
                                                           __builtin_bit_cast(at_bf16x8, utb[tt]), c, 0, 0, 0);
                *reinterpret_cast<float4*>(&dcol_s[(size_t)(tt * 16 + l15) * NPP + mt * 16 + 4 * lg]) = make_float4(c[0], c[1], c[2], c[3]);
#endif
            }
        } else {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                f32x4 c = {0.f, 0.f, 0.f, 0.f};
#ifdef T2AMD_DCOL_ROWMAJOR
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) c = __builtin_amdgcn_mfma_f32_16x16x4f32(ut[tt][dt][r], dp[dt][r], c, 0, 0, 0);
                *reinterpret_cast<float4*>(&dcol_s[(size_t)pos * DCL + tt * 16 + 4 * lg]) = make_float4(c[0], c[1], c[2], c[3]);
#else
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) c = __builtin_amdgcn_mfma_f32_16x16x4f32(dp[dt][r], ut[tt][dt][r], c, 0, 0, 0);
                *reinterpret_cast<float4*>(&dcol_s[(size_t)(tt * 16 + l15) * NPP + mt * 16 + 4 * lg]) = make_float4(c[0], c[1], c[2], c[3]);
#endif
            }
        }
        *reinterpret_cast<float4*>(&dpre_s[(size_t)pos * dpl + 4 * lg]) = make_float4(dp[0][0], dp[0][1], dp[0][2], dp[0][3]);
        *reinterpret_cast<float4*>(&dpre_s[(size_t)pos * dpl + 16 + 4 * lg]) = make_float4(dp[1][0], dp[1][1], dp[1][2], dp[1][3]);
        if (pos < len) {
            o0.x += dp[0][0]; o0.y += dp[0][1]; o0.z += dp[0][2]; o0.w += dp[0][3];
            o1.x += dp[1][0]; o1.y += dp[1][1]; o1.z += dp[1][2]; o1.w += dp[1][3];
            *reinterpret_cast<float4*>(dpmb + (long long)pos * AD) = o0;
            *reinterpret_cast<float4*>(dpmb + (long long)pos * AD + 16) = o1;
        }
    }
    T2_TS(50);
    T2_STAGE_RETURN(2);
    // W_q rows of the closing dh product (first 1024 columns): independent of everything above, fetched now so
    // that the round trip hides behind the reductions, the dU product and col2im
    // CELL: the attention LSTM's cell belongs to wave 1.  Its operands that do not depend on dq are issued here, BEFORE
    // the W_q prefetch: loads return in order, so whatever is issued after the W_q rows would have to land before the
    // wave may touch them -- issued first, these are simply there by then.
    const int cq_q4 = Hq >> 4;                   // float4 columns (= 4-unit groups) per workgroup, <= 64
    const bool cq_on = CELL && wv == 1 && lane < cq_q4;      // (waves 4 / 5, which have one position tile fewer: no gain)
    const int cq_j = (ds * cq_q4 + lane) * 4;
    CellOperands cq_r;          // (only touched under cq_on)
    Slab4 cq_s0, cq_s2;
    if constexpr (CELL) {
        if (cq_on) {
            const t2amd_lstm_bwd& cq = cells.cq();
            cq_r = cell_bwd_issue(cq, b, cq_j);
            cq_s0 = addend_issue4<PERSIST>(cq.dh[0], b, cq_j);
            cq_s2 = addend_issue4<PERSIST>(cq.dh[2], b, cq_j);
        }
    }
    // bf16 mode (a.Wq16): the rows are streamed as bf16 -- 16 bytes = EIGHT columns, half the bytes of the stream whose issue
    // alone took 1.2 us of this launch (8 waves x 16 instructions x 1 KB through the CU's 64 B/clk path); a thread then owns
    // 8 columns and only half of the lanes load.  Same fmaf chain over the 16 dims per column, same order of the partial sums.
#ifdef T2AMD_BWD_WQ_F32                       // A/B builds: the f32 rows in both modes (round 3)
    const bool wq16 = false;
#else
    const bool wq16 = a.Wq16 != nullptr;
#endif
    float4 wq_pre[16];
    if constexpr (CELL) {
        // group g = tid >> 6 takes dims 16g .. 16g+15 of ALL 128, lane c4 the float4 column ds*Hq/16 + c4
        const int H4 = Hq >> 2, QC = Hq >> 4;
        const int g = tid >> 6, c4 = tid & 63;
        if (wq16) {
            const int H8 = Hq >> 3, QC8 = QC >> 1;               // 16-byte units of 8 bf16 columns
            const float4* __restrict__ W8 = reinterpret_cast<const float4*>(a.Wq16) + (long long)(g * 16) * H8 + ds * QC8;
#pragma unroll
            for (int dd = 0; dd < 16; ++dd) wq_pre[dd] = W8[(long long)dd * H8 + (c4 < QC8 ? c4 : 0)];
        } else {
            const float4* __restrict__ W4 = reinterpret_cast<const float4*>(a.Wq) + (long long)(g * 16) * H4 + ds * QC;
#pragma unroll
            for (int dd = 0; dd < 16; ++dd) wq_pre[dd] = W4[(long long)dd * H4 + (c4 < QC ? c4 : 0)];
        }
    } else {
        const int H4 = Hq >> 2;
        const int half = tid >> 8, t8 = tid & 255;
        if (wq16) {
            const int H8 = Hq >> 3;
            const float4* __restrict__ W8 = reinterpret_cast<const float4*>(a.Wq16) + (long long)(dbase + half * 16) * H8;
#pragma unroll
            for (int dd = 0; dd < 16; ++dd) wq_pre[dd] = W8[(long long)dd * H8 + (t8 < H8 ? t8 : 0)];
        } else {
            const float4* __restrict__ W4 = reinterpret_cast<const float4*>(a.Wq) + (long long)(dbase + half * 16) * H4;
#pragma unroll
            for (int dd = 0; dd < 16; ++dd) wq_pre[dd] = W4[(long long)dd * H4 + (t8 < H4 ? t8 : 0)];
        }
    }
    T2_TS(56);
    // ... and the read-modify-write operand of this wave's dU tile (wave w owns tap tile w&3, dim tile w>>2)
    float dU_old[4];
    {
        const int tap = (wv & 3) * 16 + l15;
        const float* dUg = a.dU_acc + ((long long)b * AD + dbase + (wv >> 2) * 16 + 4 * lg) * NTAP + (tap < NTAP ? tap : 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) dU_old[r] = dUg[(long long)r * NTAP];
    }
    // dv / dq: reduce over the positions held by the 16 lanes of a lane group
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = row16_sum(dva[dt][r]), y = row16_sum(dqa[dt][r]);
            if (l15 == 0) {
                red_s[(wv * 2 + 0) * DSL + dt * 16 + 4 * lg + r] = x;
                red_s[(wv * 2 + 1) * DSL + dt * 16 + 4 * lg + r] = y;
            }
        }
    T2_TS(57);
    __syncthreads();
    T2_TS(58);
    if (tid < DSL) {
        float dvs = 0.f, dqs = 0.f;
#pragma unroll
        for (int w = 0; w < KB2_NW; ++w) {
            dvs += red_s[(w * 2 + 0) * DSL + tid];
            dqs += red_s[(w * 2 + 1) * DSL + tid];
        }
        a.dv_acc[(long long)b * AD + dbase + tid] = dv_old + dvs;
        if (FUSED && __any(poison)) dqs = __builtin_nanf("");       // abandoned hand-off (see above): poison the step
        dq_s[tid] = dqs;
        // CELL: the partners read this slice (device-scope store, drained before the token in the dU phase below)
        if constexpr (CELL) __hip_atomic_store(&a.dq_out[(long long)b * a.ld_dq + dbase + tid], dqs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else a.dq_out[(long long)b * a.ld_dq + dbase + tid] = dqs;
    }
    // CELL: the independent cell (decoder LSTM of step t-1) belongs to the last wave: operands issued here, consumed
    // after the wave's dU tile -- its HBM round trip hides behind the MFMAs, its arithmetic behind col2im, where the last
    // waves have little or nothing to do (2 Ti outputs over 512 threads)
    CellOperands cx_r;          // (only touched under cx_on)
    Slab4 cx_s0, cx_s1, cx_s2;
    const int cx_j = CELL ? (ds * p.cx_q4 + lane) * 4 : 0;
    const bool cx_on = CELL && wv == KB2_NW - 1 && lane < p.cx_q4;
    if constexpr (CELL) {
        if (cx_on) {
            const t2amd_lstm_bwd& cx = cells.cx();
            cx_r = cell_bwd_issue(cx, b, cx_j);
            cx_s0 = addend_issue4<PERSIST>(cx.dh[0], b, cx_j);
            cx_s1 = addend_issue4<PERSIST>(cx.dh[1], b, cx_j);
            cx_s2 = addend_issue4<PERSIST>(cx.dh[2], b, cx_j);
        }
    }
    T2_TS(51);
    T2_STAGE_RETURN(3);
    // dU[d][tap] += sum_pos dpre[pos][d] * win[c(tap)][pos + k(tap)]: wave w owns (tap tile w&3, dim tile w>>2)
    {
        const int tt = wv & 3, dt = wv >> 2;
        const int tap = tt * 16 + l15;
        const int toff = tap_offset(tap, TIP);
        float* dUg = a.dU_acc + ((long long)b * AD + dbase + dt * 16 + 4 * lg) * NTAP + (tap < NTAP ? tap : 0);
        float old[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) old[r] = dU_old[r];                    // fetched right after the tile loop
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
        const float* ap = dpre_s + (size_t)lg * dpl + dt * 16 + l15;
        const float* bp = win_s + toff + lg;
        // npos is a multiple of 16: chunks of 16 positions (four k-steps), the next chunk's eight LDS operands are
        // read while the current chunk's four MFMAs issue (an unpipelined loop pays the LDS latency per MFMA pair)
        if (use16) {
            // bf16: K = 32 positions per MFMA; lane (row/col l15, group lg) supplies positions s + 8*lg .. + 7
            const float* ap16 = dpre_s + dt * 16 + l15;
            const float* bp16 = win_s + toff;
            for (int s = 0; s < npos; s += 32) {
                float av[8], bv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ps = s + 8 * lg + e;
                    const int pc = ps < npos ? ps : 0;          // rows past npos were never written: read row 0, use 0
                    av[e] = ap16[(size_t)pc * dpl];
                    bv[e] = bp16[pc];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) av[e] = (s + 8 * lg + e < npos) ? av[e] : 0.f;
                uint4 af, bf;
                af.x = t2_cvt_pk_bf16(av[0], av[1]); af.y = t2_cvt_pk_bf16(av[2], av[3]);
                af.z = t2_cvt_pk_bf16(av[4], av[5]); af.w = t2_cvt_pk_bf16(av[6], av[7]);
                bf.x = t2_cvt_pk_bf16(bv[0], bv[1]); bf.y = t2_cvt_pk_bf16(bv[2], bv[3]);
                bf.z = t2_cvt_pk_bf16(bv[4], bv[5]); bf.w = t2_cvt_pk_bf16(bv[6], bv[7]);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(at_bf16x8, af),
                                                            __builtin_bit_cast(at_bf16x8, bf), c0, 0, 0, 0);
            }
        }
        float a_cur[4], b_cur[4], a_nxt[4], b_nxt[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { a_cur[j] = ap[(size_t)(4 * j) * dpl]; b_cur[j] = bp[4 * j]; }
        for (int s = 0; s < (use16 ? 0 : npos); s += 16) {
            const int sn = (s + 16 < npos) ? s + 16 : s;        // clamped: the last prefetch re-reads this chunk
#pragma unroll
            for (int j = 0; j < 4; ++j) { a_nxt[j] = ap[(size_t)(sn + 4 * j) * dpl]; b_nxt[j] = bp[sn + 4 * j]; }
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[0], b_cur[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[1], b_cur[1], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[2], b_cur[2], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[3], b_cur[3], c1, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) { a_cur[j] = a_nxt[j]; b_cur[j] = b_nxt[j]; }
        }
        // npos is a multiple of 16, so no tail
        if constexpr (CELL) {
            // second hand-off: wave 0 stored the slice's dq above; its outstanding vector-memory operations at this point
            // are that store, the W_q prefetch and dU_old, which the stores below wait for anyway -- so draining them
            // here costs nothing, and the token travels while col2im runs
            if (wv == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                unsigned* flags2 = reinterpret_cast<unsigned*>(a.ws + (long long)B * Ti + 2ll * NTS * B) + b * NTS;
                if (tid == 0) __hip_atomic_store(flags2 + ds, p.token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (tap < NTAP) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dUg[(long long)r * NTAP] = old[r] + (c0[r] + c1[r]);
        }
    }
    T2_TS(52);
    T2_STAGE_RETURN(4);
    if constexpr (CELL) {
        if (cx_on) {
            const t2amd_lstm_bwd& cx = cells.cx();
            const float4 d0 = addend_finish4<PERSIST>(cx_s0, cx.dh[0], b, cx_j), d1 = addend_finish4<PERSIST>(cx_s1, cx.dh[1], b, cx_j);
            const float4 d2 = addend_finish4<PERSIST>(cx_s2, cx.dh[2], b, cx_j);
            cell_bwd_finish<PERSIST>(cx, cx_r, d0, d1, d2, b, cx_j);
        }
    }
    // col2im: partial carry dwin[c][ti'] = sum_k dcol[ti' - k + 15][c*31 + k] over this slice's dims
    {
        float* __restrict__ out = a.dwin_part + (((long long)ds * B + b) * 2) * Ti;
        for (int i = tid; i < 2 * Ti; i += KB2_NT) {
            const int c = i >= Ti;
            const int tip_ = i - c * Ti;
            // 31 independent LDS reads (row clamped, contribution selected): a branch per tap serialises them
            float v[LK];
#pragma unroll
            for (int k = 0; k < LK; ++k) {
                int row = tip_ - k + HALO;
                row = row < 0 ? 0 : (row > npos - 1 ? npos - 1 : row);
#ifdef T2AMD_DCOL_ROWMAJOR
                v[k] = dcol_s[(size_t)row * DCL + c * LK + k];
#else
                v[k] = dcol_s[(size_t)(c * LK + k) * NPP + row];
#endif
            }
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < LK; ++k) {
                const int row = tip_ - k + HALO;
                s += (row >= 0 && row < npos) ? v[k] : 0.f;
            }
            st_xwg<PERSIST>(out + i, s);        // the next step's K_b1 phase of the utterance's other workgroups reads it
        }
    }
    T2_TS(53);
    T2_STAGE_RETURN(5);
    if constexpr (CELL) {
        // ---- the attention LSTM's cell: W_q^T dq for this workgroup's quarter of the columns, then the cell ----
        const int QC = cq_q4;
        const int g = tid >> 6, c4 = tid & 63;
        // Wave 6 collects the utterance's dq: it has no col2im outputs for Ti <= 192 and so no stores in flight that a
        // poll would have to queue behind (vector-memory operations complete in order).  The partners published their
        // tokens before col2im, one phase ago: normally the first poll succeeds.  Bounded like the first hand-off; an
        // abandoned wait turns the cell's gradients into NaN.  The same wave reads dq right behind the polls (program
        // order within a wave), so no barrier separates the two.
        if (wv == 6) {
            bool bad = false;
            if (lane < NTS) {
                const unsigned* flags2 = reinterpret_cast<const unsigned*>(a.ws + (long long)B * Ti + 2ll * NTS * B) + b * NTS;
                const long long t0_ = wall_clock64();
                unsigned spins_ = 0;
                while (__hip_atomic_load(flags2 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.token) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins_ & 255u) == 0 && wall_clock64() - t0_ > 5000000ll) { bad = true; t2_attn_gave_up(); break; }
                }
            }
            bad = __any(bad);
            const float* dqg = a.dq_out + (long long)b * a.ld_dq;
            const float q0 = __hip_atomic_load(dqg + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float q1 = __hip_atomic_load(dqg + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dqall_s[lane] = bad ? __builtin_nanf("") : q0;
            dqall_s[64 + lane] = bad ? __builtin_nanf("") : q1;
        }
        __syncthreads();
        T2_TS(54);
        float4* part_s = reinterpret_cast<float4*>(dh_s);      // [8 groups][QC]
        if (wq16) {
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int dd = 0; dd < 16; ++dd) {
                const float gq = dqall_s[g * 16 + dd];
                const unsigned u0 = __float_as_uint(wq_pre[dd].x), u1 = __float_as_uint(wq_pre[dd].y);
                const unsigned u2 = __float_as_uint(wq_pre[dd].z), u3 = __float_as_uint(wq_pre[dd].w);
                a0.x = fmaf(gq, __uint_as_float(u0 << 16), a0.x);
                a0.y = fmaf(gq, __uint_as_float(u0 & 0xffff0000u), a0.y);
                a0.z = fmaf(gq, __uint_as_float(u1 << 16), a0.z);
                a0.w = fmaf(gq, __uint_as_float(u1 & 0xffff0000u), a0.w);
                a1.x = fmaf(gq, __uint_as_float(u2 << 16), a1.x);
                a1.y = fmaf(gq, __uint_as_float(u2 & 0xffff0000u), a1.y);
                a1.z = fmaf(gq, __uint_as_float(u3 << 16), a1.z);
                a1.w = fmaf(gq, __uint_as_float(u3 & 0xffff0000u), a1.w);
            }
            if (c4 < (QC >> 1)) { part_s[g * QC + 2 * c4] = a0; part_s[g * QC + 2 * c4 + 1] = a1; }
        } else {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int dd = 0; dd < 16; ++dd) {
                const float gq = dqall_s[g * 16 + dd];
                acc.x = fmaf(gq, wq_pre[dd].x, acc.x);
                acc.y = fmaf(gq, wq_pre[dd].y, acc.y);
                acc.z = fmaf(gq, wq_pre[dd].z, acc.z);
                acc.w = fmaf(gq, wq_pre[dd].w, acc.w);
            }
            if (c4 < QC) part_s[g * QC + c4] = acc;
        }
        __syncthreads();
        if (cq_on) {
            // slab k of a separate launch = (dims 32k .. 32k+15) + (dims 32k+16 .. 32k+31); slabs are added in index order
            float4 d1;
            {
                const float4 p0 = part_s[0 * QC + c4], p1 = part_s[1 * QC + c4];
                d1 = make_float4(0.f + (p0.x + p1.x), 0.f + (p0.y + p1.y), 0.f + (p0.z + p1.z), 0.f + (p0.w + p1.w));
            }
#pragma unroll
            for (int k = 1; k < NSL; ++k) {
                const float4 p0 = part_s[(2 * k) * QC + c4], p1 = part_s[(2 * k + 1) * QC + c4];
                d1.x += p0.x + p1.x; d1.y += p0.y + p1.y; d1.z += p0.z + p1.z; d1.w += p0.w + p1.w;
            }
            const t2amd_lstm_bwd& cq = cells.cq();
            const float4 d0 = addend_finish4<PERSIST>(cq_s0, cq.dh[0], b, cq_j), d2 = addend_finish4<PERSIST>(cq_s2, cq.dh[2], b, cq_j);
            cell_bwd_finish<PERSIST>(cq, cq_r, d0, d1, d2, b, cq_j);
        }
        T2_TS(55);
        return;
    }
    __syncthreads();   // dq_s
    // partial dh = sum_{d in slice} dq[d] * W_q[d][:]; the two halves of the block take 16 dims each
    {
        const int H4 = Hq >> 2;
        const int half = tid >> 8, t8 = tid & 255;
        const float4* __restrict__ W4 = reinterpret_cast<const float4*>(a.Wq) + (long long)(dbase + half * 16) * H4;
        float* __restrict__ dh = a.dh_out + (long long)ds * a.dh_split_stride + (long long)b * a.ld_dh;
        if (wq16) {
            // bf16 rows: thread t8 of each half owns the 8 columns 8 k8 .. 8 k8 + 7 (same per-column fmaf chain, half 0 + half 1)
            const int H8 = Hq >> 3;
            const float4* __restrict__ W8 = reinterpret_cast<const float4*>(a.Wq16) + (long long)(dbase + half * 16) * H8;
            for (int k0 = 0; k0 < H8; k0 += 256) {
                const int k8 = k0 + t8;
                const bool ok = k8 < H8;
                float4 w[16];
                if (k0 == 0) {
#pragma unroll
                    for (int dd = 0; dd < 16; ++dd) w[dd] = wq_pre[dd];
                } else {
#pragma unroll
                    for (int dd = 0; dd < 16; ++dd) w[dd] = W8[(long long)dd * H8 + (ok ? k8 : 0)];
                }
                float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int dd = 0; dd < 16; ++dd) {
                    const float gq = dq_s[half * 16 + dd];
                    const unsigned u0 = __float_as_uint(w[dd].x), u1 = __float_as_uint(w[dd].y);
                    const unsigned u2 = __float_as_uint(w[dd].z), u3 = __float_as_uint(w[dd].w);
                    a0.x = fmaf(gq, __uint_as_float(u0 << 16), a0.x);
                    a0.y = fmaf(gq, __uint_as_float(u0 & 0xffff0000u), a0.y);
                    a0.z = fmaf(gq, __uint_as_float(u1 << 16), a0.z);
                    a0.w = fmaf(gq, __uint_as_float(u1 & 0xffff0000u), a0.w);
                    a1.x = fmaf(gq, __uint_as_float(u2 << 16), a1.x);
                    a1.y = fmaf(gq, __uint_as_float(u2 & 0xffff0000u), a1.y);
                    a1.z = fmaf(gq, __uint_as_float(u3 << 16), a1.z);
                    a1.w = fmaf(gq, __uint_as_float(u3 & 0xffff0000u), a1.w);
                }
                if (half == 1 && ok) {
                    *reinterpret_cast<float4*>(dh_s + k8 * 8) = a0;
                    *reinterpret_cast<float4*>(dh_s + k8 * 8 + 4) = a1;
                }
                __syncthreads();
                if (half == 0 && ok) {
                    const float4 o0 = *reinterpret_cast<const float4*>(dh_s + k8 * 8), o1 = *reinterpret_cast<const float4*>(dh_s + k8 * 8 + 4);
                    a0.x += o0.x; a0.y += o0.y; a0.z += o0.z; a0.w += o0.w;
                    a1.x += o1.x; a1.y += o1.y; a1.z += o1.z; a1.w += o1.w;
                    *reinterpret_cast<float4*>(dh + k8 * 8) = a0;
                    *reinterpret_cast<float4*>(dh + k8 * 8 + 4) = a1;
                }
                if (k0 + 256 < H8) __syncthreads();
            }
        } else
        for (int k0 = 0; k0 < H4; k0 += 256) {
            const int k4 = k0 + t8;
            const bool ok = k4 < H4;
            float4 w[16];
            if (k0 == 0) {
#pragma unroll
                for (int dd = 0; dd < 16; ++dd) w[dd] = wq_pre[dd];
            } else {
#pragma unroll
                for (int dd = 0; dd < 16; ++dd) w[dd] = W4[(long long)dd * H4 + (ok ? k4 : 0)];
            }
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int dd = 0; dd < 16; ++dd) {
                const float g = dq_s[half * 16 + dd];
                acc.x = fmaf(g, w[dd].x, acc.x);
                acc.y = fmaf(g, w[dd].y, acc.y);
                acc.z = fmaf(g, w[dd].z, acc.z);
                acc.w = fmaf(g, w[dd].w, acc.w);
            }
            if (half == 1 && ok) *reinterpret_cast<float4*>(dh_s + k4 * 4) = acc;
            __syncthreads();
            if (half == 0 && ok) {
                const float4 o = *reinterpret_cast<const float4*>(dh_s + k4 * 4);
                acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
                *reinterpret_cast<float4*>(dh + k4 * 4) = acc;
            }
        }
    }
    T2_TS(54);
}

template <bool FUSED, bool M16, bool CELL, bool GRAN>
__global__ __launch_bounds__(KB2_NT) void attn_bwd_main_kernel(AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    bool ts_on = false;
    attn_bwd_main_body<FUSED, M16, CELL, GRAN, false>(p, smem, (int)blockIdx.x, (int)blockIdx.y, ts_on, KernargCells(), KbNoHook());
}

static int g_attn_bwd_lds = 0, g_attn_bwd_lds_fused = 0, g_attn_bwd_lds_cell = 0;
#define T2_ATTN_GRANULES_DEFAULT 1      // measured on MI355X: 64.1 vs 65.3 ms per training step (profiles/r02_y_ab_fold_granules.json)
static int g_attn_gran = -1;               // -1: environment / default; 0 / 1: t2amd_set_attn_bwd_granules
extern "C" int t2amd_set_attn_bwd_granules(int on) {
    T2_REQUIRE(on == 0 || on == 1 || on == -1, "set_attn_bwd_granules: -1 (default), 0 or 1");
    g_attn_gran = on;
    return T2AMD_OK;
}
static unsigned g_attn_bwd_token = 0;      // one launch counter for both one-launch forms: a token never repeats in ws (never zero)
static int g_attn_bwd_fused = -1;          // -1: T2AMD_ATTN_FUSED_BWD / default (one launch); 0 / 1: t2amd_set_attn_bwd_fused
extern "C" int t2amd_set_attn_bwd_fused(int on) {
    T2_REQUIRE(on == 0 || on == 1 || on == -1, "set_attn_bwd_fused: -1 (default), 0 or 1");
    g_attn_bwd_fused = on;
    return T2AMD_OK;
}

extern "C" long long t2amd_attn_bwd_ws_floats(int B, int Ti) {
    const long long goff = ((long long)B * Ti + 12ll * B + 1) / 2 * 2;
    return goff + 2ll * B * Ti + 2ll * NTS * B;
}

static int attn_bwd_step_impl(const t2amd_attn_bwd* a, void* stream) {
    T2_REQUIRE(a && a->dctx_total && a->q && a->Wq && a->U && a->v && a->pm && a->memory && a->w &&
                   a->cum_before && a->dwin_part && a->dcum_acc && a->d_pm && a->dU_acc && a->dv_acc &&
                   a->dq_out && a->dh_out && a->ws,
               "attn_bwd: null pointer");
    T2_REQUIRE(a->B > 0 && a->Ti > 0 && a->E % 4 == 0 && a->Hq % 4 == 0, "attn_bwd: bad dims");
    T2_REQUIRE(t2_aligned16(a->Wq) && t2_aligned16(a->memory) && t2_aligned16(a->pm) && t2_aligned16(a->d_pm) &&
                   t2_aligned16(a->dh_out) && a->ld_dh % 4 == 0 && a->dh_split_stride % 4 == 0,
               "attn_bwd: alignment");
    AttnBwdParams p;
    p.a = *a;
    for (int i = 0; i < 3; ++i)
        if (p.a.dctx[i].p && p.a.dctx[i].nsplit < 1) p.a.dctx[i].nsplit = 1;
    p.tip = attn_tip(a->Ti);
    p.np = ((a->Ti + 15) / 16) * 16;
    p.dbg = attn_dbg_stage();
    p.ts = attn_ts_buffer();
    hipStream_t s = (hipStream_t)stream;
    const size_t lds1 = sizeof(float) * ((size_t)a->E + 2 * (size_t)((a->Ti + NTS - 1) / NTS) + 8);
    const size_t lds2 = sizeof(float) * (2 * (size_t)p.tip + p.np + DCOL_FLOATS(p.np) + (size_t)p.np * DPL + KB2_NW * 2 * DSL + DSL +
                                         DSL * NTAP + (size_t)a->Hq);
    T2_REQUIRE(lds1 <= 64 * 1024, "attn_bwd: E too large");
    T2_REQUIRE(lds2 <= 160 * 1024, "attn_bwd: Ti needs more than 160 KiB of LDS");
    // the folded cells (t2amd_attn_bwd.cell_q / cell_x)
    T2_REQUIRE(a->cell_q || !a->cell_x, "attn_bwd: cell_x needs cell_q");
    bool fold = false;
    if (a->cell_q) {
        const t2amd_lstm_bwd* cq = a->cell_q;
        T2_PROPAGATE(t2amd_check_lstm_bwd_(cq));
        T2_REQUIRE(cq->H == a->Hq && cq->B == a->B && !cq->lens, "attn_bwd: cell_q must be the [B][Hq] cell of this step, without lens");
        T2_REQUIRE(cq->dh[1].p == a->dh_out && cq->dh[1].ld == a->ld_dh && cq->dh[1].nsplit == NSL &&
                       cq->dh[1].split_stride == a->dh_split_stride,
                   "attn_bwd: cell_q->dh[1] must describe the dh_out slabs");
        fold = a->Hq % 16 == 0 && a->Hq <= 1024;
        if (a->cell_x) {
            T2_PROPAGATE(t2amd_check_lstm_bwd_(a->cell_x));
            T2_REQUIRE(a->cell_x->B == a->B && !a->cell_x->lens, "attn_bwd: cell_x must have this step's batch, without lens");
            fold = fold && a->cell_x->H % 16 == 0 && a->cell_x->H <= 1024;
        }
    }
    // CELL form: dh_s grows from [Hq] to [2 Hq] (eight 16-dim partials per column), + the utterance's dq + a flag
    const size_t lds2c = lds2 + sizeof(float) * ((size_t)a->Hq + AD + 4);
    if ((int)lds2 > 64 * 1024 && (int)lds2 > g_attn_bwd_lds && !t2amd_validate_only_flag_()) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_main_kernel<false, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        (void)hipFuncSetAttribute((const void*)attn_bwd_main_kernel<false, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        g_attn_bwd_lds = (int)lds2;
    }
    T2_REQUIRE(!a->memory16 || (t2_aligned16(a->memory16) && a->E % 8 == 0), "attn_bwd: memory16 must be 16-byte aligned, E a multiple of 8");
    T2_REQUIRE(!a->Wq16 || (t2_aligned16(a->Wq16) && a->Hq % 32 == 0), "attn_bwd: Wq16 must be 16-byte aligned, Hq a multiple of 32");
    // One launch (default since round 2; T2AMD_ATTN_FUSED_BWD=0 restores the two launches): K_b1 runs as the first phase
    // of K_b2's launch and the four workgroups of an utterance hand their dw slices to each other through memory
    // (write-through stores drained by every wave, a per-launch token, a short pause (T2AMD_ATTN_FUSED_DELAY, s_sleep
    // units, default 16), then polls).  Round 1 measured 71.6 vs 72.1 ms per training step -- with a hand-off that did not
    // drain the stores before the token and was therefore racy; with the drain round 2 measures 71.5 vs 72.1 ms over 16
    // steps, twice (profiles/r02_i_fused_bwd_after_fix.txt).  Bit-identical to the two-launch path (tests).
    static const bool fused_env = [] { const char* e = getenv("T2AMD_ATTN_FUSED_BWD"); return !(e && e[0] == '0'); }();
    const bool fused = g_attn_bwd_fused < 0 ? fused_env : g_attn_bwd_fused != 0;
    // pre-poll pause in s_sleep units: 16 for the token form; the granule form is flat from 4 to 16 (63.7 / 63.55 / 63.8 ms
    // per training step at 4 / 8 / 16; 64.2 at 0, 64.9 at 64: profiles/r02_y_granule_delay_sweep.txt)
    static const int fused_delay_env = [] { const char* e = getenv("T2AMD_ATTN_FUSED_DELAY"); const int v = e ? atoi(e) : -1; return v > 100 ? 100 : v; }();
    // First hand-off as granules (T2AMD_ATTN_GRANULES=0/1): needs the granule block of ws (t2amd_attn_bwd.ws_floats),
    // 8-byte aligned, and one granule per thread (Ti <= 512: always true within the LDS limit above)
    static const bool gran_env = [] { const char* e = getenv("T2AMD_ATTN_GRANULES"); return e ? e[0] != '0' : T2_ATTN_GRANULES_DEFAULT != 0; }();
    p.gran_off = ((long long)a->B * a->Ti + 12ll * a->B + 1) / 2 * 2;
    const bool gran = (g_attn_gran < 0 ? gran_env : g_attn_gran != 0) && a->Ti + NTS <= KB2_NT &&
                      a->ws_floats >= p.gran_off + 2ll * a->B * a->Ti + 2ll * NTS * a->B &&
                      (reinterpret_cast<uintptr_t>(a->ws + p.gran_off) & 7u) == 0;
    p.fused_delay = fused_delay_env >= 0 ? fused_delay_env : (gran ? 8 : 16);
    p.cx_q4 = 0;
    p.cq = t2amd_lstm_bwd{};
    p.cx = t2amd_lstm_bwd{};
    if (fused && fold && lds1 + lds2c <= 160 * 1024) {
        // one launch for the attention backward AND the step's two LSTM cell backwards
        if (++g_attn_bwd_token == 0) ++g_attn_bwd_token;
        p.token = g_attn_bwd_token;
        p.cq = *a->cell_q;
        if (a->cell_x) { p.cx = *a->cell_x; p.cx_q4 = a->cell_x->H >> 4; }
        for (int i = 0; i < 3; ++i) {
            if (p.cq.dh[i].p && p.cq.dh[i].nsplit < 1) p.cq.dh[i].nsplit = 1;
            if (p.cx.dh[i].p && p.cx.dh[i].nsplit < 1) p.cx.dh[i].nsplit = 1;
        }
        const size_t l2a = (lds2c + 15) / 16 * 16;
        p.kb1_smem_off = (int)(l2a / sizeof(float));
        const size_t ldsf = l2a + lds1;
        if ((int)ldsf > 64 * 1024 && (int)ldsf > g_attn_bwd_lds_cell && !t2amd_validate_only_flag_()) {
            (void)hipFuncSetAttribute((const void*)attn_bwd_main_kernel<true, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf);
            (void)hipFuncSetAttribute((const void*)attn_bwd_main_kernel<true, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf);
            (void)hipFuncSetAttribute((const void*)attn_bwd_main_kernel<true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf);
            (void)hipFuncSetAttribute((const void*)attn_bwd_main_kernel<true, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf);
            g_attn_bwd_lds_cell = (int)ldsf;
        }
        if (gran) {
            if (a->memory16) T2_LAUNCH_ROLE(4, (attn_bwd_main_kernel<true, true, true, true>), dim3(NSL, a->B), dim3(KB2_NT), ldsf, s, p);
            else T2_LAUNCH_ROLE(4, (attn_bwd_main_kernel<true, false, true, true>), dim3(NSL, a->B), dim3(KB2_NT), ldsf, s, p);
        } else {
            if (a->memory16) T2_LAUNCH_ROLE(4, (attn_bwd_main_kernel<true, true, true, false>), dim3(NSL, a->B), dim3(KB2_NT), ldsf, s, p);
            else T2_LAUNCH_ROLE(4, (attn_bwd_main_kernel<true, false, true, false>), dim3(NSL, a->B), dim3(KB2_NT), ldsf, s, p);
        }
        T2_LAUNCH_CHECK();
        return T2AMD_OK;
    }
    if (fused && lds1 + lds2 <= 160 * 1024) {
        if (++g_attn_bwd_token == 0) ++g_attn_bwd_token;
        p.token = g_attn_bwd_token;
        const size_t l2a = (lds2 + 15) / 16 * 16;
        p.kb1_smem_off = (int)(l2a / sizeof(float));
        const size_t ldsf = l2a + lds1;
        if ((int)ldsf > 64 * 1024 && (int)ldsf > g_attn_bwd_lds_fused && !t2amd_validate_only_flag_()) {
            (void)hipFuncSetAttribute((const void*)attn_bwd_main_kernel<true, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf);
            (void)hipFuncSetAttribute((const void*)attn_bwd_main_kernel<true, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf);
            (void)hipFuncSetAttribute((const void*)attn_bwd_main_kernel<true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf);
            (void)hipFuncSetAttribute((const void*)attn_bwd_main_kernel<true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf);
            g_attn_bwd_lds_fused = (int)ldsf;
        }
        if (gran) {
            if (a->memory16) T2_LAUNCH_ROLE(4, (attn_bwd_main_kernel<true, true, false, true>), dim3(NSL, a->B), dim3(KB2_NT), ldsf, s, p);
            else T2_LAUNCH_ROLE(4, (attn_bwd_main_kernel<true, false, false, true>), dim3(NSL, a->B), dim3(KB2_NT), ldsf, s, p);
        } else {
            if (a->memory16) T2_LAUNCH_ROLE(4, (attn_bwd_main_kernel<true, true, false, false>), dim3(NSL, a->B), dim3(KB2_NT), ldsf, s, p);
            else T2_LAUNCH_ROLE(4, (attn_bwd_main_kernel<true, false, false, false>), dim3(NSL, a->B), dim3(KB2_NT), ldsf, s, p);
        }
        T2_LAUNCH_CHECK();
    } else {
        t2amd_profile_mark_(4, 0, s);          // role 4: the attention backward pair of one time step (bench.py roofline)
        if (a->memory16) T2_LAUNCH(attn_bwd_dw_kernel<true>, dim3(NTS, a->B), dim3(KB1_NT), lds1, s, p);
        else T2_LAUNCH(attn_bwd_dw_kernel<false>, dim3(NTS, a->B), dim3(KB1_NT), lds1, s, p);
        // (M16 of the separate K_b2 launch touches no memory rows: it carries the MODE -- the tanh form the forward used, common.h)
        if (a->memory16) T2_LAUNCH((attn_bwd_main_kernel<false, true, false, false>), dim3(NSL, a->B), dim3(KB2_NT), lds2, s, p);
        else T2_LAUNCH((attn_bwd_main_kernel<false, false, false, false>), dim3(NSL, a->B), dim3(KB2_NT), lds2, s, p);
        t2amd_profile_mark_(4, 1, s);
        T2_LAUNCH_CHECK();
    }
    // cells that were asked for but not folded (geometry, or the two-launch form): the separate launch, from here
    if (a->cell_q) return t2amd_lstm_pointwise_bwd2_f32(a->cell_q, a->cell_x, stream);
    return T2AMD_OK;
}
extern "C" int t2amd_attention_step_bwd_f32(const t2amd_attn_bwd* a, void* stream) {
    return attn_bwd_step_impl(a, stream);
}

// (Round 6, VERDICT r05 item 9 "delete or win": the persistent BACKWARD loop -- dec_train_bwd_persistent_kernel, its per-step descriptor
// staging and its ABI entries, opt-in since round 4 -- stood here.  Two rounds at 1 ms BEHIND the launch chain it was meant to replace
// (28.45 vs 27.4 ms of backward loop, profiles/r04_j_ab_train_bwd_persistent.json; DESIGN 5.1 / 5.2 say why: the step is two all-to-all
// edges around an 18 us issue-bound phase, and the hand-offs cost what the kernel boundaries did).  The measurement stays under
// profiles/, the code is in the history (last present in the commit before this one).  What remains of it are template parameters of
// shared device functions -- PERSIST in attn_bwd_main_body / kb1_*, SC1 in cell_bwd.h, XPLAIN and the negative gate_seg in
// skinny_wide.h -- which no kernel instantiates with `true` any more.)

// ---------------------------------------------------------------------------------------
// Fold / unfold of the location layer:  U[128][62]
// ---------------------------------------------------------------------------------------
__global__ void fold_location_kernel(const float* __restrict__ wd, const float* __restrict__ wc,
                                     float* __restrict__ U) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < AD * NTAP) {
        const int d = i / NTAP, ck = i - d * NTAP;
        float s = 0.f;
        for (int f = 0; f < T2AMD_LOC_FILTERS; ++f) s = fmaf(wd[d * T2AMD_LOC_FILTERS + f], wc[f * NTAP + ck], s);
        U[i] = s;
    }
}

extern "C" int t2amd_fold_location_f32(const float* wdense, const float* wconv, float* U, void* stream) {
    T2_REQUIRE(wdense && wconv && U && t2_aligned16(U), "fold_location: bad pointers");
    T2_LAUNCH(fold_location_kernel, dim3((AD * NTAP + 255) / 256), dim3(256), 0, (hipStream_t)stream, wdense, wconv, U);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// Two launches (round 6; one workgroup did all of it in 98 us per training step): (1) dU and dv summed over the utterances, one thread
// per element, same add order as before; (2) the two small products, one thread per output, the same fmaf chains -- same bits.
__global__ __launch_bounds__(256) void unfold_location_reduce_kernel(const float* __restrict__ dU_acc, const float* __restrict__ dv_acc,
                                                                     int nb, float* __restrict__ dv, float* __restrict__ dUsum) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < AD * NTAP) {
        // utterances eight at a time, loads first (a plain loop is one waited load per utterance); same add order
        float s = 0.f;
        int b = 0;
        for (; b + 8 <= nb; b += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = dU_acc[(long long)(b + j) * AD * NTAP + i];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; b < nb; ++b) s += dU_acc[(long long)b * AD * NTAP + i];
        dUsum[i] = s;                  // may alias dU_acc's slot 0: this thread alone reads and writes element i
    } else if (i < AD * NTAP + AD) {
        const int d = i - AD * NTAP;
        float s = 0.f;
        for (int b = 0; b < nb; ++b) s += dv_acc[(long long)b * AD + d];
        dv[d] = s;
    }
}
__global__ __launch_bounds__(256) void unfold_location_products_kernel(const float* __restrict__ dUsum, const float* __restrict__ wd,
                                                                       const float* __restrict__ wc, float* __restrict__ dwd,
                                                                       float* __restrict__ dwc) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < AD * T2AMD_LOC_FILTERS) {
        const int d = i / T2AMD_LOC_FILTERS, f = i - d * T2AMD_LOC_FILTERS;
        float s = 0.f;
        for (int ck = 0; ck < NTAP; ++ck) s = fmaf(dUsum[d * NTAP + ck], wc[f * NTAP + ck], s);
        dwd[i] = s;
    } else if (i < AD * T2AMD_LOC_FILTERS + T2AMD_LOC_FILTERS * NTAP) {
        const int o = i - AD * T2AMD_LOC_FILTERS;
        const int f = o / NTAP, ck = o - f * NTAP;
        float s = 0.f;
        for (int d = 0; d < AD; ++d) s = fmaf(wd[d * T2AMD_LOC_FILTERS + f], dUsum[d * NTAP + ck], s);
        dwc[o] = s;
    }
}

extern "C" int t2amd_unfold_location_grads_f32(const float* dU_acc, const float* dv_acc, int nb,
                                               const float* wdense, const float* wconv, float* dwdense,
                                               float* dwconv, float* dv, void* stream) {
    T2_REQUIRE(dU_acc && dv_acc && nb > 0 && wdense && wconv && dwdense && dwconv && dv, "unfold_location: bad args");
    // dU_acc[0] is reused as the reduction target after its own contribution has been read:
    // write the sum into slot 0 (the caller treats dU_acc as scratch after this call).
    float* dUsum = const_cast<float*>(dU_acc);
    T2_LAUNCH(unfold_location_reduce_kernel, dim3(t2_cdiv(AD * NTAP + AD, 256)), dim3(256), 0, (hipStream_t)stream, dU_acc, dv_acc, nb,
              dv, dUsum);
    T2_LAUNCH(unfold_location_products_kernel, dim3(t2_cdiv(AD * T2AMD_LOC_FILTERS + T2AMD_LOC_FILTERS * NTAP, 256)), dim3(256), 0,
              (hipStream_t)stream, dUsum, wdense, wconv, dwdense, dwconv);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}
