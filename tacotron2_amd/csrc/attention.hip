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

#define AD T2AMD_ATT_DIM       // 128
#define NTAP T2AMD_LOC_TAPS    // 62
#define LK T2AMD_LOC_KERNEL    // 31
#define HALO 15
#define NSL T2AMD_ATT_SLICES   // 4 slices per utterance in every kernel
#define DSL (AD / NSL)         // 32 attention dims per K_e / K_b2 workgroup
#define DCL 68                 // dcol_s row stride (floats)
#define DPL 48                 // dpre_s row stride (floats): 48*lg mod 64 = {0,48,32,16}: conflict-free A reads

static inline int attn_tip(int Ti) { return (((Ti + 15) / 16) * 16 + 2 * HALO + 2 + 3) / 4 * 4; }

// LDS offset (without the position) of location tap `tap` = c*31+k in the two-window image
// win_s[2][TIP] (halo coordinates: win_s[c][ti + k] = w_c[ti + k - 15]).  Taps 62, 63 are padding.
__device__ __forceinline__ int tap_offset(int tap, int TIP) {
    if (tap >= NTAP) return 0;
    return tap < LK ? tap : TIP + tap - LK;
}

__device__ __forceinline__ void stage_windows(float* win_s, int TIP, int Ti, const float* wprev, const float* cum,
                                              int tid, int nthreads) {
    for (int i = tid; i < TIP; i += nthreads) {
        const int ti = i - HALO;
        const bool in = (ti >= 0 && ti < Ti);
        win_s[i] = (in && wprev) ? wprev[ti] : 0.f;
        win_s[TIP + i] = in ? cum[ti] : 0.f;
    }
}

// loc^T tile: acc[dt][r] = sum_tap U[dt*16 + 4*lg + r][tap] * win[c(tap)][pos + k(tap)],  pos = lane&15
__device__ __forceinline__ void loc_tile(const float (&ua)[2][16], const float* __restrict__ win_s, int TIP,
                                         int pos, int lg, f32x4& acc0, f32x4& acc1) {
    acc0 = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const float bw = win_s[tap_offset(kk * 4 + lg, TIP) + pos];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[0][kk], bw, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[1][kk], bw, acc1, 0, 0, 0);
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

struct AttnFwdParams { t2amd_attn_fwd a; int tip; };

// ---------------------------------------------------------------------------------------
// K_e: partial energies over 32 attention dims
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_energy_kernel(AttnFwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const t2amd_attn_fwd& a = p.a;
    const int ds = blockIdx.x, b = blockIdx.y;
    if (a.active && !a.active[b]) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int Ti = a.Ti, Hq = a.Hq, TIP = p.tip;
    float* win_s = smem;             // [2][TIP]
    float* q_s = win_s + 2 * TIP;    // [32]
    const int len = a.lens ? a.lens[b] : Ti;

    stage_windows(win_s, TIP, Ti, a.w_prev ? a.w_prev + (long long)b * a.ld_wprev : nullptr,
                  a.cum + (long long)b * Ti, tid, 256);
    {   // q[d] = W_q[d][:] . h   for the slice's 32 dims: 8 threads per row, 128 contiguous bytes per group
        const int d = tid >> 3, part = tid & 7;
        const float4* __restrict__ W4 = reinterpret_cast<const float4*>(a.Wq + (long long)(ds * DSL + d) * Hq);
        const float4* __restrict__ h4 = reinterpret_cast<const float4*>(a.h + (long long)b * a.ld_h);
        float acc = 0.f;
        const int n4 = Hq >> 2;
#pragma unroll 4
        for (int i = part; i < n4; i += 8) {
            const float4 w = W4[i];
            const float4 x = h4[i];
            acc = fmaf(w.x, x.x, acc);
            acc = fmaf(w.y, x.y, acc);
            acc = fmaf(w.z, x.z, acc);
            acc = fmaf(w.w, x.w, acc);
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (part == 0) {
            q_s[d] = acc;
            if (a.q_out) a.q_out[(long long)b * a.ld_q + ds * DSL + d] = acc;
        }
    }
    float ua[2][16];
    load_u_frag(ua, a.U, ds * DSL, l15, lg);
    float vv[2][4];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) vv[dt][r] = a.v[ds * DSL + dt * 16 + 4 * lg + r];
    __syncthreads();
    float qv[2][4];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) qv[dt][r] = q_s[dt * 16 + 4 * lg + r];

    float* __restrict__ eout = a.ws + ((long long)ds * a.B + b) * Ti;
    const float* __restrict__ pmb = a.pm + (long long)b * Ti * AD + ds * DSL + 4 * lg;
    const int nmt = (len + 15) >> 4;
    for (int mt = wv; mt < nmt; mt += 4) {
        const int pos = mt * 16 + l15;
        float4 pm0 = make_float4(0.f, 0.f, 0.f, 0.f), pm1 = pm0;
        if (pos < Ti) {
            pm0 = *reinterpret_cast<const float4*>(pmb + (long long)pos * AD);
            pm1 = *reinterpret_cast<const float4*>(pmb + (long long)pos * AD + 16);
        }
        f32x4 acc0, acc1;
        loc_tile(ua, win_s, TIP, pos, lg, acc0, acc1);
        float e = vv[0][0] * tanhf(acc0[0] + qv[0][0] + pm0.x);
        e = fmaf(vv[0][1], tanhf(acc0[1] + qv[0][1] + pm0.y), e);
        e = fmaf(vv[0][2], tanhf(acc0[2] + qv[0][2] + pm0.z), e);
        e = fmaf(vv[0][3], tanhf(acc0[3] + qv[0][3] + pm0.w), e);
        e = fmaf(vv[1][0], tanhf(acc1[0] + qv[1][0] + pm1.x), e);
        e = fmaf(vv[1][1], tanhf(acc1[1] + qv[1][1] + pm1.y), e);
        e = fmaf(vv[1][2], tanhf(acc1[2] + qv[1][2] + pm1.z), e);
        e = fmaf(vv[1][3], tanhf(acc1[3] + qv[1][3] + pm1.w), e);
        e += __shfl_xor(e, 16, 64);
        e += __shfl_xor(e, 32, 64);
        if (lg == 0 && pos < Ti) eout[pos] = e;
    }
}

// ---------------------------------------------------------------------------------------
// K_c: softmax over the utterance + one quarter of the context channels
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_context_kernel(AttnFwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const t2amd_attn_fwd& a = p.a;
    const int cs = blockIdx.x, b = blockIdx.y;
    if (a.active && !a.active[b]) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int Ti = a.Ti, E = a.E, B = a.B;
    float* w_s = smem;                    // [Ti rounded to 4]
    float* red_s = w_s + ((Ti + 3) & ~3); // [8]
    float* part_s = red_s + 8;            // [parts][EC]
    const int len = a.lens ? a.lens[b] : Ti;

    const float* __restrict__ e0 = a.ws + (long long)b * Ti;
    const long long es = (long long)B * Ti;
    float lmax = -INFINITY;
    for (int ti = tid; ti < Ti; ti += 256) {
        float e = -INFINITY;
        if (ti < len) e = ((e0[ti] + e0[es + ti]) + e0[2 * es + ti]) + e0[3 * es + ti];
        w_s[ti] = e;
        lmax = fmaxf(lmax, e);
    }
    lmax = wave_reduce_max(lmax);
    if (lane == 0) red_s[wv] = lmax;
    __syncthreads();
    const float gmax = fmaxf(fmaxf(red_s[0], red_s[1]), fmaxf(red_s[2], red_s[3]));
    float lsum = 0.f;
    for (int ti = tid; ti < Ti; ti += 256) {
        const float ex = (ti < len) ? expf(w_s[ti] - gmax) : 0.f;
        w_s[ti] = ex;
        lsum += ex;
    }
    lsum = wave_reduce_sum(lsum);
    if (lane == 0) red_s[4 + wv] = lsum;
    __syncthreads();
    const float inv = 1.0f / (((red_s[4] + red_s[5]) + red_s[6]) + red_s[7]);
    {
        float* wout = a.w_out + (long long)b * a.ld_wout;
        float* cum = a.cum + (long long)b * Ti;
        float* csave = a.cum_save ? a.cum_save + (long long)b * Ti : nullptr;
        for (int ti = tid; ti < Ti; ti += 256) {
            const float w = w_s[ti] * inv;
            w_s[ti] = w;
            if (cs == 0) {
                wout[ti] = w;
                const float c_old = cum[ti];
                if (csave) csave[ti] = c_old;
                cum[ti] = c_old + w;
            }
        }
    }
    __syncthreads();
    // context channels [cs*EC, (cs+1)*EC)
    const int EC = E / NSL, EC4 = EC >> 2, E4 = E >> 2;
    int parts = 256 / EC4;
    if (parts > 16) parts = 16;
    const int c4 = tid % EC4, part = tid / EC4;
    if (part < parts) {
        const float4* __restrict__ M4 = reinterpret_cast<const float4*>(a.memory) + (long long)b * Ti * E4 + cs * EC4 + c4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int ti = part;
        for (; ti + 3 * parts < len; ti += 4 * parts) {
            const float4 m0 = M4[(long long)ti * E4];
            const float4 m1 = M4[(long long)(ti + parts) * E4];
            const float4 m2 = M4[(long long)(ti + 2 * parts) * E4];
            const float4 m3 = M4[(long long)(ti + 3 * parts) * E4];
            const float w0 = w_s[ti], w1 = w_s[ti + parts], w2 = w_s[ti + 2 * parts], w3 = w_s[ti + 3 * parts];
            acc.x = fmaf(w0, m0.x, acc.x); acc.y = fmaf(w0, m0.y, acc.y); acc.z = fmaf(w0, m0.z, acc.z); acc.w = fmaf(w0, m0.w, acc.w);
            acc.x = fmaf(w1, m1.x, acc.x); acc.y = fmaf(w1, m1.y, acc.y); acc.z = fmaf(w1, m1.z, acc.z); acc.w = fmaf(w1, m1.w, acc.w);
            acc.x = fmaf(w2, m2.x, acc.x); acc.y = fmaf(w2, m2.y, acc.y); acc.z = fmaf(w2, m2.z, acc.z); acc.w = fmaf(w2, m2.w, acc.w);
            acc.x = fmaf(w3, m3.x, acc.x); acc.y = fmaf(w3, m3.y, acc.y); acc.z = fmaf(w3, m3.z, acc.z); acc.w = fmaf(w3, m3.w, acc.w);
        }
        for (; ti < len; ti += parts) {
            const float4 m = M4[(long long)ti * E4];
            const float w = w_s[ti];
            acc.x = fmaf(w, m.x, acc.x); acc.y = fmaf(w, m.y, acc.y); acc.z = fmaf(w, m.z, acc.z); acc.w = fmaf(w, m.w, acc.w);
        }
        *reinterpret_cast<float4*>(&part_s[part * EC + c4 * 4]) = acc;
    }
    __syncthreads();
    for (int c = tid; c < EC; c += 256) {
        float s = 0.f;
        for (int q = 0; q < parts; ++q) s += part_s[q * EC + c];
        a.ctx_out[(long long)b * a.ld_ctx + cs * EC + c] = s;
    }
}

extern "C" int t2amd_attention_step_fwd_f32(const t2amd_attn_fwd* a, void* stream) {
    T2_REQUIRE(a && a->h && a->Wq && a->U && a->v && a->pm && a->memory && a->cum && a->w_out && a->ctx_out && a->ws,
               "attn_fwd: null pointer");
    T2_REQUIRE(a->B > 0 && a->Ti > 0 && a->Ti <= 8192, "attn_fwd: Ti out of range");
    T2_REQUIRE(a->E % (4 * NSL) == 0 && a->E >= 4 * NSL && a->E <= 4096, "attn_fwd: E must be a multiple of 16, <= 4096");
    T2_REQUIRE(a->Hq % 32 == 0 && a->Hq > 0, "attn_fwd: Hq must be a multiple of 32");
    T2_REQUIRE(t2_aligned16(a->Wq) && t2_aligned16(a->memory) && t2_aligned16(a->pm) && t2_aligned16(a->h) &&
                   a->ld_h % 4 == 0,
               "attn_fwd: Wq/memory/pm/h must be 16-byte aligned");
    AttnFwdParams p;
    p.a = *a;
    p.tip = attn_tip(a->Ti);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds_e = sizeof(float) * (2 * (size_t)p.tip + DSL);
    const int EC = a->E / NSL;
    int parts = 256 / (EC / 4);
    if (parts > 16) parts = 16;
    const size_t lds_c = sizeof(float) * ((size_t)((a->Ti + 3) & ~3) + 8 + (size_t)parts * EC);
    T2_REQUIRE(lds_e <= 64 * 1024 && lds_c <= 64 * 1024, "attn_fwd: Ti too large for the LDS windows");
    T2_LAUNCH(attn_energy_kernel, dim3(NSL, a->B), dim3(256), lds_e, s, p);
    T2_LAUNCH(attn_context_kernel, dim3(NSL, a->B), dim3(256), lds_c, s, p);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// =========================================================================================
// Backward of one attention step.
// =========================================================================================
struct AttnBwdParams { t2amd_attn_bwd a; int tip; int np; };

// K_b1: dctx, dw[ti] = dctx . memory[ti] + carries, partial sum_ti w dw over a quarter of the positions
__global__ __launch_bounds__(256) void attn_bwd_dw_kernel(AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const t2amd_attn_bwd& a = p.a;
    const int ts = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int Ti = a.Ti, E = a.E, B = a.B;
    float* dctx_s = smem;          // [E]
    float* red_s = dctx_s + E;     // [4]
    const int len = a.lens ? a.lens[b] : Ti;
    for (int c = tid; c < E; c += 256) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const t2amd_addend& ad = a.dctx[i];
            if (ad.p) {
                const float* q = ad.p + (long long)b * ad.ld + c;
                for (int k = 0; k < ad.nsplit; ++k) s += q[(long long)k * ad.split_stride];
            }
        }
        dctx_s[c] = s;
        if (ts == 0) a.dctx_total[(long long)b * a.ld_dctx_total + c] = s;
    }
    __syncthreads();
    const int tsz = (Ti + NSL - 1) / NSL;
    const int t0 = ts * tsz;
    int t1 = t0 + tsz;
    if (t1 > Ti) t1 = Ti;
    const int E4 = E >> 2;
    const float4* __restrict__ M4 = reinterpret_cast<const float4*>(a.memory) + (long long)b * Ti * E4;
    const float* __restrict__ wrow = a.w + (long long)b * a.ld_w;
    const float* __restrict__ dwx = a.d_w_extra ? a.d_w_extra + (long long)b * a.ld_dwextra : nullptr;
    const long long ps = (long long)B * 2 * Ti;                 // stride between dim-slice partials
    const float* __restrict__ cw = a.dwin_part + ((long long)b * 2 + 0) * Ti;
    const float* __restrict__ cc = a.dwin_part + ((long long)b * 2 + 1) * Ti;
    float* __restrict__ dcum = a.dcum_acc + (long long)b * Ti;
    float* __restrict__ dwo = a.ws + (long long)b * Ti;
    float psum = 0.f;
    for (int ti = t0 + wv; ti < t1; ti += 4) {
        float s = 0.f;
        if (ti < len) {
            for (int c4 = lane; c4 < E4; c4 += 64) {
                const float4 m = M4[(long long)ti * E4 + c4];
                const float4 g = *reinterpret_cast<const float4*>(&dctx_s[c4 * 4]);
                s = fmaf(m.x, g.x, s);
                s = fmaf(m.y, g.y, s);
                s = fmaf(m.z, g.z, s);
                s = fmaf(m.w, g.w, s);
            }
            s = wave_reduce_sum(s);
        }
        if (lane == 0) {
            const float carry_w = ((cw[ti] + cw[ps + ti]) + cw[2 * ps + ti]) + cw[3 * ps + ti];
            const float carry_c = ((cc[ti] + cc[ps + ti]) + cc[2 * ps + ti]) + cc[3 * ps + ti];
            const float dc = dcum[ti] + carry_c;
            dcum[ti] = dc;
            float dw = s + carry_w + dc;
            if (dwx) dw += dwx[ti];
            dwo[ti] = dw;
            psum = fmaf(wrow[ti], dw, psum);
        }
    }
    if (lane == 0) red_s[wv] = psum;
    __syncthreads();
    if (tid == 0) a.ws[(long long)B * Ti + (long long)ts * B + b] = ((red_s[0] + red_s[1]) + red_s[2]) + red_s[3];
}

// K_b2: everything that lives in attention-dim space, for 32 dims
__global__ __launch_bounds__(256) void attn_bwd_main_kernel(AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const t2amd_attn_bwd& a = p.a;
    const int ds = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int Ti = a.Ti, Hq = a.Hq, B = a.B, TIP = p.tip, NP = p.np;
    float* win_s = smem;                      // [2][TIP]
    float* de_s = win_s + 2 * TIP;            // [NP]
    float* dcol_s = de_s + NP;                // [NP][DCL]
    float* dpre_s = dcol_s + (size_t)NP * DCL;  // [NP][DPL]
    float* red_s = dpre_s + (size_t)NP * DPL;   // [4 waves][2][32]
    float* dq_s = red_s + 4 * 2 * DSL;        // [32]
    const int len = a.lens ? a.lens[b] : Ti;
    const int nmt = (len + 15) >> 4;
    const int npos = nmt * 16;                // positions covered by the MFMA tiles

    {
        const float* sd = a.ws + (long long)B * Ti;
        const float sdot = ((sd[b] + sd[B + b]) + sd[2 * B + b]) + sd[3 * B + b];
        const float* __restrict__ wrow = a.w + (long long)b * a.ld_w;
        const float* __restrict__ dwi = a.ws + (long long)b * Ti;
        for (int ti = tid; ti < NP; ti += 256) de_s[ti] = (ti < len) ? wrow[ti] * (dwi[ti] - sdot) : 0.f;
    }
    stage_windows(win_s, TIP, Ti, a.w_prev ? a.w_prev + (long long)b * a.ld_wprev : nullptr,
                  a.cum_before + (long long)b * Ti, tid, 256);

    const int dbase = ds * DSL;
    float ua[2][16];
    load_u_frag(ua, a.U, dbase, l15, lg);
    // U^T as the A operand of dcol^T = U^T dpre: A[i = tap][k = lg], k-step (dt, r) <-> dim dt*16 + 4*lg + r
    float ut[4][2][4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tap = tt * 16 + l15;
                ut[tt][dt][r] = tap < NTAP ? a.U[(long long)(dbase + dt * 16 + 4 * lg + r) * NTAP + tap] : 0.f;
            }
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
    __syncthreads();

    const float* __restrict__ pmb = a.pm + (long long)b * Ti * AD + dbase + 4 * lg;
    float* __restrict__ dpmb = a.d_pm + (long long)b * Ti * AD + dbase + 4 * lg;
    for (int mt = wv; mt < nmt; mt += 4) {
        const int pos = mt * 16 + l15;
        float4 pm0 = make_float4(0.f, 0.f, 0.f, 0.f), pm1 = pm0;
        if (pos < Ti) {
            pm0 = *reinterpret_cast<const float4*>(pmb + (long long)pos * AD);
            pm1 = *reinterpret_cast<const float4*>(pmb + (long long)pos * AD + 16);
        }
        f32x4 acc0, acc1;
        loc_tile(ua, win_s, TIP, pos, lg, acc0, acc1);
        const float de = de_s[pos];
        float dp[2][4];
        const float pmv[2][4] = {{pm0.x, pm0.y, pm0.z, pm0.w}, {pm1.x, pm1.y, pm1.z, pm1.w}};
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = (dt ? acc1[r] : acc0[r]) + qv[dt][r] + pmv[dt][r];
                const float th = tanhf(x);
                const float g = de * vv[dt][r] * (1.f - th * th);
                dva[dt][r] = fmaf(de, th, dva[dt][r]);
                dqa[dt][r] += g;
                dp[dt][r] = g;
            }
        if (pos < len) {
            float4 o0 = *reinterpret_cast<float4*>(dpmb + (long long)pos * AD);
            float4 o1 = *reinterpret_cast<float4*>(dpmb + (long long)pos * AD + 16);
            o0.x += dp[0][0]; o0.y += dp[0][1]; o0.z += dp[0][2]; o0.w += dp[0][3];
            o1.x += dp[1][0]; o1.y += dp[1][1]; o1.z += dp[1][2]; o1.w += dp[1][3];
            *reinterpret_cast<float4*>(dpmb + (long long)pos * AD) = o0;
            *reinterpret_cast<float4*>(dpmb + (long long)pos * AD + 16) = o1;
        }
        // dcol^T[tap][pos] = sum_d U[d][tap] dpre[d][pos]: B operand = this lane's own dpre registers
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) c = __builtin_amdgcn_mfma_f32_16x16x4f32(ut[tt][dt][r], dp[dt][r], c, 0, 0, 0);
            *reinterpret_cast<float4*>(&dcol_s[(size_t)pos * DCL + tt * 16 + 4 * lg]) = make_float4(c[0], c[1], c[2], c[3]);
        }
        *reinterpret_cast<float4*>(&dpre_s[(size_t)pos * DPL + 4 * lg]) = make_float4(dp[0][0], dp[0][1], dp[0][2], dp[0][3]);
        *reinterpret_cast<float4*>(&dpre_s[(size_t)pos * DPL + 16 + 4 * lg]) = make_float4(dp[1][0], dp[1][1], dp[1][2], dp[1][3]);
    }
    // dv / dq: reduce over the positions held by the 16 lanes of a lane group
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = dva[dt][r], y = dqa[dt][r];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                x += __shfl_xor(x, off, 64);
                y += __shfl_xor(y, off, 64);
            }
            if (l15 == 0) {
                red_s[(wv * 2 + 0) * DSL + dt * 16 + 4 * lg + r] = x;
                red_s[(wv * 2 + 1) * DSL + dt * 16 + 4 * lg + r] = y;
            }
        }
    __syncthreads();
    if (tid < DSL) {
        const float dvs = ((red_s[0 * DSL + tid] + red_s[2 * DSL + tid]) + red_s[4 * DSL + tid]) + red_s[6 * DSL + tid];
        const float dqs = ((red_s[1 * DSL + tid] + red_s[3 * DSL + tid]) + red_s[5 * DSL + tid]) + red_s[7 * DSL + tid];
        a.dv_acc[(long long)b * AD + dbase + tid] += dvs;
        dq_s[tid] = dqs;
        a.dq_out[(long long)b * a.ld_dq + dbase + tid] = dqs;
    }
    // dU[d][tap] += sum_pos dpre[pos][d] * win[c(tap)][pos + k(tap)]   (wave w owns tap tile w)
    {
        const int tt = wv;
        const int tap = tt * 16 + l15;
        const int toff = tap_offset(tap, TIP);
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < npos; s += 4) {
            const float a0 = dpre_s[(size_t)(s + lg) * DPL + l15];
            const float a1 = dpre_s[(size_t)(s + lg) * DPL + 16 + l15];
            const float bw = win_s[toff + s + lg];
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bw, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bw, c1, 0, 0, 0);
        }
        if (tap < NTAP) {
            float* dUg = a.dU_acc + ((long long)b * AD + dbase) * NTAP + tap;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dUg[(long long)(4 * lg + r) * NTAP] += c0[r];
                dUg[(long long)(16 + 4 * lg + r) * NTAP] += c1[r];
            }
        }
    }
    // col2im: partial carry dwin[c][ti'] = sum_k dcol[ti' - k + 15][c*31 + k] over this slice's dims
    {
        float* __restrict__ out = a.dwin_part + (((long long)ds * B + b) * 2) * Ti;
        for (int i = tid; i < 2 * Ti; i += 256) {
            const int c = i >= Ti;
            const int tip_ = i - c * Ti;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < LK; ++k) {
                const int row = tip_ - k + HALO;
                if (row >= 0 && row < npos) s += dcol_s[(size_t)row * DCL + c * LK + k];
            }
            out[i] = s;
        }
    }
    __syncthreads();   // dq_s
    // partial dh = sum_{d in slice} dq[d] * W_q[d][:]
    {
        const int H4 = Hq >> 2;
        const float4* __restrict__ W4 = reinterpret_cast<const float4*>(a.Wq) + (long long)dbase * H4;
        float* __restrict__ dh = a.dh_out + (long long)ds * a.dh_split_stride + (long long)b * a.ld_dh;
        for (int k4 = tid; k4 < H4; k4 += 256) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
            for (int dd = 0; dd < DSL; ++dd) {
                const float g = dq_s[dd];
                const float4 w = W4[(long long)dd * H4 + k4];
                acc.x = fmaf(g, w.x, acc.x);
                acc.y = fmaf(g, w.y, acc.y);
                acc.z = fmaf(g, w.z, acc.z);
                acc.w = fmaf(g, w.w, acc.w);
            }
            *reinterpret_cast<float4*>(dh + k4 * 4) = acc;
        }
    }
}

static int g_attn_bwd_lds = 0;

extern "C" int t2amd_attention_step_bwd_f32(const t2amd_attn_bwd* a, void* stream) {
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
    hipStream_t s = (hipStream_t)stream;
    const size_t lds1 = sizeof(float) * ((size_t)a->E + 4);
    const size_t lds2 = sizeof(float) * (2 * (size_t)p.tip + p.np + (size_t)p.np * (DCL + DPL) + 8 * DSL + DSL);
    T2_REQUIRE(lds1 <= 64 * 1024, "attn_bwd: E too large");
    T2_REQUIRE(lds2 <= 160 * 1024, "attn_bwd: Ti needs more than 160 KiB of LDS");
    if ((int)lds2 > 64 * 1024 && (int)lds2 > g_attn_bwd_lds && !t2amd_validate_only_flag_()) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_main_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        g_attn_bwd_lds = (int)lds2;
    }
    T2_LAUNCH(attn_bwd_dw_kernel, dim3(NSL, a->B), dim3(256), lds1, s, p);
    T2_LAUNCH(attn_bwd_main_kernel, dim3(NSL, a->B), dim3(256), lds2, s, p);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

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

__global__ void unfold_location_kernel(const float* __restrict__ dU_acc, const float* __restrict__ dv_acc,
                                       int nb, const float* __restrict__ wd, const float* __restrict__ wc,
                                       float* __restrict__ dwd, float* __restrict__ dwc, float* __restrict__ dv,
                                       float* __restrict__ dUsum) {
    // single workgroup: first reduce dU over utterances, then the two small products
    const int tid = threadIdx.x;
    for (int i = tid; i < AD * NTAP; i += blockDim.x) {
        float s = 0.f;
        for (int b = 0; b < nb; ++b) s += dU_acc[(long long)b * AD * NTAP + i];
        dUsum[i] = s;
    }
    for (int i = tid; i < AD; i += blockDim.x) {
        float s = 0.f;
        for (int b = 0; b < nb; ++b) s += dv_acc[(long long)b * AD + i];
        dv[i] = s;
    }
    __syncthreads();
    for (int i = tid; i < AD * T2AMD_LOC_FILTERS; i += blockDim.x) {
        const int d = i / T2AMD_LOC_FILTERS, f = i - d * T2AMD_LOC_FILTERS;
        float s = 0.f;
        for (int ck = 0; ck < NTAP; ++ck) s = fmaf(dUsum[d * NTAP + ck], wc[f * NTAP + ck], s);
        dwd[i] = s;
    }
    for (int i = tid; i < T2AMD_LOC_FILTERS * NTAP; i += blockDim.x) {
        const int f = i / NTAP, ck = i - f * NTAP;
        float s = 0.f;
        for (int d = 0; d < AD; ++d) s = fmaf(wd[d * T2AMD_LOC_FILTERS + f], dUsum[d * NTAP + ck], s);
        dwc[i] = s;
    }
}

extern "C" int t2amd_unfold_location_grads_f32(const float* dU_acc, const float* dv_acc, int nb,
                                               const float* wdense, const float* wconv, float* dwdense,
                                               float* dwconv, float* dv, void* stream) {
    T2_REQUIRE(dU_acc && dv_acc && nb > 0 && wdense && wconv && dwdense && dwconv && dv, "unfold_location: bad args");
    // dU_acc[0] is reused as the reduction target after its own contribution has been read:
    // write the sum into slot 0 (the caller treats dU_acc as scratch after this call).
    T2_LAUNCH(unfold_location_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, dU_acc, dv_acc, nb,
                       wdense, wconv, dwdense, dwconv, dv, const_cast<float*>(dU_acc));
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}
