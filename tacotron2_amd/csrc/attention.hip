// Location-sensitive attention step kernels for gfx950 (one workgroup per utterance).
//
// Reference arithmetic (model.py:43-86, 22-26, 358-365):
//   q      = W_q h_att                                   (128)
//   loc    = Dense(Conv1d([w_prev ; w_cum]))             (Ti x 128)   2->32 ch, k=31, then 32->128
//   e[ti]  = v . tanh(q + loc[ti] + processed_memory[ti])
//   w      = softmax(mask(e));  ctx = w @ memory;  w_cum += w
// The conv and the dense layer are folded once per optimiser step into one 62-tap filter per
// attention dim, U[d][c*31+k] = sum_f Wd[d][f] Wc[f][c][k] (t2amd_fold_location_f32): a lane owns
// one attention dim d and keeps U[d][:] in 62 VGPRs, the two weight windows sit in LDS with a
// 15-element zero halo, and a wave evaluates one text position at a time (64 dims, 2 waves per
// position), so the energy reduction over d is a wave reduction and processed_memory /
// memory rows are read as whole coalesced lines.
#include "common.h"

#define AD T2AMD_ATT_DIM       // 128
#define NTAP T2AMD_LOC_TAPS    // 62
#define LK T2AMD_LOC_KERNEL    // 31
#define HALO 15

__device__ __forceinline__ float attn_preact(const float (&u)[NTAP], float qd, float pmv,
                                             const float* __restrict__ w0, const float* __restrict__ w1) {
    // w0/w1 point at the window start (position ti - 15 in halo coordinates)
    float acc = qd + pmv;
#pragma unroll
    for (int k = 0; k < LK; ++k) acc = fmaf(u[k], w0[k], acc);
#pragma unroll
    for (int k = 0; k < LK; ++k) acc = fmaf(u[LK + k], w1[k], acc);
    return acc;
}

struct AttnFwdParams { t2amd_attn_fwd a; int tip; int scratch; };

__global__ __launch_bounds__(1024) void attn_fwd_kernel(AttnFwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const t2amd_attn_fwd& a = p.a;
    const int b = blockIdx.x;
    if (a.active && !a.active[b]) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int Ti = a.Ti, E = a.E, Hq = a.Hq, TIP = p.tip;

    float* h_s = smem;                       // [Hq]
    float* scratch = h_s + Hq;               // [max(32*128, 8*E)]
    float* q_s = scratch + p.scratch;        // [128]
    float* win_s = q_s + AD;                 // [2][TIP]
    float* e_s = win_s + 2 * TIP;            // [2*Ti]
    float* red_s = e_s + 2 * Ti;             // [32]

    // ---- stage 0: stage h and the two weight windows --------------------------------------
    const float* hrow = a.h + (long long)b * a.ld_h;
    for (int k = tid; k < Hq; k += 1024) h_s[k] = hrow[k];
    const float* wprev = a.w_prev ? a.w_prev + (long long)b * a.ld_wprev : nullptr;
    float* cum = a.cum + (long long)b * Ti;
    for (int i = tid; i < TIP; i += 1024) {
        const int ti = i - HALO;
        const bool in = (ti >= 0 && ti < Ti);
        win_s[i] = (in && wprev) ? wprev[ti] : 0.f;
        win_s[TIP + i] = in ? cum[ti] : 0.f;
    }
    __syncthreads();

    // ---- stage 1: q = W_q h  (WqT is [Hq][128]) --------------------------------------------
    {
        const int d4 = tid & 31, part = tid >> 5;          // 32 parts x 32 float4
        const int kper = Hq / 32;
        const float4* __restrict__ W4 = reinterpret_cast<const float4*>(a.WqT);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int k0 = part * kper;
#pragma unroll 8
        for (int k = 0; k < kper; ++k) {
            const float hv = h_s[k0 + k];
            const float4 w = W4[(long long)(k0 + k) * (AD / 4) + d4];
            acc.x = fmaf(hv, w.x, acc.x);
            acc.y = fmaf(hv, w.y, acc.y);
            acc.z = fmaf(hv, w.z, acc.z);
            acc.w = fmaf(hv, w.w, acc.w);
        }
        *reinterpret_cast<float4*>(&scratch[part * AD + d4 * 4]) = acc;
    }
    __syncthreads();
    if (tid < AD) {
        float s = 0.f;
#pragma unroll
        for (int part = 0; part < 32; ++part) s += scratch[part * AD + tid];
        q_s[tid] = s;
        if (a.q_out) a.q_out[(long long)b * a.ld_q + tid] = s;
    }
    __syncthreads();

    // ---- stage 2: energies ------------------------------------------------------------------
    {
        const int half = wv & 1;
        const int d = half * 64 + lane;
        float u[NTAP];
        const float* urow = a.U + (long long)d * NTAP;
#pragma unroll
        for (int j = 0; j < NTAP; ++j) u[j] = urow[j];
        const float vd = a.v[d];
        const float qd = q_s[d];
        const float* pmb = a.pm + (long long)b * Ti * AD + d;
        for (int ti = (wv >> 1); ti < Ti; ti += 8) {
            const float pmv = pmb[(long long)ti * AD];
            const float acc = attn_preact(u, qd, pmv, win_s + ti, win_s + TIP + ti);
            float e = vd * tanhf(acc);
            e = wave_reduce_sum(e);
            if (lane == 0) e_s[2 * ti + half] = e;
        }
    }
    __syncthreads();

    // ---- stage 3: masked softmax over Ti ---------------------------------------------------
    const int len = a.lens ? a.lens[b] : Ti;
    float lmax = -INFINITY;
    for (int ti = tid; ti < Ti; ti += 1024) {
        float e = e_s[2 * ti] + e_s[2 * ti + 1];
        if (ti >= len) e = -INFINITY;
        e_s[2 * ti] = e;
        lmax = fmaxf(lmax, e);
    }
    lmax = wave_reduce_max(lmax);
    if (lane == 0) red_s[wv] = lmax;
    __syncthreads();
    float gmax = red_s[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) gmax = fmaxf(gmax, red_s[i]);
    float lsum = 0.f;
    for (int ti = tid; ti < Ti; ti += 1024) {
        const float ex = (ti < len) ? expf(e_s[2 * ti] - gmax) : 0.f;
        e_s[2 * ti + 1] = ex;
        lsum += ex;
    }
    lsum = wave_reduce_sum(lsum);
    if (lane == 0) red_s[16 + wv] = lsum;
    __syncthreads();
    float gsum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) gsum += red_s[16 + i];
    const float inv = 1.0f / gsum;

    // ---- stage 4: weights out, cumulative update -------------------------------------------
    float* wout = a.w_out + (long long)b * a.ld_wout;
    float* csave = a.cum_save ? a.cum_save + (long long)b * Ti : nullptr;
    for (int ti = tid; ti < Ti; ti += 1024) {
        const float w = e_s[2 * ti + 1] * inv;
        e_s[2 * ti] = w;
        wout[ti] = w;
        const float c_old = win_s[TIP + HALO + ti];
        if (csave) csave[ti] = c_old;
        cum[ti] = c_old + w;
    }
    __syncthreads();

    // ---- stage 5: context = w @ memory -----------------------------------------------------
    {
        const int E4 = E >> 2;
        const int parts = 1024 / E4 > 8 ? 8 : 1024 / E4;
        const int c4 = tid % E4, part = tid / E4;
        if (part < parts) {
            const float4* __restrict__ M4 = reinterpret_cast<const float4*>(a.memory) + (long long)b * Ti * E4 + c4;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
            for (int ti = part; ti < len; ti += parts) {
                const float w = e_s[2 * ti];
                const float4 m = M4[(long long)ti * E4];
                acc.x = fmaf(w, m.x, acc.x);
                acc.y = fmaf(w, m.y, acc.y);
                acc.z = fmaf(w, m.z, acc.z);
                acc.w = fmaf(w, m.w, acc.w);
            }
            *reinterpret_cast<float4*>(&scratch[part * E + c4 * 4]) = acc;
        }
        __syncthreads();
        for (int c = tid; c < E; c += 1024) {
            float s = 0.f;
            for (int q = 0; q < parts; ++q) s += scratch[q * E + c];
            a.ctx_out[(long long)b * a.ld_ctx + c] = s;
        }
    }
}

static int g_attn_attr_done = 0;

extern "C" int t2amd_attention_step_fwd_f32(const t2amd_attn_fwd* a, void* stream) {
    T2_REQUIRE(a && a->h && a->WqT && a->U && a->v && a->pm && a->memory && a->cum && a->w_out && a->ctx_out,
               "attn_fwd: null pointer");
    T2_REQUIRE(a->B > 0 && a->Ti > 0 && a->Ti <= 4096, "attn_fwd: Ti out of range");
    T2_REQUIRE(a->E % 4 == 0 && a->E >= 4 && a->E <= 4096 && a->Hq % 32 == 0 && a->Hq <= 4096, "attn_fwd: bad E/Hq");
    T2_REQUIRE(t2_aligned16(a->WqT) && t2_aligned16(a->memory), "attn_fwd: WqT/memory must be 16-byte aligned");
    AttnFwdParams p;
    p.a = *a;
    p.tip = ((a->Ti + 2 * HALO + 3) / 4) * 4;
    const int E4 = a->E / 4;
    int parts = 1024 / E4 > 8 ? 8 : 1024 / E4;
    T2_REQUIRE(parts >= 1, "attn_fwd: E too large");
    int scratch = parts * a->E;
    if (scratch < 32 * AD) scratch = 32 * AD;
    p.scratch = scratch;
    const size_t lds = sizeof(float) * ((size_t)a->Hq + scratch + AD + 2 * p.tip + 2 * a->Ti + 32);
    T2_REQUIRE(lds <= 160 * 1024, "attn_fwd: Ti/Hq/E need more than 160 KiB of LDS");
    if (lds > 64 * 1024)
        if (!t2amd_validate_only_flag_()) (void)hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    T2_LAUNCH(attn_fwd_kernel, dim3(a->B), dim3(1024), lds, (hipStream_t)stream, p);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// =========================================================================================
// Backward of one attention step.
// =========================================================================================
#define BT 512          // threads
#define BW 8            // waves
#define TC 64           // text positions per chunk
#define DP_LD 136       // dpre_s row stride (8 mod 64 banks: conflict-free b128 fragment reads)
#define DC_LD 68

struct AttnBwdParams { t2amd_attn_bwd a; const float* UT; int tip; };

__global__ __launch_bounds__(BT) void attn_bwd_kernel(AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const t2amd_attn_bwd& a = p.a;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int Ti = a.Ti, E = a.E, Hq = a.Hq, TIP = p.tip;

    float* dpre_s = smem;                         // [TC][DP_LD]   (later aliased as dU reduce [128][63])
    float* dcol_s = dpre_s + TC * DP_LD;          // [TC][DC_LD]
    float* dctx_s = dcol_s + TC * DC_LD;          // [E]
    float* win_s = dctx_s + E;                    // [2][TIP]
    float* dwin_s = win_s + 2 * TIP;              // [2][TIP]
    float* w_s = dwin_s + 2 * TIP;                // [Ti]
    float* de_s = w_s + Ti;                       // [Ti]
    float* dq_s = de_s + Ti;                      // [128]
    float* red_s = dq_s + AD;                     // [32]

    const int len = a.lens ? a.lens[b] : Ti;

    // ---- a. total context gradient, windows ------------------------------------------------
    for (int c = tid; c < E; c += BT) {
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
        a.dctx_total[(long long)b * a.ld_dctx_total + c] = s;
    }
    const float* wrow = a.w + (long long)b * a.ld_w;
    const float* wprev = a.w_prev ? a.w_prev + (long long)b * a.ld_wprev : nullptr;
    const float* cumb = a.cum_before + (long long)b * Ti;
    for (int i = tid; i < TIP; i += BT) {
        const int ti = i - HALO;
        const bool in = (ti >= 0 && ti < Ti);
        win_s[i] = (in && wprev) ? wprev[ti] : 0.f;
        win_s[TIP + i] = in ? cumb[ti] : 0.f;
        dwin_s[i] = 0.f;
        dwin_s[TIP + i] = 0.f;
    }
    for (int ti = tid; ti < Ti; ti += BT) w_s[ti] = wrow[ti];
    __syncthreads();

    // ---- b. dw[ti] = dctx . memory[ti] + carries ---------------------------------------------
    {
        const int E4 = E >> 2;
        const float4* __restrict__ M4 = reinterpret_cast<const float4*>(a.memory) + (long long)b * Ti * E4;
        const float* dwc = a.dw_carry + (long long)b * Ti;
        const float* dcc = a.dcum_carry + (long long)b * Ti;
        const float* dwx = a.d_w_extra ? a.d_w_extra + (long long)b * a.ld_dwextra : nullptr;
        for (int ti = wv; ti < Ti; ti += BW) {
            float s = 0.f;
            if (ti < len) {
                for (int c4 = lane; c4 < E4; c4 += 64) {
                    const float4 m = M4[(long long)ti * E4 + c4];
                    const float4 g = *reinterpret_cast<const float4*>(&dctx_s[c4 * 4]);
                    s += m.x * g.x + m.y * g.y + m.z * g.z + m.w * g.w;
                }
                s = wave_reduce_sum(s);
            }
            if (lane == 0) {
                float extra = dwc[ti] + dcc[ti];
                if (dwx) extra += dwx[ti];
                de_s[ti] = s + extra;      // holds dw for now
            }
        }
    }
    __syncthreads();
    // softmax backward: de = w * (dw - sum(w*dw))
    float part = 0.f;
    for (int ti = tid; ti < Ti; ti += BT) part += w_s[ti] * de_s[ti];
    part = wave_reduce_sum(part);
    if (lane == 0) red_s[wv] = part;
    __syncthreads();
    float sdot = 0.f;
#pragma unroll
    for (int i = 0; i < BW; ++i) sdot += red_s[i];
    for (int ti = tid; ti < Ti; ti += BT) de_s[ti] = w_s[ti] * (de_s[ti] - sdot);
    __syncthreads();

    // ---- c. chunks of TC positions ----------------------------------------------------------
    const int half = wv & 1;
    const int d = half * 64 + lane;
    float u[NTAP], dU[NTAP];
    {
        const float* urow = a.U + (long long)d * NTAP;
#pragma unroll
        for (int j = 0; j < NTAP; ++j) { u[j] = urow[j]; dU[j] = 0.f; }
    }
    const float vd = a.v[d];
    const float qd = a.q[(long long)b * a.ld_q + d];
    float dv_acc = 0.f, dq_acc = 0.f;
    const float* pmb = a.pm + (long long)b * Ti * AD + d;
    float* dpmb = a.d_pm + (long long)b * Ti * AD + d;
    const int l15 = lane & 15, lg = lane >> 4;

    for (int c0 = 0; c0 < Ti; c0 += TC) {
        // stage A: recompute pre-activation, dpre; 128 (ti, half) tasks over 8 waves
        for (int it = 0; it < TC / (BW / 2); ++it) {
            const int tl = (wv >> 1) + (BW / 2) * it;
            const int ti = c0 + tl;
            float dpre = 0.f;
            if (ti < len) {
                const float* w0 = win_s + ti;
                const float* w1 = win_s + TIP + ti;
                const float acc = attn_preact(u, qd, pmb[(long long)ti * AD], w0, w1);
                const float th = tanhf(acc);
                const float de = de_s[ti];
                dv_acc = fmaf(de, th, dv_acc);
                dpre = de * vd * (1.f - th * th);
                dq_acc += dpre;
#pragma unroll
                for (int k = 0; k < LK; ++k) dU[k] = fmaf(dpre, w0[k], dU[k]);
#pragma unroll
                for (int k = 0; k < LK; ++k) dU[LK + k] = fmaf(dpre, w1[k], dU[LK + k]);
                dpmb[(long long)ti * AD] += dpre;
            }
            dpre_s[tl * DP_LD + d] = dpre;
        }
        __syncthreads();
        // stage B: dcol[TC][64] = dpre[TC][128] . UT[64][128]^T on MFMA (2 column tiles per wave)
        {
            const int rt = wv >> 1;
            const int ct0 = (wv & 1) * 2;
            f32x4 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < AD / 16; ++s) {
                const float4 av = *reinterpret_cast<const float4*>(&dpre_s[(rt * 16 + l15) * DP_LD + s * 16 + lg * 4]);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const float4 bv = *reinterpret_cast<const float4*>(
                        p.UT + (long long)((ct0 + ct) * 16 + l15) * AD + s * 16 + lg * 4);
                    acc[ct][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[ct][0], 0, 0, 0);
                    acc[ct][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[ct][1], 0, 0, 0);
                    acc[ct][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[ct][0], 0, 0, 0);
                    acc[ct][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[ct][1], 0, 0, 0);
                }
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    dcol_s[(rt * 16 + lg * 4 + r) * DC_LD + (ct0 + ct) * 16 + l15] = acc[ct][0][r] + acc[ct][1][r];
        }
        __syncthreads();
        // stage C: col2im — dwin[c][ti'] += sum_k dcol[ti'-k+15][c*31+k]
        if (tid < 2 * (TC + 2 * HALO)) {
            const int c = tid / (TC + 2 * HALO);
            const int off = tid - c * (TC + 2 * HALO);     // ti' = c0 - 15 + off
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < LK; ++k) {
                const int row = off - k;                   // = ti' - k + 15 - c0
                if (row >= 0 && row < TC) s += dcol_s[row * DC_LD + c * LK + k];
            }
            const int idx = c0 + off;                      // halo coordinates (ti' + 15)
            if (idx < TIP) dwin_s[c * TIP + idx] += s;
        }
        __syncthreads();
    }

    // ---- d. reduce per-lane accumulators over the 4 waves of each half (fixed order) -------
    float* dUr = dpre_s;       // [128][63]
    for (int rank = 0; rank < BW / 2; ++rank) {
        if ((wv >> 1) == rank) {
            float* row = dUr + d * 63;
            if (rank == 0) {
#pragma unroll
                for (int j = 0; j < NTAP; ++j) row[j] = dU[j];
                row[NTAP] = dv_acc;
                dq_s[d] = dq_acc;
            } else {
#pragma unroll
                for (int j = 0; j < NTAP; ++j) row[j] += dU[j];
                row[NTAP] += dv_acc;
                dq_s[d] += dq_acc;
            }
        }
        __syncthreads();
    }
    {
        float* dUg = a.dU_acc + (long long)b * AD * NTAP;
        for (int i = tid; i < AD * NTAP; i += BT) {
            const int dd = i / NTAP, j = i - dd * NTAP;
            dUg[i] += dUr[dd * 63 + j];
        }
        if (tid < AD) {
            a.dv_acc[(long long)b * AD + tid] += dUr[tid * 63 + NTAP];
            a.dq_out[(long long)b * a.ld_dq + tid] = dq_s[tid];
        }
    }
    // ---- e. dh = Wq^T dq  (Wq is [128][Hq]) -------------------------------------------------
    {
        const int H4 = Hq >> 2;
        const float4* __restrict__ W4 = reinterpret_cast<const float4*>(a.Wq);
        for (int k4 = tid; k4 < H4; k4 += BT) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
            for (int dd = 0; dd < AD; ++dd) {
                const float g = dq_s[dd];
                const float4 w = W4[(long long)dd * H4 + k4];
                acc.x = fmaf(g, w.x, acc.x);
                acc.y = fmaf(g, w.y, acc.y);
                acc.z = fmaf(g, w.z, acc.z);
                acc.w = fmaf(g, w.w, acc.w);
            }
            *reinterpret_cast<float4*>(a.dh_out + (long long)b * a.ld_dh + k4 * 4) = acc;
        }
    }
    // ---- f. carries for step t-1 ------------------------------------------------------------
    {
        float* dwc = a.dw_carry + (long long)b * Ti;
        float* dcc = a.dcum_carry + (long long)b * Ti;
        for (int ti = tid; ti < Ti; ti += BT) {
            const float dcum_in = dcc[ti];
            dwc[ti] = dwin_s[HALO + ti];
            dcc[ti] = dcum_in + dwin_s[TIP + HALO + ti];
        }
    }
}

extern "C" int t2amd_attention_step_bwd_f32(const t2amd_attn_bwd* a, void* stream);

// UT is produced by t2amd_fold_location_f32 right behind U: layout U[128][62] then UT[64][128].
static int attn_bwd_launch(const t2amd_attn_bwd* a, const float* UT, void* stream) {
    T2_REQUIRE(a && a->dctx_total && a->q && a->Wq && a->U && a->v && a->pm && a->memory && a->w &&
                   a->cum_before && a->dw_carry && a->dcum_carry && a->d_pm && a->dU_acc && a->dv_acc &&
                   a->dq_out && a->dh_out,
               "attn_bwd: null pointer");
    T2_REQUIRE(a->B > 0 && a->Ti > 0 && a->E % 4 == 0 && a->Hq % 4 == 0, "attn_bwd: bad dims");
    T2_REQUIRE(t2_aligned16(a->Wq) && t2_aligned16(a->memory) && t2_aligned16(UT) &&
                   t2_aligned16(a->dh_out) && a->ld_dh % 4 == 0,
               "attn_bwd: alignment");
    AttnBwdParams p;
    p.a = *a;
    for (int i = 0; i < 3; ++i)
        if (p.a.dctx[i].p && p.a.dctx[i].nsplit < 1) p.a.dctx[i].nsplit = 1;
    p.UT = UT;
    p.tip = ((a->Ti + 2 * HALO + 3) / 4) * 4;
    const size_t lds = sizeof(float) * ((size_t)TC * DP_LD + TC * DC_LD + a->E + 4 * p.tip + 2 * a->Ti + AD + 32);
    T2_REQUIRE((size_t)TC * DP_LD >= (size_t)AD * 63, "attn_bwd: alias size");
    T2_REQUIRE(lds <= 160 * 1024, "attn_bwd: needs more than 160 KiB of LDS");
    if (lds > 64 * 1024)
        if (!t2amd_validate_only_flag_()) (void)hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    T2_LAUNCH(attn_bwd_kernel, dim3(a->B), dim3(BT), lds, (hipStream_t)stream, p);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

extern "C" int t2amd_attention_step_bwd_f32(const t2amd_attn_bwd* a, void* stream) {
    T2_REQUIRE(a && a->U, "attn_bwd: null args");
    return attn_bwd_launch(a, a->U + AD * NTAP, stream);
}

// ---------------------------------------------------------------------------------------
// Fold / unfold of the location layer.
//   out: U[128][62] followed by UT[64][128] (UT[ck][d] = U[d][ck], rows 62..63 zero)
// ---------------------------------------------------------------------------------------
__global__ void fold_location_kernel(const float* __restrict__ wd, const float* __restrict__ wc,
                                     float* __restrict__ U) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float* UT = U + AD * NTAP;
    if (i < AD * 64) {
        const int d = i >> 6, ck = i & 63;
        float s = 0.f;
        if (ck < NTAP) {
            for (int f = 0; f < T2AMD_LOC_FILTERS; ++f) s = fmaf(wd[d * T2AMD_LOC_FILTERS + f], wc[f * NTAP + ck], s);
            U[d * NTAP + ck] = s;
        }
        UT[ck * AD + d] = s;
    }
}

extern "C" int t2amd_fold_location_f32(const float* wdense, const float* wconv, float* U, void* stream) {
    T2_REQUIRE(wdense && wconv && U && t2_aligned16(U), "fold_location: bad pointers");
    T2_LAUNCH(fold_location_kernel, dim3(AD * 64 / 256), dim3(256), 0, (hipStream_t)stream, wdense, wconv, U);
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
