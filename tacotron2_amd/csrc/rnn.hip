// Recurrent-cell kernels for gfx950: batch x K x 4H "skinny" GEMM on the exact-f32 MFMA
// (v_mfma_f32_16x16x4_f32) with the LSTM cell fused into the epilogue.
//
// One workgroup (4 waves) owns 64 batch rows x 16 gate columns = 4 hidden units x {i,f,g,o},
// over the whole K (no split: the cell needs the complete pre-activation), so 4H/16
// workgroups stream disjoint 16-row slices of the packed weight [4H][K] exactly once per
// step from HBM/MALL; the activations (64 x K) are re-read by every workgroup from L2.
// Both operands sit row-major (K contiguous) in LDS with a row stride of 72 floats:
// a lane reads ONE ds_read_b128 = 4 consecutive k for (row = lane&15, k-group = lane>>4) and
// feeds element j to MFMA j of a group of four — the k order inside a 16-wide block is
// permuted identically for A and B, which a dot product does not care about, and stride
// 72 (= 8 mod 64 banks) makes the b128 reads conflict-free.
//
// Replaces torch.nn.LSTMCell + F.dropout (reference model.py:352-356, 366-371) and the
// per-timestep work of nn.LSTM (model.py:181-188).
#include "common.h"

#define SK_BK 64      // k per LDS tile
#define SK_LD 72      // LDS row stride (floats)
#define SK_ROWS 64    // batch rows per workgroup

struct SkinnyParams {
    t2amd_seg x[3];
    int nseg;
    const float* W;
    int Ktot, B;
    int H;            // LSTM: hidden size; plain: unused
    int N;            // plain: output columns
    // LSTM epilogue
    const float* gin; long long ld_gin;
    const float* bias;
    const float* c_prev; long long ld_cprev;
    float* gates_out; long long ld_gates;
    float* c_out; long long ld_c;
    float* h_out; long long ld_h;
    const uint8_t* keep; long long ld_keep; float keep_scale;
    const int* lens; int t;
    // plain epilogue
    float* Y; long long ldy; int nsplit; long long split_stride; int ktiles_per_split;
};

// TAG only gives each role its own kernel symbol, so that rocprofv3 --kernel-trace --stats reports
// the decoder's attention-LSTM (1) / decoder-LSTM (2) launches separately from the rest (0).
template <bool LSTM, int TAG>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(SkinnyParams p) {
    __shared__ __attribute__((aligned(16))) float Xs[2][SK_ROWS][SK_LD];
    __shared__ __attribute__((aligned(16))) float Ws[2][16][SK_LD];
    __shared__ float Os[SK_ROWS][17];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int rowbase = blockIdx.y * SK_ROWS;
    const int B = p.B;

    // weight row handled by this thread's W-tile load: c = tid>>4 (0..15), kq = tid&15
    const int wc = tid >> 4;
    const int wkq = tid & 15;
    long long wrow;
    bool wrow_ok = true;
    if (LSTM) {
        const int j0 = blockIdx.x * 4;
        wrow = (long long)(wc >> 2) * p.H + j0 + (wc & 3);
    } else {
        wrow = (long long)blockIdx.x * 16 + wc;
        wrow_ok = wrow < p.N;
    }
    const float* __restrict__ wsrc = p.W + wrow * p.Ktot + wkq * 4;

    int kt_beg = 0, kt_end = p.Ktot / SK_BK;
    if (!LSTM) {
        kt_beg = blockIdx.z * p.ktiles_per_split;
        kt_end = kt_beg + p.ktiles_per_split;
        const int all = p.Ktot / SK_BK;
        if (kt_end > all) kt_end = all;
    }

    float4 rx[4];
    float4 rw;

    auto load_tile = [&](int kt) {
        // locate segment (widths are multiples of 64, so a tile never straddles segments)
        int koff = kt * SK_BK;
        const float* sp = p.x[0].p;
        long long sld = p.x[0].ld;
        if (p.nseg > 1 && koff >= p.x[0].width) {
            koff -= p.x[0].width;
            sp = p.x[1].p;
            sld = p.x[1].ld;
            if (p.nseg > 2 && koff >= p.x[1].width) {
                koff -= p.x[1].width;
                sp = p.x[2].p;
                sld = p.x[2].ld;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i;
            const int r = f >> 4, kq = f & 15;
            const int gr = rowbase + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sp != nullptr && gr < B)
                v = *reinterpret_cast<const float4*>(sp + (long long)gr * sld + koff + kq * 4);
            rx[i] = v;
        }
        rw = wrow_ok ? *reinterpret_cast<const float4*>(wsrc + (long long)kt * SK_BK)
                     : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i;
            const int r = f >> 4, kq = f & 15;
            *reinterpret_cast<float4*>(&Xs[buf][r][kq * 4]) = rx[i];
        }
        *reinterpret_cast<float4*>(&Ws[buf][wc][wkq * 4]) = rw;
    };

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int l15 = lane & 15, lg = lane >> 4;

    if (kt_end > kt_beg) {
        load_tile(kt_beg);
        store_tile(0);
    }
    __syncthreads();
    int cur = 0;
    for (int kt = kt_beg; kt < kt_end; ++kt) {
        const bool more = (kt + 1 < kt_end);
        if (more) load_tile(kt + 1);
#pragma unroll
        for (int s = 0; s < SK_BK / 16; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(&Xs[cur][wave * 16 + l15][s * 16 + lg * 4]);
            const float4 b = *reinterpret_cast<const float4*>(&Ws[cur][l15][s * 16 + lg * 4]);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc1, 0, 0, 0);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // D layout (16x16): col = lane&15, row = (lane>>4)*4 + reg
    if (!LSTM) {
        float* __restrict__ Y = p.Y + (long long)blockIdx.z * p.split_stride;
        const int gn = blockIdx.x * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = rowbase + wave * 16 + lg * 4 + r;
            if (gr < B && gn < p.N) Y[(long long)gr * p.ldy + gn] = acc0[r] + acc1[r];
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Os[wave * 16 + lg * 4 + r][l15] = acc0[r] + acc1[r];
    __syncthreads();

    // LSTM cell: thread -> (row = tid>>2, unit jj = tid&3); gate g lives in column g*4+jj
    const int row = tid >> 2, jj = tid & 3;
    const int gr = rowbase + row;
    if (gr >= B) return;
    const int j = blockIdx.x * 4 + jj;
    const int H = p.H;
    bool valid = true;
    if (p.lens) valid = p.t < p.lens[gr];
    float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, cn = 0.f, hn = 0.f;
    if (valid) {
        float pi = Os[row][jj], pf = Os[row][4 + jj], pg = Os[row][8 + jj], po = Os[row][12 + jj];
        if (p.gin) {
            const float* g = p.gin + (long long)gr * p.ld_gin;
            pi += g[j];
            pf += g[H + j];
            pg += g[2 * H + j];
            po += g[3 * H + j];
        }
        if (p.bias) {
            pi += p.bias[j];
            pf += p.bias[H + j];
            pg += p.bias[2 * H + j];
            po += p.bias[3 * H + j];
        }
        gi = t2_sigmoid(pi);
        gf = t2_sigmoid(pf);
        gg = tanhf(pg);
        go = t2_sigmoid(po);
        const float cp = p.c_prev ? p.c_prev[(long long)gr * p.ld_cprev + j] : 0.f;
        cn = gf * cp + gi * gg;
        hn = go * tanhf(cn);
        if (p.keep) hn = p.keep[(long long)gr * p.ld_keep + j] ? hn * p.keep_scale : 0.f;
    }
    float* go_ = p.gates_out + (long long)gr * p.ld_gates;
    go_[j] = gi;
    go_[H + j] = gf;
    go_[2 * H + j] = gg;
    go_[3 * H + j] = go;
    p.c_out[(long long)gr * p.ld_c + j] = cn;
    p.h_out[(long long)gr * p.ld_h + j] = hn;
}

static int check_segs(const t2amd_seg* x, int nseg, int Ktot) {
    if (nseg < 1 || nseg > 3) T2_FAIL("skinny: nseg must be 1..3");
    int sum = 0;
    for (int i = 0; i < nseg; ++i) {
        if (x[i].width <= 0 || x[i].width % SK_BK != 0) T2_FAIL("skinny: segment widths must be positive multiples of 64");
        if (x[i].p && (!t2_aligned16(x[i].p) || x[i].ld % 4 != 0)) T2_FAIL("skinny: segment must be 16-byte aligned with ld % 4 == 0");
        sum += x[i].width;
    }
    if (sum != Ktot) T2_FAIL("skinny: segment widths do not add up to Ktot");
    return T2AMD_OK;
}

extern "C" int t2amd_lstm_step_fwd_f32(const t2amd_lstm_step* a, void* stream) {
    T2_REQUIRE(a != nullptr, "lstm_step: null args");
    T2_PROPAGATE(check_segs(a->x, a->nseg, a->Ktot));
    T2_REQUIRE(a->W && t2_aligned16(a->W), "lstm_step: W must be 16-byte aligned");
    T2_REQUIRE(a->H > 0 && a->H % 4 == 0 && a->B > 0, "lstm_step: H must be a multiple of 4");
    T2_REQUIRE(a->gates_out && a->c_out && a->h_out, "lstm_step: null outputs");
    SkinnyParams p = {};
    for (int i = 0; i < 3; ++i) p.x[i] = a->x[i];
    p.nseg = a->nseg;
    p.W = a->W; p.Ktot = a->Ktot; p.B = a->B; p.H = a->H; p.N = 4 * a->H;
    p.gin = a->gin; p.ld_gin = a->ld_gin; p.bias = a->bias;
    p.c_prev = a->c_prev; p.ld_cprev = a->ld_cprev;
    p.gates_out = a->gates_out; p.ld_gates = a->ld_gates;
    p.c_out = a->c_out; p.ld_c = a->ld_c; p.h_out = a->h_out; p.ld_h = a->ld_h;
    p.keep = a->keep; p.ld_keep = a->ld_keep; p.keep_scale = a->keep_scale;
    p.lens = a->lens; p.t = a->t;
    dim3 grid(a->H / 4, t2_cdiv(a->B, SK_ROWS), 1);
    hipStream_t s = (hipStream_t)stream;
    t2amd_profile_mark_(a->tag, 0, s);
    if (a->tag == 1) T2_LAUNCH((skinny_gemm_kernel<true, 1>), grid, dim3(256), 0, s, p);
    else if (a->tag == 2) T2_LAUNCH((skinny_gemm_kernel<true, 2>), grid, dim3(256), 0, s, p);
    else T2_LAUNCH((skinny_gemm_kernel<true, 0>), grid, dim3(256), 0, s, p);
    t2amd_profile_mark_(a->tag, 1, s);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

extern "C" int t2amd_skinny_gemm_f32(const t2amd_skinny_gemm* a, void* stream) {
    T2_REQUIRE(a != nullptr, "skinny_gemm: null args");
    T2_PROPAGATE(check_segs(a->x, a->nseg, a->Ktot));
    T2_REQUIRE(a->W && t2_aligned16(a->W) && a->Y, "skinny_gemm: bad pointers");
    T2_REQUIRE(a->N > 0 && a->B > 0 && a->nsplit >= 1, "skinny_gemm: bad dims");
    SkinnyParams p = {};
    for (int i = 0; i < 3; ++i) p.x[i] = a->x[i];
    p.nseg = a->nseg;
    p.W = a->W; p.Ktot = a->Ktot; p.B = a->B; p.N = a->N; p.H = 0;
    p.Y = a->Y; p.ldy = a->ldy; p.nsplit = a->nsplit; p.split_stride = a->split_stride;
    const int ktiles = a->Ktot / SK_BK;
    p.ktiles_per_split = t2_cdiv(ktiles, a->nsplit);
    dim3 grid(t2_cdiv(a->N, 16), t2_cdiv(a->B, SK_ROWS), a->nsplit);
    hipStream_t s = (hipStream_t)stream;
    if (a->tag == 1) T2_LAUNCH((skinny_gemm_kernel<false, 1>), grid, dim3(256), 0, s, p);
    else if (a->tag == 2) T2_LAUNCH((skinny_gemm_kernel<false, 2>), grid, dim3(256), 0, s, p);
    else T2_LAUNCH((skinny_gemm_kernel<false, 0>), grid, dim3(256), 0, s, p);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// ---------------------------------------------------------------------------------------
// LSTM cell backward (pointwise part): given dL/dh' (dropped-out hidden) and the carried
// dL/dc, produce the gate pre-activation gradients and the new dL/dc carry.
// ---------------------------------------------------------------------------------------
struct LstmBwdParams { t2amd_lstm_bwd a; };

__device__ __forceinline__ float addend_sum(const t2amd_addend& ad, int row, int col) {
    if (!ad.p) return 0.f;
    float s = 0.f;
    const float* q = ad.p + (long long)row * ad.ld + col;
    for (int k = 0; k < ad.nsplit; ++k) s += q[(long long)k * ad.split_stride];
    return s;
}

__global__ void lstm_pointwise_bwd_kernel(LstmBwdParams p) {
    const t2amd_lstm_bwd& a = p.a;
    const int H = a.H;
    const long long n = (long long)a.B * H;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n;
         idx += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / H);
        const int j = (int)(idx - (long long)b * H);
        float* dg = a.dgates + (long long)b * a.ld_dgates;
        float* dcp = a.dc + (long long)b * a.ld_dc + j;
        bool valid = true;
        if (a.lens) valid = a.t < a.lens[b];
        if (!valid) {
            dg[j] = 0.f; dg[H + j] = 0.f; dg[2 * H + j] = 0.f; dg[3 * H + j] = 0.f;
            *dcp = 0.f;
            continue;
        }
        float dh = addend_sum(a.dh[0], b, j) + addend_sum(a.dh[1], b, j) + addend_sum(a.dh[2], b, j);
        if (a.keep) dh = a.keep[(long long)b * a.ld_keep + j] ? dh * a.keep_scale : 0.f;
        const float* g = a.gates + (long long)b * a.ld_gates;
        const float gi = g[j], gf = g[H + j], gg = g[2 * H + j], go = g[3 * H + j];
        const float c = a.c[(long long)b * a.ld_c + j];
        const float cprev = a.c_prev ? a.c_prev[(long long)b * a.ld_cprev + j] : 0.f;
        const float tc = tanhf(c);
        const float d_o = dh * tc;
        const float dc = *dcp + dh * go * (1.f - tc * tc);
        dg[j] = dc * gg * gi * (1.f - gi);
        dg[H + j] = dc * cprev * gf * (1.f - gf);
        dg[2 * H + j] = dc * gi * (1.f - gg * gg);
        dg[3 * H + j] = d_o * go * (1.f - go);
        *dcp = dc * gf;
    }
}

extern "C" int t2amd_lstm_pointwise_bwd_f32(const t2amd_lstm_bwd* a, void* stream) {
    T2_REQUIRE(a && a->gates && a->c && a->dc && a->dgates, "lstm_bwd: null args");
    T2_REQUIRE(a->B > 0 && a->H > 0, "lstm_bwd: bad dims");
    LstmBwdParams p;
    p.a = *a;
    for (int i = 0; i < 3; ++i)
        if (p.a.dh[i].p && p.a.dh[i].nsplit < 1) p.a.dh[i].nsplit = 1;
    const long long n = (long long)a->B * a->H;
    int blocks = t2_cdiv(n, 256);
    if (blocks > 2048) blocks = 2048;
    T2_LAUNCH(lstm_pointwise_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}
