// Small-batch (B <= 8) decode-step kernels for gfx950: free-running inference at B = 1 (reference
// model.py:418-454, BASELINE config 4) is a chain of matrix-VECTOR products — every weight byte is used
// once per step, so the bound is the HBM/MALL weight stream, not the MFMA: the 64-row MFMA tiles of
// rnn.hip would spend 63/64 of their work on padding.
//
// One workgroup (4 waves) owns 16 weight rows; the LSTM variant picks them as 4 hidden units x
// {i,f,g,o} so that the cell is fused in the epilogue (same ownership as the MFMA kernel).  The B input
// vectors are staged once in LDS ([B][K], K <= a few thousand); wave w streams its four weight rows with
// 16-byte loads, 1 KiB contiguous per instruction and row, all of a pass issued before the first use,
// and keeps 4 x B partial dot products per lane; a wave reduction per (row, b) finishes the product.
//
// Replaces, for small batches: nn.LSTMCell (model.py:352-356, 366-371), Prenet linears (:99),
// linear_projection + gate_layer (:373-378).
#include "common.h"

struct SmallParams {
    t2amd_seg x[3];
    int nseg;
    const float* W;      // [N][ldw] rows K-contiguous
    long long ldw;
    int Ktot, B, N, H;   // LSTM: N = 4H, row g*H + j; plain: N rows
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
    float* Y; long long ldy; int act;
    int xvec;            // 1: input rows are 16-byte aligned (float4 staging), 0: scalar staging
    int w16;             // 1: W holds bf16 (ldw in elements)
    // optional stop test fused into the projection (plain epilogue): row `gate_row` is the gate logit
    int* out_lengths; uint8_t* active; int* done_count;
    int fin_t, fin_max_steps, gate_row; float fin_thr;
};

// W16: the weight rows are bf16 (t2amd_lstm_step.bf16 == 2: weights only -- the input vectors, sums, cell and outputs
// stay f32): a 16-byte load is eight k, the stream this kernel is bound by halves.
template <bool LSTM, int NB, bool W16>
__global__ __launch_bounds__(256) void small_batch_kernel(SmallParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                                   // [NB][Ktot]
    float* os = smem + (size_t)NB * p.Ktot;             // [16][NB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.Ktot, K4 = K >> 2, B = p.B;

    // stage the input vectors (absent segments are zeros)
    for (int i = tid; i < NB * K4; i += 256) {
        const int b = i / K4, k4 = i - b * K4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < B) {
            int k = k4 * 4;
            const float* sp = p.x[0].p;
            long long sld = p.x[0].ld;
            if (p.nseg > 1 && k >= p.x[0].width) {
                k -= p.x[0].width;
                sp = p.x[1].p;
                sld = p.x[1].ld;
                if (p.nseg > 2 && k >= p.x[1].width) {
                    k -= p.x[1].width;
                    sp = p.x[2].p;
                    sld = p.x[2].ld;
                }
            }
            if (sp) {
                const float* src = sp + (long long)b * sld + k;
                if (p.xvec) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3];
                }
            }
        }
        *reinterpret_cast<float4*>(&xs[(size_t)b * K + k4 * 4]) = v;
    }

    // the four weight rows of this wave
    const float4* wr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        long long row;
        if (LSTM) {
            row = (long long)wave * p.H + blockIdx.x * 4 + u;          // gate `wave`, unit 4*bx + u
        } else {
            row = (long long)blockIdx.x * 16 + wave * 4 + u;
            if (row > p.N - 1) row = p.N - 1;                          // ragged last block: never stored
        }
        wr[u] = W16 ? reinterpret_cast<const float4*>(reinterpret_cast<const unsigned short*>(p.W) + row * p.ldw)
                    : reinterpret_cast<const float4*>(p.W + row * p.ldw);
    }
    __syncthreads();

    float acc[4][NB];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[u][b] = 0.f;

    if constexpr (W16) {
        // units of 8 k (16 bytes of bf16): same two-pieces-in-flight structure
        const int K8 = K >> 3;
        auto fma8 = [&](const float4& wv, int b, int k8, float sacc) -> float {
            const float4 x0 = *reinterpret_cast<const float4*>(&xs[(size_t)b * K + k8 * 8]);
            const float4 x1 = *reinterpret_cast<const float4*>(&xs[(size_t)b * K + k8 * 8 + 4]);
            const unsigned u0 = __float_as_uint(wv.x), u1 = __float_as_uint(wv.y), u2 = __float_as_uint(wv.z), u3 = __float_as_uint(wv.w);
            sacc = fmaf(__uint_as_float(u0 << 16), x0.x, sacc); sacc = fmaf(__uint_as_float(u0 & 0xffff0000u), x0.y, sacc);
            sacc = fmaf(__uint_as_float(u1 << 16), x0.z, sacc); sacc = fmaf(__uint_as_float(u1 & 0xffff0000u), x0.w, sacc);
            sacc = fmaf(__uint_as_float(u2 << 16), x1.x, sacc); sacc = fmaf(__uint_as_float(u2 & 0xffff0000u), x1.y, sacc);
            sacc = fmaf(__uint_as_float(u3 << 16), x1.z, sacc); sacc = fmaf(__uint_as_float(u3 & 0xffff0000u), x1.w, sacc);
            return sacc;
        };
        int k8 = lane;
        for (; k8 + 64 < K8; k8 += 128) {
            float4 w0[4], w1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                w0[u] = wr[u][k8];
                w1[u] = wr[u][k8 + 64];
            }
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u][b] = fma8(w1[u], b, k8 + 64, fma8(w0[u], b, k8, acc[u][b]));
        }
        for (; k8 < K8; k8 += 64) {
            float4 w0[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w0[u] = wr[u][k8];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u][b] = fma8(w0[u], b, k8, acc[u][b]);
        }
    }
    int k4 = W16 ? K4 : lane;
    for (; k4 + 64 < K4; k4 += 128) {          // two 1-KiB pieces per row in flight: 8 loads before the first use
        float4 w0[4], w1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            w0[u] = wr[u][k4];
            w1[u] = wr[u][k4 + 64];
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float4 x0 = *reinterpret_cast<const float4*>(&xs[(size_t)b * K + k4 * 4]);
            const float4 x1 = *reinterpret_cast<const float4*>(&xs[(size_t)b * K + (k4 + 64) * 4]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float s = acc[u][b];
                s = fmaf(w0[u].x, x0.x, s); s = fmaf(w0[u].y, x0.y, s); s = fmaf(w0[u].z, x0.z, s); s = fmaf(w0[u].w, x0.w, s);
                s = fmaf(w1[u].x, x1.x, s); s = fmaf(w1[u].y, x1.y, s); s = fmaf(w1[u].z, x1.z, s); s = fmaf(w1[u].w, x1.w, s);
                acc[u][b] = s;
            }
        }
    }
    for (; k4 < K4; k4 += 64) {
        float4 w0[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w0[u] = wr[u][k4];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float4 x0 = *reinterpret_cast<const float4*>(&xs[(size_t)b * K + k4 * 4]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float s = acc[u][b];
                s = fmaf(w0[u].x, x0.x, s); s = fmaf(w0[u].y, x0.y, s); s = fmaf(w0[u].z, x0.z, s); s = fmaf(w0[u].w, x0.w, s);
                acc[u][b] = s;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float s = wave_reduce_sum(acc[u][b]);
            if (lane == 0) os[(wave * 4 + u) * NB + b] = s;
        }
    __syncthreads();

    if (!LSTM) {
        // thread -> (row r of 16, b)
        if (tid < 16 * NB) {
            const int r = tid / NB, b = tid - r * NB;
            const int n = blockIdx.x * 16 + r;
            if (b < B && n < p.N) {
                float v = os[r * NB + b];
                if (p.bias) v += p.bias[n];
                if (p.act == 1) v = fmaxf(v, 0.f);
                if (p.keep) v = p.keep[(long long)b * p.ld_keep + n] ? v * p.keep_scale : 0.f;
                p.Y[(long long)b * p.ldy + n] = v;
                // stop test after the frame is emitted: sigmoid(gate) > threshold (strict); the stopping frame is
                // part of the output (reference model.py:439-444)
                if (p.out_lengths && n == p.gate_row && p.active[b]) {
                    const float sg = 1.0f / (1.0f + expf(-v));
                    if (sg > p.fin_thr || p.fin_t + 1 >= p.fin_max_steps) {
                        p.out_lengths[b] = p.fin_t + 1;
                        p.active[b] = 0;
                        atomicAdd(p.done_count, 1);
                    }
                }
            }
        }
        return;
    }
    // LSTM cell: thread -> (unit u of 4, b); gate g lives in os row g*4 + u
    if (tid < 4 * NB) {
        const int u = tid & 3, b = tid >> 2;
        if (b >= B) return;
        const int H = p.H, j = blockIdx.x * 4 + u;
        bool valid = true;
        if (p.lens) valid = p.t < p.lens[b];
        float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, cn = 0.f, hn = 0.f;
        if (valid) {
            float pi = os[(0 * 4 + u) * NB + b], pf = os[(1 * 4 + u) * NB + b];
            float pg = os[(2 * 4 + u) * NB + b], po = os[(3 * 4 + u) * NB + b];
            if (p.gin) {
                const float* g = p.gin + (long long)b * p.ld_gin;
                pi += g[j]; pf += g[H + j]; pg += g[2 * H + j]; po += g[3 * H + j];
            }
            if (p.bias) {
                pi += p.bias[j]; pf += p.bias[H + j]; pg += p.bias[2 * H + j]; po += p.bias[3 * H + j];
            }
            gi = t2_sigmoid(pi);
            gf = t2_sigmoid(pf);
            gg = tanhf(pg);
            go = t2_sigmoid(po);
            const float cp = p.c_prev ? p.c_prev[(long long)b * p.ld_cprev + j] : 0.f;
            cn = gf * cp + gi * gg;
            hn = go * tanhf(cn);
            if (p.keep) hn = p.keep[(long long)b * p.ld_keep + j] ? hn * p.keep_scale : 0.f;
        }
        if (p.gates_out) {
            float* go_ = p.gates_out + (long long)b * p.ld_gates;
            go_[j] = gi;
            go_[H + j] = gf;
            go_[2 * H + j] = gg;
            go_[3 * H + j] = go;
        }
        p.c_out[(long long)b * p.ld_c + j] = cn;
        p.h_out[(long long)b * p.ld_h + j] = hn;
    }
}

static int small_check_segs(const t2amd_seg* x, int nseg, int Ktot) {
    if (nseg < 1 || nseg > 3) T2_FAIL("small: nseg must be 1..3");
    int sum = 0;
    for (int i = 0; i < nseg; ++i) {
        if (x[i].width <= 0 || x[i].width % 4 != 0) T2_FAIL("small: segment widths must be positive multiples of 4");
        if (x[i].p && (!t2_aligned16(x[i].p) || x[i].ld % 4 != 0)) T2_FAIL("small: segment must be 16-byte aligned with ld % 4 == 0");
        sum += x[i].width;
    }
    if (sum != Ktot) T2_FAIL("small: segment widths do not add up to Ktot");
    return T2AMD_OK;
}

static int g_small_lds = 0;

template <bool LSTM>
static int small_launch(const SmallParams& p, int grid, void* stream) {
    int nb = 1;
    while (nb < p.B) nb *= 2;
    const size_t lds = sizeof(float) * ((size_t)nb * p.Ktot + 16 * nb);
    T2_REQUIRE(lds <= 160 * 1024, "small: B*K does not fit in LDS");
    hipStream_t s = (hipStream_t)stream;
#define T2_SMALL_CASE(NB)                                                                                       \
    case NB:                                                                                                    \
        if ((int)lds > 64 * 1024 && (int)lds > g_small_lds && !t2amd_validate_only_flag_()) {                   \
            (void)hipFuncSetAttribute((const void*)small_batch_kernel<LSTM, NB, false>,                         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                  \
            (void)hipFuncSetAttribute((const void*)small_batch_kernel<LSTM, NB, true>,                          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                  \
        }                                                                                                       \
        if (p.w16) T2_LAUNCH((small_batch_kernel<LSTM, NB, true>), dim3(grid), dim3(256), lds, s, p);           \
        else T2_LAUNCH((small_batch_kernel<LSTM, NB, false>), dim3(grid), dim3(256), lds, s, p);                \
        break;
    switch (nb) {
        T2_SMALL_CASE(1)
        T2_SMALL_CASE(2)
        T2_SMALL_CASE(4)
        T2_SMALL_CASE(8)
        default: T2_FAIL("small: B must be <= 8");
    }
#undef T2_SMALL_CASE
    if ((int)lds > g_small_lds) g_small_lds = (int)lds;
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

extern "C" int t2amd_lstm_step_small_f32(const t2amd_lstm_step* a, void* stream) {
    T2_REQUIRE(a != nullptr, "lstm_step_small: null args");
    T2_PROPAGATE(small_check_segs(a->x, a->nseg, a->Ktot));
    T2_REQUIRE(a->W && t2_aligned16(a->W) && a->Ktot % 4 == 0, "lstm_step_small: W must be 16-byte aligned, K % 4 == 0");
    T2_REQUIRE(a->H > 0 && a->H % 4 == 0 && a->B > 0 && a->B <= 8, "lstm_step_small: H % 4 == 0 and 1 <= B <= 8");
    T2_REQUIRE(a->c_out && a->h_out, "lstm_step_small: null outputs");       // gates_out may be NULL
    SmallParams p = {};
    for (int i = 0; i < 3; ++i) p.x[i] = a->x[i];
    p.nseg = a->nseg;
    p.W = a->W; p.ldw = a->Ktot; p.Ktot = a->Ktot; p.B = a->B; p.H = a->H; p.N = 4 * a->H;
    p.gin = a->gin; p.ld_gin = a->ld_gin; p.bias = a->bias;
    p.c_prev = a->c_prev; p.ld_cprev = a->ld_cprev;
    p.gates_out = a->gates_out; p.ld_gates = a->ld_gates;
    p.c_out = a->c_out; p.ld_c = a->ld_c; p.h_out = a->h_out; p.ld_h = a->ld_h;
    p.keep = a->keep; p.ld_keep = a->ld_keep; p.keep_scale = a->keep_scale;
    p.lens = a->lens; p.t = a->t; p.xvec = 1;
    T2_REQUIRE(a->bf16 == 0 || a->bf16 == 2, "lstm_step_small: bf16 must be 0 or 2 (bf16 weights, f32 inputs)");
    T2_REQUIRE(a->bf16 == 0 || a->Ktot % 8 == 0, "lstm_step_small: bf16 weights need K % 8 == 0");
    p.w16 = a->bf16 == 2 ? 1 : 0;
    t2amd_profile_mark_(a->tag, 0, (hipStream_t)stream);
    const int rc = small_launch<true>(p, a->H / 4, stream);
    t2amd_profile_mark_(a->tag, 1, (hipStream_t)stream);
    return rc;
}

static int linear_small_impl(const t2amd_small_linear* a, int* out_lengths, uint8_t* active, int* done_count, int t,
                             int max_steps, float thr, int gate_row, void* stream);

extern "C" int t2amd_linear_small_f32(const t2amd_small_linear* a, void* stream) {
    return linear_small_impl(a, nullptr, nullptr, nullptr, 0, 0, 0.f, -1, stream);
}

// internal (loops.hip): frame + gate projection with the per-utterance stop test fused in
int t2amd_proj_finish_small_(const t2amd_small_linear* a, int* out_lengths, uint8_t* active, int* done_count, int t,
                             int max_steps, float thr, int gate_row, void* stream) {
    T2_REQUIRE(out_lengths && active && done_count, "proj_finish_small: null state");
    return linear_small_impl(a, out_lengths, active, done_count, t, max_steps, thr, gate_row, stream);
}

static int linear_small_impl(const t2amd_small_linear* a, int* out_lengths, uint8_t* active, int* done_count, int t,
                             int max_steps, float thr, int gate_row, void* stream) {
    T2_REQUIRE(a && a->X && a->W && a->Y, "linear_small: null args");
    T2_REQUIRE(a->B > 0 && a->B <= 8 && a->N > 0 && a->K > 0 && a->K % 4 == 0, "linear_small: 1 <= B <= 8, K % 4 == 0");
    T2_REQUIRE(t2_aligned16(a->W) && a->ldw % 4 == 0, "linear_small: W must be 16-byte aligned with ldw % 4 == 0");
    T2_REQUIRE(a->act == 0 || a->act == 1, "linear_small: act must be 0 or 1");
    SmallParams p = {};
    p.x[0].p = a->X; p.x[0].ld = a->ldx; p.x[0].width = a->K;
    p.nseg = 1;
    p.W = a->W; p.ldw = a->ldw; p.Ktot = a->K; p.B = a->B; p.N = a->N; p.H = 0;
    p.bias = a->bias; p.act = a->act;
    p.keep = a->keep; p.ld_keep = a->ldkeep; p.keep_scale = a->keep_scale;
    p.Y = a->Y; p.ldy = a->ldy;
    p.xvec = (t2_aligned16(a->X) && a->ldx % 4 == 0) ? 1 : 0;
    p.out_lengths = out_lengths; p.active = active; p.done_count = done_count;
    p.fin_t = t; p.fin_max_steps = max_steps; p.fin_thr = thr; p.gate_row = gate_row;
    return small_launch<false>(p, t2_cdiv(a->N, 16), stream);
}
