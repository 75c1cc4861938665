// HBM-bound kernels of the Tacotron 2 path for gfx950: BatchNorm statistics / apply / backward
// over channel-last rows, column sums, embedding gather + deterministic scatter, Philox
// keep-masks and the layout shuffles at the model boundary.  All are coalesced along the
// channel (fastest) dimension; reductions over rows are two-stage with fp64 partials so the
// result does not depend on launch geometry beyond a fixed 64-way split.
#include "common.h"

#define RB 64   // row-blocks of the two-stage column reductions

// ---------------------------------------------------------------------------------------
// column reductions
// ---------------------------------------------------------------------------------------
// partial[rb][n] = sum over rows r = rb, rb+RB, ... of f(x[r][n]);  two quantities.
template <int MODE>   // 0: (x, x^2)   1: (x) only
__global__ void colreduce_partial_kernel(const float* __restrict__ x, long long ldx, int M, int N,
                                         double* __restrict__ ws) {
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rsub = threadIdx.x >> 6;          // 0..3
    const int rb = blockIdx.y;
    __shared__ double s1[4][64], s2[4][64];
    double a = 0.0, q = 0.0;
    if (col < N) {
        // four rows per trip, loaded before the first is used: one load in flight per thread left the kernel
        // latency-bound at ~1.5 TB/s
        const long long step = (long long)RB * 4;
        long long r = (long long)rb * 4 + rsub;
        for (; r + 3 * step < M; r += 4 * step) {
            const float v0 = x[r * ldx + col], v1 = x[(r + step) * ldx + col];
            const float v2 = x[(r + 2 * step) * ldx + col], v3 = x[(r + 3 * step) * ldx + col];
            a += (double)v0; a += (double)v1; a += (double)v2; a += (double)v3;
            if (MODE == 0) {
                q += (double)v0 * (double)v0; q += (double)v1 * (double)v1;
                q += (double)v2 * (double)v2; q += (double)v3 * (double)v3;
            }
        }
        for (; r < M; r += step) {
            const float v = x[r * ldx + col];
            a += (double)v;
            if (MODE == 0) q += (double)v * (double)v;
        }
    }
    s1[rsub][threadIdx.x & 63] = a;
    s2[rsub][threadIdx.x & 63] = q;
    __syncthreads();
    if (rsub == 0 && col < N) {
        const int c = threadIdx.x & 63;
        ws[(long long)rb * N + col] = s1[0][c] + s1[1][c] + s1[2][c] + s1[3][c];
        if (MODE == 0) ws[(long long)(RB + rb) * N + col] = s2[0][c] + s2[1][c] + s2[2][c] + s2[3][c];
    }
}

// The same reduction on 16-byte loads (round 5): a workgroup is 16 column lanes of four columns (the same 64 columns per
// blockIdx.x) x 16 row lanes; row lane j of row block rb takes rows rb*16 + j, + 1024, ... four at a time, every load issued
// before the first use (64 B in flight per thread and tensor).  A wave's load instruction covers four 256-byte row segments.
// Needs N % 4 == 0, ldx % 4 == 0 and a 16-byte-aligned base; the scalar kernel above stays for everything else.
#define VCL 16      // column lanes (x 4 columns)
#define VRL 16      // row lanes
struct Dbl4 { double v[4]; };
__device__ __forceinline__ void vred_finish(double (*s1)[64], double (*s2)[64], const Dbl4& a, const Dbl4& q, bool two,
                                            int cl, int rl, int rb, int col0, int N, double* __restrict__ ws) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s1[rl][cl * 4 + j] = a.v[j];
        if (two) s2[rl][cl * 4 + j] = q.v[j];
    }
    __syncthreads();
    const int c = threadIdx.x;
    if (c < 64 && col0 + c < N) {
        double s = 0.0, t = 0.0;
#pragma unroll
        for (int r = 0; r < VRL; ++r) {
            s += s1[r][c];
            if (two) t += s2[r][c];
        }
        ws[(long long)rb * N + col0 + c] = s;
        if (two) ws[(long long)(RB + rb) * N + col0 + c] = t;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void colreduce_partial_vec_kernel(const float* __restrict__ x, long long ldx, int M, int N,
                                                                    double* __restrict__ ws) {
    const int cl = threadIdx.x & (VCL - 1), rl = threadIdx.x / VCL;
    const int col0 = blockIdx.x * 64, col = col0 + cl * 4;
    const int rb = blockIdx.y;
    __shared__ double s1[VRL][64], s2[MODE == 0 ? VRL : 1][64];
    Dbl4 a = {}, q = {};
    if (col < N) {
        const long long step = (long long)RB * VRL;
        long long r = (long long)rb * VRL + rl;
        auto add = [&](const float4& v) {
            a.v[0] += (double)v.x; a.v[1] += (double)v.y; a.v[2] += (double)v.z; a.v[3] += (double)v.w;
            if (MODE == 0) {
                q.v[0] += (double)v.x * (double)v.x; q.v[1] += (double)v.y * (double)v.y;
                q.v[2] += (double)v.z * (double)v.z; q.v[3] += (double)v.w * (double)v.w;
            }
        };
        for (; r + 3 * step < M; r += 4 * step) {
            const float4 v0 = *reinterpret_cast<const float4*>(x + r * ldx + col);
            const float4 v1 = *reinterpret_cast<const float4*>(x + (r + step) * ldx + col);
            const float4 v2 = *reinterpret_cast<const float4*>(x + (r + 2 * step) * ldx + col);
            const float4 v3 = *reinterpret_cast<const float4*>(x + (r + 3 * step) * ldx + col);
            add(v0); add(v1); add(v2); add(v3);
        }
        for (; r < M; r += step) add(*reinterpret_cast<const float4*>(x + r * ldx + col));
    }
    vred_finish(s1, s2, a, q, MODE == 0, cl, rl, rb, col0, N, ws);
}

static inline bool vec4_ok(const void* p, long long ld, int N) { return N % 4 == 0 && ld % 4 == 0 && t2_aligned16(p); }
static const bool g_ew_scalar = [] { const char* e = getenv("T2AMD_ELEMENTWISE_SCALAR"); return e && atoi(e) != 0; }();   // A/B

__global__ void bn_stats_finalize_kernel(const double* __restrict__ ws, int M, int N, float* mean,
                                         float* invstd, float* rmean, float* rvar, float momentum, float eps) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double s = 0.0, q = 0.0;
    for (int rb = 0; rb < RB; ++rb) {
        s += ws[(long long)rb * N + n];
        q += ws[(long long)(RB + rb) * N + n];
    }
    const double mu = s / M;
    double var = q / M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[n] = (float)mu;
    invstd[n] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) rmean[n] = (1.f - momentum) * rmean[n] + momentum * (float)mu;
    if (rvar) {
        const double unb = (M > 1) ? var * ((double)M / (double)(M - 1)) : var;
        rvar[n] = (1.f - momentum) * rvar[n] + momentum * (float)unb;
    }
}

extern "C" int t2amd_bn_stats_f32(const float* x, long long ldx, int M, int N, double* ws, float* mean,
                                  float* invstd, float* running_mean, float* running_var, float momentum,
                                  float eps, void* stream) {
    T2_REQUIRE(x && ws && mean && invstd && M > 0 && N > 0, "bn_stats: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (vec4_ok(x, ldx, N) && !g_ew_scalar)
        T2_LAUNCH((colreduce_partial_vec_kernel<0>), dim3(t2_cdiv(N, 64), RB), dim3(256), 0, s, x, ldx, M, N, ws);
    else
        T2_LAUNCH((colreduce_partial_kernel<0>), dim3(t2_cdiv(N, 64), RB), dim3(256), 0, s, x, ldx, M, N, ws);
    T2_LAUNCH(bn_stats_finalize_kernel, dim3(t2_cdiv(N, 128)), dim3(128), 0, s, ws, M, N, mean, invstd,
                       running_mean, running_var, momentum, eps);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

__global__ void bn_eval_invstd_kernel(const float* rvar, float* invstd, int N, float eps) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) invstd[n] = (float)(1.0 / sqrt((double)rvar[n] + (double)eps));
}

extern "C" int t2amd_bn_eval_invstd_f32(const float* running_var, float* invstd, int N, float eps, void* stream) {
    T2_REQUIRE(running_var && invstd && N > 0, "bn_eval_invstd: bad args");
    T2_LAUNCH(bn_eval_invstd_kernel, dim3(t2_cdiv(N, 128)), dim3(128), 0, (hipStream_t)stream, running_var,
                       invstd, N, eps);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

__global__ void colsum_finalize_kernel(const double* __restrict__ ws, int N, float* out, int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double s = 0.0;
    for (int rb = 0; rb < RB; ++rb) s += ws[(long long)rb * N + n];
    out[n] = accumulate ? out[n] + (float)s : (float)s;
}

// Column sums of a bf16 slab (round 6: the LSTM bias gradients of the bf16 mode from the gate-gradient slabs the weight-gradient
// products already read, half the bytes of the f32 slabs: 2 x 456 MB instead of 2 x 912 MB per training step).  Thread = eight
// columns (one 16-byte load) x a row lane; sums in double, partition [RB][N] as above.
__global__ __launch_bounds__(256) void colreduce_partial_bf16_kernel(const unsigned short* __restrict__ x, long long ldx, int M, int N,
                                                                     double* __restrict__ ws) {
    constexpr int CL = 8, RL = 32;                       // 8 column lanes x 8 columns = 64 columns; 32 row lanes
    const int cl = threadIdx.x & (CL - 1), rl = threadIdx.x / CL;
    const int col0 = blockIdx.x * 64, col = col0 + cl * 8;
    const int rb = blockIdx.y;
    __shared__ double s1[RL][64];
    double a[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
    if (col < N) {
        auto add = [&](const uint4& v) {
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[2 * j] += (double)__uint_as_float(w[j] << 16);
                a[2 * j + 1] += (double)__uint_as_float(w[j] & 0xffff0000u);
            }
        };
        const long long step = (long long)RB * RL;
        long long r = (long long)rb * RL + rl;
        for (; r + 3 * step < M; r += 4 * step) {
            const uint4 v0 = *reinterpret_cast<const uint4*>(x + r * ldx + col);
            const uint4 v1 = *reinterpret_cast<const uint4*>(x + (r + step) * ldx + col);
            const uint4 v2 = *reinterpret_cast<const uint4*>(x + (r + 2 * step) * ldx + col);
            const uint4 v3 = *reinterpret_cast<const uint4*>(x + (r + 3 * step) * ldx + col);
            add(v0); add(v1); add(v2); add(v3);
        }
        for (; r < M; r += step) add(*reinterpret_cast<const uint4*>(x + r * ldx + col));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[rl][cl * 8 + j] = a[j];
    __syncthreads();
    const int c = threadIdx.x;
    if (c < 64 && col0 + c < N) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < RL; ++r) t += s1[r][c];
        ws[(long long)rb * N + col0 + c] = t;
    }
}

extern "C" int t2amd_colsum_bf16(const void* x16, long long ldx, int M, int N, double* ws, float* out, int accumulate, void* stream) {
    T2_REQUIRE(x16 && ws && out && M > 0 && N > 0, "colsum_bf16: bad args");
    T2_REQUIRE(N % 8 == 0 && ldx % 8 == 0 && t2_aligned16(x16), "colsum_bf16: N, ldx multiples of 8 and a 16-byte-aligned base");
    hipStream_t s = (hipStream_t)stream;
    T2_LAUNCH(colreduce_partial_bf16_kernel, dim3(t2_cdiv(N, 64), RB), dim3(256), 0, s, (const unsigned short*)x16, ldx, M, N, ws);
    T2_LAUNCH(colsum_finalize_kernel, dim3(t2_cdiv(N, 128)), dim3(128), 0, s, ws, N, out, accumulate);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

extern "C" int t2amd_colsum_f32(const float* x, long long ldx, int M, int N, double* ws, float* out,
                                int accumulate, void* stream) {
    T2_REQUIRE(x && ws && out && M > 0 && N > 0, "colsum: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (vec4_ok(x, ldx, N) && !g_ew_scalar)
        T2_LAUNCH((colreduce_partial_vec_kernel<1>), dim3(t2_cdiv(N, 64), RB), dim3(256), 0, s, x, ldx, M, N, ws);
    else
        T2_LAUNCH((colreduce_partial_kernel<1>), dim3(t2_cdiv(N, 64), RB), dim3(256), 0, s, x, ldx, M, N, ws);
    T2_LAUNCH(colsum_finalize_kernel, dim3(t2_cdiv(N, 128)), dim3(128), 0, s, ws, N, out, accumulate);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// ---------------------------------------------------------------------------------------
// BatchNorm apply (+ activation + dropout)
// ---------------------------------------------------------------------------------------
__global__ void bn_act_fwd_kernel(const float* __restrict__ x, long long ldx, float* __restrict__ y,
                                  long long ldy, int M, int N, const float* __restrict__ mean,
                                  const float* __restrict__ invstd, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, int act, const uint8_t* __restrict__ keep,
                                  long long ldkeep, float keep_scale, const int* __restrict__ lens, int T) {
    const long long total = (long long)M * N;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / N;
        const int n = (int)(i - r * N);
        float v = (x[r * ldx + n] - mean[n]) * invstd[n] * gamma[n] + beta[n];
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = tanhf(v);
        if (keep) v = keep[r * ldkeep + n] ? v * keep_scale : 0.f;
        if (lens) {
            const int b = (int)(r / T), t = (int)(r - (long long)b * T);
            if (t >= lens[b]) v = 0.f;
        }
        y[r * ldy + n] = v;
    }
}

// 16-byte form: thread = (column lane of four columns, row lane); the per-channel parameters are loaded once per thread, four
// rows are in flight per thread, and no element pays a 64-bit division (the scalar kernel's i / N).  Same expression per element
// as above, so the same bits.
// IMG (round 6): the output also leaves as the bf16 halo image the next convolution's window product reads
// (t2amd_cast_halo_bf16's layout over utterances of IT rows, halo rows zeroed here) -- the cast pass's re-read of y is gone.
template <bool IMG>
__global__ __launch_bounds__(256) void bn_act_fwd_vec_kernel(const float* __restrict__ x, long long ldx, float* __restrict__ y,
                                                             long long ldy, int M, int N, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int act,
                                                             const uint8_t* __restrict__ keep, long long ldkeep, float keep_scale,
                                                             const int* __restrict__ lens, int T,
                                                             unsigned short* __restrict__ img, int IT, int pad, int nb) {
    const int cl = threadIdx.x & (VCL - 1), rl = threadIdx.x / VCL;
    const int col = blockIdx.x * 64 + cl * 4;
    if (col >= N) return;
    const float4 mu = *reinterpret_cast<const float4*>(mean + col), is = *reinterpret_cast<const float4*>(invstd + col);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + col), be = *reinterpret_cast<const float4*>(beta + col);
    const int step = gridDim.y * VRL;
    auto one = [&](float xv, float m, float i, float g, float b, unsigned kp, bool dead) -> float {
        float v = (xv - m) * i * g + b;
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = tanhf(v);
        if (keep) v = kp ? v * keep_scale : 0.f;
        return dead ? 0.f : v;
    };
    auto row = [&](int r, const float4& xv, unsigned kp) {
        bool dead = false;
        if (lens) {
            const int b = r / T, t = r - b * T;
            dead = t >= lens[b];
        }
        float4 o;
        o.x = one(xv.x, mu.x, is.x, ga.x, be.x, kp & 0xffu, dead);
        o.y = one(xv.y, mu.y, is.y, ga.y, be.y, (kp >> 8) & 0xffu, dead);
        o.z = one(xv.z, mu.z, is.z, ga.z, be.z, (kp >> 16) & 0xffu, dead);
        o.w = one(xv.w, mu.w, is.w, ga.w, be.w, kp >> 24, dead);
        *reinterpret_cast<float4*>(y + (long long)r * ldy + col) = o;
        if constexpr (IMG) {
            const int Tp = IT + 2 * pad;
            const int b = r / IT, t = r - b * IT;
            unsigned short* ib = img + ((long long)b * Tp) * N + col;
            uint2 v;
            v.x = (unsigned)t2_f32_to_bf16(o.x) | ((unsigned)t2_f32_to_bf16(o.y) << 16);
            v.y = (unsigned)t2_f32_to_bf16(o.z) | ((unsigned)t2_f32_to_bf16(o.w) << 16);
            *reinterpret_cast<uint2*>(ib + (long long)(pad + t) * N) = v;
            const uint2 z = make_uint2(0u, 0u);
            if (t < pad) *reinterpret_cast<uint2*>(ib + (long long)t * N) = z;
            if (t >= IT - pad) *reinterpret_cast<uint2*>(ib + (long long)(t + 2 * pad) * N) = z;
            if (b == nb - 1 && t < 2 * pad) *reinterpret_cast<uint2*>(ib + (long long)(Tp + t) * N) = z;
        }
    };
    int r = blockIdx.y * VRL + rl;
    for (; r + 3 * step < M; r += 4 * step) {
        float4 xv[4];
        unsigned kp[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xv[j] = *reinterpret_cast<const float4*>(x + (long long)(r + j * step) * ldx + col);
            if (keep) kp[j] = *reinterpret_cast<const unsigned*>(keep + (long long)(r + j * step) * ldkeep + col);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) row(r + j * step, xv[j], kp[j]);
    }
    for (; r < M; r += step) {
        const float4 xv = *reinterpret_cast<const float4*>(x + (long long)r * ldx + col);
        const unsigned kp = keep ? *reinterpret_cast<const unsigned*>(keep + (long long)r * ldkeep + col) : 0u;
        row(r, xv, kp);
    }
}

// rows of workgroups for the 16-byte elementwise forms: enough for ~8 workgroups per CU, never more than the rows give
static inline int vec_row_blocks(int M, int N) {
    int by = 2048 / t2_cdiv(N, 64);
    if (by < 1) by = 1;
    const int need = t2_cdiv(M, VRL);
    return by < need ? by : need;
}

extern "C" int t2amd_bn_act_fwd_f32(const float* x, long long ldx, float* y, long long ldy, int M, int N,
                                    const float* mean, const float* invstd, const float* gamma,
                                    const float* beta, int act, const uint8_t* keep, long long ldkeep,
                                    float keep_scale, const int* lens, int row_valid_T, void* stream) {
    T2_REQUIRE(x && y && mean && invstd && gamma && beta && M > 0 && N > 0, "bn_act_fwd: bad args");
    T2_REQUIRE(!lens || row_valid_T > 0, "bn_act_fwd: lens needs row_valid_T");
    if (!g_ew_scalar && vec4_ok(x, ldx, N) && vec4_ok(y, ldy, N) && t2_aligned16(mean) && t2_aligned16(invstd) && t2_aligned16(gamma) &&
        t2_aligned16(beta) && (!keep || (ldkeep % 4 == 0 && (reinterpret_cast<uintptr_t>(keep) & 3u) == 0))) {
        T2_LAUNCH((bn_act_fwd_vec_kernel<false>), dim3(t2_cdiv(N, 64), vec_row_blocks(M, N)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy,
                  M, N, mean, invstd, gamma, beta, act, keep, ldkeep, keep_scale, lens, row_valid_T, (unsigned short*)nullptr, 0, 0, 0);
        T2_LAUNCH_CHECK();
        return T2AMD_OK;
    }
    int blocks = t2_cdiv((long long)M * N, 256);
    if (blocks > 8192) blocks = 8192;
    T2_LAUNCH(bn_act_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, M, N, mean,
                       invstd, gamma, beta, act, keep, ldkeep, keep_scale, lens, row_valid_T);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// y AND its bf16 halo image in one pass (see bn_act_fwd_vec_kernel<true>); training-mode layers of the bf16 mode's convolution stacks.
extern "C" int t2amd_bn_act_fwd_img_f32(const float* x, long long ldx, float* y, long long ldy, int M, int N, const float* mean,
                                        const float* invstd, const float* gamma, const float* beta, int act, const uint8_t* keep,
                                        long long ldkeep, float keep_scale, void* y_img16, int T, int pad, void* stream) {
    T2_REQUIRE(x && y && mean && invstd && gamma && beta && y_img16 && M > 0 && N > 0, "bn_act_fwd_img: bad args");
    T2_REQUIRE(T > 0 && pad >= 0 && M % T == 0 && T >= 2 * pad, "bn_act_fwd_img: rows must be whole utterances of T >= 2 pad frames");
    T2_REQUIRE(vec4_ok(x, ldx, N) && vec4_ok(y, ldy, N) && t2_aligned16(mean) && t2_aligned16(invstd) && t2_aligned16(gamma) &&
                   t2_aligned16(beta) && (reinterpret_cast<uintptr_t>(y_img16) & 7u) == 0 &&
                   (!keep || (ldkeep % 4 == 0 && (reinterpret_cast<uintptr_t>(keep) & 3u) == 0)),
               "bn_act_fwd_img: needs N % 4 == 0 and 16-byte-aligned rows (the vector kernel)");
    T2_LAUNCH((bn_act_fwd_vec_kernel<true>), dim3(t2_cdiv(N, 64), vec_row_blocks(M, N)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy,
              M, N, mean, invstd, gamma, beta, act, keep, ldkeep, keep_scale, (const int*)nullptr, 0, (unsigned short*)y_img16, T, pad, M / T);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// backward stage 1: dbn = dy * keep*scale * act'(y);  dy <- dbn;  partial sums of dbn and dbn*xhat
__global__ void bn_act_bwd_stage1_kernel(float* __restrict__ dy, long long lddy, const float* __restrict__ y,
                                         long long ldy, const float* __restrict__ x, long long ldx, int M, int N,
                                         const float* __restrict__ mean, const float* __restrict__ invstd, int act,
                                         const uint8_t* __restrict__ keep, long long ldkeep, float keep_scale,
                                         double* __restrict__ ws) {
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rsub = threadIdx.x >> 6;
    const int rb = blockIdx.y;
    __shared__ double s1[4][64], s2[4][64];
    double a = 0.0, q = 0.0;
    if (col < N) {
        const float mu = mean[col], is = invstd[col];
        auto gate = [&](float g, float yv, unsigned kp) -> float {
            if (keep) {
                if (kp) {
                    g *= keep_scale;
                    if (act == 1) g = (yv > 0.f) ? g : 0.f;
                    else if (act == 2) { const float th = yv / keep_scale; g *= (1.f - th * th); }
                } else {
                    g = 0.f;
                }
            } else {
                if (act == 1) g = (yv > 0.f) ? g : 0.f;
                else if (act == 2) g *= (1.f - yv * yv);
            }
            return g;
        };
        // two rows per trip, every load issued before the first use (see colreduce_partial_kernel)
        const long long step = (long long)RB * 4;
        long long r = (long long)rb * 4 + rsub;
        for (; r + step < M; r += 2 * step) {
            const long long r1 = r + step;
            float g0 = dy[r * lddy + col], g1 = dy[r1 * lddy + col];
            const float y0 = y[r * ldy + col], y1 = y[r1 * ldy + col];
            const float x0 = x[r * ldx + col], x1 = x[r1 * ldx + col];
            unsigned k0 = 1, k1 = 1;
            if (keep) { k0 = keep[r * ldkeep + col]; k1 = keep[r1 * ldkeep + col]; }
            g0 = gate(g0, y0, k0);
            g1 = gate(g1, y1, k1);
            dy[r * lddy + col] = g0;
            dy[r1 * lddy + col] = g1;
            a += (double)g0; q += (double)g0 * (double)((x0 - mu) * is);
            a += (double)g1; q += (double)g1 * (double)((x1 - mu) * is);
        }
        for (; r < M; r += step) {
            float g = dy[r * lddy + col];
            const float yv = y[r * ldy + col];
            g = gate(g, yv, keep ? keep[r * ldkeep + col] : 1u);
            dy[r * lddy + col] = g;
            const float xhat = (x[r * ldx + col] - mu) * is;
            a += (double)g;
            q += (double)g * (double)xhat;
        }
    }
    s1[rsub][threadIdx.x & 63] = a;
    s2[rsub][threadIdx.x & 63] = q;
    __syncthreads();
    if (rsub == 0 && col < N) {
        const int c = threadIdx.x & 63;
        ws[(long long)rb * N + col] = s1[0][c] + s1[1][c] + s1[2][c] + s1[3][c];
        ws[(long long)(RB + rb) * N + col] = s2[0][c] + s2[1][c] + s2[2][c] + s2[3][c];
    }
}

// 16-byte form of stage 1 (layout of colreduce_partial_vec_kernel; two rows in flight per thread: 104 B).  The gate is the
// scalar kernel's, element by element.
__global__ __launch_bounds__(256) void bn_act_bwd_stage1_vec_kernel(float* __restrict__ dy, long long lddy, const float* __restrict__ y,
                                                                    long long ldy, const float* __restrict__ x, long long ldx, int M,
                                                                    int N, const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd, int act,
                                                                    const uint8_t* __restrict__ keep, long long ldkeep,
                                                                    float keep_scale, double* __restrict__ ws) {
    const int cl = threadIdx.x & (VCL - 1), rl = threadIdx.x / VCL;
    const int col0 = blockIdx.x * 64, col = col0 + cl * 4;
    const int rb = blockIdx.y;
    __shared__ double s1[VRL][64], s2[VRL][64];
    Dbl4 a = {}, q = {};
    if (col < N) {
        const float4 mu4 = *reinterpret_cast<const float4*>(mean + col), is4 = *reinterpret_cast<const float4*>(invstd + col);
        const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, is[4] = {is4.x, is4.y, is4.z, is4.w};
        auto gate = [&](float g, float yv, unsigned kp) -> float {
            if (keep) {
                if (kp) {
                    g *= keep_scale;
                    if (act == 1) g = (yv > 0.f) ? g : 0.f;
                    else if (act == 2) { const float th = yv / keep_scale; g *= (1.f - th * th); }
                } else {
                    g = 0.f;
                }
            } else {
                if (act == 1) g = (yv > 0.f) ? g : 0.f;
                else if (act == 2) g *= (1.f - yv * yv);
            }
            return g;
        };
        auto row = [&](long long r, const float4& g4, const float4& y4, const float4& x4, unsigned kp) {
            float g[4] = {g4.x, g4.y, g4.z, g4.w};
            const float yv[4] = {y4.x, y4.y, y4.z, y4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                g[j] = gate(g[j], yv[j], keep ? ((kp >> (8 * j)) & 0xffu) : 1u);
                a.v[j] += (double)g[j];
                q.v[j] += (double)g[j] * (double)((xv[j] - mu[j]) * is[j]);
            }
            *reinterpret_cast<float4*>(dy + r * lddy + col) = make_float4(g[0], g[1], g[2], g[3]);
        };
        const long long step = (long long)RB * VRL;
        long long r = (long long)rb * VRL + rl;
        for (; r + step < M; r += 2 * step) {
            const long long r1 = r + step;
            const float4 g0 = *reinterpret_cast<const float4*>(dy + r * lddy + col), g1 = *reinterpret_cast<const float4*>(dy + r1 * lddy + col);
            const float4 y0 = *reinterpret_cast<const float4*>(y + r * ldy + col), y1 = *reinterpret_cast<const float4*>(y + r1 * ldy + col);
            const float4 x0 = *reinterpret_cast<const float4*>(x + r * ldx + col), x1 = *reinterpret_cast<const float4*>(x + r1 * ldx + col);
            unsigned k0 = 0, k1 = 0;
            if (keep) {
                k0 = *reinterpret_cast<const unsigned*>(keep + r * ldkeep + col);
                k1 = *reinterpret_cast<const unsigned*>(keep + r1 * ldkeep + col);
            }
            row(r, g0, y0, x0, k0);
            row(r1, g1, y1, x1, k1);
        }
        for (; r < M; r += step) {
            const float4 g0 = *reinterpret_cast<const float4*>(dy + r * lddy + col);
            const float4 y0 = *reinterpret_cast<const float4*>(y + r * ldy + col);
            const float4 x0 = *reinterpret_cast<const float4*>(x + r * ldx + col);
            const unsigned k0 = keep ? *reinterpret_cast<const unsigned*>(keep + r * ldkeep + col) : 0u;
            row(r, g0, y0, x0, k0);
        }
    }
    vred_finish(s1, s2, a, q, true, cl, rl, rb, col0, N, ws);
}

__global__ __launch_bounds__(256) void bn_act_bwd_stage2_vec_kernel(float* __restrict__ dy, long long lddy, const float* __restrict__ x,
                                                                    long long ldx, int M, int N, const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                    const float* __restrict__ dgamma, const float* __restrict__ dbeta) {
    const int cl = threadIdx.x & (VCL - 1), rl = threadIdx.x / VCL;
    const int col = blockIdx.x * 64 + cl * 4;
    if (col >= N) return;
    const float invM = 1.0f / (float)M;
    float mu[4], is[4], ga[4], dg[4], db[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mu[j] = mean[col + j]; is[j] = invstd[col + j]; ga[j] = gamma[col + j]; dg[j] = dgamma[col + j]; db[j] = dbeta[col + j];
    }
    auto row = [&](int r, const float4& g4, const float4& x4) {
        const float g[4] = {g4.x, g4.y, g4.z, g4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xhat = (xv[j] - mu[j]) * is[j];
            o[j] = ga[j] * is[j] * (g[j] - db[j] * invM - xhat * dg[j] * invM);
        }
        *reinterpret_cast<float4*>(dy + (long long)r * lddy + col) = make_float4(o[0], o[1], o[2], o[3]);
    };
    const int step = gridDim.y * VRL;
    int r = blockIdx.y * VRL + rl;
    for (; r + 3 * step < M; r += 4 * step) {
        float4 g[4], xv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            g[j] = *reinterpret_cast<const float4*>(dy + (long long)(r + j * step) * lddy + col);
            xv[j] = *reinterpret_cast<const float4*>(x + (long long)(r + j * step) * ldx + col);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) row(r + j * step, g[j], xv[j]);
    }
    for (; r < M; r += step)
        row(r, *reinterpret_cast<const float4*>(dy + (long long)r * lddy + col), *reinterpret_cast<const float4*>(x + (long long)r * ldx + col));
}

// Stage 2 with its two followers folded in (round 6, bf16 mode's convolution backward): the BatchNorm-backward output dx leaves as the
// bf16 halo image the window products read (t2amd_cast_halo_bf16's layout, halo rows zeroed here) and as the column sums that are
// the convolution's bias gradient (t2amd_colsum_f32's partition and order: row lane rl of row block rb sums rows rb*VRL + rl,
// + RB*VRL, ... in double, then the row lanes, then the row blocks -- bit-identical to the separate pass), instead of as an f32
// slab that one kernel re-reads to cast and another to sum.  F32OUT: also keep the f32 slab (a consumer that needs it).
template <bool F32OUT>
__global__ __launch_bounds__(256) void bn_act_bwd_stage2_img_kernel(float* __restrict__ dy, long long lddy, const float* __restrict__ x,
                                                                    long long ldx, int M, int N, const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                    const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                                    unsigned short* __restrict__ img, int T, int pad, int nb,
                                                                    double* __restrict__ ws) {
    const int cl = threadIdx.x & (VCL - 1), rl = threadIdx.x / VCL;
    const int col0 = blockIdx.x * 64, col = col0 + cl * 4;
    const int rb = blockIdx.y;
    __shared__ double s1[VRL][64], s2[1][64];
    Dbl4 a = {}, q = {};
    if (col < N) {
        const float invM = 1.0f / (float)M;
        const int Tp = T + 2 * pad;
        float mu[4], is[4], ga[4], dg[4], db[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            mu[j] = mean[col + j]; is[j] = invstd[col + j]; ga[j] = gamma[col + j]; dg[j] = dgamma[col + j]; db[j] = dbeta[col + j];
        }
        auto row = [&](long long r, const float4& g4, const float4& x4) {
            const float g[4] = {g4.x, g4.y, g4.z, g4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xhat = (xv[j] - mu[j]) * is[j];
                o[j] = ga[j] * is[j] * (g[j] - db[j] * invM - xhat * dg[j] * invM);
                a.v[j] += (double)o[j];
            }
            if (F32OUT) *reinterpret_cast<float4*>(dy + r * lddy + col) = make_float4(o[0], o[1], o[2], o[3]);
            const int b = (int)(r / T), t = (int)(r - (long long)b * T);
            unsigned short* ib = img + ((long long)b * Tp) * N + col;
            uint2 v;
            v.x = (unsigned)t2_f32_to_bf16(o[0]) | ((unsigned)t2_f32_to_bf16(o[1]) << 16);
            v.y = (unsigned)t2_f32_to_bf16(o[2]) | ((unsigned)t2_f32_to_bf16(o[3]) << 16);
            *reinterpret_cast<uint2*>(ib + (long long)(pad + t) * N) = v;
            const uint2 z = make_uint2(0u, 0u);
            if (t < pad) *reinterpret_cast<uint2*>(ib + (long long)t * N) = z;                               // halo in front of the utterance
            if (t >= T - pad) *reinterpret_cast<uint2*>(ib + (long long)(t + 2 * pad) * N) = z;               // halo behind it
            if (b == nb - 1 && t < 2 * pad) *reinterpret_cast<uint2*>(ib + (long long)(Tp + t) * N) = z;      // the 2 pad rows past the image
        };
        const long long step = (long long)RB * VRL;
        long long r = (long long)rb * VRL + rl;
        for (; r + 3 * step < M; r += 4 * step) {
            float4 g[4], xv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                g[j] = *reinterpret_cast<const float4*>(dy + (r + j * step) * lddy + col);
                xv[j] = *reinterpret_cast<const float4*>(x + (r + j * step) * ldx + col);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) row(r + j * step, g[j], xv[j]);
        }
        for (; r < M; r += step)
            row(r, *reinterpret_cast<const float4*>(dy + r * lddy + col), *reinterpret_cast<const float4*>(x + r * ldx + col));
    }
    vred_finish(s1, s2, a, q, false, cl, rl, rb, col0, N, ws);
}

__global__ void bn_bwd_finalize_kernel(const double* __restrict__ ws, int N, float* dgamma, float* dbeta) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double s = 0.0, q = 0.0;
    for (int rb = 0; rb < RB; ++rb) {
        s += ws[(long long)rb * N + n];
        q += ws[(long long)(RB + rb) * N + n];
    }
    dbeta[n] = (float)s;
    dgamma[n] = (float)q;
}

// stage 2: dx = gamma*invstd * (dbn - dbeta/M - xhat*dgamma/M), in place over dy
__global__ void bn_act_bwd_stage2_kernel(float* __restrict__ dy, long long lddy, const float* __restrict__ x,
                                         long long ldx, int M, int N, const float* __restrict__ mean,
                                         const float* __restrict__ invstd, const float* __restrict__ gamma,
                                         const float* __restrict__ dgamma, const float* __restrict__ dbeta) {
    const long long total = (long long)M * N;
    const float invM = 1.0f / (float)M;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / N;
        const int n = (int)(i - r * N);
        const float is = invstd[n];
        const float xhat = (x[r * ldx + n] - mean[n]) * is;
        const float g = dy[r * lddy + n];
        dy[r * lddy + n] = gamma[n] * is * (g - dbeta[n] * invM - xhat * dgamma[n] * invM);
    }
}

extern "C" int t2amd_bn_act_bwd_f32(float* dy, long long lddy, const float* y, long long ldy, const float* x,
                                    long long ldx, int M, int N, const float* mean, const float* invstd,
                                    const float* gamma, int act, const uint8_t* keep, long long ldkeep,
                                    float keep_scale, double* ws, float* dgamma, float* dbeta, void* stream) {
    T2_REQUIRE(dy && y && x && mean && invstd && gamma && ws && dgamma && dbeta && M > 0 && N > 0, "bn_act_bwd: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (!g_ew_scalar && vec4_ok(dy, lddy, N) && vec4_ok(y, ldy, N) && vec4_ok(x, ldx, N) && t2_aligned16(mean) && t2_aligned16(invstd) &&
        (!keep || (ldkeep % 4 == 0 && (reinterpret_cast<uintptr_t>(keep) & 3u) == 0))) {
        T2_LAUNCH(bn_act_bwd_stage1_vec_kernel, dim3(t2_cdiv(N, 64), RB), dim3(256), 0, s, dy, lddy, y, ldy, x, ldx, M, N, mean, invstd,
                  act, keep, ldkeep, keep_scale, ws);
        T2_LAUNCH(bn_bwd_finalize_kernel, dim3(t2_cdiv(N, 128)), dim3(128), 0, s, ws, N, dgamma, dbeta);
        T2_LAUNCH(bn_act_bwd_stage2_vec_kernel, dim3(t2_cdiv(N, 64), vec_row_blocks(M, N)), dim3(256), 0, s, dy, lddy, x, ldx, M, N, mean,
                  invstd, gamma, dgamma, dbeta);
        T2_LAUNCH_CHECK();
        return T2AMD_OK;
    }
    T2_LAUNCH(bn_act_bwd_stage1_kernel, dim3(t2_cdiv(N, 64), RB), dim3(256), 0, s, dy, lddy, y, ldy, x, ldx, M,
                       N, mean, invstd, act, keep, ldkeep, keep_scale, ws);
    T2_LAUNCH(bn_bwd_finalize_kernel, dim3(t2_cdiv(N, 128)), dim3(128), 0, s, ws, N, dgamma, dbeta);
    int blocks = t2_cdiv((long long)M * N, 256);
    if (blocks > 8192) blocks = 8192;
    T2_LAUNCH(bn_act_bwd_stage2_kernel, dim3(blocks), dim3(256), 0, s, dy, lddy, x, ldx, M, N, mean, invstd,
                       gamma, dgamma, dbeta);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// The same backward with stage 2's output leaving as the bf16 halo image + the bias gradient (see bn_act_bwd_stage2_img_kernel).
extern "C" int t2amd_bn_act_bwd_img_f32(float* dy, long long lddy, const float* y, long long ldy, const float* x, long long ldx, int M,
                                        int N, const float* mean, const float* invstd, const float* gamma, int act, const uint8_t* keep,
                                        long long ldkeep, float keep_scale, double* ws, float* dgamma, float* dbeta, void* dx_img16,
                                        int T, int pad, float* dbias, int keep_f32, void* stream) {
    T2_REQUIRE(dy && y && x && mean && invstd && gamma && ws && dgamma && dbeta && dx_img16 && dbias && M > 0 && N > 0, "bn_act_bwd_img: bad args");
    T2_REQUIRE(T > 0 && pad >= 0 && M % T == 0 && T >= 2 * pad, "bn_act_bwd_img: rows must be whole utterances of T >= 2 pad frames");
    T2_REQUIRE(vec4_ok(dy, lddy, N) && vec4_ok(y, ldy, N) && vec4_ok(x, ldx, N) && t2_aligned16(mean) && t2_aligned16(invstd) &&
                   (reinterpret_cast<uintptr_t>(dx_img16) & 7u) == 0 &&
                   (!keep || (ldkeep % 4 == 0 && (reinterpret_cast<uintptr_t>(keep) & 3u) == 0)),
               "bn_act_bwd_img: needs N % 4 == 0 and 16-byte-aligned rows (the vector kernels)");
    hipStream_t s = (hipStream_t)stream;
    T2_LAUNCH(bn_act_bwd_stage1_vec_kernel, dim3(t2_cdiv(N, 64), RB), dim3(256), 0, s, dy, lddy, y, ldy, x, ldx, M, N, mean, invstd,
              act, keep, ldkeep, keep_scale, ws);
    T2_LAUNCH(bn_bwd_finalize_kernel, dim3(t2_cdiv(N, 128)), dim3(128), 0, s, ws, N, dgamma, dbeta);
    if (keep_f32)
        T2_LAUNCH((bn_act_bwd_stage2_img_kernel<true>), dim3(t2_cdiv(N, 64), RB), dim3(256), 0, s, dy, lddy, x, ldx, M, N, mean, invstd,
                  gamma, dgamma, dbeta, (unsigned short*)dx_img16, T, pad, M / T, ws);
    else
        T2_LAUNCH((bn_act_bwd_stage2_img_kernel<false>), dim3(t2_cdiv(N, 64), RB), dim3(256), 0, s, dy, lddy, x, ldx, M, N, mean, invstd,
                  gamma, dgamma, dbeta, (unsigned short*)dx_img16, T, pad, M / T, ws);
    T2_LAUNCH(colsum_finalize_kernel, dim3(t2_cdiv(N, 128)), dim3(128), 0, s, ws, N, dbias, 0);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// ---------------------------------------------------------------------------------------
// embedding
// ---------------------------------------------------------------------------------------
__global__ void embedding_fwd_kernel(const long long* __restrict__ ids, const float* __restrict__ table,
                                     float* __restrict__ out, long long rows, int dim, int nsym) {
    const long long total = rows * dim;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / dim;
        const int c = (int)(i - r * dim);
        long long id = ids[r];
        if (id < 0) id = 0;
        if (id >= nsym) id = nsym - 1;
        out[i] = table[id * dim + c];
    }
}

extern "C" int t2amd_embedding_fwd_f32(const long long* ids, const float* table, float* out, long long rows,
                                       int dim, int n_symbols, void* stream) {
    T2_REQUIRE(ids && table && out && rows > 0 && dim > 0 && n_symbols > 0, "embedding_fwd: bad args");
    int blocks = t2_cdiv(rows * dim, 256);
    if (blocks > 4096) blocks = 4096;
    T2_LAUNCH(embedding_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ids, table, out, rows, dim,
                       n_symbols);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// One workgroup per (symbol, 32-column slab): no atomics, fixed summation order.  Rows are taken in chunks of
// EMB_CHUNK: the workgroup first compacts the chunk's matching row numbers into LDS (ballot + prefix, ascending),
// then eight row lanes x 32 columns sum them -- lane l takes list entries l, l+8, ... four at a time with the loads
// issued first -- and the eight lane sums are added in lane order.  (The first version tested every row inside the
// accumulation loop, 390 us for 11 k rows; a single accumulation chain per column still took 300 us because the
// padding symbol owns ~40 % of the rows.)
#define EMB_CHUNK 4096
#define EMB_COLS 32
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const long long* __restrict__ ids, const float* __restrict__ dout,
                                                            float* __restrict__ dtable, long long rows, int dim) {
    const int sym = blockIdx.x;
    const int cl = threadIdx.x & (EMB_COLS - 1), rl = threadIdx.x / EMB_COLS;      // column in slab, row lane 0..7
    const int c = blockIdx.y * EMB_COLS + cl;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ int list[EMB_CHUNK];
    __shared__ int wcount[4];
    __shared__ int total;
    __shared__ float lsum[8][EMB_COLS];
    float acc = 0.f;
    for (long long r0 = 0; r0 < rows; r0 += EMB_CHUNK) {
        if (threadIdx.x == 0) total = 0;
        __syncthreads();
        const int lim = (rows - r0 < EMB_CHUNK) ? (int)(rows - r0) : EMB_CHUNK;
        // this thread's 16 ids of the chunk, all loads issued before the compaction loop consumes them
        long long myid[EMB_CHUNK / 256];
#pragma unroll
        for (int j = 0; j < EMB_CHUNK / 256; ++j) {
            const int k = j * 256 + threadIdx.x;
            myid[j] = ids[r0 + (k < lim ? k : 0)];
        }
#pragma unroll
        for (int j = 0; j < EMB_CHUNK / 256; ++j) {
            const int base = j * 256;
            if (base >= lim) break;
            const int k = base + threadIdx.x;
            const bool hit = (k < lim) && (myid[j] == sym);
            const unsigned long long m = __ballot(hit);
            if (lane == 0) wcount[wave] = __popcll(m);
            __syncthreads();
            int off = total;
            for (int w = 0; w < wave; ++w) off += wcount[w];
            if (hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = k;
            __syncthreads();
            if (threadIdx.x == 0) total += wcount[0] + wcount[1] + wcount[2] + wcount[3];
            __syncthreads();
        }
        const int n = total;
        if (c < dim) {
            const float* __restrict__ src = dout + r0 * dim + c;
            int k = rl;
            for (; k + 24 < n; k += 32) {
                const float v0 = src[(long long)list[k] * dim], v1 = src[(long long)list[k + 8] * dim];
                const float v2 = src[(long long)list[k + 16] * dim], v3 = src[(long long)list[k + 24] * dim];
                acc += v0; acc += v1; acc += v2; acc += v3;
            }
            for (; k < n; k += 8) acc += src[(long long)list[k] * dim];
        }
        __syncthreads();
    }
    lsum[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c < dim) {
        float s = lsum[0][cl];
#pragma unroll
        for (int l = 1; l < 8; ++l) s += lsum[l][cl];
        dtable[(long long)sym * dim + c] = s;
    }
}

extern "C" int t2amd_embedding_bwd_f32(const long long* ids, const float* dout, float* dtable, float* ws,
                                       long long rows, int dim, int n_symbols, void* stream) {
    T2_REQUIRE(ids && dout && dtable && rows > 0 && dim > 0 && n_symbols > 0, "embedding_bwd: bad args");
    (void)ws;      // kept in the signature: earlier versions needed a partial-sum workspace
    T2_LAUNCH(embedding_bwd_kernel, dim3(n_symbols, t2_cdiv(dim, EMB_COLS)), dim3(256), 0, (hipStream_t)stream,
              ids, dout, dtable, rows, dim);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// ---------------------------------------------------------------------------------------
// Philox4x32-10 keep mask
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ void philox_keep_kernel(uint8_t* __restrict__ out, long long n, float p, unsigned long long seed,
                                   unsigned long long offset) {
    const long long nquad = (n + 3) / 4;
    const bool word_ok = (reinterpret_cast<uintptr_t>(out) & 3u) == 0;
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < nquad;
         q += (long long)gridDim.x * blockDim.x) {
        const unsigned long long ctr = offset / 4 + (unsigned long long)q;
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        if (q * 4 + 3 < n && word_ok) {                 // the four bytes of a quad as one store (same bytes)
            unsigned w = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float uu = (float)(c[j] >> 8) * (1.0f / 16777216.0f);   // [0,1)
                w |= ((uu >= p) ? 1u : 0u) << (8 * j);
            }
            *reinterpret_cast<unsigned*>(out + q * 4) = w;
            continue;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long i = q * 4 + j;
            if (i < n) {
                const float uu = (float)(c[j] >> 8) * (1.0f / 16777216.0f);   // [0,1)
                out[i] = (uu >= p) ? 1 : 0;
            }
        }
    }
}

extern "C" int t2amd_philox_keep_mask(uint8_t* out, long long n, float p, unsigned long long seed,
                                      unsigned long long offset, void* stream) {
    T2_REQUIRE(out && n > 0 && p >= 0.f && p < 1.f, "philox: bad args");
    T2_REQUIRE(offset % 4 == 0, "philox: offset must be a multiple of 4");
    int blocks = t2_cdiv((n + 3) / 4, 256);
    if (blocks > 8192) blocks = 8192;
    T2_LAUNCH(philox_keep_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, n, p, seed, offset);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// ---------------------------------------------------------------------------------------
// fill / copy / transpose
// ---------------------------------------------------------------------------------------
__global__ void fill_kernel(float* p, long long n, float v) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        p[i] = v;
}
// f32 -> bf16 (round to nearest even), n a multiple of 4, both 8-byte aligned
__global__ void cast_bf16_kernel(const float4* __restrict__ src, uint2* __restrict__ dst, long long n4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        uint2 o;
        o.x = (unsigned)t2_f32_to_bf16(v.x) | ((unsigned)t2_f32_to_bf16(v.y) << 16);
        o.y = (unsigned)t2_f32_to_bf16(v.z) | ((unsigned)t2_f32_to_bf16(v.w) << 16);
        dst[i] = o;
    }
}

extern "C" int t2amd_cast_bf16_f32(const float* src, void* dst, long long n, void* stream) {
    T2_REQUIRE(src && dst && n > 0 && n % 4 == 0, "cast_bf16: n must be a positive multiple of 4");
    T2_REQUIRE(t2_aligned16(src) && (reinterpret_cast<uintptr_t>(dst) & 7u) == 0, "cast_bf16: alignment");
    int blocks = t2_cdiv(n / 4, 256);
    if (blocks > 4096) blocks = 4096;
    T2_LAUNCH(cast_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(src),
              reinterpret_cast<uint2*>(dst), n / 4);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// f32 [rows][K] -> split-bf16 image (header: t2amd_split_bf16x3_f32).  One thread per 4 consecutive k: one float4 in, two 8-byte
// stores out (4 hi, 4 lo of the same group of 16).
__global__ void split_bf16x3_kernel(const float* __restrict__ src, long long lds_, unsigned short* __restrict__ dst, long long ldd,
                                    long long rows, int K4) {
    const long long total = rows * K4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / K4;
        const int k = (int)(i - r * K4) * 4;
        const float4 v = *reinterpret_cast<const float4*>(src + r * lds_ + k);
        unsigned short h[4], l[4];
        t2_split_bf16(v.x, h[0], l[0]); t2_split_bf16(v.y, h[1], l[1]);
        t2_split_bf16(v.z, h[2], l[2]); t2_split_bf16(v.w, h[3], l[3]);
        unsigned short* const d = dst + r * ldd * 2 + t2_x3_pos(k);
        *reinterpret_cast<uint2*>(d) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
        *reinterpret_cast<uint2*>(d + 16) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
    }
}
extern "C" int t2amd_split_bf16x3_f32(const float* src, long long lds, void* dst, long long ldd, long long rows, int K, void* stream) {
    T2_REQUIRE(src && dst && rows > 0 && K > 0 && K % 16 == 0, "split_bf16x3: K must be a positive multiple of 16");
    T2_REQUIRE(lds >= K && lds % 4 == 0 && ldd >= K && ldd % 16 == 0, "split_bf16x3: row strides (lds % 4 == 0, ldd % 16 == 0, both >= K)");
    T2_REQUIRE(t2_aligned16(src) && t2_aligned16(dst), "split_bf16x3: alignment");
    int blocks = t2_cdiv(rows * (K / 4), 256);
    if (blocks > 8192) blocks = 8192;
    T2_LAUNCH(split_bf16x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, lds, reinterpret_cast<unsigned short*>(dst), ldd,
              rows, K / 4);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

extern "C" int t2amd_fill_f32(float* p, long long n, float v, void* stream) {
    T2_REQUIRE(p && n > 0, "fill: bad args");
    int blocks = t2_cdiv(n, 256);
    if (blocks > 4096) blocks = 4096;
    T2_LAUNCH(fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, n, v);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

__global__ void copy2d_kernel(const float* __restrict__ src, long long lds_, const float* __restrict__ src2,
                              long long lds2, float* __restrict__ dst, long long ldd, int rows, int cols) {
    const long long total = (long long)rows * cols;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / cols;
        const int c = (int)(i - r * cols);
        float v = src[r * lds_ + c];
        if (src2) v += src2[r * lds2 + c];
        dst[r * ldd + c] = v;
    }
}
extern "C" int t2amd_copy2d_f32(const float* src, long long lds_, const float* src2, long long lds2, float* dst,
                                long long ldd, int rows, int cols, void* stream) {
    T2_REQUIRE(src && dst && rows > 0 && cols > 0, "copy2d: bad args");
    int blocks = t2_cdiv((long long)rows * cols, 256);
    if (blocks > 4096) blocks = 4096;
    T2_LAUNCH(copy2d_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, lds_, src2, lds2, dst, ldd,
                       rows, cols);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// 32x32 LDS-tiled transpose, batched
__global__ void transpose_kernel(const float* __restrict__ src, long long lds_, float* __restrict__ dst, long long ldd,
                                 int rows, int cols, long long sstride, long long dstride) {
    __shared__ float tile[32][33];
    const float* s = src + (long long)blockIdx.z * sstride;
    float* d = dst + (long long)blockIdx.z * dstride;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < rows && c < cols) ? s[(long long)r * lds_ + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < cols && r < rows) d[(long long)c * ldd + r] = tile[tx][j];
    }
}
extern "C" int t2amd_transpose_f32(const float* src, long long lds_, float* dst, long long ldd, int rows, int cols,
                                   int batch, long long sstride, long long dstride, void* stream) {
    T2_REQUIRE(src && dst && rows > 0 && cols > 0 && batch > 0, "transpose: bad args");
    dim3 grid(t2_cdiv(cols, 32), t2_cdiv(rows, 32), batch);
    T2_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "transpose: grid too large");
    T2_LAUNCH(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, lds_, dst, ldd, rows, cols, sstride,
                       dstride);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// ---------------------------------------------------------------------------------------
// model-boundary layout kernels
// ---------------------------------------------------------------------------------------
// mels [B][C][To] -> X0 [To][B][C], shifted by one frame (X0[0] = go frame = 0)
__global__ void frames_to_tm_kernel(const float* __restrict__ mels, float* __restrict__ x0, int B, int C, int To) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    // read mels[b][c][t-1] coalesced along t
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, t = t0 + tx;          // destination frame index t
        float v = 0.f;
        if (c < C && t < To && t >= 1) v = mels[((long long)b * C + c) * To + (t - 1)];
        tile[j][tx] = v;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int t = t0 + j, c = c0 + tx;
        if (t < To && c < C) x0[((long long)t * B + b) * C + c] = tile[tx][j];
    }
}
extern "C" int t2amd_frames_to_time_major_f32(const float* mels, float* x0, int B, int C, int To, void* stream) {
    T2_REQUIRE(mels && x0 && B > 0 && C > 0 && To > 0, "frames_to_tm: bad args");
    dim3 grid(t2_cdiv(To, 32), t2_cdiv(C, 32), B);
    T2_LAUNCH(frames_to_tm_kernel, grid, dim3(256), 0, (hipStream_t)stream, mels, x0, B, C, To);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

__global__ void split_projection_kernel(const float* __restrict__ pg, float* __restrict__ mel_cl,
                                        float* __restrict__ gate, const int* __restrict__ out_lens, int B, int C,
                                        int To) {
    const long long total = (long long)To * B * (C + 1);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / (C + 1);           // = t*B + b
        const int c = (int)(i - row * (C + 1));
        const int t = (int)(row / B), b = (int)(row - (long long)t * B);
        const float v = pg[i];
        if (c < C) {
            mel_cl[((long long)b * To + t) * C + c] = v;
        } else {
            const bool pad = out_lens && t >= out_lens[b];
            gate[(long long)b * To + t] = pad ? 1e3f : v;
        }
    }
}
extern "C" int t2amd_split_projection_f32(const float* pg, float* mel_cl, float* gate, const int* out_lens, int B,
                                          int C, int To, void* stream) {
    T2_REQUIRE(pg && mel_cl && gate && B > 0 && C > 0 && To > 0, "split_projection: bad args");
    int blocks = t2_cdiv((long long)To * B * (C + 1), 256);
    if (blocks > 8192) blocks = 8192;
    T2_LAUNCH(split_projection_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pg, mel_cl, gate,
                       out_lens, B, C, To);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// [B][To][C] -> [B][C][To] with padding mask; mel_cl zeroed in place at padded frames
__global__ void finalize_outputs_kernel(float* __restrict__ mel_cl, const float* __restrict__ post_cl,
                                        float* __restrict__ mel, float* __restrict__ mel_post,
                                        const int* __restrict__ out_lens, int B, int C, int To) {
    __shared__ float t1[32][33], t2[32][33];
    const int b = blockIdx.z;
    const int c0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int len = out_lens ? out_lens[b] : To;
    for (int j = ty; j < 32; j += 8) {
        const int t = t0 + j, c = c0 + tx;
        float m = 0.f, q = 0.f;
        if (t < To && c < C) {
            const long long idx = ((long long)b * To + t) * C + c;
            if (t < len) {
                m = mel_cl[idx];
                q = post_cl ? m + post_cl[idx] : 0.f;
            } else {
                mel_cl[idx] = 0.f;
            }
        }
        t1[j][tx] = m;
        t2[j][tx] = q;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, t = t0 + tx;
        if (c < C && t < To) {
            const long long o = ((long long)b * C + c) * To + t;
            mel[o] = t1[tx][j];
            if (mel_post) mel_post[o] = t2[tx][j];
        }
    }
}
extern "C" int t2amd_finalize_outputs_f32(float* mel_cl, const float* post_cl, float* mel, float* mel_post,
                                          const int* out_lens, int B, int C, int To, void* stream) {
    T2_REQUIRE(mel_cl && mel && B > 0 && C > 0 && To > 0, "finalize_outputs: bad args");
    T2_REQUIRE((post_cl != nullptr) == (mel_post != nullptr), "finalize_outputs: post_cl and mel_post go together");
    dim3 grid(t2_cdiv(C, 32), t2_cdiv(To, 32), B);
    T2_REQUIRE(grid.y <= 65535, "finalize_outputs: To too large");
    T2_LAUNCH(finalize_outputs_kernel, grid, dim3(256), 0, (hipStream_t)stream, mel_cl, post_cl, mel, mel_post,
                       out_lens, B, C, To);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// dmel, dmel_post [B][C][To] -> dpost_cl = dmel_post^T ; dmel_cl = dmel^T + dmel_post^T
__global__ void grads_to_cl_kernel(const float* __restrict__ dmel, const float* __restrict__ dmel_post,
                                   float* __restrict__ dmel_cl, float* __restrict__ dpost_cl, int B, int C, int To) {
    __shared__ float t1[32][33], t2[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, t = t0 + tx;
        float a = 0.f, q = 0.f;
        if (c < C && t < To) {
            const long long idx = ((long long)b * C + c) * To + t;
            if (dmel) a = dmel[idx];
            if (dmel_post) q = dmel_post[idx];
        }
        t1[j][tx] = a;
        t2[j][tx] = q;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int t = t0 + j, c = c0 + tx;
        if (t < To && c < C) {
            const long long o = ((long long)b * To + t) * C + c;
            dmel_cl[o] = t1[tx][j] + t2[tx][j];
            dpost_cl[o] = t2[tx][j];
        }
    }
}
extern "C" int t2amd_grads_to_channel_last_f32(const float* dmel, const float* dmel_post, float* dmel_cl,
                                               float* dpost_cl, int B, int C, int To, void* stream) {
    T2_REQUIRE(dmel_cl && dpost_cl && B > 0 && C > 0 && To > 0, "grads_to_cl: bad args");
    dim3 grid(t2_cdiv(To, 32), t2_cdiv(C, 32), B);
    T2_LAUNCH(grads_to_cl_kernel, grid, dim3(256), 0, (hipStream_t)stream, dmel, dmel_post, dmel_cl, dpost_cl, B,
                       C, To);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

__global__ void gather_dout_kernel(const float* __restrict__ dmel_cl, const float* __restrict__ dgate,
                                   float* __restrict__ dout, int B, int C, int To) {
    const long long total = (long long)To * B * (C + 1);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / (C + 1);
        const int c = (int)(i - row * (C + 1));
        const int t = (int)(row / B), b = (int)(row - (long long)t * B);
        float v;
        if (c < C) v = dmel_cl[((long long)b * To + t) * C + c];
        else v = dgate ? dgate[(long long)b * To + t] : 0.f;
        dout[i] = v;
    }
}
extern "C" int t2amd_gather_dout_f32(const float* dmel_cl, const float* dgate, float* dout, int B, int C, int To,
                                     void* stream) {
    T2_REQUIRE(dmel_cl && dout && B > 0 && C > 0 && To > 0, "gather_dout: bad args");
    int blocks = t2_cdiv((long long)To * B * (C + 1), 256);
    if (blocks > 8192) blocks = 8192;
    T2_LAUNCH(gather_dout_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dmel_cl, dgate, dout, B, C,
                       To);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

// Prenet backward elementwise part: y = relu(pre) * keep * scale  =>  dpre = (y > 0) ? dy*scale : 0
// (y > 0 exactly when the unit was kept and the relu was active).  In place over dy.
__global__ void relu_dropout_bwd_kernel(float* __restrict__ dy, const float* __restrict__ y, float scale, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dy[i] = (y[i] > 0.f) ? dy[i] * scale : 0.f;
}
__global__ void relu_dropout_bwd_vec_kernel(float4* __restrict__ dy, const float4* __restrict__ y, float scale, long long n4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 g = dy[i], v = y[i];
        dy[i] = make_float4((v.x > 0.f) ? g.x * scale : 0.f, (v.y > 0.f) ? g.y * scale : 0.f, (v.z > 0.f) ? g.z * scale : 0.f,
                            (v.w > 0.f) ? g.w * scale : 0.f);
    }
}
extern "C" int t2amd_relu_dropout_bwd_f32(float* dy, const float* y, float scale, long long n, void* stream) {
    T2_REQUIRE(dy && y && n > 0, "relu_dropout_bwd: bad args");
    if (!g_ew_scalar && n % 4 == 0 && t2_aligned16(dy) && t2_aligned16(y)) {
        int b4 = t2_cdiv(n / 4, 256);
        if (b4 > 8192) b4 = 8192;
        T2_LAUNCH(relu_dropout_bwd_vec_kernel, dim3(b4), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<float4*>(dy),
                  reinterpret_cast<const float4*>(y), scale, n / 4);
        T2_LAUNCH_CHECK();
        return T2AMD_OK;
    }
    int blocks = t2_cdiv(n, 256);
    if (blocks > 8192) blocks = 8192;
    T2_LAUNCH(relu_dropout_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, y, scale, n);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}
