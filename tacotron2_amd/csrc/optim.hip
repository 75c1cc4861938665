// Global-norm gradient clipping + Adam in two launches over ALL parameter tensors.
//
// Replaces, for the engine's training step, reference train.py:233-236
//     grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), hparams.grad_clip_thresh)
//     optimizer.step()                                    # torch.optim.Adam(lr, weight_decay), train.py:170-171
// which torch runs as ~12 multi-tensor passes over the 28.2 M parameters (norms, scale, add, lerp, mul, addcmul, sqrt,
// div, add, addcdiv): ~3 GB of HBM traffic per step.  Here: one read of the gradients for the norm (113 MB) and one
// pass that reads p, g, m, v and writes p, m, v (28 B per element, 790 MB) with the clip factor applied on the fly.
// Both kernels are HBM-bound streams; nothing is staged through LDS except the block reduction.
//
// Arithmetic follows torch's non-fused Adam (torch/optim/adam.py, _single_tensor_adam; L2 weight decay, no amsgrad):
//     g  = g * clip_coef (+ weight_decay * p)
//     m += (g - m) * (1 - beta1);   v = v * beta2 + (1 - beta2) * g * g
//     p += -(lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)          bc_k = 1 - beta_k^step, formed on the host in double
// clip_coef = min(1, max_norm / (norm + 1e-6)) as torch.nn.utils.clip_grad_norm_ forms it.
#include "common.h"

#define OPT_CHUNK 4096       // elements per workgroup: 256 threads x 16, 4-byte lanes (no alignment requirement:
                             // gradients may be views at odd offsets into the data-parallel flat buckets)

// tensor owning workgroup `blk`: first_block[] is ascending, the scan is wave-uniform (scalar unit)
__device__ __forceinline__ int opt_find_tensor(const t2amd_tensor_list& L, int blk) {
    int t = 0;
    while (t + 1 < L.count && blk >= L.first_block[t + 1]) ++t;
    return t;
}

// ws[blk] = sum of squares of this workgroup's chunk (double)
__global__ void __launch_bounds__(256) grad_sumsq_kernel(t2amd_tensor_list L, double* __restrict__ ws) {
    __shared__ double red[4];
    const int blk = blockIdx.x;
    const int t = opt_find_tensor(L, blk);
    const float* __restrict__ g = (const float*)L.grad[t];
    const long long n = L.numel[t];
    const long long base = (long long)(blk - L.first_block[t]) * OPT_CHUNK;
    double acc = 0.0;
#pragma unroll 4
    for (int k = 0; k < OPT_CHUNK / 256; ++k) {
        const long long i = base + k * 256 + threadIdx.x;
        if (i < n) {
            const float x = g[i];
            acc += (double)x * (double)x;
        }
    }
    // wave reduction in double through two 32-bit halves is not needed: shuffle the double directly
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) ws[blk] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = ||g||_2 (float), out[1] = clip coefficient; one workgroup, fixed summation order
__global__ void __launch_bounds__(256) grad_norm_finish_kernel(const double* __restrict__ ws, int nblocks, float max_norm,
                                                               float* __restrict__ out) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) acc += ws[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
        float coef = 1.0f;
        if (max_norm > 0.0f) {
            coef = max_norm / (norm + 1e-6f);
            coef = coef > 1.0f ? 1.0f : coef;
        }
        out[0] = norm;
        out[1] = coef;
    }
}

__global__ void __launch_bounds__(256) adam_step_kernel(t2amd_tensor_list L, t2amd_adam_hyper h,
                                                        const float* __restrict__ clip) {
    const int blk = blockIdx.x;
    const int t = opt_find_tensor(L, blk);
    float* __restrict__ p = (float*)L.param[t];
    const float* __restrict__ g = (const float*)L.grad[t];
    float* __restrict__ m = (float*)L.exp_avg[t];
    float* __restrict__ v = (float*)L.exp_avg_sq[t];
    const long long n = L.numel[t];
    const long long base = (long long)(blk - L.first_block[t]) * OPT_CHUNK;
    const float coef = clip ? clip[1] : 1.0f;
    // A non-finite global norm (overflow in the bf16 compute mode, a NaN batch) skips the whole update on the device:
    // weights and moments keep their values, exactly what the reference's amp loss scaler does with such a step.
    if (clip && !(fabsf(clip[0]) <= 3.0e38f)) return;
#pragma unroll 4
    for (int k = 0; k < OPT_CHUNK / 256; ++k) {
        const long long i = base + k * 256 + threadIdx.x;
        if (i < n) {
            const float pi = p[i];
            float gi = g[i] * coef;
            if (h.weight_decay != 0.0f) gi = fmaf(h.weight_decay, pi, gi);
            float mi = m[i], vi = v[i];
            mi = fmaf(gi - mi, h.one_minus_beta1, mi);
            vi = fmaf(h.one_minus_beta2 * gi, gi, vi * h.beta2);
            const float denom = sqrtf(vi) / h.bc2_sqrt + h.eps;
            m[i] = mi;
            v[i] = vi;
            p[i] = fmaf(-h.step_size, mi / denom, pi);
        }
    }
}

static int opt_check_list(const t2amd_tensor_list* L, bool need_state, int* nblocks) {
    T2_REQUIRE(L != nullptr, "optim: null tensor list");
    T2_REQUIRE(L->count > 0 && L->count <= T2AMD_MAX_TENSORS, "optim: tensor count out of range");
    long long blocks = 0;
    for (int t = 0; t < L->count; ++t) {
        T2_REQUIRE(L->numel[t] > 0, "optim: empty tensor");
        T2_REQUIRE(L->grad[t] != nullptr, "optim: null gradient");
        if (need_state) T2_REQUIRE(L->param[t] && L->exp_avg[t] && L->exp_avg_sq[t], "optim: null parameter / moment buffer");
        T2_REQUIRE(L->first_block[t] == blocks, "optim: first_block must be the running sum of ceil(numel / 4096)");
        blocks += (L->numel[t] + OPT_CHUNK - 1) / OPT_CHUNK;
    }
    T2_REQUIRE(blocks <= 0x7fffffffLL, "optim: too many chunks");
    *nblocks = (int)blocks;
    return T2AMD_OK;
}

extern "C" int t2amd_optim_chunk(void) { return OPT_CHUNK; }

extern "C" int t2amd_grad_norm_f32(const t2amd_tensor_list* L, float max_norm, double* ws, float* norm_and_coef,
                                   void* stream) {
    int nblocks = 0;
    T2_PROPAGATE(opt_check_list(L, false, &nblocks));
    T2_REQUIRE(ws && norm_and_coef, "grad_norm: null workspace / output");
    hipStream_t s = (hipStream_t)stream;
    T2_LAUNCH(grad_sumsq_kernel, dim3(nblocks), dim3(256), 0, s, *L, ws);
    T2_LAUNCH_CHECK();
    T2_LAUNCH(grad_norm_finish_kernel, dim3(1), dim3(256), 0, s, (const double*)ws, nblocks, max_norm, norm_and_coef);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

extern "C" int t2amd_adam_step_f32(const t2amd_tensor_list* L, const t2amd_adam_hyper* h, const float* norm_and_coef,
                                   void* stream) {
    int nblocks = 0;
    T2_PROPAGATE(opt_check_list(L, true, &nblocks));
    T2_REQUIRE(h != nullptr, "adam_step: null hyper-parameters");
    T2_REQUIRE(h->bc2_sqrt > 0.0f && h->eps >= 0.0f && h->beta2 >= 0.0f && h->beta2 < 1.0f, "adam_step: bad hyper-parameters");
    T2_LAUNCH(adam_step_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, *L, *h, norm_and_coef);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}
