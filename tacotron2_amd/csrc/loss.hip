// Tacotron2Loss (reference loss_function.py:8-19) as one reduction pass and one gradient pass:
//   loss = mean((mel - y)^2) + mean((mel_post - y)^2) + mean(bce_with_logits(gate, g))
// Forward reads the three mel-shaped tensors and the two gate vectors once and reduces the three sums in double
// precision with a fixed summation order (bit-reproducible); backward reads them once more and writes the three
// gradients scaled by the upstream gradient (a device scalar: no host read).  Replaces ~10 element-wise torch launches
// over 18 MB tensors ("fused loss").
#include "common.h"

#define LOSS_NT 256
#define LOSS_BLOCKS 1024

__global__ __launch_bounds__(LOSS_NT) void loss_partial_kernel(const float* __restrict__ mel, const float* __restrict__ post,
                                                               const float* __restrict__ tgt, long long n_mel,
                                                               const float* __restrict__ gate, const float* __restrict__ gtgt,
                                                               long long n_gate, double* __restrict__ ws) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    const long long stride = (long long)gridDim.x * LOSS_NT;
    const long long n4 = n_mel >> 2;
    for (long long i = (long long)blockIdx.x * LOSS_NT + threadIdx.x; i < n4; i += stride) {
        const float4 a = reinterpret_cast<const float4*>(mel)[i], b = reinterpret_cast<const float4*>(post)[i];
        const float4 y = reinterpret_cast<const float4*>(tgt)[i];
        float d;
        float p0 = 0.f, p1 = 0.f;
        d = a.x - y.x; p0 = fmaf(d, d, p0); d = a.y - y.y; p0 = fmaf(d, d, p0);
        d = a.z - y.z; p0 = fmaf(d, d, p0); d = a.w - y.w; p0 = fmaf(d, d, p0);
        d = b.x - y.x; p1 = fmaf(d, d, p1); d = b.y - y.y; p1 = fmaf(d, d, p1);
        d = b.z - y.z; p1 = fmaf(d, d, p1); d = b.w - y.w; p1 = fmaf(d, d, p1);
        s0 += (double)p0;
        s1 += (double)p1;
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * LOSS_NT + threadIdx.x; i < n_mel; i += stride) {
        const float d0 = mel[i] - tgt[i], d1 = post[i] - tgt[i];
        s0 += (double)(d0 * d0);
        s1 += (double)(d1 * d1);
    }
    for (long long i = (long long)blockIdx.x * LOSS_NT + threadIdx.x; i < n_gate; i += stride) {
        // max(x, 0) - x*y + log1p(exp(-|x|)): torch's numerically stable form of BCE with logits
        const float x = gate[i], y = gtgt[i];
        s2 += (double)(fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x))));
    }
    __shared__ double red[3][LOSS_NT / 64];
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_xor(s0, off, 64);
        s1 += __shfl_xor(s1, off, 64);
        s2 += __shfl_xor(s2, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s0;
        red[1][threadIdx.x >> 6] = s1;
        red[2][threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const double* r = red[threadIdx.x];
        ws[(long long)threadIdx.x * gridDim.x + blockIdx.x] = (r[0] + r[1]) + (r[2] + r[3]);
    }
}

// out[0] = total, out[1..3] = the three terms
__global__ __launch_bounds__(LOSS_NT) void loss_finish_kernel(const double* __restrict__ ws, int nblocks, double inv_mel,
                                                              double inv_gate, float* __restrict__ out) {
    __shared__ double red[3][LOSS_NT / 64];
    double s[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double a = 0.0;
        for (int i = threadIdx.x; i < nblocks; i += LOSS_NT) a += ws[(long long)k * nblocks + i];
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
        s[k] = a;
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) red[k][threadIdx.x >> 6] = s[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float l0 = (float)(((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * inv_mel);
        const float l1 = (float)(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * inv_mel);
        const float l2 = (float)(((red[2][0] + red[2][1]) + (red[2][2] + red[2][3])) * inv_gate);
        out[1] = l0; out[2] = l1; out[3] = l2;
        out[0] = (l0 + l1) + l2;                     // the reference adds in this order (loss_function.py:19)
    }
}

__global__ __launch_bounds__(LOSS_NT) void loss_backward_kernel(const float* __restrict__ mel, const float* __restrict__ post,
                                                                const float* __restrict__ tgt, long long n_mel,
                                                                const float* __restrict__ gate, const float* __restrict__ gtgt,
                                                                long long n_gate, const float* __restrict__ upstream,
                                                                float* __restrict__ d_mel, float* __restrict__ d_post,
                                                                float* __restrict__ d_gate) {
    const float go = upstream[0];
    const float sm = go * (2.0f / (float)n_mel), sg = go / (float)n_gate;
    const long long stride = (long long)gridDim.x * LOSS_NT;
    const long long n4 = n_mel >> 2;
    for (long long i = (long long)blockIdx.x * LOSS_NT + threadIdx.x; i < n4; i += stride) {
        const float4 a = reinterpret_cast<const float4*>(mel)[i], b = reinterpret_cast<const float4*>(post)[i];
        const float4 y = reinterpret_cast<const float4*>(tgt)[i];
        reinterpret_cast<float4*>(d_mel)[i] = make_float4(sm * (a.x - y.x), sm * (a.y - y.y), sm * (a.z - y.z), sm * (a.w - y.w));
        reinterpret_cast<float4*>(d_post)[i] = make_float4(sm * (b.x - y.x), sm * (b.y - y.y), sm * (b.z - y.z), sm * (b.w - y.w));
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * LOSS_NT + threadIdx.x; i < n_mel; i += stride) {
        d_mel[i] = sm * (mel[i] - tgt[i]);
        d_post[i] = sm * (post[i] - tgt[i]);
    }
    for (long long i = (long long)blockIdx.x * LOSS_NT + threadIdx.x; i < n_gate; i += stride) {
        const float x = gate[i];
        d_gate[i] = sg * (1.0f / (1.0f + expf(-x)) - gtgt[i]);
    }
}

extern "C" int t2amd_loss_workspace_doubles(void) { return 3 * LOSS_BLOCKS; }

extern "C" int t2amd_tacotron2_loss_fwd_f32(const float* mel, const float* post, const float* tgt, long long n_mel,
                                           const float* gate, const float* gate_tgt, long long n_gate, double* ws,
                                           float* out4, void* stream) {
    T2_REQUIRE(mel && post && tgt && gate && gate_tgt && ws && out4, "loss_fwd: null pointer");
    T2_REQUIRE(n_mel > 0 && n_gate > 0, "loss_fwd: empty tensors");
    T2_REQUIRE(t2_aligned16(mel) && t2_aligned16(post) && t2_aligned16(tgt), "loss_fwd: mel tensors must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (int)((n_mel / 4 + LOSS_NT - 1) / LOSS_NT < LOSS_BLOCKS ? ((n_mel / 4 + LOSS_NT - 1) / LOSS_NT > 0 ? (n_mel / 4 + LOSS_NT - 1) / LOSS_NT : 1) : LOSS_BLOCKS);
    T2_LAUNCH(loss_partial_kernel, dim3(blocks), dim3(LOSS_NT), 0, s, mel, post, tgt, n_mel, gate, gate_tgt, n_gate, ws);
    T2_LAUNCH(loss_finish_kernel, dim3(1), dim3(LOSS_NT), 0, s, (const double*)ws, blocks, 1.0 / (double)n_mel, 1.0 / (double)n_gate, out4);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

extern "C" int t2amd_tacotron2_loss_bwd_f32(const float* mel, const float* post, const float* tgt, long long n_mel,
                                           const float* gate, const float* gate_tgt, long long n_gate,
                                           const float* upstream, float* d_mel, float* d_post, float* d_gate, void* stream) {
    T2_REQUIRE(mel && post && tgt && gate && gate_tgt && upstream && d_mel && d_post && d_gate, "loss_bwd: null pointer");
    T2_REQUIRE(n_mel > 0 && n_gate > 0, "loss_bwd: empty tensors");
    T2_REQUIRE(t2_aligned16(mel) && t2_aligned16(post) && t2_aligned16(tgt) && t2_aligned16(d_mel) && t2_aligned16(d_post),
               "loss_bwd: mel tensors must be 16-byte aligned");
    const long long want = (n_mel / 4 + LOSS_NT - 1) / LOSS_NT;
    const int blocks = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
    T2_LAUNCH(loss_backward_kernel, dim3(blocks), dim3(LOSS_NT), 0, (hipStream_t)stream, mel, post, tgt, n_mel, gate, gate_tgt,
              n_gate, upstream, d_mel, d_post, d_gate);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}
