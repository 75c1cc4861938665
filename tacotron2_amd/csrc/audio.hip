// Mel front end on the GPU: wav -> log-mel, the arithmetic of the reference's
// TacotronSTFT.mel_spectrogram (layers.py:63-80) = STFT.transform (stft.py:77-105: reflect pad, conv1d against the
// windowed Fourier basis with stride hop, magnitude) -> mel_basis matmul -> log(clamp(., 1e-5)).
//
// The two contractions run on the dense GEMM of gemm.hip (exact-f32 MFMA):
//   spec[n][2F]  = frames[n][L] . basis[2F][L]^T     frames = the reflect-padded signal viewed with row stride `hop`
//                                                     (rows overlap: lda = hop < K = L, no frame matrix is built)
//   mel[n][n_mel] = mag[n][Fp] . mel_basis[n_mel][Fp]^T
// What is left for this file is HBM-bound element work, one pass each:
//   reflect pad (T -> T + L samples), magnitude (2F -> F, zero-filled to the padded row), log-compress + transpose.
#include "common.h"

// index into y[0..T) of padded sample i (i already shifted by -pad): torch 'reflect' (edge sample not repeated)
static __host__ __device__ __forceinline__ long long t2_reflect(long long i, long long T) {
    if (i < 0) i = -i;
    if (i >= T) i = 2 * (T - 1) - i;
    return i;
}

// host-visible copy of the index rule, for the CPU unit test of the padding arithmetic
extern "C" long long t2amd_reflect_index(long long i, long long T) { return t2_reflect(i, T); }

__global__ void __launch_bounds__(256) reflect_pad_kernel(const float* __restrict__ y, long long ldy,
                                                          float* __restrict__ out, long long ldo, int T, int pad,
                                                          int Tout) {
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= Tout) return;
    float v = 0.0f;
    if (i < (long long)T + 2 * pad) v = y[b * ldy + t2_reflect(i - pad, T)];
    out[b * ldo + i] = v;
}

// mag[r][f] = sqrt(re*re + im*im), products and sum rounded separately (torch: real**2 + imag**2, then sqrt)
__global__ void __launch_bounds__(256) magnitude_kernel(const float* __restrict__ spec, long long lds,
                                                        float* __restrict__ mag, long long ldm, long long rows,
                                                        int F, int Fpad) {
    const long long r = blockIdx.y;
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows || f >= Fpad) return;
    float v = 0.0f;
    if (f < F) {
        const float re = spec[r * lds + f], im = spec[r * lds + F + f];
        v = sqrtf(__fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im)));
    }
    mag[r * ldm + f] = v;
}

// out[b][m][j] = log(max(mel[(b*n + j)*ld + m], clip));  j is the fast index of the output (coalesced stores; the
// strided reads hit a row-major slab of n x n_mel floats that lives in L2)
__global__ void __launch_bounds__(256) mel_log_kernel(const float* __restrict__ mel, long long ld,
                                                      float* __restrict__ out, int n, int n_mel, float clip) {
    const int b = blockIdx.z, m = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float v = mel[((long long)b * n + j) * ld + m];
    out[((long long)b * n_mel + m) * n + j] = logf(fmaxf(v, clip));
}

extern "C" int t2amd_reflect_pad_f32(const float* y, long long ldy, float* out, long long ldo, int B, int T, int pad,
                                     int Tout, void* stream) {
    T2_REQUIRE(y && out, "reflect_pad: null operand");
    T2_REQUIRE(B > 0 && B <= 65535 && T > 0 && pad >= 0, "reflect_pad: bad dims");
    T2_REQUIRE(pad < T, "reflect_pad: padding must be smaller than the signal (torch reflect rule)");
    T2_REQUIRE(Tout >= T + 2 * pad && ldo >= Tout && ldy >= T, "reflect_pad: output row too short");
    dim3 grid(t2_cdiv(Tout, 256), B);
    T2_LAUNCH(reflect_pad_kernel, grid, dim3(256), 0, (hipStream_t)stream, y, ldy, out, ldo, T, pad, Tout);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

extern "C" int t2amd_stft_magnitude_f32(const float* spec, long long lds, float* mag, long long ldm, long long rows,
                                        int F, int Fpad, void* stream) {
    T2_REQUIRE(spec && mag, "stft_magnitude: null operand");
    T2_REQUIRE(rows > 0 && rows <= 65535 && F > 0 && Fpad >= F, "stft_magnitude: bad dims (rows <= 65535 per call)");
    T2_REQUIRE(lds >= 2 * (long long)F && ldm >= Fpad, "stft_magnitude: row too short");
    dim3 grid(t2_cdiv(Fpad, 256), (unsigned)rows);
    T2_LAUNCH(magnitude_kernel, grid, dim3(256), 0, (hipStream_t)stream, spec, lds, mag, ldm, rows, F, Fpad);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}

extern "C" int t2amd_mel_log_compress_f32(const float* mel, long long ld, float* out, int B, int n, int n_mel,
                                          float clip, void* stream) {
    T2_REQUIRE(mel && out, "mel_log_compress: null operand");
    T2_REQUIRE(B > 0 && B <= 65535 && n > 0 && n_mel > 0 && n_mel <= 65535 && ld >= n_mel, "mel_log_compress: bad dims");
    T2_REQUIRE(clip > 0.0f, "mel_log_compress: clip must be positive");
    dim3 grid(t2_cdiv(n, 256), n_mel, B);
    T2_LAUNCH(mel_log_kernel, grid, dim3(256), 0, (hipStream_t)stream, mel, ld, out, n, n_mel, clip);
    T2_LAUNCH_CHECK();
    return T2AMD_OK;
}
