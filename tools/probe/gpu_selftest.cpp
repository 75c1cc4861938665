// Torch-free self-test of the kernels added at the end of round 1 (csrc/audio.hip, csrc/optim.hip, the STFT-as-GEMM
// addressing with overlapping rows, the bounded-spin grid barrier), through the C ABI only.  A fresh GPU box needs
// 1-2 minutes for its first `import torch`; this binary needs seconds, so it fits a GPU budget that a pytest run does
// not.  Build (here, no GPU needed):  python tools/probe/build_selftest.py     Run (GPU box): tools/probe/gpu_selftest
// Every check compares against a plain C++ loop in this file.  Exit code = number of failed checks.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/tacotron2_amd.h"

extern "C" int t2amd_debug_grid_barrier_(unsigned* counters, int rounds, int blocks, int lds_bytes,
                                         unsigned long long* clk, int* status, void* stream);

#define HIP_OK(x)                                                                  \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(100);                                                             \
        }                                                                          \
    } while (0)

static uint32_t g_rng = 12345u;
static float frand() {   // uniform in [-1, 1)
    g_rng = g_rng * 1664525u + 1013904223u;
    return (float)((g_rng >> 8) & 0xffffff) / 8388608.0f - 1.0f;
}
template <class T>
static T* dev_copy(const std::vector<T>& h) {
    T* d = nullptr;
    HIP_OK(hipMalloc(&d, h.size() * sizeof(T) + 64));
    HIP_OK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}
template <class T>
static std::vector<T> host_copy(const T* d, size_t n) {
    std::vector<T> h(n);
    HIP_OK(hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost));
    return h;
}
static int g_fail = 0;
static void report(const char* name, bool ok, double worst, double bound) {
    printf("%-46s %s   worst %.3e  (bound %.1e)\n", name, ok ? "PASS" : "FAIL", worst, bound);
    if (!ok) ++g_fail;
}
static int rc_ok(int rc, const char* what) {
    if (rc != 0) {
        printf("%s returned %d: %s\n", what, rc, t2amd_last_error());
        ++g_fail;
    }
    return rc;
}
static long long reflect(long long i, long long T) {
    if (i < 0) i = -i;
    if (i >= T) i = 2 * (T - 1) - i;
    return i;
}

int main() {
    int ndev = 0;
    HIP_OK(hipGetDeviceCount(&ndev));
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, 0));
    printf("device 0: %s, %d CUs, ABI %d\n", prop.gcnArchName, prop.multiProcessorCount, t2amd_abi_version());

    // ---- 1. reflect pad ---------------------------------------------------------------------------------------
    const int B = 2, T = 5000, L = 1024, hop = 256, pad = L / 2;
    const int n = T / hop + 1, ldo = (T + L + 3) / 4 * 4;
    std::vector<float> y((size_t)B * T);
    for (auto& v : y) v = 0.5f * frand();
    float* d_y = dev_copy(y);
    float* d_pad = nullptr;
    HIP_OK(hipMalloc(&d_pad, (size_t)B * ldo * sizeof(float)));
    HIP_OK(hipMemset(d_pad, 0xff, (size_t)B * ldo * sizeof(float)));
    rc_ok(t2amd_reflect_pad_f32(d_y, T, d_pad, ldo, B, T, pad, ldo, nullptr), "reflect_pad");
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> padded = host_copy(d_pad, (size_t)B * ldo);
    double worst = 0;
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < ldo; ++i) {
            const float want = i < T + 2 * pad ? y[(size_t)b * T + reflect(i - pad, T)] : 0.0f;
            worst = fmax(worst, fabs((double)padded[(size_t)b * ldo + i] - want));
        }
    report("reflect_pad (B=2, T=5000, pad=512)", worst == 0.0, worst, 0.0);

    // ---- 2. STFT as a GEMM over overlapping rows (lda = hop < K) ----------------------------------------------
    const int N = 70;                                    // stand-in for the 2F = 1026 basis rows (edge tile: N % 128 != 0)
    std::vector<float> basis((size_t)N * L);
    for (auto& v : basis) v = frand() * 0.05f;
    float* d_basis = dev_copy(basis);
    float* d_spec = nullptr;
    HIP_OK(hipMalloc(&d_spec, (size_t)B * n * N * sizeof(float)));
    t2amd_gemm_desc g;
    memset(&g, 0, sizeof(g));
    g.A = d_pad; g.B = d_basis; g.C = d_spec;
    g.M = n; g.N = N; g.K = L;
    g.lda = hop; g.ldb = L; g.ldc = N;
    g.a_kcontig = 1; g.b_kcontig = 1;
    g.batch = B; g.strideA = ldo; g.strideB = 0; g.strideC = (long long)n * N;
    g.splitk = 1; g.keep_scale = 1.0f; g.precision = 0;
    rc_ok(t2amd_gemm_f32(&g, nullptr), "gemm (overlapping rows)");
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> spec = host_copy(d_spec, (size_t)B * n * N);
    worst = 0;
    double scale = 0;
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < n; ++j)
            for (int f = 0; f < N; ++f) {
                double acc = 0;
                for (int k = 0; k < L; ++k) acc += (double)padded[(size_t)b * ldo + j * hop + k] * basis[(size_t)f * L + k];
                worst = fmax(worst, fabs(acc - spec[((size_t)b * n + j) * N + f]));
                scale = fmax(scale, fabs(acc));
            }
    report("gemm frames(lda=hop).basis^T, batch 2", worst <= 2e-5 * fmax(1.0, scale), worst, 2e-5 * fmax(1.0, scale));

    // ---- 3. magnitude ------------------------------------------------------------------------------------------
    const int F = N / 2, Fpad = 48;
    float* d_mag = nullptr;
    HIP_OK(hipMalloc(&d_mag, (size_t)B * n * Fpad * sizeof(float)));
    HIP_OK(hipMemset(d_mag, 0xff, (size_t)B * n * Fpad * sizeof(float)));
    rc_ok(t2amd_stft_magnitude_f32(d_spec, N, d_mag, Fpad, (long long)B * n, F, Fpad, nullptr), "stft_magnitude");
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> mag = host_copy(d_mag, (size_t)B * n * Fpad);
    worst = 0;
    for (int r = 0; r < B * n; ++r)
        for (int f = 0; f < Fpad; ++f) {
            float want = 0.0f;
            if (f < F) {
                const float re = spec[(size_t)r * N + f], im = spec[(size_t)r * N + F + f];
                const float a = re * re, c = im * im;           // separate roundings, as the kernel
                want = sqrtf(a + c);
            }
            worst = fmax(worst, fabs((double)mag[(size_t)r * Fpad + f] - want));
        }
    report("stft_magnitude (bitwise: mul, add, sqrt)", worst == 0.0, worst, 0.0);

    // ---- 4. log compression + transpose ------------------------------------------------------------------------
    const int n_mel = 5;
    std::vector<float> mel((size_t)B * n * n_mel);
    for (auto& v : mel) v = (frand() + 1.0f) * 5e-5f;        // straddles the 1e-5 clamp
    float* d_mel = dev_copy(mel);
    float* d_out = nullptr;
    HIP_OK(hipMalloc(&d_out, (size_t)B * n_mel * n * sizeof(float)));
    rc_ok(t2amd_mel_log_compress_f32(d_mel, n_mel, d_out, B, n, n_mel, 1e-5f, nullptr), "mel_log_compress");
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> lm = host_copy(d_out, (size_t)B * n_mel * n);
    worst = 0;
    for (int b = 0; b < B; ++b)
        for (int m = 0; m < n_mel; ++m)
            for (int j = 0; j < n; ++j) {
                const float want = logf(fmaxf(mel[((size_t)b * n + j) * n_mel + m], 1e-5f));
                worst = fmax(worst, fabs((double)lm[((size_t)b * n_mel + m) * n + j] - want));
            }
    report("mel_log_compress (+ transpose)", worst <= 6e-6, worst, 6e-6);   // device vs host logf: <= 2 ulp at |log| ~ 11 (1.9e-6 measured)

    // ---- 5. global-norm clip + Adam, two steps -----------------------------------------------------------------
    const int sizes[5] = {4095, 4096, 4097, 1, 91};
    const int NT = 5, chunk = t2amd_optim_chunk();
    long long total = 3;
    for (int s : sizes) total += s + 1;
    std::vector<float> hp(total), hg(total), hm(total, 0.0f), hv(total, 0.0f);
    for (auto& v : hp) v = 0.3f * frand();
    float* d_p = dev_copy(hp);
    float* d_m = dev_copy(hm);
    float* d_v = dev_copy(hv);
    float* d_g = nullptr;
    HIP_OK(hipMalloc(&d_g, total * sizeof(float)));
    double* d_ws = nullptr;
    float* d_nc = nullptr;
    HIP_OK(hipMalloc(&d_ws, 4096 * sizeof(double)));
    HIP_OK(hipMalloc(&d_nc, 2 * sizeof(float)));
    t2amd_tensor_list TL;
    memset(&TL, 0, sizeof(TL));
    {
        long long off = 1;                                     // offset 1: nothing 16-byte aligned
        int blocks = 0;
        for (int t = 0; t < NT; ++t) {
            TL.param[t] = d_p + off; TL.grad[t] = d_g + off; TL.exp_avg[t] = d_m + off; TL.exp_avg_sq[t] = d_v + off;
            TL.numel[t] = sizes[t];
            TL.first_block[t] = blocks;
            blocks += (sizes[t] + chunk - 1) / chunk;
            off += sizes[t] + 1;
        }
        TL.count = NT;
    }
    const float lr = 1e-3f, b1 = 0.9f, b2 = 0.999f, eps = 1e-8f, wd = 1e-6f, max_norm = 1.0f;
    double worst_p = 0, worst_n = 0;
    for (int step = 1; step <= 2; ++step) {
        const float gscale = step == 1 ? 5.0f : 1e-3f;         // clip active, then inactive
        for (auto& v : hg) v = gscale * frand();
        HIP_OK(hipMemcpy(d_g, hg.data(), total * sizeof(float), hipMemcpyHostToDevice));
        t2amd_adam_hyper h;
        h.step_size = (float)(lr / (1.0 - pow((double)b1, step)));
        h.bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, step));
        h.one_minus_beta1 = 1.0f - b1; h.beta2 = b2; h.one_minus_beta2 = 1.0f - b2; h.eps = eps; h.weight_decay = wd;
        rc_ok(t2amd_grad_norm_f32(&TL, max_norm, d_ws, d_nc, nullptr), "grad_norm");
        rc_ok(t2amd_adam_step_f32(&TL, &h, d_nc, nullptr), "adam_step");
        HIP_OK(hipDeviceSynchronize());
        // host restatement (same operation order as the kernel)
        double ss = 0;
        long long off = 1;
        for (int t = 0; t < NT; ++t) { for (int i = 0; i < sizes[t]; ++i) ss += (double)hg[off + i] * hg[off + i]; off += sizes[t] + 1; }
        const float norm = (float)sqrt(ss);
        float coef = max_norm / (norm + 1e-6f);
        coef = coef > 1.0f ? 1.0f : coef;
        std::vector<float> nc = host_copy(d_nc, 2);
        worst_n = fmax(worst_n, fabs(nc[0] - norm) / norm);
        worst_n = fmax(worst_n, fabs(nc[1] - coef) / coef);
        off = 1;
        for (int t = 0; t < NT; ++t) {
            for (int i = 0; i < sizes[t]; ++i) {
                const long long k = off + i;
                float gi = hg[k] * nc[1];                      // the device's own coefficient: isolates the update arithmetic
                gi = fmaf(wd, hp[k], gi);
                hm[k] = fmaf(gi - hm[k], 1.0f - b1, hm[k]);
                hv[k] = fmaf((1.0f - b2) * gi, gi, hv[k] * b2);
                const float denom = sqrtf(hv[k]) / h.bc2_sqrt + eps;
                hp[k] = fmaf(-h.step_size, hm[k] / denom, hp[k]);
            }
            off += sizes[t] + 1;
        }
        std::vector<float> gp = host_copy(d_p, (size_t)total);
        for (long long k = 0; k < total; ++k) worst_p = fmax(worst_p, fabs((double)gp[k] - hp[k]));
    }
    report("grad_norm / clip coefficient (relative)", worst_n <= 1e-6, worst_n, 1e-6);
    report("adam_step x2, params incl. untouched gaps", worst_p <= 1e-7, worst_p, 1e-7);

    // ---- 6. device-scope barrier cost (bounded spins) ----------------------------------------------------------
    {
        const int rounds = 1000;
        unsigned* d_cnt = nullptr;
        unsigned long long* d_clk = nullptr;
        int* d_st = nullptr;
        HIP_OK(hipMalloc(&d_cnt, rounds * sizeof(unsigned)));
        HIP_OK(hipMalloc(&d_clk, sizeof(unsigned long long)));
        HIP_OK(hipMalloc(&d_st, sizeof(int)));
        const int cfg[3][2] = {{64, 4}, {256, 4}, {256, 145000}};
        for (auto& c : cfg) {
            HIP_OK(hipMemset(d_cnt, 0, rounds * sizeof(unsigned)));
            HIP_OK(hipMemset(d_st, 0, sizeof(int)));
            const int rc = t2amd_debug_grid_barrier_(d_cnt, rounds, c[0], c[1], d_clk, d_st, nullptr);
            HIP_OK(hipDeviceSynchronize());
            unsigned long long clk = host_copy(d_clk, 1)[0];
            const int st = host_copy(d_st, 1)[0];
            printf("grid barrier, %4d workgroups, %6d B LDS: rc %d status %d  %.2f us per barrier\n", c[0], c[1], rc, st,
                   clk / 100.0 / rounds);
        }
    }
    printf("%s: %d failed checks\n", g_fail ? "SELFTEST FAILED" : "SELFTEST OK", g_fail);
    return g_fail;
}
