// TEST INFRASTRUCTURE ONLY — what tools/probe/gpu_selftest.cpp needs besides the emulated kernels when it is dry-run
// on the CPU (python tests/hip_emu/dryrun_selftest.py): a plain-loop t2amd_gemm_f32 for the descriptor subset the
// self-test uses (K-contiguous operands, batch, row strides — including lda < K, the overlapping-frame view), the ABI
// version and a barrier stub.  The real GEMM is an MFMA kernel and cannot be emulated.
#include <hip/hip_runtime.h>
#include "../../include/tacotron2_amd.h"

extern "C" int t2amd_abi_version(void) { return T2AMD_ABI_VERSION; }
extern "C" int t2amd_gemm_f32(const t2amd_gemm_desc* d, void*) {
    if (!d || !d->a_kcontig || !d->b_kcontig || d->splitk > 1 || d->convA_T || d->convB_T) return T2AMD_ERR_ARG;
    for (int b = 0; b < (d->batch < 1 ? 1 : d->batch); ++b)
        for (int m = 0; m < d->M; ++m)
            for (int n = 0; n < d->N; ++n) {
                float acc = 0.0f;
                for (int k = 0; k < d->K; ++k)
                    acc = fmaf(d->A[b * d->strideA + (long long)m * d->lda + k], d->B[b * d->strideB + (long long)n * d->ldb + k], acc);
                d->C[b * d->strideC + (long long)m * d->ldc + n] = acc;
            }
    return T2AMD_OK;
}
extern "C" int t2amd_debug_grid_barrier_(unsigned*, int, int, int, unsigned long long* clk, int* status, void*) {
    *clk = 0; *status = -1;           // not emulated: workgroups run one at a time here
    return -9;
}
