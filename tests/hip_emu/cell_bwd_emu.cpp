// TEST INFRASTRUCTURE ONLY — csrc/cell_bwd.h (the LSTM cell backward shared by lstm_pointwise_bwd_kernel and by the
// folded form of the attention backward) compiled FOR THE HOST and run as the stand-alone kernel runs it: one work-item per
// (row, 4 consecutive units), 256-item workgroups.  The GPU kernels wrap exactly these device functions; here the wrapper
// is restated (the real one lives in rnn.hip next to MFMA kernels that cannot be compiled for the host).
#include <hip/hip_runtime.h>
#include "../../tacotron2_amd/csrc/cell_bwd.h"

static void emu_cell_bwd_kernel(t2amd_lstm_bwd a) {
    const int H4 = a.H >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.B * H4) return;
    const int b = (int)(idx / H4);
    const int j = (int)(idx - (long long)b * H4) * 4;
    const CellOperands r = cell_bwd_issue(a, b, j);
    const Slab4 s0 = addend_issue4(a.dh[0], b, j), s1 = addend_issue4(a.dh[1], b, j), s2 = addend_issue4(a.dh[2], b, j);
    const float4 d0 = addend_finish4(s0, a.dh[0], b, j), d1 = addend_finish4(s1, a.dh[1], b, j), d2 = addend_finish4(s2, a.dh[2], b, j);
    cell_bwd_finish(a, r, d0, d1, d2, b, j);
}

extern "C" int t2amd_emu_cell_bwd(const t2amd_lstm_bwd* a) {
    if (!a || a->B < 1 || a->H < 4 || a->H % 4) return T2AMD_ERR_ARG;
    t2amd_lstm_bwd d = *a;
    for (int i = 0; i < 3; ++i)
        if (d.dh[i].p && d.dh[i].nsplit < 1) d.dh[i].nsplit = 1;
    const int blocks = (int)(((long long)d.B * (d.H / 4) + 255) / 256);
    hipLaunchKernelGGL(emu_cell_bwd_kernel, dim3(blocks), dim3(256), 0, nullptr, d);
    return T2AMD_OK;
}
