// Host stand-in for <hip/hip_ext.h> (tests/hip_emu): csrc/common.h includes it for hipExtLaunchKernelGGL, which only the
// launch sites of the MFMA kernels use -- nothing the emulated kernels (audio, optim, cell_bwd) reach.
#pragma once
