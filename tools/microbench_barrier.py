"""Device-scope barrier cost across co-resident workgroups (python tools/microbench_barrier.py, on the GPU box).

The figure that sizes the weight-stationary persistent decode step ("next"): 6-8 barriers per
decode step.  Spins are bounded inside the kernel (status != 0 = a workgroup gave up: the grid was not co-resident).
Run under `timeout 120`."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd import native as nv

lib = nv.load()
f = lib.t2amd_debug_grid_barrier_
f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
f.restype = C.c_int
ROUNDS = 2000
for blocks, lds in ((8, 4), (64, 4), (256, 4), (256, 145000), (512, 70000), (1024, 4)):
    best = None
    for rep in range(3):
        counters = torch.zeros(ROUNDS, dtype=torch.int32, device="cuda")
        clk = torch.zeros(1, dtype=torch.int64, device="cuda")
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        rc = f(counters.data_ptr(), ROUNDS, blocks, lds, clk.data_ptr(), status.data_ptr(), nv._stream())
        torch.cuda.synchronize()
        if rc != 0 or int(status.item()) != 0:
            best = "rc=%d status=%d (not co-resident or launch error)" % (rc, int(status.item()))
            break
        us = int(clk.item()) / 100.0 / ROUNDS            # wall_clock64 ticks at 100 MHz
        best = us if best is None else min(best, us)
    print("blocks %4d  lds %6d B : %s" % (blocks, lds, ("%.2f us per barrier" % best) if isinstance(best, float) else best))
