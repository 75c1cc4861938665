import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd import native as nv
lib = nv.load()
x = torch.zeros(4, device='cuda')
f = lib.t2amd_debug_launch_chain_
f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
for blocks in (1, 256, 512, 1024):
    f(x.data_ptr(), 100, blocks, nv._stream()); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(x.data_ptr(), 2000, blocks, nv._stream()); e1.record(); torch.cuda.synchronize()
    print("blocks %4d: %.2f us per dependent trivial launch" % (blocks, e0.elapsed_time(e1) / 2000 * 1e3))
