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

# the same chain on a non-default stream (torch's current stream is the legacy default stream unless the caller
# switches): does the null stream cost extra per launch?
side0 = torch.cuda.Stream()
sp0 = C.c_void_p(side0.cuda_stream)
for blocks in (1, 256):
    f(x.data_ptr(), 100, blocks, sp0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side0):
        e0.record(); f(x.data_ptr(), 2000, blocks, sp0); e1.record()
    torch.cuda.synchronize()
    print("side stream, blocks %4d: %.2f us per dependent trivial launch" % (blocks, e0.elapsed_time(e1) / 2000 * 1e3))

g = lib.t2amd_debug_graph_chain_
g.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
g.restype = C.c_float
side = torch.cuda.Stream()          # capture is not allowed on the legacy default stream
torch.cuda.synchronize()
for blocks in (1, 256, 512):
    for n in (7, 70, 700):
        ms = g(x.data_ptr(), n, blocks, 200 if n < 100 else 20, C.c_void_p(side.cuda_stream))
        print("graph of %3d nodes, blocks %4d: %.2f us per replay = %.2f us per node" % (n, blocks, ms * 1e3, ms * 1e3 / n))
# captured on the side stream, replayed on the legacy default stream (what a library called on torch's current stream does)
ms = g(x.data_ptr(), 70, -256, 200, C.c_void_p(side.cuda_stream))
print("graph of  70 nodes replayed on the default stream: %.2f us per node" % (ms * 1e3 / 70) if ms > 0 else "default-stream replay failed: HIP error %d" % int(-ms))

