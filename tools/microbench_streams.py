"""Do kernels from two HIP streams overlap on this GPU?  Two independent attention-forward problems."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd import native as nv
dev = torch.device('cuda')
lib = nv.load()
def make(B, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    Ti, E, Hq = 177, 512, 1024
    h, mem, pm = rnd(B, Hq), rnd(B, Ti, E), rnd(B, Ti, 128)
    Wq, U, v = rnd(128, Hq) * 0.05, rnd(128 * 62) * 0.1, rnd(128)
    lens = torch.randint(60, Ti + 1, (B,), generator=g).sort(descending=True)[0].to(torch.int32).to(dev)
    wprev = torch.softmax(rnd(B, Ti), 1); cum = torch.rand(B, Ti, device=dev)
    outs = [torch.empty(B, Ti, device=dev), torch.empty(B, Ti, device=dev), torch.empty(B, E, device=dev), torch.empty(B, 128, device=dev)]
    ws = torch.empty(nv.attn_fwd_ws_floats(B, Ti), device=dev)
    cap = {}
    of = lib.t2amd_attention_step_fwd_f32
    class G:
        def __call__(self, ref, stream): cap['a'] = type(ref._obj).from_buffer_copy(ref._obj); return 0
    lib.t2amd_attention_step_fwd_f32 = G()
    nv.attention_step_fwd(h, Wq, U, v, pm, mem, lens, wprev, cum, outs[0], outs[1], outs[2], outs[3], ws)
    lib.t2amd_attention_step_fwd_f32 = of
    return cap['a'], (h, mem, pm, Wq, U, v, lens, wprev, cum, outs, ws)
of = lib.t2amd_attention_step_fwd_f32
def run(cfgs, n=300):
    # cfgs: list of (desc, stream)
    for _ in range(10):
        for d, s in cfgs: of(C.byref(d), C.c_void_p(s.cuda_stream))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    for s in set(s for _, s in cfgs): s.wait_stream(torch.cuda.current_stream())
    for _ in range(n):
        for d, s in cfgs: of(C.byref(d), C.c_void_p(s.cuda_stream))
    for s in set(s for _, s in cfgs): torch.cuda.current_stream().wait_stream(s)
    e1.record(torch.cuda.current_stream()); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
a64, k0 = make(64, 1)
a32, k1 = make(32, 2); b32, k2 = make(32, 3)
print("B=64 one stream:                 %.2f us per step" % run([(a64, s1)]))
print("B=32 one stream:                 %.2f us per step" % run([(a32, s1)]))
print("2 x B=32 on ONE stream:          %.2f us per step-pair" % run([(a32, s1), (b32, s1)]))
print("2 x B=32 on TWO streams:         %.2f us per step-pair" % run([(a32, s1), (b32, s2)]))
