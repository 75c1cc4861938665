"""Timing of the attention step launches (tools only).  T2AMD_ATTN_STAGE=n truncates the kernels after stage n."""
import os, sys, torch
os.environ.setdefault('T2AMD_ATTN_TS_PICK', '100')
if os.environ.get('T2AMD_ATTN_TS') == '1' or os.environ.get('T2AMD_ATTN_STAGE', '0') != '0':      # stamps / stage returns exist only in the instrumented build (python -m tacotron2_amd.build --stamps)
    os.environ.setdefault('T2AMD_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tacotron2_amd', 'lib', 'libtacotron2_amd_stamps.so'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd import native as nv
dev = torch.device('cuda')
B, Ti, E, Hq = 64, 177, 512, 1024
g = torch.Generator(device='cpu').manual_seed(0)
def rnd(*s): return torch.randn(*s, generator=g).to(dev)
h, mem, pm = rnd(B, Hq), rnd(B, Ti, E), rnd(B, Ti, 128)
Wq, U, v = rnd(128, Hq) * 0.05, rnd(128 * 62) * 0.1, rnd(128)
lens = torch.randint(60, Ti + 1, (B,), generator=g).sort(descending=True)[0].to(torch.int32).to(dev)
wprev = torch.softmax(rnd(B, Ti), 1); cum = torch.rand(B, Ti, device=dev)
cum_save, w_out, ctx, q = torch.empty(B, Ti, device=dev), torch.empty(B, Ti, device=dev), torch.empty(B, E, device=dev), torch.empty(B, 128, device=dev)
ws = torch.zeros(nv.attn_fwd_ws_floats(B, Ti) + nv.attn_bwd_ws_floats(B, Ti), device=dev)     # token and granule words must start at zero
dctx, dctx_total = rnd(B, E), torch.empty(B, E, device=dev)
dwin, dcum = torch.zeros(4, B, 2, Ti, device=dev), torch.zeros(B, Ti, device=dev)
d_pm, dU, dv_, dq, dh = torch.zeros(B, Ti, 128, device=dev), torch.zeros(B, 128, 62, device=dev), torch.zeros(B, 128, device=dev), torch.empty(B, 128, device=dev), torch.empty(4, B, Hq, device=dev)
import ctypes as C
lib = nv.load()
# build the descriptors once (the python wrappers rebuild them per call, which is slower than the kernels)
_cap = {}
_of, _ob = lib.t2amd_attention_step_fwd_f32, lib.t2amd_attention_step_bwd_f32
class _Grab:
    def __init__(self, key): self.key = key
    def __call__(self, ref, stream):
        _cap[self.key] = (type(ref._obj).from_buffer_copy(ref._obj), stream); return 0
lib.t2amd_attention_step_fwd_f32 = _Grab('f'); lib.t2amd_attention_step_bwd_f32 = _Grab('b')
nv.attention_step_fwd(h, Wq, U, v, pm, mem, lens, wprev, cum, cum_save, w_out, ctx, q, ws)
nv.attention_step_bwd([dctx], dctx_total, None, q, Wq, U, v, pm, mem, lens, w_out, wprev, cum_save, dwin, dcum, d_pm, dU, dv_, dq, dh, ws)
lib.t2amd_attention_step_fwd_f32, lib.t2amd_attention_step_bwd_f32 = _of, _ob
fa, fs = _cap['f']; ba, bs = _cap['b']
if os.environ.get('T2AMD_MB_BF16') == '1':      # the engine's bf16 compute mode: split-bf16 location conv, bf16 gradient products
    fa.loc_split_bf16 = 1; ba.bf16 = 1
def fwd(): _of(C.byref(fa), fs)
def bwd(): _ob(C.byref(ba), bs)
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("stage", os.environ.get("T2AMD_ATTN_STAGE", "0"), "fwd (K_e+K_c) %.2f us   bwd (K_b1+K_b2) %.2f us" % (timeit(fwd), timeit(bwd)))

# the same launches replayed from a hipGraph (what a captured decoder loop would pay per step).  Timing only: the
# in-launch hand-offs carry their launch token as a kernel argument, so a replay presents the tokens of the capture again
# and its polls pass on the previous replay's granules (same inputs, same values here) -- never capture them in a product.
if os.environ.get("T2AMD_ATTN_STAGE", "0") == "0":
    side = torch.cuda.Stream()
    sp = C.c_void_p(side.cuda_stream)
    lib.t2amd_debug_capture_end_.restype = C.c_float
    lib.t2amd_debug_capture_end_.argtypes = [C.c_void_p, C.c_int]
    lib.t2amd_debug_capture_begin_.argtypes = [C.c_void_p]
    torch.cuda.synchronize()
    for name, desc, fn in (("fwd", fa, _of), ("bwd", ba, _ob)):
        n = 100
        rc = lib.t2amd_debug_capture_begin_(sp)
        assert rc == 0, rc
        for _ in range(n):
            fn(C.byref(desc), sp)
        ms = lib.t2amd_debug_capture_end_(sp, 20)
        print("graph replay %s: %.2f us per step" % (name, ms * 1e3 / n))

# in-kernel phase timestamps (T2AMD_ATTN_TS=1): 100 MHz wall clock at the phase boundaries of workgroup (0,0)
if os.environ.get("T2AMD_ATTN_TS") == "1":
    fwd(); bwd(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    lib.t2amd_debug_attn_ts_.argtypes = [C.c_void_p]
    rc = lib.t2amd_debug_attn_ts_(buf)
    assert rc == 0, rc
    names = {0: "K_e ", 16: "K_c ", 32: "K_b1", 48: "K_b2"}
    for base, nm in names.items():
        ts = [buf[base + i] for i in range(15) if buf[base + i]]
        print(nm, "phase boundaries (us from kernel entry):", " ".join("%.2f" % ((t - ts[0]) / 100.0) for t in ts))
