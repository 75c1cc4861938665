"""One attention-backward step (t2amd_attention_step_bwd_f32) repeated from the same inputs, every output compared bit for bit with
the first run -- optionally with a side stream of this process multiplying matrices (INPROC=<GEMMs per repeat>), with register-file
poison kernels beside it (REGPOISON=<launches per repeat>), or while `python tools/gpu_hammer.py <seconds> mm` runs in another
process.  This is how round 5 pinned the nondeterminism of tests/test_zz9_dp_gpu.py on ONE kernel and ONE instruction pattern
(DESIGN.md section 5.3).

    INPROC=30 python tools/stress_attn_bwd.py <fused 0|1> <B> <Ti> <repeats> <steps per repeat> [m16]
"""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd import native as nv
nv.load()
fused = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B, Ti = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3, 23)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 300
nsteps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
m16 = len(sys.argv) > 6 and sys.argv[6] == 'm16'
nv.set_attn_bwd_fused(fused)
dev = torch.device('cuda')
g = torch.Generator().manual_seed(5)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
E, Hq = 512, 1024
mem, pm = rnd(B, Ti, E), rnd(B, Ti, 128)
Wq, U, v = rnd(128, Hq) * 0.05, rnd(128 * 62) * 0.1, rnd(128)
lens = torch.tensor(([Ti, max(1, Ti - 6), max(1, Ti // 2)] * B)[:B], dtype=torch.int32, device=dev)
if os.environ.get('FULL_LENS'): lens.fill_(Ti)
w = torch.softmax(rnd(B, Ti), 1); wprev = torch.softmax(rnd(B, Ti), 1); cum = torch.rand(B, Ti, generator=g).to(dev)
q, dctx, dwx = rnd(B, 128), rnd(B, E), rnd(B, Ti)
ws0 = torch.zeros(nv.attn_bwd_ws_floats(B, Ti) , device=dev)
dwin0, dcum0 = rnd(4, B, 2, Ti), rnd(B, Ti)
names = ("tot", "dwin", "dcum", "d_pm", "dU", "dv", "dq", "dh", "ws")
side = torch.cuda.Stream()
ha = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16); hb = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
import ctypes as C
_lib = nv.load()
_pr = _lib.t2amd_debug_poison_regs_
_pr.argtypes = [C.c_int, C.c_uint, C.c_int, C.c_void_p]; _pr.restype = C.c_int
def once():
    if os.environ.get('REGPOISON'):
        assert _pr(2048, 0x7fc07fc0, int(os.environ['REGPOISON']), C.c_void_p(side.cuda_stream)) == 0
    if os.environ.get('INPROC'):
        with torch.cuda.stream(side):
            for _ in range(int(os.environ['INPROC'])):
                hc = ha @ hb
    ws = ws0.clone(); dwin = dwin0.clone(); dcum = dcum0.clone()
    d_pm, dU, dv_ = torch.zeros(B, Ti, 128, device=dev), torch.zeros(B, 128, 62, device=dev), torch.zeros(B, 128, device=dev)
    dq, dh, tot = torch.zeros(B, 128, device=dev), torch.zeros(4, B, Hq, device=dev), torch.zeros(B, E, device=dev)
    for step in range(nsteps):
        nv.attention_step_bwd([dctx], tot, dwx, q, Wq, U, v, pm, mem, lens, w, wprev, cum, dwin, dcum, d_pm, dU, dv_, dq, dh, ws, bf16=m16, memory16=mem.bfloat16() if m16 else None)
    torch.cuda.synchronize()
    out = [t.clone() for t in (tot, dwin, dcum, d_pm, dU, dv_, dq, dh, ws[:B * Ti + 12 * B])]
    torch.cuda.synchronize()
    return out
first = once()
bad = {}
for r in range(reps):
    o = once()
    for n, a, b in zip(names, first, o):
        if not torch.equal(a, b):
            d = (a != b)
            bad.setdefault(n, []).append((r, int(d.sum()), [int(x) for x in d.nonzero()[0].tolist()], float((a - b).abs().max())))
if 'ws' in bad:
    o = None
    for r in range(3):
        o = once()
        d = (first[8] != o[8]).nonzero().flatten().tolist()
        print('ws diff idx', d, [round(float(first[8][i]), 4) for i in d], [round(float(o[8][i]), 4) for i in d])
print('nan in first ws:', bool(torch.isnan(first[8]).any()), 'nonfinite anywhere in last:', any(bool((~torch.isfinite(t)).any()) for t in o) if 'o' in dir() and o is not None else None)
print(json.dumps(dict(nsteps=nsteps, m16=m16, fused=fused, B=B, Ti=Ti, reps=reps, unequal={k: (len(v), v[:1]) for k, v in bad.items()})))
