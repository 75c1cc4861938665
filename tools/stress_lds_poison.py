"""Does any kernel of the training step depend on what it FINDS in LDS, on wave timing, or on what shares its SIMD?  One process, the tiny three-utterance
step of tests/test_zz9_dp_gpu.py run REPEATS times; from the second run on, a side stream keeps launching short kernels that fill
LDS with NaN patterns beside the step's kernels (t2amd_debug_poison_lds_) -- the stand-in for another process's kernels on the
same CUs.  Every run must give the bits of the first.

    python tools/stress_lds_poison.py [fp32|bf16] [repeats] [separate|fused|<comma list of afwd0,abwd0,fold0,encp0,fwdp0>]

POISON_LAUNCHES=<n> LDS-poison launches per step (default 2000, 0 = none); INPROC=<n> 2048^3 bf16 GEMMs per step on a side stream --
the disturbance that exposed the packed-FMA fault of DESIGN.md section 5.3; or run `python tools/gpu_hammer.py <s> mm` beside it.
"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 200
forms = sys.argv[3] if len(sys.argv) > 3 else "separate"

import golden_util as gu                                     # noqa: E402
from tacotron2_amd import engine, native as nv                # noqa: E402
from tacotron2_amd.engine import MaskSource                   # noqa: E402
from tacotron2_amd.loss_function import Tacotron2Loss         # noqa: E402
from tacotron2_amd.model import Tacotron2                     # noqa: E402

lib = nv.load()
dev = torch.device("cuda", 0)
toggles = ("afwd0", "abwd0", "fold0", "encp0", "fwdp0") if forms == "separate" else () if forms == "fused" else tuple(forms.split(","))
# "separate" = what distributed.apply_gradient_allreduce selects on a shared GPU; or a comma list of single toggles
if "afwd0" in toggles: nv.set_attn_fwd_fused(0)
if "abwd0" in toggles: nv.set_attn_bwd_fused(0)
if "fold0" in toggles: nv.set_bptt_cell_fold(0)
if "encp0" in toggles: engine.ENCODER_BATCH_PERSISTENT = False
if "fwdp0" in toggles:
    engine.TRAIN_FWD_PERSISTENT = False
hp = gu.make_hparams("")
shard = gu.make_train_batch([23, 17, 9], [41, 33, 20], hp.n_mel_channels, 500)
ms = MaskSource(None, dev)
ms.seed, (B, Ti, To) = 1000, (3, 23, 41)
masks = dict(enc=[ms.get('enc', i, (B, Ti, 512), 0.5) for i in range(3)],
             prenet=[ms.get('prenet', i, (To, B, 256), 0.5) for i in range(2)],
             att=ms.get('att', None, (To, B, 1024), 0.1), dec=ms.get('dec', None, (To, B, 1024), 0.1),
             post=[ms.get('post', i, (B, To, c), 0.5) for i, c in enumerate([512] * 4 + [80])])
torch.manual_seed(1234)
model = Tacotron2(hp).to(dev).train()
model.precision = precision
crit = Tacotron2Loss()
side = torch.cuda.Stream()
f = lib.t2amd_debug_poison_lds_
f.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_int, C.c_void_p]
f.restype = C.c_int


ha = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
hb = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)


def run(poison):
    if poison > 0 and os.environ.get("INPROC"):
        with torch.cuda.stream(side):
            for _ in range(int(os.environ["INPROC"])):
                ha @ hb
    model.zero_grad()
    model.dropout_masks = masks
    x, y = model.parse_batch(tuple(t.clone() for t in shard))
    if poison > 0:
        for lds in (160 * 1024, 64 * 1024, 96 * 1024, 32 * 1024):
            assert f(512, lds, 0x7fc07fc0, poison // 4, C.c_void_p(side.cuda_stream)) == 0
    loss = crit(model(x), y)
    loss.backward()
    torch.cuda.synchronize()
    return loss.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}


loss0, first = run(0)
notes = []
for rep in range(repeats):
    l2, g2 = run(int(os.environ.get("POISON_LAUNCHES", "2000")))
    diff = [k for k in first if not torch.equal(g2[k], first[k])]
    if diff or not torch.equal(l2, loss0):
        k0 = diff[0] if diff else None
        notes.append(dict(repeat=rep, loss_equal=bool(torch.equal(l2, loss0)), tensors=len(diff), of=len(first),
                          finite=bool(all(torch.isfinite(g2[k]).all() for k in g2)),
                          equal=[k for k in first if k not in diff] if len(diff) > 30 else None, differing=diff if len(diff) <= 30 else None,
                          max_rel=float((g2[k0].double() - first[k0].double()).abs().max() / (first[k0].double().abs().max() + 1e-30)) if k0 else 0.0))
out = dict(precision=precision, forms=forms, repeats=repeats, unequal=len(notes), notes=notes[:4], env={k: v for k, v in os.environ.items() if k.startswith("T2AMD_")})
print(json.dumps(dict(out, notes=[dict(n, equal=None if n["equal"] is None else len(n["equal"])) for n in notes[:2]]))[:1500])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "stress_%s_%s_%s.json" % (precision, forms, os.environ.get("STRESS_TAG", "x"))), "w"), indent=1)
