"""Forward-only sweep of the persistent forward loop's knobs in one process (all read per call): where the next tile's prefetch is
issued (T2AMD_DTP_PREFETCH=0/1/2) and the two pre-poll pauses (T2AMD_DTP_DELAY_L / _T).  python tools/ab_dtp_sweep.py"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd import engine, native
from tacotron2_amd.hparams import create_hparams
from tacotron2_amd.model import Tacotron2
from tacotron2_amd.synth import synth_batch
native.load()
dev = torch.device("cuda", 0)
hp = create_hparams()
torch.manual_seed(1234)
m = Tacotron2(hp).to(dev).train()
m.precision = os.environ.get("AB_PRECISION", "bf16")
batch = tuple(t.to(dev) for t in synth_batch(64, 1234))
x, _ = m.parse_batch(batch)
engine.TRAIN_FWD_PERSISTENT = True


def fwd(n=6):
    with torch.no_grad():
        m(x); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            m(x)
        torch.cuda.synchronize()
    return round(1e3 * (time.perf_counter() - t0) / n, 3)


out = {}
for rep in range(2):
    for pf in ("0", "1", "2"):
        os.environ["T2AMD_DTP_PREFETCH"] = pf
        out.setdefault("prefetch_%s" % pf, []).append(fwd())
print(json.dumps(out), flush=True)
if os.environ.get("AB_QUICK") == "1":
    sys.exit(0)
best = min(("1", "2"), key=lambda k: min(out["prefetch_" + k]))
os.environ["T2AMD_DTP_PREFETCH"] = best
sw = {}
for dl in (0, 2, 4, 8, 16):
    for dt in (0, 4, 8, 16):
        os.environ["T2AMD_DTP_DELAY_L"], os.environ["T2AMD_DTP_DELAY_T"] = str(dl), str(dt)
        sw["L%d_T%d" % (dl, dt)] = fwd(4)
out["pause_sweep_prefetch_%s" % best] = sw
print(json.dumps(sw), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/ab_dtp_sweep_%s.json" % m.precision, "w"), indent=1)
