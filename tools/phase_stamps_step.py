"""In-situ phase stamps: run two full training steps (B=64 synthetic batch) with T2AMD_ATTN_TS=1 and print, for the
last launch of every instrumented kernel, the wall-clock offsets (us) of its phase boundaries as seen by thread 0 of
workgroup 0.     T2AMD_ATTN_TS=1 python tools/phase_stamps_step.py [--precision bf16|fp32]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("T2AMD_ATTN_TS", "1")
# the stamps are compiled in only in the instrumented build: python -m tacotron2_amd.build --stamps
os.environ.setdefault("T2AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tacotron2_amd", "lib", "libtacotron2_amd_stamps.so"))
import torch
from tacotron2_amd import native as nv
from tacotron2_amd.hparams import create_hparams
from tacotron2_amd.model import Tacotron2
from tacotron2_amd.loss_function import Tacotron2Loss
from tacotron2_amd.synth import synth_batch

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16")
args = ap.parse_args()
dev = torch.device("cuda:0")
hp = create_hparams()
torch.manual_seed(hp.seed)
model = Tacotron2(hp).to(dev)
model.precision = args.precision
crit = Tacotron2Loss()
model.train()
for i in range(2):
    batch = tuple(t.to(dev) for t in synth_batch(hp.batch_size, 1234 + i))
    model.zero_grad()
    x, y = model.parse_batch(batch)
    loss = crit(model(x), y)
    loss.backward()
torch.cuda.synchronize()
lib = nv.load()
buf = (C.c_ulonglong * 128)()
lib.t2amd_debug_attn_ts_.argtypes = [C.c_void_p]
assert lib.t2amd_debug_attn_ts_(buf) == 0, "stamps are off: set T2AMD_ATTN_TS=1"
names = [(0, "K_e   (entry, prologue done, tiles done)"),
         (16, "K_c   (entry, max, sum, weights, context partials, end)"),
         (32, "K_b1  (entry, operands staged, dw done, end)"),
         (48, "K_b2  (entry, prologue, tiles, reduce, dU, col2im, dh | folded cells: dq collected, end)"),
         (64, "LSTM pair  (entry, first DMA, tile 0 landed, k loop, partial sums, end)"),
         (80, "dgrad pair (entry, first DMA, tile 0 landed, k loop, partial sums, end)"),
         (96, "cell bwd   (entry, operands landed, end)")]
for base, nm in names:
    ts = [buf[base + i] for i in range(15) if buf[base + i]]
    if ts:
        print("%-78s %s" % (nm, " ".join("%.2f" % ((t - ts[0]) / 100.0) for t in ts)))
if buf[48] and buf[32]:
    print("K_b1 phase inside K_b2: main-body prologue issued %.2f | entry %.2f | operands staged %.2f | dw done %.2f | end %.2f   (us from K_b2's entry)" %
          tuple((buf[i] - buf[48]) / 100.0 for i in (62, 32, 33, 34, 35)))
# K_b2's first hand-off in detail (round 5: slots 59-61)
if buf[48] and buf[59]:
    print("K_b2 hand-off: independent work done %.2f | wave 0's granules carry the token %.2f | barrier behind them %.2f | staging done %.2f" %
          tuple((buf[i] - buf[48]) / 100.0 for i in (59, 60, 61, 49)))
# K_b2's reduce phase in detail (slots 56-58: W_q / cell prefetch issued, lane-group sums written, barrier passed)
if buf[48] and buf[56]:
    print("K_b2 detail: tiles done %.2f | prefetch issued %.2f | sums in LDS %.2f | barrier passed %.2f | dq stored %.2f" %
          tuple((buf[i] - buf[48]) / 100.0 for i in (50, 56, 57, 58, 51)))
