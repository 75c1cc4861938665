"""rocprofv3 target: three training steps with the backward loop on the chain, three with the persistent launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tacotron2_amd import engine, native
from tacotron2_amd.hparams import create_hparams
from tacotron2_amd.loss_function import Tacotron2Loss
from tacotron2_amd.model import Tacotron2
from tacotron2_amd.synth import synth_batch
native.load()
dev = torch.device("cuda", 0)
hp = create_hparams()
torch.manual_seed(1234)
m = Tacotron2(hp).to(dev).train()
m.precision = "bf16"
crit = Tacotron2Loss()
batch = tuple(t.to(dev) for t in synth_batch(64, 1234))
import time
for persistent in (False, True, False, True):
    engine.TRAIN_BWD_PERSISTENT = persistent
    ts = []
    for i in range(4):
        m.zero_grad(); x, y = m.parse_batch(batch); loss = crit(m(x), y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss.backward()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
    print("persistent" if persistent else "chain", "backward: host enqueue ms / total ms:", ["%.2f / %.2f" % t for t in ts[1:]], m.last_train_decoder_bwd_path, flush=True)
