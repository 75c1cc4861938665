"""Where a decode step of BASELINE configs[4] (256 utterances, bf16 mode) spends its time, launch by launch, WITHOUT in-kernel stamps:
every launch of the step is timed back to back on its own (event pair around replays of a hipGraph of 100 identical launches: the device time per launch,
dispatch ramp and drain included) over a sweep of its loop length -- K for the weight-streaming kernels, Ti for the attention kernels --
and a straight line through the points splits it into a FIXED cost per launch and a cost per loop step.

    python tools/microbench_config5.py          # writes gpurun_out/microbench_config5.json
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd import native as nv

nv.load()
dev = torch.device("cuda")
B, H, E, A = 256, 1024, 512, 128
g = torch.Generator().manual_seed(7)


def rnd(*shape, scale=0.05):
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def timeit(fn, n=100, replays=10):
    """us per call, the calls captured into ONE hipGraph (as the engine's decode loop replays them) so that the host's launch rate
    -- about 20 us per call through ctypes -- is not what is measured."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for _ in range(n):
                fn()
    graph.replay()
    torch.cuda.synchronize()
    best = None
    for _ in range(5):                                  # the minimum of five windows: one window in ~20 lands on a stall of tens of us
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(replays):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        t = 1e3 * e0.elapsed_time(e1) / (n * replays)
        best = t if best is None else min(best, t)
    return best


def fit(xs, ys):
    n = len(xs)
    mx, my = sum(xs) / n, sum(ys) / n
    slope = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
    return {"per_loop_step_us": round(slope, 3), "fixed_us": round(my - slope * mx, 2)}


out = {"what": "BASELINE configs[4] decode step, B = 256, bf16 mode: us per launch by loop length; fixed = intercept, per_loop_step = slope"}

# ---- the two LSTM launches (skinny_wide64_kernel: 64 x 64 tiles, one round of 256 workgroups) ----------------------------------
pts = {}
SEGS = {512: [512], 1024: [1024], 1536: [512, 1024], 1792: [256, 512, 1024], 2048: [1024, 1024], 2560: [1024, 512, 1024]}
for K, widths in SEGS.items():
    xs = [rnd(B, w, scale=1.0).bfloat16() for w in widths]
    W = rnd(4 * H, K).bfloat16()
    gin, bias, c_prev = rnd(B, 4 * H), rnd(4 * H), rnd(B, H)
    gates = torch.empty(B, 4 * H, device=dev)
    c_out, h_out = torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
    h16 = torch.empty(B, H, device=dev, dtype=torch.bfloat16)
    pts[K] = timeit(lambda: nv.lstm_step_fwd(xs, widths, W, H, B, gates, c_out, h_out, gin=gin, bias=bias, c_prev=c_prev, bf16=True, h16_out=h16))
out["lstm_64x64_tile"] = {"us_by_K": {k: round(v, 2) for k, v in pts.items()}, "k_tile": 128,
                          **fit([k / 128 for k in pts], list(pts.values())),
                          "in_the_step": "LSTM_a K = 1792 (14 k-tiles), LSTM_d K = 2560 (20)"}

# ---- the two plain launches (skinny_wide_kernel<false>: prenet layer 2, projection + folded prenet layer 1 + stop test) ------
pts = {}
for K, N in ((256, 256), (512, 337), (1024, 337), (1536, 337)):
    x = rnd(B, K, scale=1.0).bfloat16()
    W = rnd(N, K).bfloat16()
    Y = torch.empty(1, B, N, device=dev)
    bias = rnd(N)
    pts[(K, N)] = timeit(lambda: nv.skinny_gemm([x], [K], W, N, B, Y, bf16=True, bias=bias))
out["plain_64x32_tile"] = {"us_by_K_N": {"%d_%d" % k: round(v, 2) for k, v in pts.items()}, "k_tile": 128,
                           **fit([k[0] / 128 for k in pts], list(pts.values())),
                           "in_the_step": "prenet layer 2 K = 256 (2 k-tiles), projection K = 1536 (12)"}

# ---- the attention step (attn_energy4_kernel + attn_context_kernel: two launches per call) -----------------------------------
pts = {}
for Ti in (64, 128, 177):
    mem = rnd(B, Ti, E, scale=1.0)
    pm = rnd(B, Ti, A, scale=1.0)
    h = rnd(B, H, scale=1.0)
    Wq, U, v = rnd(A, H), rnd(A, 62), rnd(A)
    lens = torch.full((B,), Ti, dtype=torch.int32, device=dev)
    wprev, cum = torch.zeros(B, Ti, device=dev), torch.zeros(B, Ti, device=dev)
    w_out, ctx_out, q_out = torch.empty(B, Ti, device=dev), torch.empty(B, E, device=dev), torch.empty(B, A, device=dev)
    ws = torch.zeros(nv.attn_fwd_ws_floats(B, Ti), device=dev)
    m16, wq16 = mem.bfloat16(), Wq.bfloat16()
    pts[Ti] = timeit(lambda: nv.attention_step_fwd(h, Wq, U, v, pm, mem, lens, wprev, cum, None, w_out, ctx_out, q_out, ws,
                                                  bf16=True, memory16=m16, Wq16=wq16))
out["attention_step_two_launches"] = {"us_by_Ti": {k: round(v, 2) for k, v in pts.items()},
                                      **fit([k / 16 for k in pts], list(pts.values())), "loop_step": "one 16-position tile per utterance",
                                      "in_the_step": "Ti = 177 (K_e4 14.8 us + K_c 8.4 us by rocprofv3)"}
# an empty dependent launch on this stream, for scale
e = torch.zeros(1, device=dev)
out["trivial_launch_us"] = round(timeit(lambda: e.add_(1.0)), 2)
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/microbench_config5.json", "w"), indent=1)
