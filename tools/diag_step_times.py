#!/usr/bin/env python
"""Where do the milliseconds between bench.py's fresh-batch loop and its replay legs go?  

Runs the driver's loop shape (W warm-up + K timed steps on K + W DIFFERENT synthetic batches, never synchronising inside
the loop), with one event per step on the launch stream and a snapshot of the caching allocator's counters per step, then
replays the same timed batches (every shape already seen) the way bench.py's optimizer_ab leg does.  Writes
gpurun_out/<tag>.json:

  pass1 / pass2   per-step GPU time between consecutive step-end events (ms), wall time of the loop, and per step the
                  deltas of num_device_alloc / num_device_free / num_alloc_retries / reserved bytes
  arena           the engine arena's own counters (chunks, bytes, growths) when the arena is on

    python tools/diag_step_times.py [--steps 20] [--warmup 5] [--tag diag_steps]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stats(dev):
    s = torch.cuda.memory_stats(dev)
    return {"num_device_alloc": s.get("num_device_alloc", 0), "num_device_free": s.get("num_device_free", 0),
            "num_alloc_retries": s.get("num_alloc_retries", 0), "reserved": s.get("reserved_bytes.all.current", 0),
            "allocated": s.get("allocated_bytes.all.current", 0),
            "alloc_calls": s.get("allocation.all.allocated", 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--tag", default="diag_steps")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--swap", default="", help="i,j: exchange batches i and j of the list (moves a shape change elsewhere)")
    ap.add_argument("--gc-off", action="store_true", help="gc.disable() during the passes")
    a = ap.parse_args()
    from tacotron2_amd import native, engine
    from tacotron2_amd.hparams import create_hparams
    from tacotron2_amd.model import Tacotron2
    from tacotron2_amd.loss_function import Tacotron2Loss
    from tacotron2_amd.optim import FusedAdam
    from tacotron2_amd.synth import synth_batch
    native.load()
    dev = torch.device("cuda", 0)
    hp = create_hparams()
    torch.manual_seed(hp.seed)
    model = Tacotron2(hp).to(dev)
    model.precision = a.precision
    opt = FusedAdam(model.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)
    crit = Tacotron2Loss()
    model.train()
    n = a.warmup + a.steps
    batches, shapes = [], []
    for i in range(n):
        b = synth_batch(64, 1234 + i)
        batches.append(tuple(t.to(dev) for t in b))
        shapes.append((int(b[1].max()), int(b[4].max()), int(b[4].sum())))
    torch.cuda.synchronize()
    if a.swap:
        i_, j_ = (int(v) for v in a.swap.split(","))
        batches[i_], batches[j_] = batches[j_], batches[i_]
        shapes[i_], shapes[j_] = shapes[j_], shapes[i_]
    import gc
    cur_step = [-1]
    gc_log, gc_t0 = [], [0.0]

    def gc_cb(phase, info):
        if phase == "start":
            gc_t0[0] = time.perf_counter()
        else:
            gc_log.append({"generation": info["generation"], "ms": round(1e3 * (time.perf_counter() - gc_t0[0]), 3),
                           "collected": info["collected"], "at_step": cur_step[0]})
    gc.callbacks.append(gc_cb)
    if a.gc_off:
        gc.disable()
    phases = {}

    def step(batch):
        t = [time.perf_counter()]
        model.zero_grad()
        x, y = model.parse_batch(batch)
        t.append(time.perf_counter())
        out_ = model(x)
        t.append(time.perf_counter())
        loss = crit(out_, y)
        t.append(time.perf_counter())
        loss.backward()
        t.append(time.perf_counter())
        opt.step(clip_norm=hp.grad_clip_thresh)
        t.append(time.perf_counter())
        phases[cur_step[0]] = [round(1e3 * (t[i + 1] - t[i]), 2) for i in range(5)]
        return loss

    def run_pass(idx, label):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(idx) + 1)]
        snaps = [stats(dev)]
        host = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evs[0].record()
        for j, i in enumerate(idx):
            h0 = time.perf_counter()
            cur_step[0] = i
            step(batches[i])
            evs[j + 1].record()
            host.append(1e3 * (time.perf_counter() - h0))
            snaps.append(stats(dev))
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        gpu = [evs[j].elapsed_time(evs[j + 1]) for j in range(len(idx))]
        per = []
        for j, i in enumerate(idx):
            d = {k: snaps[j + 1][k] - snaps[j][k] for k in snaps[0]}
            per.append({"batch": i, "Ti": shapes[i][0], "To": shapes[i][1], "gpu_ms": round(gpu[j], 3),
                        "host_enqueue_ms": round(host[j], 3), "d_device_alloc": d["num_device_alloc"],
                        "d_device_free": d["num_device_free"], "d_retries": d["num_alloc_retries"],
                        "d_reserved_MB": round(d["reserved"] / 2 ** 20, 1), "alloc_calls": d["alloc_calls"],
                        "host_phases_ms[parse,fwd,loss,bwd,opt]": phases.get(i)})
        frames = sum(shapes[i][2] for i in idx)
        return {"label": label, "wall_ms_per_step": 1e3 * wall / len(idx), "frames_per_s": frames / wall,
                "sum_gpu_ms_per_step": sum(gpu) / len(idx), "device_allocs": sum(p["d_device_alloc"] for p in per),
                "reserved_GB_end": snaps[-1]["reserved"] / 2 ** 30, "steps": per}

    out = {"cmd": " ".join(sys.argv), "precision": a.precision}
    for i in range(a.warmup):
        step(batches[i])
    out["after_warmup"] = stats(dev)
    timed = list(range(a.warmup, n))
    out["pass1_fresh_batches"] = run_pass(timed, "K fresh batches after W warm-ups (the driver's loop)")
    out["pass2_same_batches_again"] = run_pass(timed, "the same K batches again (every shape seen)")
    half = timed[:max(2, a.steps // 2)]
    out["pass3_first_half_again"] = run_pass(half, "K/2 of them again (bench.py's optimizer_ab shape)")
    out["gc_collections"] = [g for g in gc_log if g["generation"] == 2 or g["ms"] > 2.0]
    out["gc_generation_counts"] = {str(k): sum(1 for g in gc_log if g["generation"] == k) for k in (0, 1, 2)}
    print("gc:", out["gc_generation_counts"], out["gc_collections"][:12])
    ar = getattr(engine, "arena_stats", None)
    if ar is not None:
        out["arena"] = ar(model)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", a.tag + ".json"), "w") as fh:
        json.dump(out, fh, indent=1)
    for k in ("pass1_fresh_batches", "pass2_same_batches_again", "pass3_first_half_again"):
        p = out[k]
        print("%-28s wall %.2f ms/step  gpu %.2f ms/step  %.1f k frames/s  device allocs %d  reserved %.1f GB" % (
            k, p["wall_ms_per_step"], p["sum_gpu_ms_per_step"], p["frames_per_s"] / 1e3, p["device_allocs"], p["reserved_GB_end"]))
    for p in out["pass1_fresh_batches"]["steps"]:
        print(p["batch"], p["Ti"], p["To"], p["gpu_ms"], p["host_enqueue_ms"], p["d_device_alloc"], p["d_reserved_MB"],
              p["host_phases_ms[parse,fwd,loss,bwd,opt]"])


if __name__ == "__main__":
    main()
