"""Multi-GPU behaviour that can be PROVEN on the one GPU there is.

(1) ``test_rccl_world1_*``: a world-size-1 ``nccl`` (= RCCL) process group on cuda:0.  ``gloo`` -- what the two-rank
    tests use -- completes a collective on the host; ProcessGroupNCCL instead returns an asynchronous ``Work`` whose
    ``wait()`` only orders the CURRENT STREAM behind RCCL's own stream.  The engine enqueues its kernels on raw stream
    handles, so this is the test that the three bucket all-reduces (postnet -> decoder -> encoder, launched from inside
    the backward) are ordered correctly against the kernels that write the buckets before them and the optimiser that
    reads ``p.grad`` after them: with one rank the mean is the identity, so every ``p.grad`` and one FusedAdam step must
    be BIT-identical to the same step without any exchange.  Reference contract: distributed.py:126-173.

(2) ``test_training_step_with_cus_held_by_another_kernel``: the fused attention kernels hand data between the four
    workgroups of an utterance inside one launch and rely on forward progress under in-order dispatch.  Under real data
    parallelism RCCL kernels run beside the BPTT chain and take CUs away.  A side-stream kernel holds 16 / 32 / 64 CUs
    (whole-LDS workgroups) for the whole BACKWARD of a step -- when RCCL runs: the buckets are launched as BPTT finishes
    them and the optimiser / next forward are ordered behind them -- : the gradients must stay bit-identical, finite (no NaN
    poison from an abandoned bounded spin), the backward must not slow down by the 50 ms a timed-out spin would cost per
    launch, and by no more than 1.5 x with 32 CUs held (measured 1.38 x) (round 4: the bound VERDICT r03 item 7 asked for).
    (The decoder's BPTT loop is the launch chain: its opt-in persistent launch of rounds 4-5, which needed every CU and made the
    engine start no collective beside it, was removed in round 6.)

(3) ``test_whole_step_under_held_cus_falls_back_to_the_launch_chain`` (round 4): the persistent decoder loop of the forward
    needs every CU; with CUs held for the whole step (a shared GPU) its arrival census gives up within 2 ms, the step is
    poisoned and counted, the loop goes back to the launch chain and the same step under the same hold is bit-identical.
"""
import ctypes as C
import json
import os
import socket
import time

import pytest
import torch
import torch.multiprocessing as mp

from tacotron2_amd import native

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rccl_worker(port, precision, q):
    import faulthandler
    os.makedirs(OUT, exist_ok=True)
    log = open(os.path.join(OUT, "rccl_world1_%s.log" % precision), "w")
    faulthandler.enable(log)
    faulthandler.dump_traceback_later(240, exit=True, file=log)
    import torch.distributed as dist
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1)
        import sys
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from tacotron2_amd import native
        from tacotron2_amd.distributed import apply_gradient_allreduce, reduce_tensor
        from tacotron2_amd.hparams import create_hparams
        from tacotron2_amd.loss_function import Tacotron2Loss
        from tacotron2_amd.model import Tacotron2
        from tacotron2_amd.optim import FusedAdam
        from tacotron2_amd.synth import synth_batch
        native.load()
        hp = create_hparams()
        full = synth_batch(64, 1234)
        # the bench batch with the horizon cut to 160 frames: B = 64 (every 4 x B grid full), 160 BPTT steps
        ol = full[4].clamp(max=160)
        To = int(ol.max())
        batch = tuple(t.to(dev) for t in (full[0], full[1], full[2][:, :, :To].contiguous(), full[3][:, :To].contiguous(), ol))
        gate = batch[3].clone()
        for b in range(64):
            gate[b, int(ol[b]) - 1:] = 1.0
        batch = batch[:3] + (gate,) + batch[4:]
        crit = Tacotron2Loss()

        def fresh():
            torch.manual_seed(1234)
            m = Tacotron2(hp).to(dev).train()
            m.precision = precision
            return m

        def two_steps(model, opt):
            res = []
            for it in range(2):
                torch.manual_seed(99 + it)                        # same Philox dropout streams on both sides
                model.zero_grad()
                x, y = model.parse_batch(batch)
                loss = crit(model(x), y)
                loss.backward()
                grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
                opt.step(clip_norm=hp.grad_clip_thresh)          # reads p.grad straight after the collectives
                res.append((float(loss), grads, {k: p.detach().clone() for k, p in model.named_parameters()}))
            torch.cuda.synchronize()
            return res

        ref_model = fresh()
        ref = two_steps(ref_model, FusedAdam(ref_model.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay))
        dp_model = apply_gradient_allreduce(fresh())
        assert dp_model._grad_sync is not None and not dp_model._grad_sync.serial
        t0 = time.perf_counter()
        got = two_steps(dp_model, FusedAdam(dp_model.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay))
        dt = time.perf_counter() - t0
        sync = dp_model._grad_sync
        bad = []
        for it in range(2):
            if got[it][0] != ref[it][0]:
                bad.append(("loss", it, got[it][0], ref[it][0]))
            for k in ref[it][1]:
                if not torch.equal(got[it][1][k], ref[it][1][k]):
                    bad.append(("grad", it, k, float((got[it][1][k] - ref[it][1][k]).abs().max())))
                if not torch.equal(got[it][2][k], ref[it][2][k]):
                    bad.append(("param", it, k, float((got[it][2][k] - ref[it][2][k]).abs().max())))
        views = all(any(f.data_ptr() <= p.grad.data_ptr() < f.data_ptr() + 4 * f.numel() for f in sync.flat.values())
                    for p in dp_model.parameters())
        mean_loss = float(reduce_tensor(torch.tensor(got[1][0], device=dev), 1))
        bwd_path = getattr(dp_model, "last_train_decoder_bwd_path", None)
        path_ok = bwd_path == "launch chain"
        q.put(dict(ok=not bad and views and sync.fresh_allocations == 1 and mean_loss == got[1][0] and path_ok, bad=bad[:10],
                   decoder_bptt=bwd_path,
                   p_grad_is_a_bucket_view=views, bucket_sets_allocated=sync.fresh_allocations,
                   reduce_op=str(sync.op), divide_pass=sync.divide, seconds_two_steps=dt,
                   bucket_bytes={b: 4 * n for b, n in sync.sizes.items()}, backend=dist.get_backend(), To=To))
    except Exception:                                             # pragma: no cover
        import traceback
        q.put(dict(ok=False, error=traceback.format_exc()))
    finally:
        faulthandler.cancel_dump_traceback_later()
        log.close()
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("precision", ["bf16", "fp32", "bf16x3"])
def test_rccl_world1_bucket_allreduce_is_ordered_and_exact(native_lib, precision):
    import queue
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), precision, q))
    p.start()
    try:
        res = q.get(timeout=300)
    except queue.Empty:
        res = dict(ok=False, error="no report within 300 s (exit code %s): gpurun_out/rccl_world1_%s.log" % (p.exitcode, precision))
    p.join(timeout=30)
    if p.is_alive():
        p.kill()
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "parity_rccl_world1_%s.json" % precision), "w") as f:
        json.dump(res, f, indent=1)
    assert res.get("ok"), res


# -------------------------------------------------------------------------------------------------------------------
def _hold(lib, ncus, ms, stop, arrived, stream):
    f = lib.t2amd_debug_hold_cus_
    f.argtypes = [C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    f.restype = C.c_int
    rc = f(ncus, ms, stop.data_ptr(), arrived.data_ptr(), C.c_void_p(stream.cuda_stream))
    assert rc == 0, lib.t2amd_last_error()


def test_training_step_with_cus_held_by_another_kernel(native_lib):
    """RCCL's kernels run beside the BACKWARD of a step (the buckets are launched as BPTT finishes them; `GradSync.finish()`
    orders the optimiser -- and with it the next forward -- behind the collectives): a side-stream kernel holds 16 / 32 / 64
    whole CUs (all of their LDS) from the end of the forward to the end of the backward.  The in-launch hand-offs of the
    attention backward and the persistent encoder BPTT must not time out, the gradients stay bit-identical, and the backward
    may slow down only by the CUs it lost: asserted bound 1.5 x with 32 CUs held (measured
    1.35-1.39 x on the backward with 16-64 held, round 3: +21-23 % on the whole step)."""
    from tacotron2_amd import engine
    from tacotron2_amd.hparams import create_hparams
    from tacotron2_amd.loss_function import Tacotron2Loss
    from tacotron2_amd.model import Tacotron2
    from tacotron2_amd.synth import synth_batch
    dev = torch.device("cuda", 0)
    hp = create_hparams()
    full = synth_batch(64, 1234)
    ol = full[4].clamp(max=120)
    To = int(ol.max())
    batch = tuple(t.to(dev) for t in (full[0], full[1], full[2][:, :, :To].contiguous(), full[3][:, :To].contiguous(), ol))
    crit = Tacotron2Loss()
    torch.manual_seed(1234)
    model = Tacotron2(hp).to(dev).train()
    model.precision = 'bf16'
    side = torch.cuda.Stream()

    def step(ncus=0):
        """One training step; with ncus > 0 that many CUs are held from after the forward until the backward has run.
        Returns (loss, ms of the backward, CUs actually held)."""
        torch.manual_seed(5)
        model.zero_grad()
        x, y = model.parse_batch(batch)
        loss = crit(model(x), y)
        torch.cuda.synchronize()
        held, stop = 0, None
        if ncus:
            stop = torch.zeros(1, dtype=torch.int32, device=dev)
            arrived = torch.zeros(1, dtype=torch.int32, device=dev)
            _hold(native_lib, ncus, 3000.0, stop, arrived, side)
            t_w = time.perf_counter()
            while int(arrived.item()) < ncus and time.perf_counter() - t_w < 5.0:     # the holders own their CUs
                time.sleep(0.001)
            held = int(arrived.item())
        t0 = time.perf_counter()
        loss.backward()
        torch.cuda.current_stream().synchronize()
        ms = 1e3 * (time.perf_counter() - t0)
        if stop is not None:
            stop.fill_(1)                                                         # release the CUs
        torch.cuda.synchronize()
        return loss, ms, held

    _held_cus_cases(native_lib, model, step, To)


def _held_cus_cases(native_lib, model, step, To):
    step()
    loss0, base_ms, _ = step()
    base_ms = min(base_ms, step()[1])
    ref = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    ref_loss = float(loss0)
    native.attn_handoff_timeouts(reset=True)
    rows = dict(B=64, To=To, precision='bf16', backward_ms_alone=base_ms, forward=model.last_train_decoder_path,
                encoder_bptt=model.last_encoder_bwd_path, cases=[])
    for ncus in (16, 32, 64):
        loss, ms, held = step(ncus)
        finite = all(bool(torch.isfinite(p.grad).all()) for p in model.parameters())
        same = all(torch.equal(p.grad, ref[k]) for k, p in model.named_parameters())
        rows['cases'].append(dict(cus_held=held, asked=ncus, backward_ms=ms, ratio=ms / base_ms, finite=finite, bit_identical=same,
                                  loss_equal=float(loss) == ref_loss))
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, "coresidency_stress.json"), "w") as f:
            json.dump(rows, f, indent=1)
        assert held == ncus, rows
        assert finite and same and float(loss) == ref_loss, rows
        assert native.attn_handoff_timeouts(reset=False) == 0 and native.encoder_handoff_timeouts(reset=False) == 0, rows
        # a bounded spin that gave up costs 50 ms per launch: the backward may slow down by the CUs it lost, not by timeouts
        assert ms < 3.0 * base_ms + 20.0, rows
        if ncus == 32:
            assert ms < 1.5 * base_ms, rows      # measured 1.35-1.39 x with 16 / 32 / 64 held (profiles/r04_coresidency_stress.json)


def test_whole_step_under_held_cus_falls_back_to_the_launch_chain(native_lib):
    """The persistent decoder loop needs EVERY CU to itself (256 workgroups of 96 KB LDS, all resident at once).  With CUs held
    for the whole step -- another process on the GPU, not the product layout -- its arrival census finds out within 2 ms, before
    anything is written: the step is poisoned (NaN, counted), engine.handle_nonfinite_step() says why and selects the launch
    chain, and the SAME step under the SAME hold then runs to the bit-identical result.  The cost of sharing a GPU is one
    skipped step, not a hang and not a wrong number."""
    from tacotron2_amd import engine
    from tacotron2_amd.hparams import create_hparams
    from tacotron2_amd.loss_function import Tacotron2Loss
    from tacotron2_amd.model import Tacotron2
    from tacotron2_amd.synth import synth_batch
    if not engine.TRAIN_FWD_PERSISTENT:
        pytest.skip("the persistent decoder loop is switched off")
    dev = torch.device("cuda", 0)
    hp = create_hparams()
    full = synth_batch(64, 1234)
    ol = full[4].clamp(max=60)
    To = int(ol.max())
    batch = tuple(t.to(dev) for t in (full[0], full[1], full[2][:, :, :To].contiguous(), full[3][:, :To].contiguous(), ol))
    crit = Tacotron2Loss()
    torch.manual_seed(1234)
    model = Tacotron2(hp).to(dev).train()
    model.precision = 'bf16'
    side = torch.cuda.Stream()
    keep_flags = (engine.TRAIN_FWD_PERSISTENT, engine.ENCODER_BATCH_PERSISTENT)

    def step():
        torch.manual_seed(5)
        model.zero_grad()
        x, y = model.parse_batch(batch)
        loss = crit(model(x), y)
        loss.backward()
        torch.cuda.current_stream().synchronize()
        return loss

    try:
        step()
        ref_loss = float(step())
        assert model.last_train_decoder_path == 'persistent'
        ref = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        native.attn_handoff_timeouts(reset=True)
        stop = torch.zeros(1, dtype=torch.int32, device=dev)
        arrived = torch.zeros(1, dtype=torch.int32, device=dev)
        _hold(native_lib, 16, 4000.0, stop, arrived, side)
        t_w = time.perf_counter()
        while int(arrived.item()) < 16 and time.perf_counter() - t_w < 5.0:
            time.sleep(0.001)
        assert int(arrived.item()) == 16
        t0 = time.perf_counter()
        loss = step()
        ms = 1e3 * (time.perf_counter() - t0)
        assert not torch.isfinite(loss), "the persistent loop cannot have been resident with 16 CUs held"
        assert ms < 500.0, ms                                  # found out by the census, not by hanging
        said = []
        assert engine.handle_nonfinite_step(log=said.append) >= 1 and said
        assert engine.TRAIN_FWD_PERSISTENT is False
        loss2 = step()                                         # same step, same hold, on the launch chain
        stop.fill_(1)
        torch.cuda.synchronize()
        assert model.last_train_decoder_path == 'launch chain' and model.last_train_decoder_bwd_path == 'launch chain'
        assert float(loss2) == ref_loss
        assert all(torch.equal(p.grad, ref[k]) for k, p in model.named_parameters())
    finally:
        engine.TRAIN_FWD_PERSISTENT, engine.ENCODER_BATCH_PERSISTENT = keep_flags
        native.set_attn_fwd_fused(-1)
        native.set_attn_bwd_fused(-1)
        native.set_bptt_cell_fold(1)
