"""Data parallelism with the REAL engine on the one GPU there is: two ``gloo`` ranks sharing ``cuda:0``
(RCCL refuses two ranks on one device; the collective is the only thing that differs from the driver's
one-rank-per-GPU RCCL run -- the engine, the bucket views it writes into, the ordering of the asynchronous
all-reduces against the engine's raw-stream kernel launches and the views handed to autograd are the same code).

Reference contract (distributed.py:126-173, train.py:20-24): after ``backward`` every ``p.grad`` is the world MEAN
of the per-rank gradients; BatchNorm statistics stay per rank; ``reduce_tensor`` is the world-mean loss.

Each rank first computes, WITHOUT any exchange, the gradients of both shards on the rank-0 weights (the engine is
bitwise deterministic, so both ranks hold the same two results), then runs its own shard data-parallel and must
find ``p.grad == (g_shard0 + g_shard1) / 2`` up to the rounding of one f32 add and one multiply.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, precision, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # gloo picks its interface by resolving the host name, which may not resolve in the test containers (a lookup that times
    # out costs a minute or more per rank, seen as a 76 s -- once > 600 s -- run of this test): bind to loopback
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    # a rank that stops making progress says where: after 240 s every thread's stack goes to its log and the rank exits
    import faulthandler
    logdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(logdir, exist_ok=True)
    log = open(os.path.join(logdir, "dp_rank%d_%s.log" % (rank, precision)), "w")
    faulthandler.enable(log)
    faulthandler.dump_traceback_later(240, exit=True, file=log)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=200))
        import golden_util as gu
        from tacotron2_amd import native
        from tacotron2_amd.distributed import apply_gradient_allreduce, reduce_tensor
        from tacotron2_amd.loss_function import Tacotron2Loss
        from tacotron2_amd.model import Tacotron2
        native.load()
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        hp = gu.make_hparams("")
        shards = [gu.make_train_batch([23, 17, 9], [41, 33, 20], hp.n_mel_channels, 500 + r) for r in range(world)]
        from tacotron2_amd.engine import MaskSource

        def masks_for(shard):
            ms = MaskSource(None, dev)
            ms.seed, (B, Ti, To) = 1000 + shard, (3, 23, 41)
            return dict(enc=[ms.get('enc', i, (B, Ti, 512), 0.5) for i in range(3)],
                        prenet=[ms.get('prenet', i, (To, B, 256), 0.5) for i in range(2)],
                        att=ms.get('att', None, (To, B, 1024), 0.1), dec=ms.get('dec', None, (To, B, 1024), 0.1),
                        post=[ms.get('post', i, (B, To, c), 0.5) for i, c in enumerate([512] * 4 + [80])])

        torch.manual_seed(1234 + 7 * rank)                    # ranks start from DIFFERENT weights on purpose
        model = Tacotron2(hp).to(dev).train()
        model.precision = precision
        crit = Tacotron2Loss()

        def run(shard):
            model.zero_grad()
            model.dropout_masks = masks_for(shard)
            x, y = model.parse_batch(tuple(t.clone() for t in shards[shard]))
            loss = crit(model(x), y)
            loss.backward()
            torch.cuda.synchronize()
            return loss.detach(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}

        model = apply_gradient_allreduce(model)               # broadcast rank 0's state; installs _grad_sync
        sync = model._grad_sync
        w0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        chk = torch.stack([v.double().sum() for v in w0.values() if v.dtype.is_floating_point]).sum().cpu()
        both = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(both, chk)
        assert all(torch.equal(both[0], b) for b in both), "state not identical after the wrap-time broadcast"

        # single-rank references on both shards (no exchange), BN buffers restored in between
        model._grad_sync = None
        single, bn_after = [], []
        for s in range(world):
            model.load_state_dict(w0)
            single.append(run(s))
            bn_after.append({k: v.detach().clone() for k, v in model.state_dict().items() if 'running_' in k})
        model.load_state_dict(w0)

        # data-parallel step on this rank's shard
        model._grad_sync = sync
        loss, grads = run(rank)
        mean_loss = reduce_tensor(loss, world)
        # the single-rank references must be the same bits on both ranks (the engine is deterministic, also with two
        # processes interleaving their kernels on one GPU)
        for s in range(world):
            chk = torch.stack([g.double().sum() for g in single[s][1].values()]).cpu()
            both = [torch.zeros_like(chk) for _ in range(world)]
            dist.all_gather(both, chk)
            assert torch.equal(both[0], both[1]), "single-rank gradients of shard %d differ between the ranks" % s
        worst, bad = 0.0, []

        def compare(grads, tag):
            nonlocal worst
            for k, g in grads.items():
                want = (single[0][1][k].double() + single[1][1][k].double()) / 2
                scale = want.abs().max().item() + 1e-30
                err = (g.double() - want).abs().max().item()
                own = (g.double() - single[rank][1][k].double()).abs().max().item()   # = what a missing exchange gives
                worst = max(worst, err / scale)
                if err > 2e-7 * scale + 1e-12:
                    bad.append((tag, k, err / scale, own / scale))
        compare(grads, 0)
        assert not bad, bad[:12]
        assert abs(float(mean_loss) - (float(single[0][0]) + float(single[1][0])) / 2) < 1e-6 * abs(float(mean_loss))
        # gradients are views of the three flat buckets (no copy back), BN statistics stay this rank's own
        names = [n for n, _ in model.named_parameters() if n.startswith('decoder.')]
        pg = dict(model.named_parameters())
        views = pg[names[0]].grad.untyped_storage().data_ptr() == pg[names[-1]].grad.untyped_storage().data_ptr()
        for k, v in bn_after[rank].items():
            assert torch.equal(model.state_dict()[k], v), k
        # DESIGN round 2 recorded ONE run of this test with a 1e-3 discrepancy that never reappeared.  The exchange is
        # therefore repeated REPEATS more times in this process pair (same weights, same shard, same masks: the same
        # expected mean every time); any repeat that differs is reported with its index and tensor.
        first = {k: g.clone() for k, g in grads.items()}
        unequal_repeats, notes = 0, []
        for rep in range(1, 1 + int(os.environ.get("T2AMD_DP_REPEATS", "24"))):
            l2, g2 = run(rank)
            compare(g2, rep)
            diff = [k for k in first if not torch.equal(g2[k], first[k])]
            if diff:
                # what a repeat that differs looked like (round 5 saw ONE in ~30 runs of this test, after the one of round 2):
                # whether this rank's OWN loss moved (its forward / inputs) or only the exchanged gradients did (the other
                # rank's step or the exchange), how many tensors and how many elements of the first one
                unequal_repeats += 1
                k0 = diff[0]
                notes.append(dict(repeat=rep, rank=rank, own_loss_equal=bool(torch.equal(l2, loss)), tensors=len(diff), of=len(first),
                                  names=[k for k in first if k not in diff] if len(diff) > len(first) // 2 else diff, names_are='EQUAL tensors' if len(diff) > len(first) // 2 else 'differing tensors',
                                  first=k0, elements=int((g2[k0] != first[k0]).sum()), numel=g2[k0].numel(),
                                  max_rel=float((g2[k0].double() - first[k0].double()).abs().max() / (first[k0].double().abs().max() + 1e-30))))
        if notes:
            import json
            with open(os.path.join(logdir, "dp_unequal_repeats_rank%d_%s.json" % (rank, precision)), "w") as f:
                json.dump(notes, f, indent=1)
        assert not bad, (notes[:3], bad[:12])
        assert unequal_repeats == 0, ("%d repeats of the same exchange were not bit-identical to the first" % unequal_repeats, notes[:3])
        # one optimiser step on the averaged gradients keeps the ranks identical
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        chk = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().cpu()
        both = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(both, chk)
        assert torch.equal(both[0], both[1]), "ranks diverged after one step on the averaged gradients"
        q.put((rank, "ok", worst, bool(views)))
    except Exception:                                         # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc(), None, None))
    finally:
        faulthandler.cancel_dump_traceback_later()
        log.close()
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_two_ranks_share_the_gpu_real_engine(native_lib, precision):
    import queue
    ctx = mp.get_context("spawn")
    attempts = []
    for attempt in range(2):
        # Two processes sharing one GPU is a configuration of this TEST (the product runs one rank per GPU).  A rank that
        # never reports -- seen once in a dozen runs, nothing in either rank's log -- is retried once and recorded; a rank
        # that reports a mismatch or an exception fails the test at once.
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, precision, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = []
        try:
            for _ in procs:
                res.append(q.get(timeout=300))
        except queue.Empty:
            res.append((-1, "no report within 300 s (exit codes %s): see gpurun_out/dp_rank*_%s.log"
                        % ([p.exitcode for p in procs], precision), None, None))
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
        attempts.append([r[1] for r in res])
        if all(r[0] >= 0 for r in res):
            break
    assert all(r[1] == "ok" for r in res), res
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_dp_%s.json" % precision), "w") as f:
        import json
        json.dump(dict(world=2, backend="gloo, both ranks on cuda:0", precision=precision,
                       worst_relative_error_vs_mean_of_single_rank_grads=max(r[2] for r in res),
                       p_grad_is_a_view_of_its_bucket=all(r[3] for r in res), attempts=attempts), f)
