"""Data-parallel path on CPU: world_size-2 gloo processes.

  * GradSync (the engine-driven bucketed all-reduce): after bucket_ready()+finish() every rank holds
    the world-mean gradient, bucketed postnet -> decoder -> encoder, and the tensors handed back
    are views into the flat bucket buffers;
  * the same through GradSync.out(): the engine's kernels write into the buckets directly;
  * apply_gradient_allreduce: state broadcast from rank 0; Tacotron2 is exchanged by the engine's backward through
    persistent buckets, any other module through hook-driven buckets (world mean, unused parameters, double wrap).
  (N ranks x B == mean of the single-rank gradients with the REAL engine: tests/test_zz9_dp_gpu.py.)
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")          # no host-name lookup (it may not resolve here)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tacotron2_amd.distributed import GradSync, apply_gradient_allreduce, bucket_of, reduce_tensor
        from tacotron2_amd.hparams import create_hparams
        from tacotron2_amd.model import Tacotron2
        import golden_util as gu

        # ---- engine-driven bucketed sync on the real parameter set (tiny dims) ----
        torch.manual_seed(100 + rank)                       # ranks start different on purpose
        model = Tacotron2(create_hparams(gu.TINY_HP))
        apply_gradient_allreduce(model)
        sd = model.state_dict()
        ref = [torch.zeros_like(v) for v in sd.values()]
        for r, v in zip(ref, sd.values()):
            r.copy_(v)
            dist.broadcast(r, 0)
        assert all(torch.equal(r, v) for r, v in zip(ref, sd.values())), "state not broadcast from rank 0"
        sync = model._grad_sync
        assert set(sync.layout) == {'postnet', 'decoder', 'encoder'}
        assert bucket_of('embedding.weight') == 'encoder'
        g = torch.Generator().manual_seed(7 + rank)
        grads = {n: torch.randn(p.shape, generator=g) for n, p in model.named_parameters()}
        mine = {n: t.clone() for n, t in grads.items()}
        sync.start()
        for b in ('postnet', 'decoder', 'encoder'):
            sync.bucket_ready(b, grads)
        sync.finish()
        g_other = torch.Generator().manual_seed(7 + (1 - rank))
        for n, p in model.named_parameters():
            other = torch.randn(p.shape, generator=g_other)
            assert torch.allclose(grads[n], (mine[n] + other) / 2, atol=1e-6), n
            assert grads[n].shape == p.shape
        names = [n for n, _ in model.named_parameters() if n.startswith('postnet.')]
        assert grads[names[0]].untyped_storage().data_ptr() == grads[names[1]].untyped_storage().data_ptr()

        # ---- the engine writes straight into the buckets: out() views, 256-byte aligned, no packing pass ----
        sync.start(torch.device('cpu'))
        mine2 = {}
        for n, p in model.named_parameters():
            o = sync.out(n, p.shape)
            assert o.shape == p.shape and o.data_ptr() % 256 == sync.flat[bucket_of(n)].data_ptr() % 256
            o.copy_(torch.randn(p.shape, generator=g))
            mine2[n] = o.clone()
        for b in ('postnet', 'decoder', 'encoder'):
            sync.bucket_ready(b)
        sync.finish()
        for n, p in model.named_parameters():
            other = torch.randn(p.shape, generator=g_other)
            assert torch.allclose(sync.out(n, p.shape), (mine2[n] + other) / 2, atol=1e-6), n

        # ---- the buckets persist across backwards; a live p.grad that still points into them forces a fresh set ----
        n_alloc = sync.fresh_allocations
        keep_ptr = sync.flat['decoder'].data_ptr()
        sync.start(torch.device('cpu'))
        assert sync.fresh_allocations == n_alloc and sync.flat['decoder'].data_ptr() == keep_ptr
        some = next(iter(model.parameters()))
        some.grad = sync.out(next(iter(dict(model.named_parameters()))), some.shape)      # gradient accumulation in progress
        sync.start(torch.device('cpu'))
        assert sync.fresh_allocations == n_alloc + 1
        some.grad = None

        # ---- any other module: hook-driven buckets, world mean in p.grad after backward (reference distributed.py:126-173)
        torch.manual_seed(100 + rank)                       # ranks start from different weights: rank 0's must win
        net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))
        unused = torch.nn.Linear(3, 3)
        net.add_module('unused', unused)
        import tacotron2_amd.distributed as D
        D.HOOK_BUCKET_BYTES, keep_bb = 64, D.HOOK_BUCKET_BYTES      # tiny buckets: several of them, launched at different times
        try:
            assert apply_gradient_allreduce(net) is net
        finally:
            D.HOOK_BUCKET_BYTES = keep_bb
        assert apply_gradient_allreduce(net) is net         # second wrap: no second set of hooks
        assert len(net._hook_sync.buckets) > 1
        w0 = [p.detach().clone() for p in net.parameters()]
        gather = [torch.zeros_like(w0[0]) for _ in range(world)]
        dist.all_gather(gather, w0[0])
        assert torch.equal(gather[0], gather[1])            # state broadcast
        ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))
        ref.load_state_dict({k: v for k, v in net.state_dict().items() if not k.startswith('unused')})
        for step in range(2):                               # twice: the per-backward state resets
            xs = [torch.randn(4, 6, generator=torch.Generator().manual_seed(50 + 10 * step + r)) for r in range(world)]
            for q_ in ref.parameters():
                q_.grad = None
            for r in range(world):
                (ref[2](ref[1](ref[0](xs[r]))).pow(2).sum() / world).backward()
            net.zero_grad()
            net[2](net[1](net[0](xs[rank]))).pow(2).sum().backward()
            for a, b in zip([net[0].weight, net[0].bias, net[2].weight, net[2].bias], ref.parameters()):
                assert torch.allclose(a.grad, b.grad, atol=1e-6), (step, a.grad, b.grad)
            assert unused.weight.grad is None
        apply_gradient_allreduce(model)                     # the reference wraps twice (train.py:79,179); harmless
        m = reduce_tensor(torch.tensor(float(rank + 1)), world)
        assert abs(m.item() - 1.5) < 1e-6
        q.put((rank, "ok"))
    except Exception as e:                                  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_data_parallel_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def _train_worker(rank, world, port, outdir, q):
    """The training driver end to end on two gloo ranks with the kernels switched off (validate-only): sampler
    sharding, state broadcast, the engine's bucketed gradient exchange inside backward, loss reduction, validation
    collectives and the rank-0-only checkpoint all run; values are meaningless."""
    try:
        import golden_util as gu
        from tacotron2_amd import native, train as tr
        from tacotron2_amd.hparams import create_hparams

        class FiniteLoss(torch.nn.Module):
            def forward(self, out, targets):
                return sum(torch.nan_to_num(o, nan=0.0, posinf=0.0, neginf=0.0).clamp(-1.0, 1.0).sum() for o in out[:3]) * 0.0 + 1.0 + rank

        def finite_clip(params, max_norm):
            for p in params:
                if p.grad is not None:
                    p.grad.zero_()
            return torch.tensor(0.5)

        tr.Tacotron2Loss = FiniteLoss
        torch.nn.utils.clip_grad_norm_ = finite_clip
        native.load()
        native.set_validate_only(True)
        hp = create_hparams(gu.TINY_HP + ",batch_size=2,iters_per_checkpoint=2,epochs=1,distributed_run=True,"
                            "dist_backend=gloo,dist_url=tcp://127.0.0.1:%d,training_files=synthetic:8:3:60,"
                            "validation_files=synthetic:4:4:60" % port)
        last = tr.train(outdir, "logs", None, False, world, rank, "g", hp, max_iterations=2)
        assert last == 1
        assert getattr(tr.load_model, "__module__", "") == "tacotron2_amd.train"
        q.put((rank, "ok"))
    except Exception:                                       # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_training_driver_world2_gloo(tmp_path):
    import json
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    out = str(tmp_path / "run")
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, out, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
    assert os.path.exists(os.path.join(out, "checkpoint_0"))          # written by rank 0 only
    recs = [json.loads(l) for l in open(os.path.join(out, "logs", "scalars.jsonl"))]
    tl = [r["training.loss"] for r in recs if "training.loss" in r]
    assert tl == [1.5, 1.5]                                            # world mean of (1 + rank)
