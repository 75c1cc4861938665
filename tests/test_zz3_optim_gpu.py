"""Global-norm clip + Adam in two HIP launches against the torch pair the reference calls."""
import json
import os

import pytest
import torch

import golden_util as gu
from tacotron2_amd.hparams import create_hparams

pytestmark = pytest.mark.gpu


def test_fused_adam_matches_torch_clip_and_adam(native_lib):
    """optim.FusedAdam (two HIP launches) against the pair the reference calls, torch.nn.utils.clip_grad_norm_ +
    torch.optim.Adam(lr, weight_decay) (train.py:170-171, 233-236), run on the CPU in float32 on the same tensors:
    odd sizes, a 1-element tensor, views at odd offsets of a flat buffer (the data-parallel bucket layout), chunk
    edges (4095 / 4096 / 4097 elements), one step with the clip active and two without.
    Tolerance: |dp| < 3e-7 (an update is ~1e-3 and float32 rounding of the few operations that may be fused
    differently is ~1e-10, but one flipped rounding of p + dp costs an ulp of p: 1.2e-7 for |p| in [1, 2)),
    moments 1e-5 relative, norm 1e-5 relative."""
    from tacotron2_amd.optim import FusedAdam
    gen = torch.Generator().manual_seed(21)
    sizes = [(4095,), (4096,), (4097,), (1,), (7, 13), (129, 257), (3, 5, 31), (10000,)]
    total = sum(int(torch.tensor(s).prod()) for s in sizes) + len(sizes) + 3
    flat_p = torch.randn(total, generator=gen) * 0.3
    cpu_params, gpu_params, off = [], [], 1                      # offset 1: nothing is 16-byte aligned
    flat_gpu = flat_p.cuda()
    for s in sizes:
        n = int(torch.tensor(s).prod())
        cpu_params.append(torch.nn.Parameter(flat_p[off:off + n].clone().view(s)))
        gpu_params.append(torch.nn.Parameter(flat_gpu[off:off + n].view(s)))   # a view into the flat device buffer
        off += n + 1
    ref = torch.optim.Adam(cpu_params, lr=1e-3, weight_decay=1e-6)
    opt = FusedAdam(gpu_params, lr=1e-3, weight_decay=1e-6)
    for it, (scale, clip) in enumerate([(5.0, 1.0), (1e-3, 1.0), (1.0, None)]):
        flat_g = torch.randn(total, generator=gen) * scale
        flat_g_gpu, off = flat_g.cuda(), 1
        for pc, pg, s in zip(cpu_params, gpu_params, sizes):
            n = pc.numel()
            pc.grad = flat_g[off:off + n].clone().view(s)
            pg.grad = flat_g_gpu[off:off + n].view(s)
            off += n + 1
        before = [g.grad.clone() for g in gpu_params]
        if clip is not None:
            n_ref = torch.nn.utils.clip_grad_norm_(cpu_params, clip)
        ref.step()
        n_got = opt.step(clip_norm=clip)
        torch.cuda.synchronize()
        if clip is not None:
            assert abs(float(n_got) - float(n_ref)) <= 1e-5 * float(n_ref), (it, float(n_got), float(n_ref))
        else:
            assert n_got is None
        for pc, pg, b in zip(cpu_params, gpu_params, before):
            assert torch.equal(pg.grad, b)                       # gradients are read, never rescaled in place
            d = (pg.detach().cpu() - pc.detach()).abs().max().item()
            assert d < 3e-7, (it, tuple(pc.shape), d)
            for key in ("exp_avg", "exp_avg_sq"):
                a, r = opt.state[pg][key].cpu(), ref.state[pc][key]
                # relative to the tensor's scale: a moment that cancels to ~0 has no relative precision of its own
                assert ((a - r).abs() <= 1e-5 * r.abs() + 1e-6 * r.abs().max()).all(), (it, key, tuple(pc.shape))
            assert float(opt.state[pg]["step"]) == float(ref.state[pc]["step"]) == it + 1
    # the neighbours of every view in the flat buffer were not touched
    off = 1
    for s in sizes:
        n = int(torch.tensor(s).prod())
        assert float(flat_gpu[off - 1]) == float(flat_p[off - 1]) and float(flat_gpu[off + n]) == float(flat_p[off + n])
        off += n + 1


def test_train_driver_with_fused_optimizer(native_lib, tmp_path):
    from tacotron2_amd import train as tr
    hpstr = gu.TINY_HP + ",batch_size=2,iters_per_checkpoint=2,epochs=2,training_files=synthetic:6:3:60," \
                         "validation_files=synthetic:3:4:60"
    runs = {}
    for name, fused in (("torch", False), ("fused", True)):
        out = tmp_path / name
        tr.train(str(out), "logs", None, False, 1, 0, "g", create_hparams(hpstr), max_iterations=3, fused_optimizer=fused)
        recs = [json.loads(l) for l in open(out / "logs" / "scalars.jsonl")]
        runs[name] = ([r["training.loss"] for r in recs if "training.loss" in r],
                      [r["grad.norm"] for r in recs if "grad.norm" in r])
    # same seed, same data order, same dropout stream: the two optimisers must walk the same trajectory
    for a, b in zip(runs["torch"][0], runs["fused"][0]):
        assert abs(a - b) <= 2e-3 * abs(a), runs
    for a, b in zip(runs["torch"][1], runs["fused"][1]):
        assert abs(a - b) <= 2e-2 * abs(a), runs
    ck = torch.load(tmp_path / "fused" / "checkpoint_2", weights_only=False)
    assert len(ck["optimizer"]["state"]) == 60
    plain = torch.optim.Adam(tr.load_model(create_hparams(hpstr)).parameters())
    plain.load_state_dict(ck["optimizer"])                       # the reference's optimiser reads the checkpoint


def test_fused_adam_skipped_step_fixes_its_own_step_counts(native_lib):
    """a step whose global norm is not finite is skipped on the device; the bias-correction step counts must
    not count it -- without the caller doing anything, and only for the parameters that step covered."""
    import math
    from tacotron2_amd.optim import FusedAdam
    torch.manual_seed(0)
    a = torch.nn.Parameter(torch.randn(300, device="cuda"))
    b = torch.nn.Parameter(torch.randn(77, 5, device="cuda"))
    c = torch.nn.Parameter(torch.randn(9, device="cuda"))            # never gets a gradient in the skipped step
    ref = [torch.nn.Parameter(p.detach().clone()) for p in (a, b, c)]
    opt = FusedAdam([a, b, c], lr=1e-2, weight_decay=1e-6)
    ropt = torch.optim.Adam(ref, lr=1e-2, weight_decay=1e-6)

    def grads(ps, seed, with_c=True, poison=False):
        g = torch.Generator(device="cuda").manual_seed(seed)
        for i, p in enumerate(ps):
            p.grad = None if (i == 2 and not with_c) else torch.randn(p.shape, device="cuda", generator=g)
        if poison:
            ps[0].grad[3] = float("nan")

    # step 1: all three parameters, finite
    grads([a, b, c], 1); grads(ref, 1)
    opt.step(clip_norm=1.0)
    torch.nn.utils.clip_grad_norm_(ref, 1.0); ropt.step()
    # step 2: NaN gradient, c has no gradient -> skipped on the device; the reference loop skips it by hand
    before = [p.detach().clone() for p in (a, b, c)]
    grads([a, b, c], 2, with_c=False, poison=True)
    n = opt.step(clip_norm=1.0)
    assert not math.isfinite(float(n))
    for p, q in zip((a, b, c), before):
        assert torch.equal(p.detach(), q)                             # weights untouched
    # step 3: finite again; nobody called undo_step_count()
    grads([a, b, c], 3); grads(ref, 3)
    opt.step(clip_norm=1.0)
    torch.nn.utils.clip_grad_norm_(ref, 1.0); ropt.step()
    torch.cuda.synchronize()
    assert [float(opt.state[p]['step']) for p in (a, b, c)] == [2.0, 2.0, 2.0]
    for p, q in zip((a, b, c), ref):
        assert (p.detach() - q.detach()).abs().max().item() < 3e-6
    # the explicit call still works and is idempotent for one step
    grads([a, b, c], 4, poison=True)
    opt.step(clip_norm=1.0)
    opt.undo_step_count(); opt.undo_step_count()
    assert [float(opt.state[p]['step']) for p in (a, b, c)] == [2.0, 2.0, 2.0]
