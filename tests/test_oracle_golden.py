"""The oracle (oracle/tacotron2_oracle.py) against the golden vectors produced by the real
reference (tests/golden/make_golden.py).  CPU only."""
import pytest
import torch

import golden_util as gu
from oracle import tacotron2_oracle as orc


def _check_sd(sd, digest):
    for k, d in digest.items():
        f = sd[k].double().reshape(-1)
        assert abs(f.sum().item() - d['sum']) <= 1e-6 * max(1.0, d['abssum']), k
        assert torch.equal(f[d['idx']].float(), d['sample']), "weights regenerated from the seed differ: %s" % k


def _rel(a, b):
    return (a.double() - b.double()).abs().max().item() / max(1.0, b.double().abs().max().item())


@pytest.mark.parametrize("name", ["tiny_train", "default_train", "tiny_train_nomask"])
def test_oracle_train_matches_reference(name):
    fx = gu.load_fixture(name)
    hp = gu.make_hparams(fx['hp'])
    sd = gu.build_state_dict(hp, fx['seed'])
    _check_sd(sd, fx['sd_digest'])
    batch = gu.make_train_batch(fx['in_lens'], fx['out_lens'], hp.n_mel_channels, fx['seed'])
    masks = gu.unpack_masks(fx['masks'])
    loss, out, grads, bufs = orc.train_step_grads(sd, hp, batch, masks)
    for i in range(4):
        assert _rel(out[i], fx['outputs'][i]) < 2e-5, i
    assert _rel(loss, fx['loss']) < 1e-5
    for k, d in fx['grad_digest'].items():
        g = grads[k].double().reshape(-1)
        # conv biases feeding BatchNorm have an exactly-zero true gradient: only rounding noise remains
        assert abs(g.norm().item() - d['l2']) <= 1e-4 * d['l2'] + 2e-6, k
        assert _rel(g[d['idx']].float(), d['sample']) < 5e-5, k
    for k, v in fx['grad_small'].items():
        assert _rel(grads[k], v) < 5e-5, k
    for k, v in fx['buffers'].items():
        assert _rel(bufs[k].float(), v.float()) < 1e-6, k


def test_oracle_inference_matches_reference():
    fx = gu.load_fixture("default_infer")
    hp = gu.make_hparams(fx['hp'])
    sd = gu.build_state_dict(hp, fx['seed'], perturb_bn=True)
    _check_sd(sd, fx['sd_digest'])
    masks = gu.unpack_mask(fx['masks'])
    out, lengths, hit = orc.tacotron2_inference(sd, hp, fx['text'], masks, fx['steps'], fx['threshold'])
    assert lengths.tolist() == fx['lengths']          # gate-stop index: exact
    for i in range(4):
        assert out[i].shape == fx['outputs'][0][i].shape
        assert _rel(out[i], fx['outputs'][0][i]) < 2e-5, i


def test_oracle_batched_inference_equals_per_utterance_reference():
    """batched result == the reference's B=1 result on each unpadded text."""
    fx = gu.load_fixture("tiny_infer_batched")
    hp = gu.make_hparams(fx['hp'])
    sd = gu.build_state_dict(hp, fx['seed'], perturb_bn=True)
    masks = gu.unpack_mask(fx['masks'])
    out, lengths, hit = orc.tacotron2_inference(sd, hp, fx['text'], masks, fx['steps'], fx['threshold'],
                                                input_lengths=torch.tensor(fx['in_lens']))
    assert lengths.tolist() == fx['lengths']
    for b, L in enumerate(fx['lengths']):
        ref = fx['outputs'][b]
        assert _rel(out[0][b, :, :L], ref[0][0]) < 5e-5
        assert _rel(out[1][b, :, :L], ref[1][0]) < 5e-5
        assert _rel(out[3][b, :L, :fx['in_lens'][b]], ref[3][0]) < 5e-5


def test_oracle_edge_cases():
    """Minimum-size input the reference accepts: a B=1 training batch with two frames (BatchNorm in
    train mode rejects a single value per channel, in the reference as well)."""
    hp = gu.make_hparams(gu.TINY_HP)
    sd = gu.build_state_dict(hp, 7)
    batch = gu.make_train_batch([3], [2], hp.n_mel_channels, 7)
    g = torch.Generator().manual_seed(3)
    masks = orc.draw_masks_train(hp, 1, 3, 2, g)
    loss, out, grads, bufs = orc.train_step_grads(sd, hp, batch, masks)
    assert out[0].shape == (1, 80, 2) and out[3].shape == (1, 2, 3)
    assert torch.isfinite(loss)
    assert abs(out[3].sum().item() - 2.0) < 1e-5       # attention weights sum to one


def test_oracle_equals_reference_digest_at_the_timed_size():
    """The oracle at BASELINE configs[1] itself -- synth_batch(64, 1234), B = 64, Ti = 177, To = 870, the batch bench.py
    times and tests/test_zz5 checks the engine on -- against the digest of the REFERENCE's results on the same weights, batch
    and dropout masks (tests/golden/make_golden_fullsize.py; the generator also compares tensor against tensor: 2e-5 / 5e-5).
    One oracle forward + backward, ~40 s of CPU.  Tolerances: integrals and samples of the outputs 2e-5 of their scale,
    gradient L2 norms 1e-4, gradient samples 1e-3 of the tensor's largest sample, loss 1e-6."""
    import os
    import sys
    sys.path.insert(0, gu.GOLDEN_DIR)
    try:
        import make_golden_fullsize as mf
    finally:
        sys.path.remove(gu.GOLDEN_DIR)
    dg = torch.load(os.path.join(gu.GOLDEN_DIR, mf.NAME + ".pt"), weights_only=False)
    threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    try:
        hp, sd, batch, masks, Ti, To = mf.fullsize_case()
        assert (Ti, To) == (dg['meta']['Ti'], dg['meta']['To'])
        loss, out, grads, _ = orc.train_step_grads(sd, hp, batch, masks)
    finally:
        torch.set_num_threads(threads)
    worst = mf.compare_to_digest(dg, out, loss, grads)
    print("oracle vs the reference's digest at B=64/To=870:", worst)
