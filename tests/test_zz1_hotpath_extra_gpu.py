"""Hot-path cases added after the main parity suite (file names sort these last, least risky first):
mask_padding=False against the reference fixture, and the notebook's ``.half()`` inference call."""
import json
import os

import pytest
import torch

import golden_util as gu
from tacotron2_amd.hparams import create_hparams

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_train_step_without_padding_mask(native_lib, precision):
    """hparams.mask_padding=False (reference model.py:490): padded frames keep their decoded values, carry loss
    and gradient.  Same checks as the masked fixtures, against tests/golden/tiny_train_nomask.pt (made by the
    reference) and the live oracle."""
    import test_parity_gpu as tp
    tp.test_train_step_matches_reference_and_oracle(native_lib, "tiny_train_nomask", precision)


def test_notebook_half_inference(native_lib):
    """inference.ipynb cells 7 + 13: ``model.cuda().eval().half()`` then ``model.inference(sequence)``."""
    from tacotron2_amd.model import Tacotron2
    hp = create_hparams("max_decoder_steps=30")
    torch.manual_seed(11)
    model = Tacotron2(hp)
    _ = model.cuda().eval().half()
    seq = torch.randint(1, 148, (1, 21)).cuda().long()
    mel, mel_post, gate, align = model.inference(seq)
    torch.cuda.synchronize()
    assert all(t.dtype == torch.float16 and t.is_cuda for t in (mel, mel_post, gate, align))
    T = mel.shape[2]
    assert 1 <= T <= 30 and mel_post.shape == mel.shape and gate.shape == (1, T, 1) and align.shape == (1, T, 21)
    assert torch.isfinite(mel.float()).all() and torch.isfinite(mel_post.float()).all()
    assert (align.float().sum(2) - 1).abs().max().item() < 5e-3        # float16 rounding of 21 weights
    assert all(p.dtype == torch.float32 for p in model.parameters())


def test_other_dropout_rates_match_oracle(native_lib):
    """p_attention_dropout / p_decoder_dropout other than the default 0.1 (hparams.py:60-61): the keep scale 1/(1-p)
    travels through the forward and the BPTT loops.  Oracle pinned for this configuration against the reference in
    tests/test_reference_dropin_cpu.py; same tolerances as the other edge shapes."""
    import test_parity_gpu as tp
    tp.test_edge_shapes_match_oracle(native_lib, gu.TINY_HP + ",p_attention_dropout=0.3,p_decoder_dropout=0.2",
                                     [9, 9, 3], [14, 6, 21], 1e-3)


SMALL_ATTENTION = ",attention_dim=96,attention_location_n_filters=16,attention_location_kernel_size=17"


def test_smaller_attention_geometry_trains_like_the_oracle(native_lib):
    """attention_dim / attention_location_n_filters / attention_location_kernel_size below the defaults the kernels are
    compiled for (hparams.py:65-70): the five attention weights run zero-embedded (engine._embed_attention), gradients are
    cropped back.  Outputs, loss and all 60 gradients against the oracle (pinned for this geometry against the reference
    in tests/test_reference_dropin_cpu.py), fp32-mode tolerances of the other edge shapes."""
    import test_parity_gpu as tp
    tp.test_edge_shapes_match_oracle(native_lib, gu.TINY_HP + SMALL_ATTENTION, [11, 8, 3], [9, 16, 5], 1e-3)
    tp.test_edge_shapes_match_oracle(native_lib, SMALL_ATTENTION.lstrip(",") + ",attention_location_kernel_size=5",
                                     [19, 2], [3, 18], 1e-3)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_smaller_attention_geometry_decodes_like_the_oracle(native_lib, precision):
    """Same geometry through Tacotron2.inference at the default widths: B = 1 (the persistent kernel in both modes since
    round 3: bf16 rows in LDS / exact f32 rows in LDS + registers) and a ragged batch of three, forced length, against the
    f32 oracle."""
    from oracle import tacotron2_oracle as orc
    from tacotron2_amd.model import Tacotron2
    steps = 36
    hp = gu.make_hparams(SMALL_ATTENTION.lstrip(",") + ",max_decoder_steps=%d" % steps)
    hp.gate_threshold = 2.0
    sd = gu.build_state_dict(hp, 77, perturb_bn=True)
    model = Tacotron2(hp)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    model.precision = precision
    for lens, seed in (([37], 3), ([30, 21, 13], 4)):
        text = gu.make_text(lens, 50 + seed)
        keep = orc.draw_masks_infer(hp, len(lens), steps, torch.Generator().manual_seed(seed))
        model.dropout_masks = dict(prenet_infer=keep.cuda())
        il = torch.tensor(lens) if len(lens) > 1 else None
        with torch.no_grad():
            out = model.inference(text.cuda(), il.cuda() if il is not None else None)
        torch.cuda.synchronize()
        if len(lens) == 1:
            assert model.last_decode_path == 'persistent', model.last_decode_path
            (omel, opost, ogate, oalign), _, _ = orc.tacotron2_inference(sd, hp, text, keep, steps, 2.0)
            refs = [(omel, out[0].float().cpu()), (oalign, out[3].float().cpu())]
        else:
            refs = []
            for b, n in enumerate(lens):
                (omel, _, _, oalign), _, _ = orc.tacotron2_inference(sd, hp, text[b:b + 1, :n], keep[:, :, b:b + 1], steps, 2.0)
                refs.append((omel[0], out[0][b].float().cpu()))
                refs.append((oalign[0], out[3][b, :, :n].float().cpu()))
        for ref, got in refs:
            assert got.shape == ref.shape, (got.shape, ref.shape)
            scale = max(float(ref.abs().mean()), 1e-3)
            err = float((got - ref).abs().mean())
            assert err < (2e-2 if precision == 'bf16' else 2e-5) * scale, (lens, precision, err, scale)


def test_reduced_precision_parameter_storage_in_inference(native_lib):
    """``model.to(torch.bfloat16)`` / ``.to(torch.float16)`` (what the reference notebook's ``.half()`` does to a stock
    nn.Module, inference.ipynb cell 7): inference accepts the stored values (widened exactly to f32 once per weight
    version), computes in the bf16 mode and returns tensors of the storage dtype.  Against the same weights held as f32
    (after the same rounding) in the bf16 compute mode the outputs agree to the rounding of the output cast."""
    import torch
    from tacotron2_amd.hparams import create_hparams
    from tacotron2_amd.model import Tacotron2
    dev = torch.device("cuda", 0)
    hp = create_hparams("max_decoder_steps=24")
    hp.gate_threshold = 2.0
    text = torch.randint(1, 148, (1, 40), generator=torch.Generator().manual_seed(3)).to(dev)
    for dt in (torch.bfloat16, torch.float16):
        torch.manual_seed(11)
        ref = Tacotron2(hp).to(dev).eval()
        with torch.no_grad():
            for p in ref.parameters():
                p.copy_(p.to(dt).float())                        # the values the reduced-precision model stores
            for b_ in ref.buffers():
                if b_.is_floating_point():
                    b_.copy_(b_.to(dt).float())
        ref.precision = 'bf16'
        torch.manual_seed(12)
        with torch.no_grad():
            want = ref.inference(text)
        torch.manual_seed(11)
        low = Tacotron2(hp).to(dev).eval().to(dt)
        assert low.embedding.weight.dtype == dt
        torch.manual_seed(12)
        with torch.no_grad():
            got = low.inference(text)
        assert all(t.dtype == dt for t in got)
        assert got[0].shape == want[0].shape
        assert (got[0].float() - want[0].to(dt).float()).abs().max().item() == 0.0
    # training keeps f32 master weights: reduced-precision storage is refused there
    low.train()
    import pytest as _pytest
    from tacotron2_amd.native import NativeError
    import golden_util as gu
    batch = gu.make_train_batch([9, 5], [12, 8], 80, 1)
    x, _ = low.parse_batch(batch)
    with _pytest.raises(NativeError, match="master weights"):
        low(x)


def test_weight_image_cache_sees_param_data_edits(native_lib, capfd):
    """packed / transposed / bf16 weight images are cached per weight version; a write through ``param.data``
    changes neither the storage address nor the version counter.  The content guard (engine._guard_weights) must notice
    it BEFORE the cached images are reused: the call after the edit equals a fresh model holding the edited weights,
    in inference (both decode paths) and in a training step; deepcopy / pickle do not carry the device images."""
    import copy
    import pickle
    from tacotron2_amd.loss_function import Tacotron2Loss
    from tacotron2_amd.model import Tacotron2
    hp = create_hparams("max_decoder_steps=24")
    hp.gate_threshold = 2.0
    torch.manual_seed(3)
    model = Tacotron2(hp).cuda().eval()
    seq = torch.randint(1, 148, (1, 19)).cuda().long()
    from oracle import tacotron2_oracle as orc
    keep = orc.draw_masks_infer(hp, 1, 24, torch.Generator().manual_seed(2)).cuda()
    for prec in ('fp32', 'bf16'):
        model.precision = prec
        model.dropout_masks = dict(prenet_infer=keep)
        a = [o.clone() for o in model.inference(seq)]
        b = model.inference(seq)                             # images reused, guard compares equal
        assert torch.equal(a[0], b[0])
        capfd.readouterr()
        with torch.no_grad():
            for n, p in model.named_parameters():            # an "EMA swap": every LSTM / projection weight through .data
                if n.startswith('decoder.') and p.dim() == 2:
                    p.data.mul_(0.97)
        got = model.inference(seq)
        assert "version counters" in capfd.readouterr().err
        fresh = Tacotron2(hp).cuda().eval()
        fresh.load_state_dict(model.state_dict())
        fresh.precision = prec
        fresh.dropout_masks = dict(prenet_infer=keep)
        want = fresh.inference(seq)
        for g_, w_ in zip(got, want):
            assert torch.equal(g_, w_), prec
        assert not torch.equal(got[0], a[0])
    # copies carry parameters, not device images or process-group state
    assert len(model._weight_cache) > 0
    twin = copy.deepcopy(model)
    assert not getattr(twin, '_weight_cache', None)
    blob = pickle.dumps(model)
    assert len(blob) < 1.3 * 4 * sum(p.numel() for p in model.parameters()) + (1 << 20)
    # training: the edit between two steps (same FusedAdam generation, same version counters)
    model.train()
    model.precision = 'bf16'
    model.dropout_masks = None
    batch = gu.make_train_batch([13, 9], [22, 17], hp.n_mel_channels, 5)

    def grads(m):
        torch.manual_seed(8)
        m.zero_grad()
        x, y = m.parse_batch(tuple(t.clone() for t in batch))
        Tacotron2Loss()(m(x), y).backward()
        return {k: p.grad.clone() for k, p in m.named_parameters()}
    grads(model)
    with torch.no_grad():
        model.decoder.attention_rnn.weight_hh.data.mul_(1.1)
    g1 = grads(model)
    fresh = Tacotron2(hp).cuda().train()
    fresh.load_state_dict(model.state_dict())
    fresh.precision = 'bf16'
    g2 = grads(fresh)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k


def test_loops_run_through_the_torch_dispatcher(native_lib):
    """The engine's loop-level calls go through torch.ops.tacotron2_amd.* (TORCH_LIBRARY, csrc/torch_ops.cpp) and give
    the same bits as the ctypes route over the same C ABI: one training step and one batched inference."""
    from tacotron2_amd import native
    from tacotron2_amd.loss_function import Tacotron2Loss
    from tacotron2_amd.model import Tacotron2
    assert native.torch_ops() is not None
    hp = create_hparams("max_decoder_steps=20")
    hp.gate_threshold = 2.0
    torch.manual_seed(4)
    model = Tacotron2(hp).cuda().train()
    batch = gu.make_train_batch([14, 9, 6], [25, 19, 12], hp.n_mel_channels, 6)
    text = gu.make_text([12, 7], 3).cuda()
    lens = torch.tensor([12, 7]).cuda()

    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}

    def run():
        res = {}
        model.load_state_dict(sd0)               # a training step moves the BatchNorm running statistics inference reads
        for prec in ('fp32', 'bf16'):
            model.train()
            model.precision = prec
            torch.manual_seed(9)
            model.zero_grad()
            x, y = model.parse_batch(tuple(t.clone() for t in batch))
            loss = Tacotron2Loss()(model(x), y)
            loss.backward()
            route_train = native.last_loop_route
            res[prec] = [loss.detach().clone()] + [p.grad.clone() for p in model.parameters()]
            model.eval()
            torch.manual_seed(10)
            res[prec] += [o.clone() for o in model.inference(text, lens)]
            res[prec + '_routes'] = (route_train, native.last_loop_route)
        return res
    keep = native._torch_ops
    a = run()
    assert a['fp32_routes'] == ('torch.ops', 'torch.ops') and a['bf16_routes'] == ('torch.ops', 'torch.ops'), a['fp32_routes']
    try:
        native._torch_ops = False
        b = run()
    finally:
        native._torch_ops = keep
    assert b['fp32_routes'] == ('ctypes', 'ctypes')
    for prec in ('fp32', 'bf16'):
        for u, v in zip(a[prec], b[prec]):
            assert torch.equal(u, v), prec


def test_attention_handoff_timeouts_are_reported_and_handled(native_lib, capfd):
    """an abandoned in-launch hand-off poisons the step with NaN; the host must be able to learn WHY.  The
    kernels count give-ups in a device word; engine.handle_nonfinite_step reads it, reports, and selects the
    separate-launch forms (no co-residency assumption), which give the same bits."""
    import ctypes as C
    from tacotron2_amd import engine, native
    from tacotron2_amd.loss_function import Tacotron2Loss
    from tacotron2_amd.model import Tacotron2
    assert native.attn_handoff_timeouts(reset=True) >= 0
    assert native.attn_handoff_timeouts(reset=False) == 0
    assert engine.handle_nonfinite_step() == 0                   # nothing timed out: forms untouched
    hp = create_hparams("")
    torch.manual_seed(21)
    model = Tacotron2(hp).cuda().train()
    model.precision = 'bf16'
    batch = gu.make_train_batch([15, 11, 8, 6], [28, 21, 17, 9], hp.n_mel_channels, 8)

    def grads():
        torch.manual_seed(3)
        model.zero_grad()
        x, y = model.parse_batch(tuple(t.clone() for t in batch))
        Tacotron2Loss()(model(x), y).backward()
        return [p.grad.clone() for p in model.parameters()]
    ref = grads()
    f = native_lib.t2amd_debug_attn_timeout_                     # what a timed-out spin does to the counter
    f.argtypes = [C.c_void_p]
    assert f(native._stream()) == 0 and f(native._stream()) == 0
    torch.cuda.synchronize()
    fold0 = native.get_bptt_cell_fold()
    # (the handler demotes the process to the chains: what it changes is put back below -- found by conftest's flag guard in round 5;
    # until then every in-process test collected after this one ran the forward loop on the launch chain)
    entry = (engine.TRAIN_FWD_PERSISTENT, engine.ENCODER_BATCH_PERSISTENT, dict(engine._DEMOTION))
    try:
        assert engine.handle_nonfinite_step() == 2
        assert "hand-off(s) timed out" in capfd.readouterr().err
        assert native.attn_handoff_timeouts(reset=False) == 0    # reset by the handler
        assert native.get_bptt_cell_fold() == 0                  # separate-launch forms selected ...
        got = grads()
        for a, b in zip(got, ref):                               # ... and they compute the same bits
            assert torch.equal(a, b)
    finally:
        native.set_attn_fwd_fused(-1)
        native.set_attn_bwd_fused(-1)
        native.set_bptt_cell_fold(fold0)
        engine.TRAIN_FWD_PERSISTENT, engine.ENCODER_BATCH_PERSISTENT = entry[:2]
        engine._DEMOTION.clear(); engine._DEMOTION.update(entry[2])


def test_weight_gradient_routes_agree_including_the_masked_postnet_input(native_lib):
    """bf16 mode: the weight gradients straight from the K-major slabs / halo images (engine.WGRAD_KK, gemm16_kk with
    transposing LDS reads) against round 2's route (transposed copies + f32-source convolution gradients) on one model, batch
    and dropout seed, at a size where both are active (postnet rows >= 4096).  Both round the same operands to bf16, so every
    gradient agrees to summation order -- including the FIRST postnet convolution's, whose saved input the reference masks in
    place after the forward (model.py:491-495): an image kept from the forward would be the unmasked one (98 % off)."""
    from tacotron2_amd import engine
    from tacotron2_amd.loss_function import Tacotron2Loss
    from tacotron2_amd.model import Tacotron2
    from tacotron2_amd.synth import synth_batch
    dev = torch.device("cuda", 0)
    hp = create_hparams()
    hp.batch_size = 8
    torch.manual_seed(hp.seed)
    model = Tacotron2(hp).to(dev)
    model.precision = "bf16"
    model.train()
    criterion = Tacotron2Loss()
    batch = tuple(t.to(dev) for t in synth_batch(8, 4321))
    start = engine.WGRAD_KK
    sd = {k: v.clone() for k, v in model.state_dict().items()}

    def grads_of(kk):
        engine.WGRAD_KK = kk
        model.load_state_dict(sd)
        torch.manual_seed(5)
        model.zero_grad()
        x, y = model.parse_batch(batch)
        loss = criterion(model(x), y)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.item()), {k: p.grad.detach().clone() for k, p in model.named_parameters()}

    try:
        l1, g1 = grads_of(True)
        l0, g0 = grads_of(False)
    finally:
        engine.WGRAD_KK = start
    assert batch[2].shape[2] * 8 >= 4096                         # the window / K-major routes were the ones running
    assert l1 == l0
    rel = {k: float((g1[k] - g0[k]).abs().max() / g0[k].abs().max().clamp_min(1e-30)) for k in g0}
    worst = max(rel, key=rel.get)
    assert rel[worst] < 2e-5, (worst, rel[worst])
    assert all(bool(torch.isfinite(v).all()) for v in g1.values())


def test_folded_batchnorm_backward_and_bf16_bias_sums_agree_with_the_separate_passes(native_lib):
    """bf16 mode, round 6: (i) the BatchNorm backward of a convolution layer writes the bf16 halo image and the bias gradient
    itself (engine.BN_BWD_IMAGE; no f32 slab, no cast pass, no column-sum pass) and the BatchNorm apply of the forward writes the
    next layer's image (engine.BN_FWD_IMAGE) -- loss and every gradient of the step must be the same BITS as with the separate passes; (ii) the two LSTM bias gradients are column sums of the bf16 gate-gradient slabs
    (engine.BIAS_GRAD16) -- only those four tensors may move, by bf16 rounding of the addends; (iii) the f32 gate-gradient slabs,
    which nothing reads any more, are dropped (engine.GATE_GRADS_BF16_ONLY) -- same bits."""
    from tacotron2_amd import engine
    from tacotron2_amd.loss_function import Tacotron2Loss
    from tacotron2_amd.model import Tacotron2
    from tacotron2_amd.synth import synth_batch
    dev = torch.device("cuda", 0)
    hp = create_hparams()
    hp.batch_size = 24
    torch.manual_seed(hp.seed)
    model = Tacotron2(hp).to(dev)
    model.precision = "bf16"
    model.train()
    criterion = Tacotron2Loss()
    batch = tuple(t.to(dev) for t in synth_batch(24, 4321))
    assert batch[2].shape[2] * 24 >= 4096 and batch[0].shape[1] * 24 >= 4096      # window / K-major routes in postnet AND encoder
    start = (engine.BN_BWD_IMAGE, engine.BN_FWD_IMAGE, engine.BIAS_GRAD16, engine.GATE_GRADS_BF16_ONLY, engine.DXD_RING)
    sd = {k: v.clone() for k, v in model.state_dict().items()}

    def grads_of(img, b16, drop32=False, ring=0):
        engine.BN_BWD_IMAGE, engine.BN_FWD_IMAGE, engine.BIAS_GRAD16, engine.GATE_GRADS_BF16_ONLY, engine.DXD_RING = img, img, b16, drop32, ring
        model.load_state_dict(sd)
        torch.manual_seed(5)
        model.zero_grad()
        x, y = model.parse_batch(batch)
        loss = criterion(model(x), y)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.item()), {k: p.grad.detach().clone() for k, p in model.named_parameters()}

    try:
        l_sep, g_sep = grads_of(False, False)
        l_img, g_img = grads_of(True, False)
        l_all, g_all = grads_of(True, True)
        l_drop, g_drop = grads_of(True, True, True)
        l_ring, g_ring = grads_of(True, True, True, 3)
        from tacotron2_amd import native
        native.set_decoder_streams(2)             # the decoder-LSTM chain on its side stream: one slab per time step again, no f32 slabs
        try:
            l_two, g_two = grads_of(True, True, True, 3)
        finally:
            native.set_decoder_streams(1)
    finally:
        engine.BN_BWD_IMAGE, engine.BN_FWD_IMAGE, engine.BIAS_GRAD16, engine.GATE_GRADS_BF16_ONLY, engine.DXD_RING = start
    assert l_sep == l_img == l_all == l_drop == l_ring
    # (iv) the decoder LSTM's input-gradient slabs as a ring of three instead of one per time step: same bits
    diff = [k for k in g_drop if not torch.equal(g_drop[k], g_ring[k])]
    assert not diff, diff
    assert l_two == l_ring
    diff = [k for k in g_ring if not torch.equal(g_ring[k], g_two[k])]
    assert not diff, diff                                          # (v) and the two-stream loop agrees bit for bit
    # (iii) with every consumer on the bf16 slabs, the f32 gate-gradient slabs are neither allocated nor written: same bits
    diff = [k for k in g_all if not torch.equal(g_all[k], g_drop[k])]
    assert not diff, diff
    diff = [k for k in g_sep if not torch.equal(g_sep[k], g_img[k])]
    assert not diff, diff                                          # (i) bit-identical, every tensor
    lstm_bias = {'decoder.attention_rnn.bias_ih', 'decoder.attention_rnn.bias_hh', 'decoder.decoder_rnn.bias_ih',
                 'decoder.decoder_rnn.bias_hh'}
    moved = {k for k in g_sep if not torch.equal(g_sep[k], g_all[k])}
    assert moved <= lstm_bias, moved - lstm_bias                   # (ii) nothing else changes
    for k in lstm_bias:
        rel = float((g_all[k] - g_sep[k]).abs().max() / g_sep[k].abs().max())
        assert rel < 2e-3, (k, rel)
        cos = float(torch.nn.functional.cosine_similarity(g_all[k].double().view(1, -1), g_sep[k].double().view(1, -1)))
        assert cos > 0.999999, (k, cos)


def test_encoder_launch_give_up_poisons_the_step_and_is_reported(native_lib, capfd):
    """The training step does not read the persistent encoder launch's status back (a host sync per step); a give-up is
    turned into a NaN in the step's data by a one-thread launch behind the kernel and counted in the library.  Here that
    finishing launch runs alone on a status word set by the test: nothing happens on 0; on a non-zero status the poison
    word is NaN, the counter says 1, engine.handle_nonfinite_step() reports it, resets the counter and moves the encoder to
    the launch chain; and a real training forward takes the persistent route with a finite result."""
    import ctypes as C
    from tacotron2_amd import engine, native
    lib = native.load()
    fn = lib.t2amd_debug_encoder_poison_
    fn.argtypes, fn.restype = [C.c_void_p, C.c_void_p, C.c_void_p], C.c_int
    native.encoder_handoff_timeouts(reset=True)
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    word = torch.ones(4, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    assert fn(status.data_ptr(), word.data_ptr(), stream) == 0
    torch.cuda.synchronize()
    assert torch.all(word == 1) and native.encoder_handoff_timeouts(reset=False) == 0
    status.fill_(3)
    assert fn(status.data_ptr(), word.data_ptr(), stream) == 0
    torch.cuda.synchronize()
    assert torch.isnan(word[0]) and torch.all(word[1:] == 1)
    assert native.encoder_handoff_timeouts(reset=False) == 1
    keep = engine.ENCODER_BATCH_PERSISTENT
    try:
        engine.ENCODER_BATCH_PERSISTENT = True
        assert engine.handle_nonfinite_step() >= 1
        assert "persistent encoder launch gave up 1 time" in capfd.readouterr().err
        assert engine.ENCODER_BATCH_PERSISTENT is False and native.encoder_handoff_timeouts(reset=False) == 0
        # a real training forward: the persistent route (opt-in for training), finite outputs, nothing counted
        engine.ENCODER_BATCH_PERSISTENT = True
        keep_train, engine.ENCODER_BATCH_PERSISTENT_TRAIN = engine.ENCODER_BATCH_PERSISTENT_TRAIN, True
        from tacotron2_amd.loss_function import Tacotron2Loss
        from tacotron2_amd.model import Tacotron2
        from tacotron2_amd.synth import synth_batch
        hp = create_hparams()
        hp.batch_size = 4
        torch.manual_seed(hp.seed)
        model = Tacotron2(hp).cuda().train()
        x, y = model.parse_batch(tuple(t.cuda() for t in synth_batch(4, 99)))
        loss = Tacotron2Loss()(model(x), y)
        loss.backward()
        torch.cuda.synchronize()
        assert model.last_encoder_path == 'persistent' and bool(torch.isfinite(loss))
        assert native.encoder_handoff_timeouts(reset=False) == 0
        engine.ENCODER_BATCH_PERSISTENT_TRAIN = keep_train
    finally:
        engine.ENCODER_BATCH_PERSISTENT = keep
