"""Drop-in boundary, host logic and C-ABI surface (CPU only, no compute calls)."""
import ctypes
import re
import os

import pytest
import torch

import golden_util as gu
from tacotron2_amd.hparams import create_hparams
from tacotron2_amd.model import Tacotron2
from tacotron2_amd import native


def test_hparams_defaults_and_parsing():
    hp = create_hparams()
    assert hp.n_symbols == 148 and hp.batch_size == 64 and hp.mask_padding is True
    assert len(hp.values()) == 48                               # 48 fields
    hp = create_hparams("batch_size=2,fp16_run=True,ignore_layers=[a.b,c],learning_rate=0.01,dist_url=tcp://x:1")
    assert hp.batch_size == 2 and hp.fp16_run is True and hp.ignore_layers == ['a.b', 'c']
    assert hp.learning_rate == 0.01 and hp.dist_url == "tcp://x:1"
    with pytest.raises(ValueError):
        create_hparams("no_such_field=1")


def test_state_dict_surface():
    torch.manual_seed(0)
    m = Tacotron2(create_hparams())
    sd = m.state_dict()
    params = dict(m.named_parameters())
    assert len(params) == 60 and len(sd) == 84
    assert sum(p.numel() for p in params.values()) == 28193153
    for k in ('embedding.weight', 'encoder.convolutions.2.0.conv.bias', 'encoder.lstm.weight_hh_l0_reverse',
              'decoder.prenet.layers.1.linear_layer.weight', 'decoder.attention_rnn.bias_hh',
              'decoder.attention_layer.location_layer.location_conv.conv.weight',
              'decoder.attention_layer.v.linear_layer.weight', 'decoder.gate_layer.linear_layer.bias',
              'postnet.convolutions.4.1.num_batches_tracked'):
        assert k in sd, k
    assert m.decoder.attention_layer.score_mask_value == -float("inf")
    assert hasattr(m, 'parse_batch') and hasattr(m, 'inference') and hasattr(m, 'parse_output')


def test_unsupported_geometry_fails_loudly():
    with pytest.raises(ValueError):
        Tacotron2(create_hparams("attention_dim=256"))                     # larger than the compiled geometry
    with pytest.raises(ValueError):
        Tacotron2(create_hparams("attention_location_kernel_size=30"))     # even: the reference's own padding breaks
    Tacotron2(create_hparams("attention_dim=64,attention_location_n_filters=8,attention_location_kernel_size=3"))
    with pytest.raises(ValueError):
        Tacotron2(create_hparams("n_frames_per_step=2"))


def test_library_exports_every_header_symbol(native_lib):
    hdr = open(os.path.join(gu.ROOT, "include", "tacotron2_amd.h")).read()
    declared = set(re.findall(r"\b(t2amd_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 34
    assert declared == set(native.SYMBOLS), declared ^ set(native.SYMBOLS)
    for name in declared:
        assert hasattr(native_lib, name), name
    assert native_lib.t2amd_abi_version() == 1
    sizes = (ctypes.c_int * 32)()
    n = native_lib.t2amd_struct_sizes(sizes, 32)
    assert [ctypes.sizeof(s) for s in native._STRUCTS] == list(sizes)[:n]


def test_shipped_library_has_no_packed_f32_instruction(native_lib):
    """the library that ships is the one built WITHOUT v_pk_{fma,mul,add}_f32 (one with a swizzled
    source on its destination pair gave wrong lanes beside another kernel's MFMA waves).  The gate lives in build() (it refuses to
    leave such a library behind); this holds the LOADED library to it as well, and the scanner to the pattern it must recognise."""
    from tacotron2_amd import build
    assert build.NO_PACKED_F32[-1] == "-packed-fp32-ops" and all(f in build.CFLAGS for f in build.NO_PACKED_F32)
    assert build._pk_hazard("v[34:35], v[10:11], v[34:35], v[66:67] op_sel:[0,1,0]")           # the proven one (round 5)
    assert build._pk_hazard("v[30:31], v[12:13], v[30:31], 0 op_sel_hi:[1,0,0]")
    assert not build._pk_hazard("v[34:35], v[10:11], v[34:35], v[66:67]")                        # same pair, no cross-half read
    assert not build._pk_hazard("v[34:35], v[10:11], v[36:37], v[66:67] op_sel:[0,1,0]")        # swizzle on another pair
    if not os.path.exists(build.OBJDUMP):
        pytest.skip("no llvm-objdump on this box")
    n, hits = build.scan_packed_f32(native.LIB_PATH)
    assert (n, hits) == (0, []), (n, hits[:3])


def test_no_cpu_fallback(native_lib):
    """CPU tensors must be refused: the product has one compute path."""
    m = Tacotron2(create_hparams(gu.TINY_HP))
    batch = gu.make_train_batch([5, 3], [7, 6], 80, 1)
    x = (batch[0], batch[1], batch[2], 5, batch[4])
    with pytest.raises(native.NativeError):
        m(x)
    with pytest.raises(native.NativeError):
        m.inference(batch[0][:1])


def test_argument_validation_errors_are_reported(native_lib):
    d = native.GemmDesc()
    rc = native_lib.t2amd_gemm_f32(ctypes.byref(d), None)
    assert rc == 1 and b"null operand" in native_lib.t2amd_last_error()


@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16x3"])
@pytest.mark.parametrize("fold", [0, 1])
@pytest.mark.parametrize("hpstr,in_lens,out_lens", [(gu.TINY_HP, [12, 9, 5], [20, 16, 11]), ("", [17, 11], [30, 23])])
def test_host_plumbing_validate_only(native_lib, hpstr, in_lens, out_lens, fold, precision):
    """Every host-side check / loop of forward, backward and inference runs (kernels skipped):
    catches shape, stride, alignment and pointer-plumbing errors without a GPU.  fold = 1: the BPTT loop hands the
    step's LSTM cell backwards to the attention-backward call (t2amd_attn_bwd.cell_q / cell_x), whose descriptor checks
    run here."""
    native.set_validate_only(True)
    start = native.get_bptt_cell_fold()
    native.set_bptt_cell_fold(fold)
    try:
        hp = create_hparams((hpstr + "," if hpstr else "") + "max_decoder_steps=6")
        m = Tacotron2(hp)
        m.precision = precision          # (all three operand modes of the time loops: f32 slabs, bf16 copies, split-bf16 images)
        batch = gu.make_train_batch(in_lens, out_lens, 80, 1)
        x, y = m.parse_batch(batch)
        out = m(x)
        assert [tuple(o.shape) for o in out] == [(len(in_lens), 80, max(out_lens))] * 2 + \
            [(len(in_lens), max(out_lens)), (len(in_lens), max(out_lens), max(in_lens))]
        (out[0].sum() + out[1].sum() + out[2].sum()).backward()
        assert all(p.grad is not None and p.grad.shape == p.shape for p in m.parameters())
        m.eval()
        o = m.inference(batch[0][:1, :in_lens[0]])
        assert o[2].dim() == 3 and o[2].shape[2] == 1
        o = m.inference(batch[0], batch[1])
        assert o[0].shape[0] == len(in_lens)
    finally:
        native.set_validate_only(False)
        native.set_bptt_cell_fold(start)


@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16x3"])
def test_batched_inference_host_plumbing_above_eight_utterances(native_lib, precision):
    """B > 8 takes the MFMA-tile route of the free-running loop (t2amd_decoder_infer_steps_f32): f32 operands, bf16 copies, or --
    round 6 -- split-bf16 images for the two LSTM steps ('bf16x3').  Validate-only: every step's descriptors are built and checked."""
    native.set_validate_only(True)
    try:
        hp = create_hparams("max_decoder_steps=5")
        m = Tacotron2(hp).eval()
        m.precision = precision
        lens = [19, 17, 15, 14, 12, 11, 9, 8, 6, 5, 3]
        text = gu.make_text(lens, 3)
        o = m.inference(text, torch.tensor(lens))
        assert o[0].shape[0] == len(lens) and m.last_decode_path.startswith("launch chain")
    finally:
        native.set_validate_only(False)


def test_attention_workspace_sizes_and_cell_descriptor_checks(native_lib):
    """t2amd_attn_{fwd,bwd}_ws_floats cover every form of the attention step (partial energies / dw slab, token blocks,
    granule blocks at 8-byte-aligned offsets), and the folded-cell descriptors of t2amd_attn_bwd are validated on the
    host: cell_q must be THIS step's [B][Hq] cell and its dh[1] must describe the dh_out slabs (validate-only: no GPU)."""
    import torch
    S = native.ATT_SLICES
    for B, Ti in ((1, 1), (3, 37), (64, 187), (256, 203), (5, 511)):
        f, w = native.attn_fwd_ws_floats(B, Ti), native.attn_bwd_ws_floats(B, Ti)
        assert f % 4 == 0 and f >= S * B * Ti + 2 * S * B * Ti
        goff = (B * Ti + 12 * B + 1) // 2 * 2
        assert w == goff + 2 * B * (Ti + S)
    native.set_validate_only(True)
    try:
        B, Ti, E, Hq = 3, 20, 64, 64
        z = lambda *s: torch.zeros(*s)
        dh = z(S, B, Hq)
        args = ([z(B, E)], z(B, E), None, z(B, 128), z(128, Hq), z(128 * 62), z(128), z(B, Ti, 128), z(B, Ti, E),
                torch.full((B,), Ti, dtype=torch.int32), z(B, Ti), None, z(B, Ti), z(S, B, 2, Ti), z(B, Ti), z(B, Ti, 128),
                z(B, 128, 62), z(B, 128), z(B, 128), dh, z(native.attn_bwd_ws_floats(B, Ti)))
        cell = lambda H, d1: native.lstm_bwd_desc(B, H, [z(B, H), d1, None], z(B, 4 * H), None, z(B, H), None, 1.0, z(B, H),
                                                  z(B, 4 * H))
        good = cell(Hq, (dh[0], S, dh.stride(0)))
        native.attention_step_bwd(*args, cell_q=good, cell_x=cell(128, None))            # accepted
        with pytest.raises(native.NativeError):                                          # dh[1] is not the dh_out slabs
            native.attention_step_bwd(*args, cell_q=cell(Hq, z(B, Hq)))
        with pytest.raises(native.NativeError):                                          # not this step's [B][Hq] cell
            native.attention_step_bwd(*args, cell_q=cell(128, (z(S, B, 128)[0], S, B * 128)))
        with pytest.raises(native.NativeError):                                          # cell_x without cell_q
            native.attention_step_bwd(*args, cell_x=good)
    finally:
        native.set_validate_only(False)


def test_splitk_policy_matches_library_tile_rule():
    """engine._choose_splitk sizes split-K against the tile edge the library reports (t2amd_gemm_tile_size): for the
    hot weight-gradient shapes the launch must fill whole rounds of 256 workgroups when it runs 256-tiles, and the
    query itself is pure host logic (no GPU)."""
    import math
    from tacotron2_amd import native as nv, engine
    nv.load()
    # (M, N, K, precision, a_km, b_kn): decoder / attention LSTM wgrads, a postnet conv wgrad, the gate projection
    shapes = [(4096, 2560, 55000, 2, True, True), (4096, 1792, 55000, 2, True, True), (512, 2560, 55000, 2, True, True),
              (4096, 2560, 55000, 1, True, True), (81, 1536, 55000, 2, True, True), (55000, 512, 2560, 2, False, False)]
    for M, N, K, prec, a_km, b_kn in shapes:
        sk = engine._choose_splitk(M, N, K, 1, prec, a_km, b_kn)
        assert 1 <= sk <= 64 and (sk == 1 or K // sk >= 256)
        tile = nv.gemm_tile_size(M, N, prec, sk, a_km, b_kn)
        assert tile in (128, 256)
        wgs = math.ceil(M / tile) * math.ceil(N / tile) * sk
        if tile == 256:
            assert wgs >= 192
            rounds = wgs / 256.0
            assert rounds / math.ceil(rounds) >= 0.8, (M, N, sk, wgs)
    # split-bf16 keeps both hi and lo images: only the weight-gradient layout may use the large tile
    assert nv.gemm_tile_size(4096, 2560, 1, 3, False, True) == 128
    assert nv.gemm_tile_size(4096, 2560, 0, 3, True, True) == 128



def test_torch_library_ops_are_registered(native_lib):
    """SURVEY 8b 'what the native layer exports': the loop-level entry points are TORCH_LIBRARY ops over the same C ABI
    (csrc/torch_ops.cpp).  Without a GPU: the registration library builds and loads, every op exists with the mutating
    schema, agrees with the C ABI's version, and the CPU key refuses loudly (there is no CPU compute path)."""
    import torch
    from tacotron2_amd import native
    ops = native.torch_ops()
    assert ops is not None, "lib/libtacotron2_amd_torch.so missing: python -m tacotron2_amd.build"
    assert ops.abi_version() == native.load().t2amd_abi_version()
    names = ("encoder_lstm_fwd", "encoder_lstm_bwd", "decoder_train_fwd", "decoder_train_bwd", "decoder_infer_steps",
             "decoder_infer_persistent", "conv_gemm", "conv_gemm16", "wgrad_gemm16")
    for n in names:
        schema = str(getattr(ops, n).default._schema)
        assert "Tensor[] reads" in schema and "Tensor(a!)[] writes" in schema, schema
    desc = torch.frombuffer(native.DecTrain(), dtype=torch.uint8)
    with pytest.raises(RuntimeError, match="no CPU compute path"):
        ops.decoder_train_fwd(desc, [torch.zeros(2)], [torch.zeros(2)])
    # validate-only (CPU test mode) keeps the ctypes route: the dispatcher route is for device tensors
    assert native._via_ops("decoder_train_fwd", [native.DecTrain()], [torch.zeros(2)], [torch.zeros(2)]) is False


def test_bucket_launch_order_around_the_backward_loop(native_lib, monkeypatch):
    """Data parallel (reference distributed.py:126-173): the engine launches the postnet bucket's all-reduce in front of the
    decoder BPTT (it travels while the loop runs), the decoder bucket behind it, the encoder bucket last.  Host logic only
    (validate-only).  (Until round 5 there was a second order for the opt-in persistent BPTT launch, removed in round 6.)"""
    native.set_validate_only(True)
    events = []

    class FakeSync(object):
        def __init__(self, model):
            self.shapes = {k: tuple(p.shape) for k, p in model.named_parameters()}

        def start(self, device=None, dtype=torch.float32):
            events.append("start")

        def out(self, name, shape):
            return torch.zeros(*shape)

        def bucket_ready(self, bucket, grads=None):
            events.append("bucket:" + bucket)

        def finish(self):
            events.append("finish")

    real = native.decoder_train_bwd_loop
    monkeypatch.setattr(native, "decoder_train_bwd_loop", lambda *a, **k: (events.append("bptt"), real(*a, **k))[1])
    try:
        for precision in ("bf16", "fp32", "bf16x3"):
            del events[:]
            m = Tacotron2(create_hparams("max_decoder_steps=6"))
            m.precision = precision
            m._grad_sync = FakeSync(m)
            x, y = m.parse_batch(gu.make_train_batch([17, 11, 9, 9, 8, 5, 3, 2], [30, 23, 12, 25, 7, 19, 30, 4], 80, 1))
            out = m(x)
            (out[0].sum() + out[1].sum() + out[2].sum()).backward()
            assert events == ["start", "bucket:postnet", "bptt", "bucket:decoder", "bucket:encoder", "finish"], events
            assert m.last_train_decoder_bwd_path == "launch chain"
    finally:
        native.set_validate_only(False)


def test_train_driver_precision_env(native_lib, monkeypatch):
    """train.load_model: T2AMD_PRECISION picks the compute mode (fp32 / bf16 / bf16x3), anything else is refused; without it
    fp16_run alone decides, as in the reference (train.py:73-81)."""
    from tacotron2_amd import train as tr
    native.set_validate_only(True)
    try:
        monkeypatch.delenv("T2AMD_PRECISION", raising=False)
        assert tr.load_model(create_hparams(gu.TINY_HP)).precision == "fp32"
        assert tr.load_model(create_hparams(gu.TINY_HP + ",fp16_run=True")).precision == "bf16"
        for prec in ("fp32", "bf16", "bf16x3"):
            monkeypatch.setenv("T2AMD_PRECISION", prec)
            assert tr.load_model(create_hparams(gu.TINY_HP + ",fp16_run=True")).precision == prec
        monkeypatch.setenv("T2AMD_PRECISION", "fp16")
        with pytest.raises(native.NativeError, match="T2AMD_PRECISION"):
            tr.load_model(create_hparams(gu.TINY_HP))
    finally:
        native.set_validate_only(False)


def test_small_batch_boundary_setting(native_lib):
    """t2amd_set_small_batch_max: -1 = by operand mode (3 rows with bf16 operands, 4 otherwise), 0 .. 8 = that many rows; anything
    else is refused.  Pure host state (no GPU)."""
    old = native.small_batch_max_setting()
    try:
        native.set_small_batch_max(-1)
        assert (native.small_batch_max(0), native.small_batch_max(1), native.small_batch_max(3)) == (4, 3, 4)
        assert native.small_batch_max_setting() == -1
        for n in (0, 5, 8):
            native.set_small_batch_max(n)
            assert native.small_batch_max(0) == n == native.small_batch_max(1) and native.small_batch_max_setting() == n
        for bad in (-2, 9):
            with pytest.raises(native.NativeError, match="small_batch_max"):
                native.set_small_batch_max(bad)
        # which kernels a batch runs on: above the boundary the tiles -- except bf16 operands on widths the bf16 tiles refuse
        native.set_small_batch_max(-1)
        full = (512, 1024, 1024, 256)
        assert [native.dec_infer_uses_tiles(B, 1, *full) for B in (3, 4, 8, 9)] == [False, True, True, True]
        assert [native.dec_infer_uses_tiles(B, 0, *full) for B in (4, 5, 9)] == [False, True, True]
        assert [native.dec_infer_uses_tiles(B, 3, *full) for B in (4, 5)] == [False, True]
        tiny = (64, 64, 64, 64)
        assert [native.dec_infer_uses_tiles(B, 1, *tiny) for B in (4, 8, 9)] == [False, False, True]
        assert [native.dec_infer_uses_tiles(B, 0, *tiny) for B in (4, 5)] == [False, True]
    finally:
        native.set_small_batch_max(old)
