"""Kernels of the path must give the same bits whatever else runs on the GPU beside them.

Round 5 found one that did not: the f32 form of the separate-launch attention backward (K_b1,
`attn_bwd_dw_kernel<false>`) accumulated with `v_pk_fma_f32` instructions whose destination pair was also a source pair read with a
cross-half `op_sel`; alone on the GPU it was right in every run of four rounds, beside a GEMM (another process's, or one on a side
stream of this process) 16 lanes of one row in two came out wrong in 60 % of the launches -- what `test_zz9_dp_gpu.py` had seen once
in round 2 and again in round 5.  This test is the in-process reproducer: the attention backward step, repeated from the same
inputs while a side stream multiplies matrices, in both launch forms and both operand precisions; every repeat must equal the first
bit for bit.  (tools/scan_pk_overlap.py finds the instruction pattern in the ISA; tools/stress_lds_poison.py is the whole-step
version.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


@pytest.fixture(scope="module")
def nv():
    from tacotron2_amd import native
    native.load()
    return native


@pytest.mark.parametrize("fused", [0, 1])
@pytest.mark.parametrize("m16", [False, True])
@pytest.mark.parametrize("B,Ti,reps", [(3, 23, 150), (64, 177, 25)])
def test_attention_backward_is_bit_stable_beside_a_gemm(nv, fused, m16, B, Ti, reps):
    E, Hq = 512, 1024
    g = torch.Generator().manual_seed(5)
    rnd = lambda *s: torch.randn(*s, generator=g).to(DEV)                         # noqa: E731
    mem, pm = rnd(B, Ti, E), rnd(B, Ti, 128)
    Wq, U, v = rnd(128, Hq) * 0.05, rnd(128 * 62) * 0.1, rnd(128)
    lens = torch.tensor(([Ti, max(1, Ti - 6), max(1, Ti // 2)] * B)[:B], dtype=torch.int32, device=DEV)
    w, wprev = torch.softmax(rnd(B, Ti), 1), torch.softmax(rnd(B, Ti), 1)
    cum = torch.rand(B, Ti, generator=g).to(DEV)
    q, dctx, dwx = rnd(B, 128), rnd(B, E), rnd(B, Ti)
    ws0 = torch.zeros(nv.attn_bwd_ws_floats(B, Ti), device=DEV)
    dwin0, dcum0 = rnd(4, B, 2, Ti), rnd(B, Ti)
    mem16 = mem.bfloat16() if m16 else None
    side = torch.cuda.Stream()
    ha = torch.randn(2048, 2048, device=DEV, dtype=torch.bfloat16)
    hb = torch.randn(2048, 2048, device=DEV, dtype=torch.bfloat16)
    names = ("dctx_total", "dwin", "dcum", "d_pm", "dU", "dv", "dq", "dh", "dw")
    saved = nv.get_attn_bwd_fused()

    def once(disturb):
        ws, dwin, dcum = ws0.clone(), dwin0.clone(), dcum0.clone()
        d_pm, dU, dv_ = torch.zeros(B, Ti, 128, device=DEV), torch.zeros(B, 128, 62, device=DEV), torch.zeros(B, 128, device=DEV)
        dq, dh, tot = torch.zeros(B, 128, device=DEV), torch.zeros(4, B, Hq, device=DEV), torch.zeros(B, E, device=DEV)
        if disturb:
            with torch.cuda.stream(side):
                for _ in range(20):
                    ha @ hb
        nv.attention_step_bwd([dctx], tot, dwx, q, Wq, U, v, pm, mem, lens, w, wprev, cum, dwin, dcum, d_pm, dU, dv_, dq, dh, ws,
                              bf16=m16, memory16=mem16)
        out = [t.clone() for t in (tot, dwin, dcum, d_pm, dU, dv_, dq, dh, ws[:B * Ti])]
        torch.cuda.synchronize()
        return out

    nv.set_attn_bwd_fused(fused)
    try:
        first = once(False)
        bad = {}
        for r in range(reps):
            for n, a, b in zip(names, first, once(True)):
                if not torch.equal(a, b):
                    bad.setdefault(n, []).append(r)
        assert not bad, {k: (len(v_), v_[:5]) for k, v_ in bad.items()}
    finally:
        nv.set_attn_bwd_fused(saved)


# ----------------------------------------------------------------------------------------------------------------------------
# Round 6: the two translation units that carried most of the 117 packed-f32 sites -- the B = 1 persistent
# decoder (decode_persist.hip) and the small-batch matrix-vector kernels (gemv.hip) -- held to the same standard.  The library is
# built without packed-f32 instructions now (build.py refuses anything else); these are the co-run checks that nothing else in
# those kernels depends on having the SIMD to itself.  The disturbance: (a) a library GEMM on a side stream (what the round-5
# reproducer used), (b) `t2amd_debug_mfma_spin_` -- MFMA waves with no LDS and ~20 registers, the only foreign work that fits on a
# SIMD beside a persistent workgroup that holds all of its CU's LDS.
# ----------------------------------------------------------------------------------------------------------------------------
class _Disturb:
    def __init__(self, nv):
        import ctypes as C
        self.C = C
        self.side = torch.cuda.Stream()
        self.ha = torch.randn(2048, 2048, device=DEV, dtype=torch.bfloat16)
        self.hb = torch.randn(2048, 2048, device=DEV, dtype=torch.bfloat16)
        self.sink = torch.zeros(4, device=DEV)
        self.spin = nv.load().t2amd_debug_mfma_spin_
        self.spin.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        self.spin.restype = C.c_int

    def __call__(self, kind, amount):
        if kind == "gemm":
            with torch.cuda.stream(self.side):
                for _ in range(amount):
                    self.ha @ self.hb
        else:                       # 512 workgroups = 2 per CU = 2 foreign waves per SIMD, ~60 us per launch
            assert self.spin(512, 2000, amount, self.C.c_void_p(self.sink.data_ptr()), self.C.c_void_p(self.side.cuda_stream)) == 0


def _infer_model(precision, steps):
    import golden_util as gu
    from tacotron2_amd.model import Tacotron2
    hp = gu.make_hparams("max_decoder_steps=%d" % steps)
    hp.gate_threshold = 2.0                                       # forced length: every frame is compared
    sd = gu.build_state_dict(hp, 321, perturb_bn=True)
    m = Tacotron2(hp)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    m.precision = precision
    return hp, m


@pytest.mark.parametrize("kind", ["gemm", "mfma_spin"])
@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_persistent_b1_decoder_is_bit_stable_beside_mfma_waves(nv, precision, kind):
    """decode_persist.hip: the whole of Decoder.inference (reference model.py:435-449) as one launch, repeated from the same text and
    dropout stream while foreign MFMA waves share the chip: every repeat equals the undisturbed run bit for bit, and every repeat
    really ran the persistent kernel (a hand-off that timed out beside the foreign work would fall back to the launch chain, whose
    summation order differs -- that is reported by the engine and counted here, not compared)."""
    import golden_util as gu
    from oracle import tacotron2_oracle as orc
    from tacotron2_amd import engine
    steps, Ti, reps = 96, 100, 10
    hp, model = _infer_model(precision, steps)
    text = gu.make_text([Ti], 55).to(DEV)
    keep = orc.draw_masks_infer(hp, 1, steps, torch.Generator().manual_seed(4)).to(DEV)
    dist = _Disturb(nv)
    old = engine.PERSISTENT_DECODE
    engine.PERSISTENT_DECODE = True

    def once(disturb):
        model.dropout_masks = dict(prenet_infer=keep)
        model._persist_backoff = 0                                # a timeout beside the foreign work must not park the next repeats
        if disturb:
            dist(kind, 12 if kind == "gemm" else 40)
        with torch.no_grad():
            out = model.inference(text)
        torch.cuda.synchronize()
        return [o.float().clone() for o in out], model.last_decode_path

    try:
        first, path = once(False)
        assert path == 'persistent', path
        bad, persistent_runs = {}, 0
        for r in range(reps):
            out, path = once(True)
            if path != 'persistent':
                continue
            persistent_runs += 1
            for n, a, b in zip(("mel", "mel_post", "gate", "align"), first, out):
                if not torch.equal(a, b):
                    bad.setdefault(n, []).append((r, int((a != b).sum())))
        assert persistent_runs >= reps // 2, persistent_runs
        assert not bad, bad
    finally:
        engine.PERSISTENT_DECODE = old


@pytest.mark.parametrize("kind", ["gemm", "mfma_spin"])
@pytest.mark.parametrize("precision,B", [("bf16", 1), ("bf16", 5), ("fp32", 8)])
def test_small_batch_launch_chain_is_bit_stable_beside_mfma_waves(nv, precision, B, kind):
    """gemv.hip + the small-batch attention / loop kernels (loops.hip): B <= 8 inference on the matrix-vector launch chain (the route B = 4 took and B = 1
    takes when the persistent kernel is off; the boundary is set to 8 rows here), repeated beside foreign MFMA waves."""
    import golden_util as gu
    from oracle import tacotron2_oracle as orc
    from tacotron2_amd import engine
    steps, reps = 48, 8
    hp, model = _infer_model(precision, steps)
    in_lens = [61, 55, 40, 33, 23, 20, 17, 15][:B]
    text = gu.make_text(in_lens, 9).to(DEV)
    lens = torch.tensor(in_lens, device=DEV)
    keep = orc.draw_masks_infer(hp, B, steps, torch.Generator().manual_seed(12)).to(DEV)
    dist = _Disturb(nv)
    old = (engine.PERSISTENT_DECODE, engine.SMALL_BATCH_PERSISTENT)
    engine.PERSISTENT_DECODE, engine.SMALL_BATCH_PERSISTENT = False, 1
    old_boundary = nv.small_batch_max_setting()
    nv.set_small_batch_max(8)                 # the matrix-vector kernels up to 8 rows (the default boundary is 3 / 4 since round 6)

    def once(disturb):
        model.dropout_masks = dict(prenet_infer=keep)
        if disturb:
            dist(kind, 12 if kind == "gemm" else 40)
        with torch.no_grad():
            out = model.inference(text, lens) if B > 1 else model.inference(text)
        torch.cuda.synchronize()
        return [o.float().clone() for o in out], model.last_decode_path

    try:
        first, path = once(False)
        assert path.startswith('launch chain'), path
        bad = {}
        for r in range(reps):
            out, path = once(True)
            assert path.startswith('launch chain'), path
            for n, a, b in zip(("mel", "mel_post", "gate", "align"), first, out):
                if not torch.equal(a, b):
                    bad.setdefault(n, []).append((r, int((a != b).sum())))
        assert not bad, bad
    finally:
        nv.set_small_batch_max(old_boundary)
        engine.PERSISTENT_DECODE, engine.SMALL_BATCH_PERSISTENT = old


@pytest.mark.parametrize("B", [1, 3, 8])
def test_small_batch_gemv_kernels_are_bit_stable_beside_mfma_waves(nv, B):
    """The matrix-vector LSTM step and the small linear of gemv.hip on their own (71 of round 5's 117 sites), at the decoder's
    real widths, 200 launches beside the spin kernel."""
    H, widths = 1024, (256, 512, 1024)
    g = torch.Generator().manual_seed(7)
    rnd = lambda *s: torch.randn(*s, generator=g).to(DEV)                         # noqa: E731
    xs = [rnd(B, w) for w in widths]
    W, gin, bias, c_prev = rnd(4 * H, sum(widths)) * 0.03, rnd(B, 4 * H), rnd(4 * H), rnd(B, H)
    Wl, bl, X = rnd(81, 1536), rnd(81), rnd(B, 1536)
    dist = _Disturb(nv)

    def once(disturb):
        gates, c, h = (torch.empty(B, 4 * H, device=DEV), torch.empty(B, H, device=DEV), torch.empty(B, H, device=DEV))
        Y = torch.empty(B, 81, device=DEV)
        if disturb:
            dist("mfma_spin", 4)
        for _ in range(25):
            nv.lstm_step_fwd(xs, list(widths), W, H, B, gates, c, h, gin=gin, bias=bias, c_prev=c_prev, small=True)
            nv.linear_small(X, Wl, Y, bias=bl)
        torch.cuda.synchronize()
        return gates, c, h, Y

    first = once(False)
    for r in range(8):
        for n, a, b in zip(("gates", "c", "h", "Y"), first, once(True)):
            assert torch.equal(a, b), (r, n, int((a != b).sum()))
