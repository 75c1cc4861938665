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
