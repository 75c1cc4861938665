"""Per-kernel parity tests: every C-ABI kernel against a plain PyTorch fp32 CPU reference of the
same operation, on seeded inputs.  Tolerances (stated per test) are fp32 round-off class: the
kernels compute in exact f32 (MFMA f32 forms are a k-ordered fmaf chain) but in a different
summation order than ATen."""
import pytest
import torch
import torch.nn.functional as F

from oracle import tacotron2_oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def nv(native_lib):
    from tacotron2_amd import native
    assert torch.cuda.is_available(), "GPU tests need a device"
    return native


def G(seed):
    return torch.Generator().manual_seed(seed)


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=G(seed)) * scale


def dv(t):
    return t.to(DEV)


def err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return (a - b).abs().max().item() / max(1.0, b.abs().max().item())


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(130, 81, 50), (300, 257, 96), (64, 4096, 256), (7, 5, 3), (256, 128, 1024)])
@pytest.mark.parametrize("a_km,b_kn", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_layouts(nv, M, N, K, a_km, b_kn):
    A = rnd(M, K, seed=1)
    B = rnd(K, N, seed=2)
    ref = A @ B
    Ad = dv(A.t().contiguous()) if a_km else dv(A)
    Bd = dv(B) if b_kn else dv(B.t().contiguous())
    C = torch.full((M, N), float('nan'), device=DEV)
    nv.gemm(C, Ad, Bd, a_km=a_km, b_kn=b_kn)
    assert err(C, ref) < 2e-5


@pytest.mark.parametrize("M,N,K,sk", [(3600, 3590, 100, 1), (1000, 900, 2100, 12), (2048, 1792, 96, 4), (300, 257, 96, 1)])
@pytest.mark.parametrize("a_km,b_kn", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_plain_bf16_tiles(nv, M, N, K, sk, a_km, b_kn):
    """precision=2 on both tile sizes (256 x 256 x 32 / 8 waves when the launch has >= 192 such tiles, else
    128 x 128): equals the f32 product of the bf16-ROUNDED operands to summation-order accuracy; ragged M, N, K
    edges, every layout, split-K partial slabs."""
    A = rnd(M, K, seed=11) * (1.0 + torch.arange(M).float().unsqueeze(1) / M)
    B = rnd(K, N, seed=12) * (0.5 + torch.arange(N).float().unsqueeze(0) / N)
    Ar, Br = A.bfloat16().double(), B.bfloat16().double()
    ref = Ar @ Br
    scale = Ar.abs() @ Br.abs()
    Ad = dv(A.t().contiguous()) if a_km else dv(A)
    Bd = dv(B) if b_kn else dv(B.t().contiguous())
    expect = 256 if (M >= 512 and N >= 512 and -(-M // 256) * -(-N // 256) * sk >= 192) else 128
    assert nv.gemm_tile_size(M, N, 2, sk, a_km, b_kn) == expect
    if sk == 1:
        C = torch.full((M, N), float('nan'), device=DEV)
        nv.gemm(C, Ad, Bd, a_km=a_km, b_kn=b_kn, fast=2)
        out = C.cpu().double()
    else:
        part = torch.full((sk, M * N), float('nan'), device=DEV)
        nv.gemm(part[0].view(M, N), Ad, Bd, a_km=a_km, b_kn=b_kn, fast=2, splitk=sk, partials=part)
        out = part.cpu().double().sum(0).view(M, N)
    rel = ((out - ref).abs() / scale).max().item()
    assert rel < 2e-6, rel


@pytest.mark.parametrize("M,N,K,sk", [(2048, 1792, 2100, 4), (1000, 900, 4100, 12)])
def test_gemm_split_bf16_big_tile_wgrad(nv, M, N, K, sk):
    """precision=1 on 256 x 256 tiles: only the weight-gradient layout (both operands M/N-contiguous: unpadded LDS
    images, 128 KB) qualifies; split-K partial slabs, ragged edges."""
    A = rnd(M, K, seed=21) * (1.0 + torch.arange(M).float().unsqueeze(1) / M)
    B = rnd(K, N, seed=22) * (0.5 + torch.arange(N).float().unsqueeze(0) / N)
    ref = A.double() @ B.double()
    scale = A.abs().double() @ B.abs().double()
    assert nv.gemm_tile_size(M, N, 1, sk, True, True) == 256
    assert nv.gemm_tile_size(M, N, 1, sk, False, True) == 128
    part = torch.full((sk, M * N), float('nan'), device=DEV)
    nv.gemm(part[0].view(M, N), dv(A.t().contiguous()), dv(B), a_km=True, b_kn=True, fast=1, splitk=sk, partials=part)
    out = part.cpu().double().sum(0).view(M, N)
    rel = ((out - ref).abs() / scale).max().item()
    assert rel < 2.5e-5, rel


@pytest.mark.parametrize("M,N,K", [(130, 81, 50), (300, 257, 96), (64, 4096, 256), (7, 5, 3), (256, 128, 1024), (512, 384, 4000)])
@pytest.mark.parametrize("a_km,b_kn", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_split_bf16(nv, M, N, K, a_km, b_kn):
    """precision=1 (Ah.Bh + Ah.Bl + Al.Bh on the bf16 MFMA): error class 2^-17 per product, far from plain
    bf16 (2^-9); asymmetric operands so that a transposed fragment layout cannot pass."""
    A = rnd(M, K, seed=1) * (1.0 + torch.arange(M).float().unsqueeze(1) / M)
    B = rnd(K, N, seed=2) * (0.5 + torch.arange(N).float().unsqueeze(0) / N)
    ref = A.double() @ B.double()
    Ad = dv(A.t().contiguous()) if a_km else dv(A)
    Bd = dv(B) if b_kn else dv(B.t().contiguous())
    C = torch.full((M, N), float('nan'), device=DEV)
    nv.gemm(C, Ad, Bd, a_km=a_km, b_kn=b_kn, fast=True)
    scale = (A.abs().double() @ B.abs().double())          # sum |a||b| bounds the rounding error
    rel = ((C.cpu().double() - ref).abs() / scale).max().item()
    assert rel < 2.5e-5, rel                                 # measured ~3e-6; plain bf16 would be ~4e-3
    C32 = torch.empty(M, N, device=DEV)
    nv.gemm(C32, Ad, Bd, a_km=a_km, b_kn=b_kn)
    rel32 = ((C32.cpu().double() - ref).abs() / scale).max().item()
    assert rel32 < 2e-6


def test_gemm_epilogue_and_views(nv):
    M, N, K = 200, 96, 72
    A, W, bias, C0 = rnd(M, K, seed=3), rnd(N + 10, K + 8, seed=4), rnd(N, seed=5), rnd(M, N, seed=6)
    keep = (torch.rand(M, N, generator=G(7)) >= 0.5).to(torch.uint8)
    Wv = W[5:5 + N, 4:4 + K]                      # strided view, 16-byte aligned offset
    ref = torch.relu(A @ Wv.t() + bias) * keep * 2.0
    C = torch.empty(M, N, device=DEV)
    nv.gemm(C, dv(A), dv(W)[5:5 + N, 4:4 + K], bias=dv(bias), act=1, keep=dv(keep), keep_scale=2.0)
    assert err(C, ref) < 2e-5
    C = dv(C0.clone())
    nv.gemm(C, dv(A), dv(W)[5:5 + N, 4:4 + K], bias=dv(bias), accumulate=True)
    assert err(C, C0 + A @ Wv.t() + bias) < 2e-5
    # unaligned view -> scalar path
    Wv2 = W[1:1 + N, 3:3 + K]
    C = torch.empty(M, N, device=DEV)
    nv.gemm(C, dv(A), dv(W)[1:1 + N, 3:3 + K])
    assert err(C, A @ Wv2.t()) < 2e-5


def test_gemm_batched_and_splitk(nv):
    Bn, M, N, K = 5, 37, 64, 90
    A, Bm = rnd(Bn, K, M, seed=8), rnd(K, Bn, N, seed=9)        # A^T per batch; B strided like DCTX[:, b, :]
    ref = torch.stack([A[b].t() @ Bm[:, b, :] for b in range(Bn)])
    Ad, Bd = dv(A), dv(Bm)
    C = torch.empty(Bn, M, N, device=DEV)
    nv.gemm(C[0], Ad[0], Bd[:, 0, :], a_km=True, b_kn=True, batch=Bn, strides=(K * M, N, M * N))
    assert err(C, ref) < 2e-5
    M, N, K = 81, 200, 5000
    A2, B2 = rnd(K, M, seed=10), rnd(K, N, seed=11)
    part = torch.empty(8, M * N, device=DEV)
    Cs = torch.empty(M, N, device=DEV)
    nv.gemm(part[0].view(M, N), dv(A2), dv(B2), a_km=True, b_kn=True, splitk=8, partials=part)
    nv.splitk_reduce(part, 8, Cs)
    assert err(Cs, A2.t() @ B2) < 5e-5


@pytest.mark.parametrize("Bn,T,Ci,Co,k", [(3, 20, 80, 128, 5), (2, 13, 128, 80, 5), (1, 4, 16, 16, 5)])
def test_conv_forward_dgrad_wgrad(nv, Bn, T, Ci, Co, k):
    """implicit-GEMM conv1d (channel-last) vs F.conv1d and its autograd."""
    x = rnd(Bn, Ci, T, seed=12).requires_grad_(True)
    W = rnd(Co, Ci, k, seed=13, scale=0.2).requires_grad_(True)
    bias = rnd(Co, seed=14)
    y = F.conv1d(x, W, bias, padding=k // 2)
    gy = rnd(Bn, Co, T, seed=15)
    y.backward(gy)
    pad = k // 2
    rows = Bn * T
    x_cl = dv(x.detach().permute(0, 2, 1).contiguous().view(rows, Ci))
    gy_cl = dv(gy.permute(0, 2, 1).contiguous().view(rows, Co))
    Wd = dv(W.detach())
    Wp = torch.empty(Co, k * Ci, device=DEV)
    nv.transpose(Wp.view(Co, k, Ci)[0], Wd[0], batch=Co, sstride=Ci * k, dstride=k * Ci)
    assert torch.equal(Wp.view(Co, k, Ci).cpu(), W.detach().permute(0, 2, 1))
    yk = torch.empty(rows, Co, device=DEV)
    nv.gemm(yk, x_cl, Wp, bias=dv(bias), convA=(T, Ci, pad, 1))
    assert err(yk.view(Bn, T, Co).permute(0, 2, 1), y) < 2e-5
    Wdg = torch.empty(Ci, k * Co, device=DEV)
    nv.transpose(Wdg.view(Ci * k, Co), Wd.view(Co, Ci * k))
    dx = torch.empty(rows, Ci, device=DEV)
    nv.gemm(dx, gy_cl, Wdg, convA=(T, Co, pad, -1))
    assert err(dx.view(Bn, T, Ci).permute(0, 2, 1), x.grad) < 2e-5
    part = torch.empty(2, Co * k * Ci, device=DEV)
    nv.gemm(part[0].view(Co, k * Ci), gy_cl, x_cl, a_km=True, b_kn=True, convB=(T, Ci, pad), splitk=2, partials=part)
    dW = torch.empty(Co, Ci, k, device=DEV)
    nv.splitk_reduce(part, 2, dW, perm_taps=k, perm_ci=Ci)
    assert err(dW, W.grad) < 5e-5
    # the same two gradient GEMMs on the split-bf16 kernel (what the engine's backward uses)
    dx2 = torch.empty(rows, Ci, device=DEV)
    nv.gemm(dx2, gy_cl, Wdg, convA=(T, Co, pad, -1), fast=True)      # Co % 32 != 0 falls back to exact f32
    assert err(dx2.view(Bn, T, Ci).permute(0, 2, 1), x.grad) < 5e-5
    nv.gemm(part[0].view(Co, k * Ci), gy_cl, x_cl, a_km=True, b_kn=True, convB=(T, Ci, pad), splitk=2, partials=part,
            fast=True)
    nv.splitk_reduce(part, 2, dW, perm_taps=k, perm_ci=Ci)
    assert err(dW, W.grad) < 5e-5


# ------------------------------------------------------------------------------------------------
# BatchNorm / elementwise / layout
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("act", [0, 1, 2])
def test_bn_act_forward_backward(nv, act):
    M, N = 333, 80
    x = (rnd(M, N, seed=20) * 1.5 + 0.3).requires_grad_(True)
    gamma = (0.5 + torch.rand(N, generator=G(21))).requires_grad_(True)
    beta = rnd(N, seed=22).requires_grad_(True)
    keep = (torch.rand(M, N, generator=G(23)) >= 0.5).to(torch.uint8)
    rm, rv = torch.zeros(N), torch.ones(N)
    y = F.batch_norm(x, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    y = torch.relu(y) if act == 1 else torch.tanh(y) if act == 2 else y
    y = y * keep * 2.0
    gy = rnd(M, N, seed=24)
    y.backward(gy)
    xd = dv(x.detach())
    ws = torch.empty(2 * 64 * N, dtype=torch.float64, device=DEV)
    mean, invstd = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
    rmd, rvd = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
    nv.bn_stats(xd, ws, mean, invstd, rmd, rvd, 0.1, 1e-5)
    assert err(rmd, rm) < 1e-6 and err(rvd, rv) < 1e-6
    yd = torch.empty(M, N, device=DEV)
    nv.bn_act_fwd(xd, yd, mean, invstd, dv(gamma.detach()), dv(beta.detach()), act, dv(keep), 2.0)
    assert err(yd, y) < 1e-5
    g = dv(gy.clone())
    dgamma, dbeta = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
    nv.bn_act_bwd(g, yd, xd, mean, invstd, dv(gamma.detach()), act, dv(keep), 2.0, ws, dgamma, dbeta)
    assert err(g, x.grad) < 2e-5
    assert err(dgamma, gamma.grad) < 2e-5 and err(dbeta, beta.grad) < 2e-5
    # eval mode
    inv2 = torch.empty(N, device=DEV)
    nv.bn_eval_invstd(rvd, inv2)
    nv.bn_act_fwd(xd, yd, rmd, inv2, dv(gamma.detach()), dv(beta.detach()), 0)
    ref = F.batch_norm(x.detach(), rm, rv, gamma.detach(), beta.detach(), training=False, eps=1e-5)
    assert err(yd, ref) < 1e-5


@pytest.mark.parametrize("act,N,B,T,pad,keep_f32", [(2, 512, 5, 203, 2, False), (1, 80, 3, 97, 2, True), (0, 128, 4, 64, 0, False),
                                                     (2, 64, 2, 4, 2, False)])
def test_bn_backward_writing_the_halo_image_and_the_bias_gradient(nv, act, N, B, T, pad, keep_f32):
    """t2amd_bn_act_bwd_img_f32 (round 6): stage 2 of the BatchNorm backward writes its output as the bf16 halo image the window
    products read and as the convolution's bias gradient.  Must be the same BITS as the three separate passes
    (bn_act_bwd -> cast_halo_bf16, colsum), halo rows zeroed in an image that starts out full of NaNs; with keep_f32 the f32
    slab as well."""
    M = B * T
    x = dv(rnd(M, N, seed=60) * 1.3 + 0.2)
    gamma, beta = dv(0.5 + torch.rand(N, generator=G(61))), dv(rnd(N, seed=62))
    keep = dv((torch.rand(M, N, generator=G(63)) >= 0.5).to(torch.uint8))
    gy = dv(rnd(M, N, seed=64))
    ws = torch.empty(2 * 64 * N, dtype=torch.float64, device=DEV)
    mean, invstd = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
    nv.bn_stats(x, ws, mean, invstd)
    y = torch.empty(M, N, device=DEV)
    nv.bn_act_fwd(x, y, mean, invstd, gamma, beta, act, keep, 2.0)
    # forward: y and its halo image in one pass == bn_act_fwd, then the cast pass
    y2 = torch.empty(M, N, device=DEV)
    yimg = torch.empty(B * (T + 2 * pad) + 2 * pad, N, dtype=torch.bfloat16, device=DEV)
    nv.cast_halo_bf16(y, yimg, T, pad)
    yimg2 = torch.full_like(yimg, float('nan'))
    nv.bn_act_fwd_img(x, y2, mean, invstd, gamma, beta, act, keep, 2.0, yimg2, T, pad)
    assert torch.equal(y, y2) and torch.equal(yimg.view(torch.int16), yimg2.view(torch.int16))
    # the three separate passes
    g = gy.clone()
    dgamma, dbeta, dbias = (torch.empty(N, device=DEV) for _ in range(3))
    nv.bn_act_bwd(g, y, x, mean, invstd, gamma, act, keep, 2.0, ws, dgamma, dbeta)
    nv.colsum(g, ws, dbias)
    img = torch.empty(B * (T + 2 * pad) + 2 * pad, N, dtype=torch.bfloat16, device=DEV)
    nv.cast_halo_bf16(g, img, T, pad)
    # the folded form
    g2 = gy.clone()
    dgamma2, dbeta2, dbias2 = (torch.empty(N, device=DEV) for _ in range(3))
    img2 = torch.full_like(img, float('nan'))
    nv.bn_act_bwd_img(g2, y, x, mean, invstd, gamma, act, keep, 2.0, ws, dgamma2, dbeta2, img2, T, pad, dbias2, keep_f32=keep_f32)
    torch.cuda.synchronize()
    assert torch.equal(dgamma, dgamma2) and torch.equal(dbeta, dbeta2)
    assert torch.equal(img.view(torch.int16), img2.view(torch.int16))
    assert torch.equal(dbias, dbias2)
    if keep_f32:
        assert torch.equal(g, g2)
    with pytest.raises(nv.NativeError):                           # rows that are not whole utterances
        nv.bn_act_bwd_img(g2[:-1], y[:-1], x[:-1], mean, invstd, gamma, act, keep[:-1], 2.0, ws, dgamma2, dbeta2,
                          torch.empty(((M - 1) // T) * (T + 2 * pad) + 2 * pad, N, dtype=torch.bfloat16, device=DEV), T, pad, dbias2)


@pytest.mark.parametrize("M,N", [(55680, 4096), (1031, 64), (7, 8)])
def test_column_sums_of_a_bf16_slab(nv, M, N):
    """t2amd_colsum_bf16: the bf16 mode's LSTM bias gradients come from the bf16 gate-gradient slabs (half the bytes of the f32
    ones).  Against the fp64 sum of the same bf16 values; accumulate adds to what is there."""
    x = (torch.randn(M, N + 8, generator=G(70)) * 0.3).to(torch.bfloat16).to(DEV)[:, :N]       # a row stride above N
    ws = torch.empty(2 * 64 * N, dtype=torch.float64, device=DEV)
    out = torch.empty(N, device=DEV)
    nv.colsum16(x, ws, out)
    ref = x.double().sum(0)
    assert float((out.double() - ref).abs().max()) <= 1e-6 * float(ref.abs().max()) + 1e-9
    nv.colsum16(x, ws, out, accumulate=True)
    assert float((out.double() - 2 * ref).abs().max()) <= 2e-6 * float(ref.abs().max()) + 1e-9


@pytest.mark.parametrize("act,N", [(2, 512), (1, 80), (0, 132)])
def test_bn_16_byte_forms_equal_the_scalar_kernels(nv, act, N):
    """The BatchNorm apply / backward and the column reductions take a 16-byte path when N % 4 == 0 and every base is 16-byte
    aligned (elementwise.hip, round 5).  The scalar kernels are reached here through views that start one float into a wider
    buffer; the elementwise results must be the same bits, the fp64 row reductions (different row order) agree to float rounding."""
    M, T = 2 * 1031, 1031
    lens = torch.tensor([1031, 517], dtype=torch.int32, device=DEV)

    def unaligned(t, dtype=torch.float32):                       # same values behind a base that is not 16-byte aligned
        buf = torch.zeros(t.shape[0], t.shape[1] + 4, dtype=dtype, device=DEV)
        v = buf[:, 1:1 + t.shape[1]]
        v.copy_(t)
        return v

    x = dv(rnd(M, N, seed=40) * 1.3 + 0.2)
    gamma, beta = dv(0.5 + torch.rand(N, generator=G(41))), dv(rnd(N, seed=42))
    keep = dv((torch.rand(M, N, generator=G(43)) >= 0.5).to(torch.uint8))
    gy = dv(rnd(M, N, seed=44))
    out = {}
    for form in ("vector", "scalar"):
        xs = x if form == "vector" else unaligned(x)
        ws = torch.empty(2 * 64 * N, dtype=torch.float64, device=DEV)
        mean, invstd = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
        nv.bn_stats(xs, ws, mean, invstd)
        y = torch.empty(M, N, device=DEV) if form == "vector" else unaligned(torch.empty(M, N, device=DEV))
        nv.bn_act_fwd(xs, y, mean, invstd, gamma, beta, act, keep, 2.0, lens, T)
        g = gy.clone() if form == "vector" else unaligned(gy)
        dgamma, dbeta = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
        nv.bn_act_bwd(g, y, xs, mean, invstd, gamma, act, keep, 2.0, ws, dgamma, dbeta)
        cs = torch.empty(N, device=DEV)
        nv.colsum(xs, ws, cs)
        out[form] = [t.clone().contiguous().cpu() for t in (mean, invstd, y, g, dgamma, dbeta, cs)]
    v, s = out["vector"], out["scalar"]
    for i in (0, 1, 4, 5, 6):                                     # statistics and parameter gradients: fp64 sums in another order
        assert err(v[i], s[i]) < 1e-6, i
    if torch.equal(v[0], s[0]) and torch.equal(v[1], s[1]):      # same statistics in -> same bits out of the apply
        assert torch.equal(v[2], s[2])
    else:
        assert err(v[2], s[2]) < 1e-6
    assert err(v[3], s[3]) < 1e-5
    assert torch.all(v[2][T + 517:] == 0)                         # rows past the second utterance's length are zeroed


def test_small_kernels(nv):
    ws = torch.empty(128 * 300, dtype=torch.float64, device=DEV)
    x = rnd(1000, 300, seed=30)
    out = torch.empty(300, device=DEV)
    nv.colsum(dv(x), ws, out)
    assert err(out, x.sum(0)) < 1e-6
    # embedding
    ids = torch.randint(0, 148, (7, 19), generator=G(31))
    table = rnd(148, 128, seed=32)
    e = torch.empty(7 * 19, 128, device=DEV)
    nv.embedding_fwd(dv(ids), dv(table), e)
    assert torch.equal(e.cpu(), table[ids.view(-1)])
    de = rnd(7 * 19, 128, seed=33)
    dt = torch.empty(148, 128, device=DEV)
    nv.embedding_bwd(dv(ids), dv(de), dt)
    ref = torch.zeros(148, 128).index_add_(0, ids.view(-1), de)
    assert err(dt, ref) < 1e-6
    # philox: deterministic, right rate, offset-consistent
    m1 = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    m2 = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    nv.philox_keep_mask(m1, 0.1, 1234, 0)
    nv.philox_keep_mask(m2, 0.1, 1234, 0)
    assert torch.equal(m1, m2)
    assert abs(m1.float().mean().item() - 0.9) < 2e-3
    nv.philox_keep_mask(m2[:4096], 0.1, 1234, 8192)
    assert torch.equal(m2[:4096], m1[8192:8192 + 4096])
    nv.philox_keep_mask(m2, 0.1, 99, 0)
    assert not torch.equal(m1, m2)
    # a ragged length behind a base that is not word-aligned (byte stores) gives the stream of the packed stores
    m3 = torch.full((4200,), 7, dtype=torch.uint8, device=DEV)
    nv.philox_keep_mask(m3[1:1 + 4099], 0.1, 1234, 0)
    assert torch.equal(m3[1:1 + 4099], m1[:4099]) and m3[0].item() == 7 and m3[4100].item() == 7
    m3.fill_(7)
    nv.philox_keep_mask(m3[:4099], 0.1, 1234, 0)
    assert torch.equal(m3[:4099], m1[:4099]) and m3[4099].item() == 7
    # copy2d / transpose / fill
    a, b = rnd(33, 70, seed=34), rnd(33, 70, seed=35)
    d = torch.zeros(40, 100, device=DEV)
    nv.copy2d(d[3:36, 10:80], dv(a), dv(b))
    assert torch.equal(d[3:36, 10:80].cpu(), a + b) and d[0].abs().sum().item() == 0
    t = torch.empty(70, 33, device=DEV)
    nv.transpose(t, dv(a))
    assert torch.equal(t.cpu(), a.t())
    nv.fill(t, 2.5)
    assert (t == 2.5).all().item()
    # relu-dropout backward
    y = torch.relu(rnd(50, 64, seed=36)) * (torch.rand(50, 64, generator=G(37)) >= 0.5) * 2.0
    g = rnd(50, 64, seed=38)
    gd = dv(g.clone())
    nv.relu_dropout_bwd(gd, dv(y), 2.0)
    assert torch.equal(gd.cpu(), torch.where(y > 0, g * 2.0, torch.zeros_like(g)))


def test_boundary_layout_kernels(nv):
    B, C, To = 3, 80, 37
    mels = rnd(B, C, To, seed=40)
    x0 = torch.empty(To, B, C, device=DEV)
    nv.frames_to_time_major(dv(mels), x0)
    ref = torch.cat((torch.zeros(1, B, C), mels.permute(2, 0, 1)[:-1]), 0)
    assert torch.equal(x0.cpu(), ref)
    pg = rnd(To, B, C + 1, seed=41)
    olens = torch.tensor([37, 20, 5], dtype=torch.int32)
    mel_cl, gate = torch.empty(B, To, C, device=DEV), torch.empty(B, To, device=DEV)
    nv.split_projection(dv(pg), mel_cl, gate, dv(olens))
    assert torch.equal(mel_cl.cpu(), pg[:, :, :C].permute(1, 0, 2))
    gref = pg[:, :, C].t().clone()
    pad = torch.arange(To)[None, :] >= olens[:, None]
    gref[pad] = 1e3
    assert torch.equal(gate.cpu(), gref)
    post = rnd(B, To, C, seed=42)
    mel, mel_post = torch.empty(B, C, To, device=DEV), torch.empty(B, C, To, device=DEV)
    mcl0 = mel_cl.cpu().clone()
    nv.finalize_outputs(mel_cl, dv(post), mel, mel_post, dv(olens))
    m_ref = mcl0.permute(0, 2, 1).clone()
    p_ref = (mcl0 + post).permute(0, 2, 1).clone()
    m_ref[pad[:, None, :].expand(-1, C, -1)] = 0
    p_ref[pad[:, None, :].expand(-1, C, -1)] = 0
    assert torch.equal(mel.cpu(), m_ref) and torch.equal(mel_post.cpu(), p_ref)
    assert torch.equal(mel_cl.cpu(), m_ref.permute(0, 2, 1))          # in-place fill (H2.3)
    dm, dp = rnd(B, C, To, seed=43), rnd(B, C, To, seed=44)
    dmel_cl, dpost_cl = torch.empty(B, To, C, device=DEV), torch.empty(B, To, C, device=DEV)
    nv.grads_to_channel_last(dv(dm), dv(dp), dmel_cl, dpost_cl)
    assert torch.equal(dmel_cl.cpu(), (dm + dp).permute(0, 2, 1)) and torch.equal(dpost_cl.cpu(), dp.permute(0, 2, 1))
    nv.grads_to_channel_last(None, dv(dp), dmel_cl, dpost_cl)
    assert torch.equal(dmel_cl.cpu(), dp.permute(0, 2, 1))
    dg = rnd(B, To, seed=45)
    dout = torch.empty(To, B, C + 1, device=DEV)
    nv.gather_dout(dmel_cl, dv(dg), dout)
    assert torch.equal(dout[:, :, :C].cpu(), dp.permute(2, 0, 1)) and torch.equal(dout[:, :, C].cpu(), dg.t())


# ------------------------------------------------------------------------------------------------
# recurrent cell
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,widths", [(5, 128, (64, 128)), (70, 256, (128, 64, 256)), (64, 1024, (1024, 512, 1024))])
def test_lstm_step_forward(nv, B, H, widths):
    K = sum(widths)
    xs = [rnd(B, w, seed=50 + i) for i, w in enumerate(widths)]
    W = rnd(4 * H, K, seed=54, scale=0.05)
    gin, bias, c_prev = rnd(B, 4 * H, seed=55), rnd(4 * H, seed=56), rnd(B, H, seed=57)
    keep = (torch.rand(B, H, generator=G(58)) >= 0.1).to(torch.uint8)
    lens = torch.randint(1, 6, (B,), generator=G(59)).to(torch.int32)
    t = 3
    pre = torch.cat(xs, 1) @ W.t() + gin + bias
    i, f, g, o = pre.chunk(4, 1)
    i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
    c = f * c_prev + i * g
    h = o * torch.tanh(c) * keep * nv.scale_for(0.1)
    gates = torch.full((B, 4 * H), float('nan'), device=DEV)
    c_out, h_out = torch.empty(B, H, device=DEV), torch.empty(B, H, device=DEV)
    nv.lstm_step_fwd([dv(x) for x in xs], list(widths), dv(W), H, B, gates, c_out, h_out, gin=dv(gin), bias=dv(bias),
                     c_prev=dv(c_prev), keep=dv(keep), keep_scale=nv.scale_for(0.1))
    assert err(gates, torch.cat((i, f, g, o), 1)) < 1e-5
    assert err(c_out, c) < 1e-5 and err(h_out, h) < 1e-5
    # zero segment (None) and packed-sequence masking
    nv.lstm_step_fwd([None] + [dv(x) for x in xs[1:]], list(widths), dv(W), H, B, gates, c_out, h_out,
                     bias=dv(bias), lens=dv(lens), t=t)
    xs0 = [torch.zeros_like(xs[0])] + xs[1:]
    pre = torch.cat(xs0, 1) @ W.t() + bias
    i, f, g, o = pre.chunk(4, 1)
    c2 = torch.sigmoid(i) * torch.tanh(g)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    valid = (t < lens).float()[:, None]
    assert err(c_out, c2 * valid) < 1e-5 and err(h_out, h2 * valid) < 1e-5


def test_skinny_gemm_and_lstm_backward(nv):
    B, N, K = 37, 200, 512
    x, W = rnd(B, K, seed=60), rnd(N, K, seed=61, scale=0.1)
    for ns in (1, 4):
        Y = torch.empty(ns, B, N, device=DEV)
        nv.skinny_gemm([dv(x)], [K], dv(W), N, B, Y, nsplit=ns)
        assert err(Y.sum(0), x @ W.t()) < 2e-5
    H = 128
    gates_pre = rnd(B, 4 * H, seed=62).requires_grad_(True)
    c_prev = rnd(B, H, seed=63).requires_grad_(True)
    keep = (torch.rand(B, H, generator=G(64)) >= 0.1).to(torch.uint8)
    sc = nv.scale_for(0.1)
    i, f, g, o = gates_pre.chunk(4, 1)
    i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
    c = f * c_prev + i * g
    h = o * torch.tanh(c) * keep * sc
    dh1, dh2, dc_in = rnd(B, H, seed=65), rnd(2, B, H, seed=66), rnd(B, H, seed=67)
    ((h * (dh1 + dh2.sum(0))).sum() + (c * dc_in).sum()).backward()
    dc = dv(dc_in.clone())
    dgates = torch.empty(B, 4 * H, device=DEV)
    dh1d, dh2d = dv(dh1), dv(dh2)          # keep the device buffers alive while the kernel runs
    a2 = nv._addend(dh2d[0], nsplit=2, split_stride=B * H)
    # use the low-level entry to exercise split addends
    st = nv.LstmBwd()
    st.B, st.H = B, H
    st.dh[0] = nv._addend(dh1d)
    st.dh[1] = a2
    st.dh[2] = nv._addend(None)
    gact = dv(torch.cat((i, f, g, o), 1).detach())
    cd, cpd, kd = dv(c.detach()), dv(c_prev.detach()), dv(keep)
    st.gates, st.ld_gates = nv.ptr(gact), 4 * H
    st.c_prev, st.ld_cprev = nv.ptr(cpd), H
    st.c, st.ld_c = nv.ptr(cd), H
    st.keep, st.ld_keep, st.keep_scale = nv.ptr(kd, torch.uint8), H, sc
    st.dc, st.ld_dc = nv.ptr(dc), H
    st.dgates, st.ld_dgates = nv.ptr(dgates), 4 * H
    import ctypes
    nv._check(nv.load().t2amd_lstm_pointwise_bwd_f32(ctypes.byref(st), nv._stream()), "lstm_bwd")
    assert err(dgates, gates_pre.grad) < 1e-5
    assert err(dc, c_prev.grad) < 1e-5


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def _attn_inputs(B, Ti, E, Hq, seed):
    sd = {
        'decoder.attention_layer.query_layer.linear_layer.weight': rnd(128, Hq, seed=seed, scale=0.05),
        'decoder.attention_layer.location_layer.location_conv.conv.weight': rnd(32, 2, 31, seed=seed + 1, scale=0.3),
        'decoder.attention_layer.location_layer.location_dense.linear_layer.weight': rnd(128, 32, seed=seed + 2, scale=0.3),
        'decoder.attention_layer.v.linear_layer.weight': rnd(1, 128, seed=seed + 3),
    }
    h, mem, pm = rnd(B, Hq, seed=seed + 4), rnd(B, Ti, E, seed=seed + 5), rnd(B, Ti, 128, seed=seed + 6)
    lens = torch.randint(max(1, Ti // 3), Ti + 1, (B,), generator=G(seed + 7))
    lens[0] = Ti
    w_prev = torch.softmax(rnd(B, Ti, seed=seed + 8), 1)
    cum = torch.rand(B, Ti, generator=G(seed + 9)) * 2
    return sd, h, mem, pm, lens, w_prev, cum


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("B,Ti,E,Hq", [(3, 37, 128, 128), (4, 175, 512, 1024), (2, 70, 512, 1024), (2, 300, 512, 1024)])
def test_attention_forward_backward(nv, B, Ti, E, Hq, bf16):
    """bf16=True: the two gradient products of the location layer (dcol = U^T dpre -> the carries, dU) round their
    operands to bf16 (t2amd_attn_bwd.bf16): bf16-class tolerance on exactly those outputs, f32-class on the rest."""
    sd, h, mem, pm, lens, w_prev, cum = _attn_inputs(B, Ti, E, Hq, 70)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    hL, pmL, wpL, cumL = (t.clone().requires_grad_(True) for t in (h, pm, w_prev, cum))
    mask = ~orc.get_mask_from_lengths(lens, Ti)
    ctx, w = orc.attention_step(hL, mem, pmL, wpL, cumL, mask, leaf, -float('inf'))
    cum_new = cumL + w
    d_ctx, d_w_extra, d_w_carry, d_cum_carry = rnd(B, E, seed=80), rnd(B, Ti, seed=81), rnd(B, Ti, seed=82), rnd(B, Ti, seed=83)
    ((ctx * d_ctx).sum() + (w * (d_w_extra + d_w_carry)).sum() + (cum_new * d_cum_carry).sum()).backward()

    Wq = dv(sd['decoder.attention_layer.query_layer.linear_layer.weight'])
    Wd = dv(sd['decoder.attention_layer.location_layer.location_dense.linear_layer.weight'])
    Wc = dv(sd['decoder.attention_layer.location_layer.location_conv.conv.weight'])
    v = dv(sd['decoder.attention_layer.v.linear_layer.weight']).view(-1)
    U = torch.empty(128 * 62, device=DEV)
    nv.fold_location(Wd, Wc, U)
    Uref = torch.einsum('df,fck->dck', sd['decoder.attention_layer.location_layer.location_dense.linear_layer.weight'],
                        sd['decoder.attention_layer.location_layer.location_conv.conv.weight']).reshape(128, 62)
    assert err(U.view(128, 62), Uref) < 1e-5

    lens32 = dv(lens.to(torch.int32))
    cum_d, cum_save = dv(cum.clone()), torch.empty(B, Ti, device=DEV)
    w_out, ctx_out, q_out = torch.empty(B, Ti, device=DEV), torch.empty(B, E, device=DEV), torch.empty(B, 128, device=DEV)
    memd, pmd, hd, wpd = dv(mem), dv(pm), dv(h), dv(w_prev)
    ws = torch.full((nv.attn_fwd_ws_floats(B, Ti) + nv.attn_bwd_ws_floats(B, Ti),), float('nan'), device=DEV)
    mem16 = memd.bfloat16() if bf16 else None       # bf16 mode also streams a bf16 copy of the encoder memory
    nv.attention_step_fwd(hd, Wq, U, v, pmd, memd, lens32, wpd, cum_d, cum_save, w_out, ctx_out, q_out, ws, bf16=bf16,
                          memory16=mem16)
    ftol = 1e-4 if bf16 else 1e-5         # bf16=True: location conv as a split-bf16 product (~2^-17 per product)
    assert err(w_out, w) < ftol
    if bf16 and Hq % 128 == 0:            # the query product from a bf16 copy of W_q: exact against the same rounding
        q16 = torch.empty(B, 128, device=DEV)
        cum2, w2, c2 = dv(cum.clone()), torch.empty(B, Ti, device=DEV), torch.empty(B, E, device=DEV)
        nv.attention_step_fwd(hd, Wq, U, v, pmd, memd, lens32, wpd, cum2, torch.empty(B, Ti, device=DEV), w2, c2, q16, ws,
                              bf16=True, memory16=mem16, Wq16=Wq.bfloat16())
        Wq_r = sd['decoder.attention_layer.query_layer.linear_layer.weight'].bfloat16().float()
        assert err(q16, h @ Wq_r.t()) < 1e-5
        assert err(w2, w) < 2e-2
    if bf16:        # context from bf16-rounded rows: exact against the same rounding, bf16-class against f32 rows
        assert err(ctx_out, torch.bmm(w_out.cpu().unsqueeze(1), mem.bfloat16().float()).squeeze(1)) < 1e-5
        assert err(ctx_out, ctx) < 5e-3
    else:
        assert err(ctx_out, ctx) < ftol
    assert err(cum_d, cum_new) < ftol and torch.equal(cum_save.cpu(), cum)
    assert err(q_out, h @ sd['decoder.attention_layer.query_layer.linear_layer.weight'].t()) < 1e-5
    assert (w_out.cpu()[mask] == 0).all()

    # backward
    dctx_total = torch.empty(B, E, device=DEV)
    # incoming carries in partial form: spread the w-carry over the slices, keep part of the cum-carry
    # in the running accumulator and part in the c = 1 partials
    S = nv.ATT_SLICES
    dwin = torch.zeros(S, B, 2, Ti)
    for s_ in range(S):
        dwin[s_, :, 0] = d_w_carry / S
        dwin[s_, :, 1] = d_cum_carry * 0.125
    dwin_d, dcum_d = dv(dwin), dv(d_cum_carry * 0.5)
    d_pm = torch.zeros(B, Ti, 128, device=DEV)
    dU_acc, dv_acc = torch.zeros(B, 128, 62, device=DEV), torch.zeros(B, 128, device=DEV)
    dq, dh = torch.empty(B, 128, device=DEV), torch.full((S, B, Hq), float('nan'), device=DEV)
    half = dv(d_ctx * 0.5)
    nv.attention_step_bwd([half, half], dctx_total, dv(d_w_extra), q_out, Wq, U, v, pmd, memd, lens32, w_out, wpd,
                          cum_save, dwin_d, dcum_d, d_pm, dU_acc, dv_acc, dq, dh, ws, bf16=bf16, memory16=mem16)
    gt = 2e-2 if bf16 else 1.0          # tolerance scale of the bf16-rounded products (relative to the f32 limits)
    cl = lambda lim, ref: max(lim, gt * 0.5 * ref.abs().max().item()) if bf16 else lim
    assert err(dctx_total, d_ctx) < 1e-6
    assert err(dh.sum(0), hL.grad) < cl(2e-5, hL.grad)
    assert err(d_pm, pmL.grad) < cl(2e-5, pmL.grad)
    assert err(dcum_d, d_cum_carry) < 1e-6                      # running accumulator now holds the full carry
    assert err(dwin_d[:, :, 0].sum(0), wpL.grad) < cl(2e-5, wpL.grad)
    assert err(dcum_d + dwin_d[:, :, 1].sum(0), cumL.grad) < cl(2e-5, cumL.grad)
    dWd, dWc, dvv = torch.empty(128, 32, device=DEV), torch.empty(32, 2, 31, device=DEV), torch.empty(1, 128, device=DEV)
    nv.unfold_location_grads(dU_acc, dv_acc, B, Wd, Wc, dWd, dWc, dvv)
    gWd = leaf['decoder.attention_layer.location_layer.location_dense.linear_layer.weight'].grad
    gWc = leaf['decoder.attention_layer.location_layer.location_conv.conv.weight'].grad
    assert err(dWd, gWd) < cl(5e-5, gWd)
    assert err(dWc, gWc) < cl(5e-5, gWc)
    gv = leaf['decoder.attention_layer.v.linear_layer.weight'].grad
    assert err(dvv, gv) < cl(5e-5, gv)
    dWq_ref = leaf['decoder.attention_layer.query_layer.linear_layer.weight'].grad
    assert err(dq.cpu().t() @ h, dWq_ref) < cl(5e-5, dWq_ref)


def test_attention_backward_fused_matches_two_launch():
    """T2AMD_ATTN_FUSED_BWD=1 (K_b1 + hand-off + K_b2 in one launch) must reproduce the two-launch backward bit for
    bit: same per-thread arithmetic, only the transport of dw between the four workgroups of an utterance differs.
    Runs in child processes because the switch is read once per process."""
    import subprocess, sys, os, tempfile
    code = r"""
import sys, torch
sys.path.insert(0, %r)
from tacotron2_amd import native as nv
dev = torch.device('cuda')
g = torch.Generator().manual_seed(5)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
B, Ti, E, Hq = 5, 150, 512, 1024
mem, pm = rnd(B, Ti, E), rnd(B, Ti, 128)
Wq, U, v = rnd(128, Hq) * 0.05, rnd(128 * 62) * 0.1, rnd(128)
lens = torch.tensor([150, 140, 97, 64, 33], dtype=torch.int32, device=dev)
w = torch.softmax(rnd(B, Ti), 1); wprev = torch.softmax(rnd(B, Ti), 1); cum = torch.rand(B, Ti, generator=g).to(dev)
q, dctx, dwx = rnd(B, 128), rnd(B, E), rnd(B, Ti)
outs = []
ws = torch.zeros(nv.attn_bwd_ws_floats(B, Ti), device=dev)
dwin, dcum = rnd(4, B, 2, Ti), rnd(B, Ti)
d_pm, dU, dv_ = torch.zeros(B, Ti, 128, device=dev), torch.zeros(B, 128, 62, device=dev), torch.zeros(B, 128, device=dev)
dq, dh, tot = torch.empty(B, 128, device=dev), torch.empty(4, B, Hq, device=dev), torch.empty(B, E, device=dev)
for step in range(3):
    nv.attention_step_bwd([dctx], tot, dwx, q, Wq, U, v, pm, mem, lens, w, wprev, cum, dwin, dcum, d_pm, dU, dv_, dq, dh, ws)
torch.cuda.synchronize()
torch.save([t.cpu() for t in (tot, dwin, dcum, d_pm, dU, dv_, dq, dh)], sys.argv[1])
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    # third run: the one-launch form with the first hand-off as {token, value} granules (T2AMD_ATTN_GRANULES=1)
    for fused, gran in (("0", "0"), ("1", "0"), ("1", "1")):
        with tempfile.NamedTemporaryFile(suffix=".pt", delete=False) as fh:
            path = fh.name
        env = dict(os.environ, T2AMD_ATTN_FUSED_BWD=fused, T2AMD_ATTN_GRANULES=gran)
        subprocess.run([sys.executable, "-c", code, path], check=True, env=env, timeout=120)
        res.append(torch.load(path))
        os.unlink(path)
    for other in res[1:]:
        for x, y in zip(res[0], other):
            assert torch.equal(x, y)


@pytest.mark.parametrize("gran", [0, 1])
@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("B,Ti,E,Hq,Hx,first", [(3, 37, 128, 128, 128, False), (5, 150, 512, 1024, 1024, False),
                                                (64, 187, 512, 1024, 1024, False), (2, 300, 512, 1024, 512, False),
                                                (4, 21, 256, 256, 0, True), (3, 60, 128, 136, 128, False)])
def test_attention_backward_folded_cells_bitwise(nv, B, Ti, E, Hq, Hx, first, bf16, gran):
    """t2amd_attn_bwd.cell_q / cell_x: the LSTM cell backwards of a BPTT step run as the closing phase of the
    attention-backward launch.  Against the separate t2amd_lstm_pointwise_bwd2_f32 launch fed by the dh_out slabs, every
    output of the attention step and of both cells (gate gradients f32 + bf16, dc carry) must be bit-identical: the
    folded kernel forms W_q^T dq from the same 16-dim partial sums in the same order.  Three chained calls per variant
    (carries, accumulators and the dc carries feed the next call; the hand-off tokens change per launch).  B = 64 fills
    the chip like the training step; Ti = 300 gives the polling wave col2im stores of its own; Hx = 0: no second cell
    (time step 0 of the loop); first: no previous cell state; Hq = 136 is a geometry the folded kernel does not cover
    (the call then launches the cells itself).  gran = 1: the folded run also takes the granule form of the first hand-off
    (t2amd_set_attn_bwd_granules), the reference run the drained-stores-and-token form."""
    S = nv.ATT_SLICES
    g = G(900 + B + Ti)
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(DEV)
    mem, pm = r(B, Ti, E), r(B, Ti, 128)
    Wq, U, v = r(128, Hq, scale=0.05), r(128 * 62, scale=0.1), r(128)
    lens = torch.randint(max(1, Ti // 3), Ti + 1, (B,), generator=g).to(torch.int32)
    lens[0] = Ti
    lens = lens.to(DEV)
    w, wprev = torch.softmax(r(B, Ti), 1), torch.softmax(r(B, Ti), 1)
    cum = torch.rand(B, Ti, generator=g).to(DEV)
    q, dctx, dwx = r(B, 128), r(B, E), r(B, Ti)
    mem16 = mem.bfloat16() if bf16 else None
    # cell operands: activated gates in (0,1) / (-1,1), cell states, dropout masks, upstream gradient slabs
    def cell_inputs(H, seed):
        gg = G(seed)
        rr = lambda *s: torch.randn(*s, generator=gg).to(DEV)
        gates = torch.cat([torch.sigmoid(rr(B, H)), torch.sigmoid(rr(B, H)), torch.tanh(rr(B, H)), torch.sigmoid(rr(B, H))], 1)
        return dict(gates=gates.contiguous(), c=rr(B, H), c_prev=None if first else rr(B, H),
                    keep=(torch.rand(B, H, generator=gg) > 0.1).to(torch.uint8).to(DEV),
                    up2=rr(2, B, H + 64), up1=rr(B, H + 32), dc0=rr(B, H))
    cqi = cell_inputs(Hq, 31)
    cxi = cell_inputs(Hx, 32) if Hx else None

    def run(fold):
        ws = torch.zeros(nv.attn_bwd_ws_floats(B, Ti), device=DEV)
        gg = G(77)
        dwin, dcum = (torch.randn(S, B, 2, Ti, generator=gg) * 0.1).to(DEV), (torch.randn(B, Ti, generator=gg) * 0.1).to(DEV)
        d_pm, dU, dv_ = torch.zeros(B, Ti, 128, device=DEV), torch.zeros(B, 128, 62, device=DEV), torch.zeros(B, 128, device=DEV)
        dq, dh = torch.empty(B, 128, device=DEV), torch.zeros(S, B, Hq, device=DEV)
        tot = torch.empty(B, E, device=DEV)
        st = {}
        for name, ci, H in (("q", cqi, Hq), ("x", cxi, Hx)):
            if ci is None:
                continue
            st[name] = dict(dc=ci['dc0'].clone(), dg=torch.full((B, 4 * H), float('nan'), device=DEV),
                            dg16=torch.zeros(B, 4 * H, dtype=torch.bfloat16, device=DEV) if bf16 else None)
        outs = []
        for step in range(3):
            descs = {}
            for name, ci, H in (("q", cqi, Hq), ("x", cxi, Hx)):
                if ci is None:
                    continue
                # dh[0]: two partial slabs (a split-K dgrad output, column offset 64), dh[1]: the dh_out slabs (cell_q) or
                # a plain addend (cell_x), dh[2]: absent for cell_x, a one-slab addend at column offset 32 for cell_q
                d0 = (ci['up2'][0, :, 64:], 2, ci['up2'].stride(0))
                d1 = (dh[0], S, dh.stride(0)) if name == "q" else ci['up1'][:, 32:]
                d2 = ci['up1'][:, 32:] if name == "q" else None
                descs[name] = nv.lstm_bwd_desc(B, H, [d0, d1, d2], ci['gates'], ci['c_prev'], ci['c'], ci['keep'], 1.0 / 0.9,
                                               st[name]['dc'], st[name]['dg'], dgates16=st[name]['dg16'])
            args = ([dctx], tot, dwx, q, Wq, U, v, pm, mem, lens, w, wprev, cum, dwin, dcum, d_pm, dU, dv_, dq, dh, ws)
            if fold:
                nv.attention_step_bwd(*args, bf16=bf16, memory16=mem16, cell_q=descs["q"], cell_x=descs.get("x"))
            else:
                nv.attention_step_bwd(*args, bf16=bf16, memory16=mem16)
                nv.lstm_pointwise_bwd2(descs["q"], descs.get("x"))
            torch.cuda.synchronize()
            outs.append([t.clone() for t in (tot, dwin, dcum, d_pm, dU, dv_, dq)] +
                        [t.clone() for n in sorted(st) for t in (st[n]['dc'], st[n]['dg'], st[n]['dg16']) if t is not None])
        return outs

    try:
        nv.set_attn_bwd_granules(0)
        ref = run(False)
        nv.set_attn_bwd_granules(gran)
        got = run(True)
    finally:
        nv.set_attn_bwd_granules(-1)
    assert all(torch.isfinite(t.float()).all() for t in ref[-1])
    for step, (a_, b_) in enumerate(zip(ref, got)):
        for i, (x, y) in enumerate(zip(a_, b_)):
            assert torch.equal(x, y), (step, i, (x.float() - y.float()).abs().max().item())


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("B,Ti,E,Hq,inactive", [(3, 37, 128, 128, False), (5, 150, 512, 1024, True), (64, 187, 512, 1024, False),
                                                (2, 300, 512, 1024, False), (1, 5, 512, 1024, False)])
def test_attention_forward_fused_matches_two_launch(nv, B, Ti, E, Hq, inactive, bf16):
    """t2amd_set_attn_fwd_fused(1): K_e and K_c of a step as ONE launch whose four workgroups per utterance exchange the
    partial energies as {token, value} granules.  Same per-thread arithmetic and summation order: weights, context (f32 and
    the bf16 copy), cumulative weights and the saved query must equal the two-launch form bit for bit, over three chained
    steps (the cumulative weights and the previous weights feed the next step; the launch token changes), with ragged
    lengths, an utterance of full length, and utterances switched off (batched inference's `active` mask)."""
    g = G(400 + B + Ti)
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(DEV)
    mem, pm = r(B, Ti, E), r(B, Ti, 128)
    Wq, U, v = r(128, Hq, scale=0.05), r(128 * 62, scale=0.1), r(128)
    lens = torch.randint(max(1, Ti // 3), Ti + 1, (B,), generator=g).to(torch.int32)
    lens[0] = Ti
    lens = lens.to(DEV)
    hs = [r(B, Hq) for _ in range(3)]
    active = None
    if inactive:
        active = torch.ones(B, dtype=torch.uint8)
        active[1] = 0
        active = active.to(DEV)
    mem16 = mem.bfloat16() if bf16 else None
    Wq16 = Wq.bfloat16() if bf16 and Hq % 128 == 0 else None

    def run(fused):
        nv.set_attn_fwd_fused(fused)
        ws = torch.zeros(nv.attn_fwd_ws_floats(B, Ti), device=DEV)
        cum = torch.zeros(B, Ti, device=DEV)
        w_prev, outs = None, []
        for step in range(3):
            cum_save, w_out = torch.zeros(B, Ti, device=DEV), torch.zeros(B, Ti, device=DEV)
            ctx, q = torch.zeros(B, E, device=DEV), torch.zeros(B, 128, device=DEV)
            nv.attention_step_fwd(hs[step], Wq, U, v, pm, mem, lens, w_prev, cum, cum_save, w_out, ctx, q, ws, active=active,
                                  bf16=bf16, memory16=mem16, Wq16=Wq16)
            torch.cuda.synchronize()
            outs.append([t.clone() for t in (w_out, ctx, q, cum, cum_save)])
            w_prev = w_out
        return outs

    try:
        ref, got = run(0), run(1)
    finally:
        nv.set_attn_fwd_fused(-1)
    assert all(torch.isfinite(t).all() for t in ref[-1])
    assert (ref[-1][0].sum(1)[active.bool() if inactive else slice(None)] - 1).abs().max().item() < 1e-4
    for step, (a_, b_) in enumerate(zip(ref, got)):
        for i, (x, y) in enumerate(zip(a_, b_)):
            assert torch.equal(x, y), (step, i, (x - y).abs().max().item())


def test_attention_energy_kernel_forms_agree():
    """K_e has three forms (attention.hip): one utterance per workgroup in the latency-shaped register allocation, the
    same with room for two workgroups per CU, and four utterances per workgroup with the W_q slice kept in registers
    (batched inference, B > 128).  Per-thread arithmetic is identical, so weights, context and the saved query must
    agree bit for bit -- with a ragged batch whose size is not a multiple of four and inactive utterances in it.
    Child processes: the A/B switch is read once per process."""
    import subprocess, sys, os, tempfile
    code = r"""
import sys, torch
sys.path.insert(0, %r)
from tacotron2_amd import native as nv
dev = torch.device('cuda')
g = torch.Generator().manual_seed(15)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
B, Ti, E, Hq = 11, 150, 512, 1024
mem, pm, h = rnd(B, Ti, E), rnd(B, Ti, 128), rnd(B, Hq)
Wq, U, v = rnd(128, Hq) * 0.05, rnd(128 * 62) * 0.1, rnd(128)
lens = torch.tensor([150, 150, 140, 120, 97, 64, 33, 20, 16, 15, 2], dtype=torch.int32, device=dev)
active = torch.tensor([1, 1, 0, 1, 1, 1, 1, 0, 1, 0, 1], dtype=torch.uint8, device=dev)
wprev = torch.softmax(rnd(B, Ti), 1); cum = torch.rand(B, Ti, generator=g).to(dev)
w_out, ctx_out, q_out = torch.zeros(B, Ti, device=dev), torch.zeros(B, E, device=dev), torch.zeros(B, 128, device=dev)
ws = torch.zeros(nv.attn_fwd_ws_floats(B, Ti), device=dev)
nv.attention_step_fwd(h, Wq, U, v, pm, mem, lens, wprev, cum, None, w_out, ctx_out, q_out, ws, active=active, bf16=True,
                      memory16=mem.bfloat16(), Wq16=Wq.bfloat16())
torch.cuda.synchronize()
torch.save([t.cpu() for t in (w_out, ctx_out, q_out, cum)], sys.argv[1])
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for form in ("0", "1", "2"):
        with tempfile.NamedTemporaryFile(suffix=".pt", delete=False) as fh:
            path = fh.name
        env = dict(os.environ, T2AMD_KE_FORM=form)
        subprocess.run([sys.executable, "-c", code, path], check=True, env=env, timeout=300)
        res.append(torch.load(path))
        os.unlink(path)
    assert float(res[0][0].sum()) > 5.0                       # eight active utterances, weights sum to one each
    for other in res[1:]:
        for x, y in zip(res[0], other):
            assert torch.equal(x, y)


def test_attention_no_mask_and_first_step(nv):
    """Inference semantics: no length mask (reference model.py:432) and zero previous weights."""
    B, Ti, E, Hq = 1, 29, 512, 1024
    sd, h, mem, pm, lens, w_prev, cum = _attn_inputs(B, Ti, E, Hq, 90)
    ctx, w = orc.attention_step(h, mem, pm, torch.zeros(B, Ti), torch.zeros(B, Ti), None, sd, -float('inf'))
    Wq = dv(sd['decoder.attention_layer.query_layer.linear_layer.weight'])
    U = torch.empty(128 * 62, device=DEV)
    nv.fold_location(dv(sd['decoder.attention_layer.location_layer.location_dense.linear_layer.weight']),
                     dv(sd['decoder.attention_layer.location_layer.location_conv.conv.weight']), U)
    cum_d = torch.zeros(B, Ti, device=DEV)
    w_out, ctx_out = torch.empty(B, Ti, device=DEV), torch.empty(B, E, device=DEV)
    ws = torch.empty(nv.attn_fwd_ws_floats(B, Ti), device=DEV)
    nv.attention_step_fwd(dv(h), Wq, U, dv(sd['decoder.attention_layer.v.linear_layer.weight']).view(-1), dv(pm), dv(mem),
                          None, None, cum_d, None, w_out, ctx_out, None, ws)
    assert err(w_out, w) < 1e-5 and err(ctx_out, ctx) < 1e-5 and err(cum_d, w) < 1e-5


# ------------------------------------------------------------------------------------------------
# small-batch (B <= 8) decode kernels
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B", [1, 2, 3, 5, 8])
def test_small_batch_lstm_step_and_linear(nv, B):
    H, widths = 256, (256, 512, 64)
    K = sum(widths)
    xs = [rnd(B, w, seed=100 + i) for i, w in enumerate(widths)]
    W, gin, bias = rnd(4 * H, K, seed=104, scale=0.05), rnd(B, 4 * H, seed=105), rnd(4 * H, seed=106)
    c_prev = rnd(B, H, seed=107)
    pre = torch.cat(xs, 1) @ W.t() + gin + bias
    i, f, g, o = pre.chunk(4, 1)
    i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
    c = f * c_prev + i * g
    h = o * torch.tanh(c)
    gates = torch.full((B, 4 * H), float('nan'), device=DEV)
    c_out, h_out = torch.empty(B, H, device=DEV), torch.empty(B, H, device=DEV)
    xd = [dv(x) for x in xs]
    nv.lstm_step_fwd(xd, list(widths), dv(W), H, B, gates, c_out, h_out, gin=dv(gin), bias=dv(bias),
                     c_prev=dv(c_prev), small=True)
    assert err(gates, torch.cat((i, f, g, o), 1)) < 1e-5
    assert err(c_out, c) < 1e-5 and err(h_out, h) < 1e-5
    # absent middle segment = zeros; the MFMA kernel must agree with the matrix-vector kernel
    g2, c2, h2 = torch.empty_like(gates), torch.empty_like(c_out), torch.empty_like(h_out)
    nv.lstm_step_fwd([xd[0], None, xd[2]], list(widths), dv(W), H, B, gates, c_out, h_out, bias=dv(bias), small=True)
    nv.lstm_step_fwd([xd[0], None, xd[2]], list(widths), dv(W), H, B, g2, c2, h2, bias=dv(bias))
    assert err(gates, g2) < 1e-5 and err(h_out, h2) < 1e-5
    # plain linear with a ragged N, an unaligned input row stride (81 floats, like the PG slab), relu + mask
    N, Kl = 81, 80
    Xbig = rnd(B, 81, seed=110)
    Wl, bl = rnd(N, Kl, seed=111), rnd(N, seed=112)
    keep = (torch.rand(B, N, generator=G(113)) >= 0.5).to(torch.uint8)
    Xd = dv(Xbig)
    Y = torch.full((B, N), float('nan'), device=DEV)
    nv.linear_small(Xd[:, :Kl], dv(Wl), Y, bias=dv(bl), act=1, keep=dv(keep), keep_scale=2.0)
    assert err(Y, torch.relu(Xbig[:, :Kl] @ Wl.t() + bl) * keep * 2.0) < 1e-5
    Y2 = torch.empty(B, N, device=DEV)
    nv.linear_small(Xd[:, :Kl], dv(Wl), Y2)
    assert err(Y2, Xbig[:, :Kl] @ Wl.t()) < 1e-5


@pytest.mark.parametrize("B,N,widths", [(16, 256, (256,)), (256, 81, (1024, 512)), (70, 33, (64, 128))])
def test_skinny_gemm_epilogue(nv, B, N, widths):
    """Plain skinny product with the nn.Linear + relu + dropout epilogue (the per-step prenet layer / mel+gate
    projection of batched inference, reference model.py:99, 373-378): ragged N, several row blocks."""
    K = sum(widths)
    xs = [rnd(B, w, seed=140 + i) for i, w in enumerate(widths)]
    W, bias = rnd(N, K, seed=144, scale=0.1), rnd(N, seed=145)
    keep = (torch.rand(B, N, generator=torch.Generator().manual_seed(146)) > 0.5).to(torch.uint8)
    X = torch.cat(xs, 1)
    Y = torch.full((1, B, N), float('nan'), device=DEV)
    nv.skinny_gemm([dv(x) for x in xs], list(widths), dv(W), N, B, Y, bias=dv(bias), act=1, keep=dv(keep), keep_scale=2.0)
    assert err(Y[0], torch.relu(X @ W.t() + bias) * keep * 2.0) < 1e-5
    Y2 = torch.full((1, B, N), float('nan'), device=DEV)
    nv.skinny_gemm([dv(x) for x in xs], list(widths), dv(W), N, B, Y2, bias=dv(bias))
    assert err(Y2[0], X @ W.t() + bias) < 1e-5


# ------------------------------------------------------------------------------------------------
# bf16 operand mode of the recurrent kernels
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,widths", [(64, 256, (256, 128, 256)), (37, 128, (128,)), (3, 64, (128, 256)),
                                        # two or more rounds of 64 x 32 workgroups: the 64 x 64 kernel (skinny_wide64_kernel)
                                        (256, 1024, (128, 256, 128)), (200, 1024, (640,)), (520, 512, (128, 128))])
def test_lstm_step_bf16_operands(nv, B, H, widths):
    """bf16 X and W on v_mfma_f32_16x16x32_bf16, f32 accumulate / cell: must equal an f32 product of the
    bf16-ROUNDED operands to f32 summation-order accuracy (products of bf16 values are exact in f32)."""
    K = sum(widths)
    xs = [rnd(B, w, seed=120 + i).bfloat16() for i, w in enumerate(widths)]
    W = (rnd(4 * H, K, seed=124, scale=0.05) * (1 + torch.arange(4 * H).float().unsqueeze(1) / H)).bfloat16()
    gin, bias, c_prev = rnd(B, 4 * H, seed=125), rnd(4 * H, seed=126), rnd(B, H, seed=127)
    pre = torch.cat([x.float() for x in xs], 1) @ W.float().t() + gin + bias
    i, f, g, o = pre.chunk(4, 1)
    i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
    c = f * c_prev + i * g
    h = o * torch.tanh(c)
    gates = torch.full((B, 4 * H), float('nan'), device=DEV)
    c_out, h_out = torch.empty(B, H, device=DEV), torch.empty(B, H, device=DEV)
    h16 = torch.empty(B, H, device=DEV, dtype=torch.bfloat16)
    nv.lstm_step_fwd([x.to(DEV) for x in xs], list(widths), W.to(DEV), H, B, gates, c_out, h_out, gin=dv(gin), bias=dv(bias),
                     c_prev=dv(c_prev), bf16=True, h16_out=h16)
    assert err(gates, torch.cat((i, f, g, o), 1)) < 1e-5
    assert err(c_out, c) < 1e-5 and err(h_out, h) < 1e-5
    assert torch.equal(h16.cpu(), h_out.cpu().bfloat16())
    # dropout mask + finished rows (t >= len): the wide kernel has its own cell epilogue
    keep = (torch.rand(B, H, generator=torch.Generator().manual_seed(129)) > 0.3).to(torch.uint8)
    lens = torch.tensor([(5 if r % 3 else 2) for r in range(B)], dtype=torch.int32)
    gates.fill_(float('nan'))
    nv.lstm_step_fwd([x.to(DEV) for x in xs], list(widths), W.to(DEV), H, B, gates, c_out, h_out, gin=dv(gin), bias=dv(bias),
                     c_prev=dv(c_prev), keep=keep.to(DEV), keep_scale=1.0 / 0.7, lens=lens.to(DEV), t=3, bf16=True, h16_out=h16)
    live = (3 < lens).float().unsqueeze(1)
    assert err(gates, torch.cat((i, f, g, o), 1) * live) < 1e-5
    assert err(c_out, c * live) < 1e-5 and err(h_out, h * keep.float() / 0.7 * live) < 1e-5
    assert torch.equal(h16.cpu(), h_out.cpu().bfloat16())
    # plain (dgrad-shaped) product with split-K
    N = 200
    W2 = rnd(N, K, seed=128, scale=0.1).bfloat16()
    for ns in (1, 2):
        if (K // 128) % ns:
            continue
        Y = torch.empty(ns, B, N, device=DEV)
        nv.skinny_gemm([x.to(DEV) for x in xs], list(widths), W2.to(DEV), N, B, Y, nsplit=ns, bf16=True)
        assert err(Y.sum(0), torch.cat([x.float() for x in xs], 1) @ W2.float().t()) < 2e-5


# ------------------------------------------------------------------------------------------------
# split-bf16 operand images and the 'bf16x3' form of the wide tile (round 6; csrc/skinny_wide.h SW_X3)
# ------------------------------------------------------------------------------------------------
def split_image_ref(x):
    """CPU restatement of t2amd_split_bf16x3_f32: [..., K] f32 -> [..., 2 K] bf16, per 16 k: 16 hi = bf16(x), 16 lo = bf16(x - hi)."""
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    g = x.shape[-1] // 16
    hi, lo = hi.reshape(x.shape[:-1] + (g, 16)), lo.reshape(x.shape[:-1] + (g, 16))
    return torch.cat((hi, lo), -1).reshape(x.shape[:-1] + (2 * x.shape[-1],))


def unsplit(img):
    """hi + lo of an image as f32 (what the three products of the tile add up to, to ~2^-17)."""
    g = img.shape[-1] // 32
    v = img.float().reshape(img.shape[:-1] + (g, 2, 16))
    return (v[..., 0, :] + v[..., 1, :]).reshape(img.shape[:-1] + (g * 16,))


@pytest.mark.parametrize("rows,K", [(7, 16), (64, 1536), (300, 4096), (1, 80 * 16)])
def test_split_bf16x3_image(nv, rows, K):
    x = rnd(rows, K, seed=300) * torch.logspace(-3, 2, K).unsqueeze(0)
    out = torch.empty(rows, 2 * K, device=DEV, dtype=torch.bfloat16)
    nv.split_bf16x3(dv(x), out)
    ref = split_image_ref(x)
    assert torch.equal(out.cpu().view(torch.int16), ref.view(torch.int16))
    rel = ((unsplit(out.cpu()) - x).abs() / x.abs().clamp_min(1e-30)).max().item()
    assert rel < 2.0 ** -16, rel                       # 16 mantissa bits: 2^-17 per half, both roundings


@pytest.mark.parametrize("B,H,widths", [(64, 256, (256, 128, 256)), (37, 128, (128,)), (3, 64, (128, 256)), (64, 1024, (1024, 512, 1024))])
def test_lstm_step_bf16x3_operands(nv, B, H, widths):
    """Split-bf16 X and W: hi.hi + lo.hi + hi.lo on v_mfma_f32_32x32x16_bf16, f32 accumulate.  (i) The kernel's arithmetic: against
    the f64 sum of exactly those three products of the images' values it must agree to f32 summation-order accuracy (products of
    bf16 values are exact in f32) -- the bf16 test's bar.  (ii) The mode's accuracy: against the f64 product of the f32 operands the
    pre-activations carry ~2^-17 per product (here ~1e-5 of a pre-activation of ~3 after K = 640..2560 random-sign products), 1/256 of
    the bf16 mode's error on the same operands.  (iii) The split image of h equals the split of the h it wrote."""
    K = sum(widths)
    xs = [rnd(B, w, seed=120 + i) for i, w in enumerate(widths)]
    W = rnd(4 * H, K, seed=124, scale=0.05) * (1 + torch.arange(4 * H).float().unsqueeze(1) / H)
    gin, bias, c_prev = rnd(B, 4 * H, seed=125), rnd(4 * H, seed=126), rnd(B, H, seed=127)

    def cell(pre):
        i, f, g, o = pre.chunk(4, 1)
        i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
        c = f * c_prev.double() + i * g
        return torch.cat((i, f, g, o), 1), c, o * torch.tanh(c)

    X = torch.cat(xs, 1)
    Xh, Wh = X.bfloat16().double(), W.bfloat16().double()
    Xl, Wl = (X - X.bfloat16().float()).bfloat16().double(), (W - W.bfloat16().float()).bfloat16().double()
    add = gin.double() + bias.double()
    gates3, c3, h3ref = cell(Xh @ Wh.t() + Xl @ Wh.t() + Xh @ Wl.t() + add)          # what the tile computes
    gates_t, c_t, h_t = cell(X.double() @ W.double().t() + add)                      # the f32 operands' own product
    gates = torch.full((B, 4 * H), float('nan'), device=DEV)
    c_out, h_out = torch.empty(B, H, device=DEV), torch.empty(B, H, device=DEV)
    h3 = torch.empty(B, 2 * H, device=DEV, dtype=torch.bfloat16)
    x3 = [torch.empty(B, 2 * w, device=DEV, dtype=torch.bfloat16) for w in widths]
    for xi, x in zip(x3, xs):
        nv.split_bf16x3(dv(x), xi)
    W3 = torch.empty(4 * H, 2 * K, device=DEV, dtype=torch.bfloat16)
    nv.split_bf16x3(dv(W), W3)
    nv.lstm_step_fwd(x3, list(widths), W3, H, B, gates, c_out, h_out, gin=dv(gin), bias=dv(bias), c_prev=dv(c_prev), bf16=3, h16_out=h3)
    assert err(gates, gates3) < 1e-5 and err(c_out, c3) < 1e-5 and err(h_out, h3ref) < 1e-5          # (i)
    e3 = (gates.cpu().double() - gates_t).abs().max().item()                                           # (ii)
    e16 = (cell(X.bfloat16().double() @ W.bfloat16().double().t() + add)[0] - gates_t).abs().max().item()
    assert e3 < 2e-4 and e3 < 0.02 * e16 + 1e-7, (e3, e16)
    assert torch.equal(h3.cpu().view(torch.int16), split_image_ref(h_out.cpu()).view(torch.int16))     # (iii)
    i, f, g, o = gates3.chunk(4, 1)
    c, h = c3, h3ref
    # dropout mask + finished rows: the epilogue's other branch
    keep = (torch.rand(B, H, generator=torch.Generator().manual_seed(129)) > 0.3).to(torch.uint8)
    lens = torch.tensor([(5 if r % 3 else 2) for r in range(B)], dtype=torch.int32)
    nv.lstm_step_fwd(x3, list(widths), W3, H, B, gates, c_out, h_out, gin=dv(gin), bias=dv(bias), c_prev=dv(c_prev),
                     keep=keep.to(DEV), keep_scale=1.0 / 0.7, lens=lens.to(DEV), t=3, bf16=3, h16_out=h3)
    live = (3 < lens).double().unsqueeze(1)
    assert err(h_out, h * keep.double() / 0.7 * live) < 1e-5
    assert torch.equal(h3.cpu().view(torch.int16), split_image_ref(h_out.cpu()).view(torch.int16))
    # plain (dgrad-shaped) product with split-K
    N = 200
    W2 = rnd(N, K, seed=128, scale=0.1)
    W23 = torch.empty(N, 2 * K, device=DEV, dtype=torch.bfloat16)
    nv.split_bf16x3(dv(W2), W23)
    W2h, W2l = W2.bfloat16().double(), (W2 - W2.bfloat16().float()).bfloat16().double()
    ref3 = Xh @ W2h.t() + Xl @ W2h.t() + Xh @ W2l.t()
    for ns in (1, 2, 4):
        if (K // 64) % ns:
            continue
        Y = torch.empty(ns, B, N, device=DEV)
        nv.skinny_gemm(x3, list(widths), W23, N, B, Y, nsplit=ns, bf16=3)
        assert err(Y.sum(0), ref3) < 1e-5
        assert err(Y.sum(0), X.double() @ W2.double().t()) < 2e-5


@pytest.mark.parametrize("fused", [0, 1])
def test_attention_context_leaves_as_a_split_image_too(nv, fused):
    """t2amd_attn_fwd.ctx16_x3: K_c writes the split-bf16 image of the context next to the f32 context ('bf16x3' mode: the LSTM
    tiles' operand); the step itself is the exact-f32 form, bit-identical with and without the image."""
    B, Ti, E, Hq = 5, 150, 512, 1024
    sd, h, mem, pm, lens, w_prev, cum = _attn_inputs(B, Ti, E, Hq, 70)
    Wq = dv(sd['decoder.attention_layer.query_layer.linear_layer.weight'])
    Wd = dv(sd['decoder.attention_layer.location_layer.location_dense.linear_layer.weight'])
    Wc = dv(sd['decoder.attention_layer.location_layer.location_conv.conv.weight'])
    v = dv(sd['decoder.attention_layer.v.linear_layer.weight']).view(-1)
    U = torch.empty(128 * 62, device=DEV)
    nv.fold_location(Wd, Wc, U)
    lens32 = dv(lens.to(torch.int32))
    saved = nv.get_attn_fwd_fused() if hasattr(nv, 'get_attn_fwd_fused') else None
    nv.set_attn_fwd_fused(fused)
    try:
        outs = []
        for with_img in (False, True):
            ws = torch.zeros(nv.attn_fwd_ws_floats(B, Ti), device=DEV)
            cum_d, cum_save = dv(cum.clone()), torch.empty(B, Ti, device=DEV)
            w_out, ctx_out, q_out = torch.empty(B, Ti, device=DEV), torch.empty(B, E, device=DEV), torch.empty(B, 128, device=DEV)
            img = torch.full((B, 2 * E), float('nan'), device=DEV, dtype=torch.bfloat16) if with_img else None
            nv.attention_step_fwd(dv(h), Wq, U, v, dv(pm), dv(mem), lens32, dv(w_prev), cum_d, cum_save, w_out, ctx_out, q_out, ws,
                                  ctx_x3_out=img)
            torch.cuda.synchronize()
            outs.append((w_out, ctx_out, cum_d, img))
        for a, b in zip(outs[0][:3], outs[1][:3]):
            assert torch.equal(a, b)
        assert torch.equal(outs[1][3].cpu().view(torch.int16), split_image_ref(outs[1][1].cpu()).view(torch.int16))
    finally:
        nv.set_attn_fwd_fused(-1 if saved is None else saved)


def test_cell_backward_writes_the_split_image_of_the_gate_gradients(nv):
    B, H = 5, 64
    g = torch.sigmoid(rnd(B, 4 * H, seed=310)).to(DEV)
    c, cp, dh, dc = (rnd(B, H, seed=311 + k).to(DEV) for k in range(4))
    dg = torch.empty(B, 4 * H, device=DEV)
    dg3 = torch.full((B, 8 * H), float('nan'), device=DEV, dtype=torch.bfloat16)
    a = nv.lstm_bwd_desc(B, H, [dh], g, cp, c, None, 1.0, dc.clone(), dg, dgates16=dg3, x3=True)
    nv.lstm_pointwise_bwd2(a)
    torch.cuda.synchronize()
    assert torch.isfinite(dg).all()
    assert torch.equal(dg3.cpu().view(torch.int16), split_image_ref(dg.cpu()).view(torch.int16))
    # the plain bf16 copy is untouched by the new flag
    dg16 = torch.empty(B, 4 * H, device=DEV, dtype=torch.bfloat16)
    a = nv.lstm_bwd_desc(B, H, [dh], g, cp, c, None, 1.0, dc.clone(), dg, dgates16=dg16)
    nv.lstm_pointwise_bwd2(a)
    assert torch.equal(dg16.cpu(), dg.cpu().bfloat16())


# ------------------------------------------------------------------------------------------------
# bf16-resident product (csrc/gemm16.hip)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,pad", [(256, 256, 64, 0), (300, 520, 192, 64), (1024, 768, 1280, 0), (70, 33, 128, 8)])
def test_gemm16_tn_matches_f32_product_of_the_bf16_operands(nv, M, N, K, pad):
    """C = A . B^T with bf16 K-contiguous operands through LDS-DMA (256 x 256 x 64 tiles, ragged edges, padded row strides):
    products of bf16 values are exact in f32, so the result must equal an f32 matmul of the same operands to summation-order
    accuracy.  Asymmetric random operands: a transposed or mis-swizzled fragment cannot pass."""
    A = (rnd(M, K + pad, seed=300) * (1 + torch.arange(M).float().unsqueeze(1) / M)).bfloat16()
    B = (rnd(N, K + pad, seed=301) * (1 + torch.arange(K + pad).float().unsqueeze(0) / K)).bfloat16()
    ref = A[:, :K].float() @ B[:, :K].float().t()
    Ad, Bd = A.to(DEV), B.to(DEV)
    Cm = torch.full((M, N), float('nan'), device=DEV)
    nv.gemm16_tn(Cm, Ad[:, :K] if pad else Ad, Bd[:, :K] if pad else Bd)
    tol = 2e-6 * K ** 0.5 * float(ref.abs().max())
    assert (Cm.cpu() - ref).abs().max().item() < tol
    bias = rnd(N, seed=302)
    nv.gemm16_tn(Cm, Ad[:, :K] if pad else Ad, Bd[:, :K] if pad else Bd, accumulate=True, bias=dv(bias))
    assert (Cm.cpu() - (2 * ref + bias)).abs().max().item() < 2 * tol
    if K % 128 == 0:
        sk = 2
        part = torch.full((sk, M * N), float('nan'), device=DEV)
        nv.gemm16_tn(Cm, Ad[:, :K] if pad else Ad, Bd[:, :K] if pad else Bd, splitk=sk, partials=part)
        assert (part.view(sk, M, N).sum(0).cpu() - ref).abs().max().item() < tol


@pytest.mark.parametrize("M,N,K,pad", [(256, 256, 64, 0), (304, 520, 192, 8), (1024, 768, 1285, 0), (72, 40, 130, 24), (8, 8, 1, 0),
                                       (512, 264, 63, 0)])
def test_gemm16_kk_matches_f32_product_of_the_bf16_operands(nv, M, N, K, pad):
    """C = A^T . B with bf16 K-MAJOR operands ([K][M], [K][N]): LDS-DMA of 64 k-rows x 512 B, fragments by the transposing LDS
    read (ds_read_b64_tr_b16).  Products of bf16 values are exact in f32, so the result equals an f32 matmul of the same
    operands to summation-order accuracy.  Asymmetric operands (a ramp along m and along k): a transposed fragment, a wrong k
    order between A and B or a mis-swizzled slot cannot pass; K that is not a multiple of 64 exercises the zeroed tail rows
    (the rows behind K are NaN in A and B: nothing past K may be multiplied), padded row strides the clamped columns."""
    A = (rnd(K, M + pad, seed=330) * (1 + torch.arange(M + pad).float().unsqueeze(0) / M) * (1 + torch.arange(K).float().unsqueeze(1) / K)).bfloat16()
    B = (rnd(K, N + pad, seed=331) * (1 + 2 * torch.arange(N + pad).float().unsqueeze(0) / N)).bfloat16()
    ref = A[:, :M].float().t() @ B[:, :N].float()
    Ad = torch.full((K + 70, M + pad), float('nan'), dtype=torch.bfloat16, device=DEV)
    Bd = torch.full((K + 70, N + pad), float('nan'), dtype=torch.bfloat16, device=DEV)
    Ad[:K] = A.to(DEV)
    Bd[:K] = B.to(DEV)
    Cm = torch.full((M, N), float('nan'), device=DEV)
    nv.gemm16_kk(Cm, Ad[:K, :M], Bd[:K, :N], K)
    tol = 2e-6 * max(K, 16) ** 0.5 * float(ref.abs().max())
    assert (Cm.cpu() - ref).abs().max().item() < tol
    bias = rnd(N, seed=332)
    nv.gemm16_kk(Cm, Ad[:K, :M], Bd[:K, :N], K, accumulate=True, bias=dv(bias))
    assert (Cm.cpu() - (2 * ref + bias)).abs().max().item() < 2 * tol
    for sk in (2, 3):
        if K < 64 * sk:
            continue
        part = torch.full((sk, M * N), float('nan'), device=DEV)
        nv.gemm16_kk(Cm, Ad[:K, :M], Bd[:K, :N], K, splitk=sk, partials=part)
        assert (part.view(sk, M, N).sum(0).cpu() - ref).abs().max().item() < tol
        wide = torch.full((M, N + 24), 5.0, device=DEV)                      # the partials reduced into a column block
        nv.splitk_reduce2d(part, sk, wide[:, 16:16 + N])
        assert (wide[:, 16:16 + N].cpu() - ref).abs().max().item() < tol
        assert torch.all(wide[:, :16].cpu() == 5.0) and torch.all(wide[:, 16 + N:].cpu() == 5.0)
        nv.splitk_reduce2d(part, sk, wide[:, 16:16 + N], accumulate=True)
        assert (wide[:, 16:16 + N].cpu() - 2 * ref).abs().max().item() < 2 * tol


def test_gemm16_kk_shifted_rows_are_a_row_offset(nv):
    """dG[B:]^T . x[:rows - B] -- the weight gradient of an input the step reads from the PREVIOUS time step -- is the same
    product with A starting B rows further down: no copy, no padded image."""
    rows, Bsz, G4, W = 640, 64, 264, 136
    dG = rnd(rows, G4, seed=340).bfloat16()
    x = rnd(rows, W, seed=341).bfloat16()
    ref = dG[Bsz:].float().t() @ x[:rows - Bsz].float()
    Cm = torch.full((G4, W), float('nan'), device=DEV)
    nv.gemm16_kk(Cm, dG.to(DEV)[Bsz:], x.to(DEV), rows - Bsz)
    assert (Cm.cpu() - ref).abs().max().item() < 2e-6 * rows ** 0.5 * float(ref.abs().max())


@pytest.mark.parametrize("sk", [1, 3])
def test_gemm16_kk_group_shares_one_launch(nv, sk):
    """The input blocks of an LSTM's weight gradient as ONE launch: three products with their own widths, row offsets (the
    block read from the previous time step starts Bsz rows further down in dG), K and destinations (column blocks of two
    matrices) -- each equals its own f32 product; widths that are not multiples of the tile (the last column tile of a problem
    must not spill into the next problem's columns)."""
    rows, Bsz, G4 = 777, 8, 520
    widths = (264, 40, 512)
    dG = rnd(rows, G4, seed=360).bfloat16()
    xs = [(rnd(rows, w, seed=361 + i) * (1 + torch.arange(w).float().unsqueeze(0) / w)).bfloat16() for i, w in enumerate(widths)]
    shifted = (False, True, True)
    dGd = dG.to(DEV)
    dWih = torch.full((G4, widths[0] + widths[1]), float('nan'), device=DEV)
    dWhh = torch.full((G4, widths[2]), float('nan'), device=DEV)
    dests = (dWih[:, :widths[0]], dWih[:, widths[0]:], dWhh)
    probs, parts = [], []
    for x, sh, dst, w in zip(xs, shifted, dests, widths):
        A = dGd[Bsz:] if sh else dGd
        K = rows - Bsz if sh else rows
        if sk == 1:
            probs.append(dict(Cm=dst, A16=A, B16=x.to(DEV), K=K, M=G4, N=w))
        else:
            pt = torch.full((sk, G4 * w), float('nan'), device=DEV)
            parts.append(pt)
            probs.append(dict(Cm=pt[0].view(G4, w), A16=A, B16=x.to(DEV), K=K, M=G4, N=w, splitk=sk, partials=pt))
    nv.gemm16_kk_group(probs)
    for pt, dst in zip(parts, dests):
        nv.splitk_reduce2d(pt, sk, dst)
    for x, sh, dst in zip(xs, shifted, dests):
        ref = (dG[Bsz:].float().t() @ x[:rows - Bsz].float()) if sh else (dG.float().t() @ x.float())
        assert (dst.cpu() - ref).abs().max().item() < 2e-6 * rows ** 0.5 * float(ref.abs().max())
    with pytest.raises(Exception):
        nv.gemm16_kk_group(probs + probs)                                  # more than four problems


@pytest.mark.parametrize("B,T,Ci,Co,k", [(3, 37, 64, 96, 5), (5, 200, 128, 304, 5), (2, 9, 64, 64, 3), (4, 50, 80, 512, 5), (4, 50, 512, 80, 5)])
def test_gemm16_kk_conv_weight_gradient_over_the_halo_images(nv, B, T, Ci, Co, k):
    """dW[co][tap][ci] = sum_{b,t} g[b,t,co] x[b,t+tap-pad,ci] as ONE K-major product over the two bf16 halo images the
    convolution's forward and data gradient already make: A = g's image from row `pad` on ([K][Co]), B[k][n] = x's image
    flat[k Ci + n], n < k Ci (overlapping rows: ldb = Ci).  Equals autograd of F.conv1d on the bf16-rounded operands; the halos
    keep utterances apart."""
    import torch.nn.functional as F
    pad = (k - 1) // 2
    Tp = T + 2 * pad
    x = rnd(B * T, Ci, seed=350)
    g = rnd(B * T, Co, seed=351)
    xb = x.bfloat16().float().view(B, T, Ci).transpose(1, 2).requires_grad_(False)
    Wt = torch.zeros(Co, Ci, k, requires_grad=True)
    F.conv1d(xb, Wt, None, padding=pad).backward(g.bfloat16().float().view(B, T, Co).transpose(1, 2))
    ref = Wt.grad.permute(0, 2, 1).reshape(Co, k * Ci)                    # [co][tap Ci + ci]
    ximg = torch.full((B * Tp + 2 * pad, Ci), float('nan'), dtype=torch.bfloat16, device=DEV)
    gimg = torch.full((B * Tp + 2 * pad, Co), float('nan'), dtype=torch.bfloat16, device=DEV)
    nv.cast_halo_bf16(dv(x), ximg, T, pad)
    nv.cast_halo_bf16(dv(g), gimg, T, pad)
    K = B * Tp
    dW = torch.full((Co, k * Ci), float('nan'), device=DEV)
    nv.gemm16_kk(dW, gimg[pad:], ximg, K, M=Co, N=k * Ci, lda=Co, ldb=Ci)
    tol = 3e-6 * (B * T) ** 0.5 * float(ref.abs().max())
    assert (dW.cpu() - ref).abs().max().item() < tol
    sk = 2
    part = torch.full((sk, Co * k * Ci), float('nan'), device=DEV)
    nv.gemm16_kk(dW, gimg[pad:], ximg, K, M=Co, N=k * Ci, lda=Co, ldb=Ci, splitk=sk, partials=part)
    out = torch.empty(Co, Ci, k, device=DEV)
    nv.splitk_reduce(part, sk, out, perm_taps=k, perm_ci=Ci)              # (co, tap, ci) -> torch's (co, ci, tap)
    assert (out.cpu() - Wt.grad).abs().max().item() < tol


@pytest.mark.parametrize("rows,cols,rpad", [(130, 70, 192), (64, 64, 64), (1000, 257, 1024), (512, 256, 576)])
def test_transpose_cast_bf16(nv, rows, cols, rpad):
    src = rnd(rows, cols + 3, seed=310)[:, :cols]                       # a row stride that is not the width: scalar paths
    src4 = rnd(rows, ((cols + 3) // 4) * 4, seed=311)[:, :cols]         # 16-byte-aligned rows: the vector paths
    for s in (src, src.bfloat16(), src4, src4.bfloat16()):
        dst = torch.full((cols, rpad + 8), 7.0, device=DEV, dtype=torch.bfloat16)
        nv.transpose_cast_bf16(s.to(DEV), dst[:, :rpad])
        want = torch.zeros(cols, rpad, dtype=torch.bfloat16)
        want[:, :rows] = s.bfloat16().t()
        assert torch.equal(dst[:, :rpad].cpu(), want)
        assert torch.all(dst[:, rpad:].cpu() == 7.0)


@pytest.mark.parametrize("B,T,Ci,Co,k", [(3, 37, 64, 96, 5), (5, 200, 128, 300, 5), (2, 9, 64, 64, 3), (4, 60, 80, 512, 5), (3, 41, 512, 80, 5),
                                         (2, 5, 40, 40, 3)])
def test_conv16_window_product_matches_conv1d(nv, B, T, Ci, Co, k):
    """nn.Conv1d over channel-last rows as a product of sliding windows of a bf16 image with zero halo rows (gemm16 window
    mode): equals F.conv1d on the bf16-rounded operands, utterance by utterance (nothing leaks across the halos), with
    bias and accumulation; and the tap-reversed weights give the data gradient."""
    import torch.nn.functional as F
    pad = (k - 1) // 2
    x = rnd(B * T, Ci, seed=320)
    W = rnd(Co, Ci, k, seed=321, scale=0.2)
    bias = rnd(Co, seed=322)
    xb, Wb = x.bfloat16().float(), W.bfloat16().float()
    ref = F.conv1d(xb.view(B, T, Ci).transpose(1, 2), Wb, bias, padding=pad).transpose(1, 2).reshape(B * T, Co)
    img = torch.full((B * (T + 2 * pad) + 2 * pad, Ci), float('nan'), dtype=torch.bfloat16, device=DEV)   # poisoned: the kernel
    nv.cast_halo_bf16(dv(x), img, T, pad)                                                                # writes the halos itself
    halo = torch.ones(B * (T + 2 * pad) + 2 * pad, dtype=torch.bool)
    for b_ in range(B):
        halo[b_ * (T + 2 * pad) + pad:b_ * (T + 2 * pad) + pad + T] = False
    assert torch.all(img.cpu()[halo] == 0) and torch.isfinite(img.float()).all()
    Wp16 = nv.pack_conv_bf16(dv(W))                                                       # [co][tap Ci + ci], zero behind k Ci
    Kp = (k * Ci + 63) // 64 * 64
    want = torch.zeros(Co, Kp, dtype=torch.bfloat16)
    want[:, :k * Ci] = W.permute(0, 2, 1).reshape(Co, k * Ci).bfloat16()
    assert torch.equal(Wp16.cpu(), want)
    y = torch.full((B * T, Co), float('nan'), device=DEV)
    nv.conv16(y, img, Wp16, B, T, pad, bias=dv(bias))
    tol = 3e-6 * (k * Ci) ** 0.5 * float(ref.abs().max())
    assert (y.cpu() - ref).abs().max().item() < tol
    nv.conv16(y, img, Wp16, B, T, pad, accumulate=True)
    assert (y.cpu() - (2 * ref - bias)).abs().max().item() < 2 * tol
    # data gradient of the same convolution: g (rows, Co) -> dx (rows, Ci)
    g = rnd(B * T, Co, seed=323)
    gb = g.bfloat16().float()
    dref = F.conv_transpose1d(gb.view(B, T, Co).transpose(1, 2), Wb, padding=pad).transpose(1, 2).reshape(B * T, Ci)
    if Co % 8 == 0:
        gimg = torch.full((B * (T + 2 * pad) + 2 * pad, Co), float('nan'), dtype=torch.bfloat16, device=DEV)
        nv.cast_halo_bf16(dv(g), gimg, T, pad)
        Wd16 = nv.pack_conv_bf16(dv(W), reversed=True)                                    # [ci][(k - 1 - tap) Co + co]
        wantd = torch.zeros(Ci, (k * Co + 63) // 64 * 64, dtype=torch.bfloat16)
        wantd[:, :k * Co] = W.flip(2).permute(1, 2, 0).reshape(Ci, k * Co).bfloat16()
        assert torch.equal(Wd16.cpu(), wantd)
        dx = torch.full((B * T, Ci), float('nan'), device=DEV)
        nv.conv16(dx, gimg, Wd16, B, T, pad)
        assert (dx.cpu() - dref).abs().max().item() < 3e-6 * (k * Co) ** 0.5 * float(dref.abs().max())
