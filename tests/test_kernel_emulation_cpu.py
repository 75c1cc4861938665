"""The element-wise HIP kernels of csrc/audio.hip and csrc/optim.hip EXECUTED ON THE CPU, from the same source the
GPU compiles, through a host stand-in for the HIP runtime (tests/hip_emu: one OS thread per work-item, pthread
barriers for __syncthreads / wave shuffles).  Test infrastructure only: nothing here is reachable from the package.

Why: these kernels were written when no GPU was at hand.  Running their bodies and their launch geometry on the CPU
against the oracle (mel front end: the fixture made by the reference's stft.py / layers.py; optimiser: the pair the
reference calls, clip_grad_norm_ + torch.optim.Adam) pins index arithmetic, bounds, chunking, reductions and
rounding before the `-m gpu` tests run the real thing.  The two GEMMs of the mel path are MFMA kernels and are NOT
emulated: torch.matmul stands in for them here (they have their own GPU unit tests).
"""
import contextlib
import ctypes
import os
import sys

import pytest
import torch

import golden_util as gu
from tacotron2_amd import native

sys.path.insert(0, os.path.join(gu.ROOT, "tests", "hip_emu"))


@contextlib.contextmanager
def emulated_kernels():
    import build_emu
    emu = ctypes.CDLL(build_emu.build())
    assert emu.t2amd_emulated() == 1
    for name, at in native._argtypes().items():
        if hasattr(emu, name):
            fn = getattr(emu, name)
            fn.argtypes, fn.restype = at, ctypes.c_int
    emu.t2amd_last_error.restype = ctypes.c_char_p
    emu.t2amd_reflect_index.argtypes, emu.t2amd_reflect_index.restype = [ctypes.c_longlong] * 2, ctypes.c_longlong
    saved = (native._lib, native._validate_only, native.gemm, native.transpose)

    def gemm_standin(Cm, A, B, batch=1, strides=(0, 0, 0), **kw):          # C[M,N] = A[M,K] . B[N,K]^T per batch item
        assert not kw, kw
        for b in range(batch):
            a = torch.as_strided(A, A.shape, A.stride(), A.storage_offset() + b * strides[0])
            w = torch.as_strided(B, B.shape, B.stride(), B.storage_offset() + b * strides[1])
            c = torch.as_strided(Cm, Cm.shape, Cm.stride(), Cm.storage_offset() + b * strides[2])
            c.copy_(a @ w.t())

    def transpose_standin(dst, src, batch=1, sstride=0, dstride=0):
        for b in range(batch):
            s = torch.as_strided(src, src.shape, src.stride(), src.storage_offset() + b * sstride)
            d = torch.as_strided(dst, dst.shape, dst.stride(), dst.storage_offset() + b * dstride)
            d.copy_(s.t())

    native._lib, native._validate_only = emu, True                        # CPU pointers allowed, kernels DO run (emulated)
    native.gemm, native.transpose = gemm_standin, transpose_standin
    try:
        yield emu
    finally:
        native._lib, native._validate_only, native.gemm, native.transpose = saved


def _golden(name):
    return torch.load(os.path.join(gu.GOLDEN_DIR, name), weights_only=False)


def test_reflect_pad_kernel_matches_torch():
    with emulated_kernels():
        for B, T, pad in ((2, 777, 64), (1, 300, 299), (3, 1000, 0)):
            y = torch.randn(B, T)
            Tout = (T + 2 * pad + 3) // 4 * 4 + 4
            out = torch.full((B, Tout), 7.0)
            native.reflect_pad(y, out, pad)
            ref = torch.nn.functional.pad(y.view(B, 1, T), (pad, pad), mode='reflect').view(B, -1) if pad else y
            assert torch.equal(out[:, :T + 2 * pad], ref)
            assert (out[:, T + 2 * pad:] == 0).all()                      # row slack is zeroed
        with pytest.raises(native.NativeError, match="smaller than the signal"):
            native.reflect_pad(torch.zeros(1, 8), torch.zeros(1, 24), 8)


def test_magnitude_and_log_kernels_match_torch():
    with emulated_kernels():
        rows, F, Fpad = 70, 33, 48
        spec = torch.randn(rows, 2 * F + 3)                                # row stride larger than 2F
        mag = torch.full((rows, Fpad), -1.0)
        native.stft_magnitude(spec, mag, F)
        a, b = spec[:, :F].numpy(), spec[:, F:2 * F].numpy()
        import numpy as np
        assert np.array_equal(mag[:, :F].numpy(), np.sqrt(a * a + b * b))   # IEEE mul, add, sqrt, each rounded once
        ref = torch.sqrt(spec[:, :F] ** 2 + spec[:, F:2 * F] ** 2)          # torch's vectorised sqrt: within an ulp of that
        assert ((mag[:, :F] - ref).abs() <= 1.2e-7 * ref).all() and (mag[:, F:] == 0).all()
        B, n, n_mel = 2, 300, 5
        mel = torch.rand(B * n, n_mel) * 1e-4                              # straddles the 1e-5 clamp
        out = torch.empty(B, n_mel, n)
        native.mel_log_compress(mel, out, 1e-5)
        ref = torch.log(torch.clamp(mel, min=1e-5)).view(B, n, n_mel).transpose(1, 2)
        assert (out - ref).abs().max().item() < 1e-6


def test_mel_front_end_host_path_with_emulated_kernels_matches_reference_fixture():
    """tacotron2_amd.audio.TacotronSTFT end to end on the CPU: its own host code (padded row length, the overlapping
    frame view with row stride hop, batch strides, padded mel basis), the three emulated kernels, torch.matmul for
    the two GEMMs — against the reference's mel (tests/golden/audio_demo.pt).  Bound: the float32 summation-order
    noise of this pipeline measured on CPU (max 2e-5) with margin."""
    from tacotron2_amd.audio import TacotronSTFT
    g = _golden("audio_demo.pt")
    with emulated_kernels():
        stft = TacotronSTFT().cpu()
        for key_in, key_out in (("y", "mel"), ("y_odd", "mel_odd")):
            out = stft.mel_spectrogram(g[key_in])
            assert tuple(out.shape) == tuple(g[key_out].shape)
            d = (out - g[key_out]).abs()
            assert d.max().item() < 1e-4 and d.mean().item() < 2e-6, (d.max().item(), d.mean().item())
        mag = stft.stft_fn.transform_magnitude(g["y"])
        assert torch.allclose(mag.sum(dim=1), g["mag_row_sums"], rtol=1e-4, atol=1e-3)
        hp = dict(filter_length=512, hop_length=128, win_length=400, n_mel_channels=40, sampling_rate=16000,
                  mel_fmin=50.0, mel_fmax=7600.0)
        from oracle import audio_oracle as ao
        y = g["y"][:, :5000]
        d = (TacotronSTFT(**hp).cpu().mel_spectrogram(y) - ao.mel_spectrogram(y, **hp)).abs()
        assert d.max().item() < 1e-4, d.max().item()


def test_fused_adam_kernels_emulated_match_torch_clip_and_adam():
    """optim.FusedAdam with its two kernels emulated, against torch.nn.utils.clip_grad_norm_ + torch.optim.Adam on
    the same tensors: chunk edges, a 1-element tensor, views at odd offsets of a flat buffer, clip active / inactive /
    absent.  Same tolerances as the GPU test (tests/test_zz3_optim_gpu.py)."""
    from tacotron2_amd.optim import FusedAdam
    gen = torch.Generator().manual_seed(21)
    sizes = [(4095,), (4096,), (4097,), (1,), (7, 13), (129, 257), (3, 5, 31)]
    numel = [int(torch.tensor(s).prod()) for s in sizes]
    total = sum(numel) + len(sizes) + 3
    flat_p = torch.randn(total, generator=gen) * 0.3
    flat_e = flat_p.clone()
    ref_params, emu_params, off = [], [], 1
    for s, n in zip(sizes, numel):
        ref_params.append(torch.nn.Parameter(flat_p[off:off + n].clone().view(s)))
        emu_params.append(torch.nn.Parameter(flat_e[off:off + n].view(s)))
        off += n + 1
    ref = torch.optim.Adam(ref_params, lr=1e-3, weight_decay=1e-6)
    with emulated_kernels():
        opt = FusedAdam(emu_params, lr=1e-3, weight_decay=1e-6)
        for it, (scale, clip) in enumerate([(5.0, 1.0), (1e-3, 1.0), (1.0, None)]):
            flat_g, off = torch.randn(total, generator=gen) * scale, 1
            for pr, pe, s, n in zip(ref_params, emu_params, sizes, numel):
                pr.grad = flat_g[off:off + n].clone().view(s)
                pe.grad = flat_g[off:off + n].view(s)
                off += n + 1
            if clip is not None:
                n_ref = torch.nn.utils.clip_grad_norm_(ref_params, clip)
            ref.step()
            n_got = opt.step(clip_norm=clip)
            if clip is not None:
                assert abs(float(n_got) - float(n_ref)) <= 1e-5 * float(n_ref), (it, float(n_got), float(n_ref))
                coef = float(opt._norm[1])
                assert abs(coef - min(1.0, clip / (float(n_ref) + 1e-6))) <= 1e-5 * coef
            for pr, pe in zip(ref_params, emu_params):
                d = (pe.detach() - pr.detach()).abs().max().item()
                assert d < 3e-7, (it, tuple(pr.shape), d)
                for key in ("exp_avg", "exp_avg_sq"):
                    a, r = opt.state[pe][key], ref.state[pr][key]
                    # relative to the tensor's scale: a moment that cancels to ~0 has no relative precision of its own
                    assert ((a - r).abs() <= 1e-5 * r.abs() + 1e-6 * r.abs().max()).all(), (it, key, tuple(pr.shape))
        off = 1                                                            # neighbours in the flat buffer untouched
        for n in numel:
            assert float(flat_e[off - 1]) == float(flat_p[off - 1]) and float(flat_e[off + n]) == float(flat_p[off + n])
            off += n + 1


@pytest.mark.parametrize("B,H,first,with_keep,splits", [(3, 8, False, True, (2, 4, 1)), (5, 64, True, False, (1, 0, 0)),
                                                        (2, 260, False, True, (6, 1, 3))])
def test_emulated_lstm_cell_backward_matches_autograd(B, H, first, with_keep, splits):
    """csrc/cell_bwd.h -- the cell backward shared by lstm_pointwise_bwd_kernel and the folded attention backward -- run
    on the CPU from the GPU's source against torch autograd of the reference's arithmetic: nn.LSTMCell's cell
    (model.py:351-352, 366-370) followed by F.dropout (:353, :371), in float64.  Inputs as the loops present them: the
    upstream gradient of the dropped-out h as up to three addends, each a sum of partial slabs (split-K dgrad outputs) at
    a column offset, the carried dL/dc, activated gates and cell states saved by the forward, a byte keep-mask."""
    import build_emu
    emu = ctypes.CDLL(build_emu.build())
    emu.t2amd_emu_cell_bwd.argtypes, emu.t2amd_emu_cell_bwd.restype = [ctypes.POINTER(native.LstmBwd)], ctypes.c_int
    g = torch.Generator().manual_seed(11 + H)
    rnd = lambda *s: torch.randn(*s, generator=g)
    pre = rnd(B, 4 * H).double().requires_grad_(True)
    c_prev = (torch.zeros(B, H) if first else rnd(B, H)).double().requires_grad_(True)
    keep = (torch.rand(B, H, generator=g) > 0.3).to(torch.uint8) if with_keep else None
    scale = native.scale_for(0.3) if with_keep else 1.0
    i, f, gg, o = pre[:, :H].sigmoid(), pre[:, H:2 * H].sigmoid(), pre[:, 2 * H:3 * H].tanh(), pre[:, 3 * H:].sigmoid()
    c = f * c_prev + i * gg
    h = o * c.tanh()
    hd = h * keep.double() * scale if with_keep else h
    # three addends of the gradient wrt the dropped-out h: n partial slabs each, read at a column offset of a wider slab
    addends, dh_total = [], torch.zeros(B, H, dtype=torch.float64)
    for n, off in zip(splits, (4, 0, 8)):
        if n == 0:
            addends.append(None)
            continue
        slab = rnd(n, B, H + off).contiguous()
        addends.append((slab[0, :, off:], n, slab.stride(0)))
        dh_total = dh_total + slab[:, :, off:].double().sum(0)
    dc_in = rnd(B, H)
    # the loss whose gradients the cell backward forms: upstream dh on the dropped-out h, the carried dL/dc on c
    ((hd * dh_total).sum() + (c * dc_in.double()).sum()).backward(inputs=[pre, c_prev])
    gates = torch.cat([i, f, gg, o], 1).detach().float().contiguous()
    dc = dc_in.clone()
    dgates = torch.full((B, 4 * H), float('nan'))
    dg16 = torch.zeros(B, 4 * H, dtype=torch.bfloat16)
    saved = native._validate_only
    native._validate_only = True                                 # descriptors over CPU tensors (the emulated kernel reads them)
    c32, cp32 = c.detach().float().contiguous(), c_prev.detach().float().contiguous()   # (the descriptor holds raw pointers)
    try:
        desc = native.lstm_bwd_desc(B, H, addends, gates, None if first else cp32, c32, keep, scale, dc, dgates, dgates16=dg16)
    finally:
        native._validate_only = saved
    assert emu.t2amd_emu_cell_bwd(ctypes.byref(desc)) == 0
    err = lambda a, b: ((a.double() - b).abs().max() / (b.abs().max() + 1e-30)).item()
    assert err(dgates, pre.grad) < 2e-6
    assert err(dc, c_prev.grad) < 2e-6                            # the new carry: dL/dc_{t-1}
    assert torch.equal(dg16, dgates.bfloat16())                  # the dgrad GEMM's bf16 operand: round-to-nearest-even
