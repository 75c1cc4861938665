"""CPU coverage of the rows widened after the hot path: mel front end oracle + tables,
batch collation / dataset, training driver (checkpoint, warm start, resume, validation) and launcher.

No GPU arithmetic here: the oracle is checked against the fixtures made from the reference
(tests/golden/make_golden_audio.py); product host code runs in the library's validate-only mode
(every argument check and host loop runs, no kernel is launched, values are meaningless)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

import golden_util as gu
from tacotron2_amd import native
from tacotron2_amd.hparams import create_hparams


def _golden(name):
    return torch.load(os.path.join(gu.GOLDEN_DIR, name), weights_only=False)


# ---- mel front end ------------------------------------------------------------------------------
def test_audio_oracle_matches_reference_fixture():
    from oracle import audio_oracle as ao
    g = _golden("audio_demo.pt")
    assert torch.equal(ao.mel_spectrogram(g["y"]), g["mel"])                 # fixture = reference stft.py/layers.py
    assert torch.equal(ao.mel_spectrogram(g["y_odd"]), g["mel_odd"])
    assert g["mel"].shape == (2, 80, 9000 // 256 + 1) and g["mel_odd"].shape == (1, 80, 4321 // 256 + 1)
    assert torch.equal(ao.stft_magnitude(g["y"]).sum(dim=1), g["mag_row_sums"])


def test_product_tables_match_oracle_and_reference_digest():
    from oracle import audio_oracle as ao
    from tacotron2_amd import audio
    g = _golden("audio_demo.pt")
    fb = audio.fourier_basis(1024, 1024)
    assert fb.shape == (1026, 1024) and fb.dtype == np.float32
    assert np.abs(fb - ao.forward_basis(1024, 1024)[:, 0, :].numpy()).max() <= 2 ** -23
    assert torch.allclose(torch.from_numpy(fb).double().sum(dim=1), g["basis_digest"], atol=1e-4)
    # win_length < filter_length: window centred in the frame
    fb2 = audio.fourier_basis(16, 8)
    assert np.abs(fb2 - ao.forward_basis(16, 8)[:, 0, :].numpy()).max() <= 2 ** -23
    assert np.all(fb2[:, :4] == 0) and np.all(fb2[:, 12:] == 0)
    for args in [(22050, 1024, 80, 0.0, 8000.0), (16000, 512, 40, 50.0, None), (22050, 1024, 128, 0.0, None)]:
        a, b = audio.mel_filterbank(*args), ao.librosa_mel(*args)
        assert a.shape == b.shape and np.abs(a - b).max() < 1e-14


def test_mel_filterbank_structure():
    """The table has no reference artefact to be pinned to (librosa is absent): check what the published
    definition implies — triangles on the Slaney scale, unit area in Hz, linear spacing below 1 kHz."""
    from tacotron2_amd import audio
    fb = audio.mel_filterbank(22050, 1024, 80, 0.0, 8000.0)
    assert fb.shape == (80, 513) and (fb >= 0).all()
    hz = np.linspace(0, 22050 / 2, 513)
    assert fb[:, hz > 8000.0 + 22050 / 1024].max() == 0.0
    peaks = hz[fb.argmax(axis=1)]
    assert np.all(np.diff(peaks) > 0)
    for row in fb:                                           # single-peaked
        nz = np.nonzero(row)[0]
        assert np.all(np.diff(nz) == 1)
        k = row.argmax()
        assert np.all(np.diff(row[nz[0]:k + 1]) >= 0) and np.all(np.diff(row[k:nz[-1] + 1]) <= 0)
    edges = audio._mel_to_hz(np.linspace(audio._hz_to_mel(0.0), audio._hz_to_mel(8000.0), 82))
    low = edges[edges < 1000.0]
    assert np.allclose(np.diff(low), np.diff(low)[0])        # linear part
    assert abs(audio._hz_to_mel(1000.0) - 15.0) < 1e-12 and abs(audio._mel_to_hz(15.0) - 1000.0) < 1e-9
    # area normalisation: integral of each triangle over Hz is 1 (sampled on the FFT grid -> approximately)
    area = fb.sum(axis=1) * (hz[1] - hz[0])
    assert np.all(np.abs(area[20:] - 1.0) < 0.15)


def test_mel_filterbank_against_an_independent_public_implementation():
    """The one table of the front end with no artefact of the reference to be held against (reference layers.py:50-53 calls
    librosa.filters.mel, which is not installed and not vendored): the oracle's restatement of librosa 0.6.0 (Slaney scale,
    norm=1) is checked against an INDEPENDENT public implementation that is in this image -- Hugging Face
    transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney"), documented as equivalent to librosa's -- on the
    reference's configuration (hparams.py:36-42: 22,050 Hz, n_fft 1024, 80 mels, 0-8000 Hz) and on two others: equal to 1e-12.
    The product's table (tacotron2_amd.audio) is held to the oracle's by test_product_tables_match_oracle_and_reference_digest."""
    audio_utils = pytest.importorskip("transformers.audio_utils")
    import numpy as np
    from oracle import audio_oracle as ao
    for sr, n_fft, n_mels, fmin, fmax in ((22050, 1024, 80, 0.0, 8000.0), (16000, 512, 40, 50.0, 7600.0), (22050, 2048, 128, 0.0, None)):
        ref = audio_utils.mel_filter_bank(num_frequency_bins=1 + n_fft // 2, num_mel_filters=n_mels, min_frequency=fmin,
                                          max_frequency=fmax if fmax is not None else sr / 2.0, sampling_rate=sr,
                                          norm="slaney", mel_scale="slaney")
        mine = np.asarray(ao.librosa_mel(sr, n_fft, n_mels, fmin, fmax), dtype=np.float64)
        assert mine.shape == (n_mels, 1 + n_fft // 2) and ref.shape == mine.shape[::-1]
        assert float(np.abs(ref.T - mine).max()) < 1e-12 * max(1.0, float(np.abs(mine).max()))


def test_reflect_index_rule(native_lib):
    T, pad = 37, 9
    ref = torch.nn.functional.pad(torch.arange(T, dtype=torch.float32).view(1, 1, T), (pad, pad), mode='reflect').view(-1)
    got = [native.reflect_index(i - pad, T) for i in range(T + 2 * pad)]
    assert got == [int(v) for v in ref.tolist()]


def test_audio_host_plumbing_validate_only(native_lib):
    from tacotron2_amd.audio import TacotronSTFT
    native.set_validate_only(True)
    try:
        stft = TacotronSTFT().cpu()
        assert set(dict(stft.named_buffers())) >= {"mel_basis", "stft_fn.forward_basis"}
        assert stft.mel_basis.shape == (80, 513) and stft.stft_fn.forward_basis.shape == (1026, 1, 1024)
        for T in (9000, 4321, 513):
            out = stft.mel_spectrogram(torch.zeros(3, T))
            assert out.shape == (3, 80, T // 256 + 1)
        assert stft.stft_fn.transform_magnitude(torch.zeros(2, 2048)).shape == (2, 513, 9)
        with pytest.raises(ValueError):
            stft.mel_spectrogram(torch.zeros(1, 512))        # cannot reflect-pad by 512
        with pytest.raises(ValueError):
            stft.mel_spectrogram(torch.zeros(4000))
    finally:
        native.set_validate_only(False)
    with pytest.raises(native.NativeError):                  # no CPU path
        TacotronSTFT().cpu().mel_spectrogram(torch.zeros(1, 4096))


def test_audio_argument_errors(native_lib):
    rc = native_lib.t2amd_reflect_pad_f32(None, 0, None, 0, 1, 10, 2, 14, None)
    assert rc == 1 and b"null operand" in native_lib.t2amd_last_error()


# ---- data path ----------------------------------------------------------------------------------
def test_collate_matches_reference_fixture():
    from tacotron2_amd.data_utils import TextMelCollate
    g = _golden("collate.pt")
    for r, ref in g["collated"].items():
        out = TextMelCollate(r)(g["items"])
        assert len(out) == 5
        for a, b in zip(out, ref):
            assert a.dtype == b.dtype and torch.equal(a, b)
        assert out[2].shape[2] % r == 0
    text, il, mel, gate, ol = TextMelCollate(1)(g["items"])
    assert il.tolist() == sorted(il.tolist(), reverse=True)
    for b in range(len(ol)):
        assert gate[b, :ol[b] - 1].sum() == 0 and gate[b, ol[b] - 1:].min() == 1
        assert mel[b, :, ol[b]:].abs().sum() == 0 and text[b, il[b]:].sum() == 0


def test_loader_filelist_shuffle_npy_and_synthetic(tmp_path, monkeypatch):
    monkeypatch.setitem(sys.modules, "text", None)           # no text frontend importable, whatever ran before
    from tacotron2_amd.data_utils import TextMelLoader, TextMelCollate
    g = _golden("collate.pt")
    lines = []
    for i in range(50):
        p = tmp_path / ("f%d.npy" % i)
        np.save(p, np.full((80, 5 + i % 7), float(i), dtype=np.float32))
        lines.append("%s|ids: %s" % (p, " ".join(str(1 + (i + k) % 147) for k in range(3 + i % 5))))
    fl = tmp_path / "list.txt"
    fl.write_text("\n".join(lines) + "\n", encoding="utf-8")
    hp = create_hparams("load_mel_from_disk=True")
    ds = TextMelLoader(str(fl), hp)
    order = [int(os.path.basename(r[0])[1:-4]) for r in ds.audiopaths_and_text]
    assert order == g["shuffle_1234"]                        # reference: random.seed(1234); random.shuffle
    text, mel = ds[0]
    i = order[0]
    assert text.dtype == torch.int32 and text.tolist() == [1 + (i + k) % 147 for k in range(3 + i % 5)]
    assert mel.shape == (80, 5 + i % 7) and float(mel[0, 0]) == float(i)
    np.save(tmp_path / "bad.npy", np.zeros((40, 5), dtype=np.float32))
    with pytest.raises(AssertionError):
        ds.get_mel(str(tmp_path / "bad.npy"))
    with pytest.raises(RuntimeError):
        ds.get_text("plain words need a text frontend")
    with pytest.raises(RuntimeError):
        ds.get_text("1984")                                  # digits are text, not symbol id 1984 (explicit 'ids:' opt-in)
    with pytest.raises(ValueError):
        ds.get_text("ids: 5 1984")                           # out of range for the embedding
    ds2 = TextMelLoader(str(fl), hp, text_to_sequence=lambda t, cleaners: [len(t), len(cleaners)])
    assert ds2.get_text("abc").tolist() == [3, 1]
    syn = TextMelLoader("synthetic:6:7", create_hparams())
    assert len(syn) == 6
    a, b = syn[2], syn[2]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[1].shape[0] == 80
    assert 15 <= a[0].numel() <= 187 and 90 <= a[1].shape[1] <= 870 and int(a[0].min()) >= 1
    batch = TextMelCollate(1)([syn[i] for i in range(6)])
    assert batch[0].shape[0] == 6 and batch[2].shape[1] == 80


# ---- training driver ----------------------------------------------------------------------------
class _FiniteLoss(torch.nn.Module):
    """Validate-only outputs are uninitialised memory: a criterion that is finite whatever they hold."""

    def forward(self, out, targets):
        return sum(torch.nan_to_num(o, nan=0.0, posinf=0.0, neginf=0.0).clamp(-1.0, 1.0).sum() for o in out[:3]) * 0.0 + 1.0


def _finite_clip(params, max_norm):
    for p in params:
        if p.grad is not None:
            p.grad.zero_()
    return torch.tensor(0.5)


def test_train_driver_checkpoint_resume_warm_start(native_lib, tmp_path, monkeypatch, capsys):
    from tacotron2_amd import train as tr
    monkeypatch.setattr(tr, "Tacotron2Loss", _FiniteLoss)
    monkeypatch.setattr(torch.nn.utils, "clip_grad_norm_", _finite_clip)
    native.set_validate_only(True)
    try:
        hpstr = gu.TINY_HP + ",batch_size=2,iters_per_checkpoint=2,epochs=2,training_files=synthetic:6:3:60," \
                             "validation_files=synthetic:3:4:60"
        out = tmp_path / "run"
        last = tr.train(str(out), "logs", None, False, 1, 0, "g", create_hparams(hpstr), max_iterations=3)
        assert last == 2
        assert sorted(os.listdir(out)) == ["checkpoint_0", "checkpoint_2", "logs"]
        ck = torch.load(out / "checkpoint_2", weights_only=False)
        assert set(ck) == {"iteration", "state_dict", "optimizer", "learning_rate"} and ck["iteration"] == 2
        assert len(ck["state_dict"]) == 84 and ck["learning_rate"] == 1e-3
        recs = [json.loads(l) for l in open(out / "logs" / "scalars.jsonl")]
        assert [r["iteration"] for r in recs if "training.loss" in r] == [0, 1, 2]
        assert [r["iteration"] for r in recs if "validation.loss" in r] == [0, 2]
        assert any(k.startswith("param.rms/decoder.") for k in recs[1])
        # resume: next iteration is saved + 1, epoch offset from the loader length (3 batches per epoch)
        last = tr.main(["-o", str(out), "-l", "logs", "-c", str(out / "checkpoint_2"), "--hparams", hpstr,
                        "--max_iterations", "5"])
        assert last == 4 and os.path.exists(out / "checkpoint_4")
        assert "Epoch: 1" in capsys.readouterr().out
        # warm start: every tensor from the checkpoint except the ignored layer
        hp = create_hparams(hpstr)
        torch.manual_seed(77)
        m = tr.load_model(hp)
        before = m.embedding.weight.detach().clone()
        tr.warm_start_model(str(out / "checkpoint_2"), m, hp.ignore_layers)
        sd = m.state_dict()
        assert torch.equal(sd["embedding.weight"], before)
        assert all(torch.equal(sd[k], v) for k, v in ck["state_dict"].items() if k != "embedding.weight")
        tr.warm_start_model(str(out / "checkpoint_2"), m, [])
        assert torch.equal(m.state_dict()["embedding.weight"], ck["state_dict"]["embedding.weight"])
        opt = torch.optim.Adam(m.parameters(), lr=0.5)
        _, opt, lr, it = tr.load_checkpoint(str(out / "checkpoint_2"), m, opt)
        assert (lr, it) == (1e-3, 2) and len(opt.state_dict()["state"]) == 60
        with pytest.raises(AssertionError):
            tr.load_checkpoint(str(out / "nope"), m, opt)
        # fp16_run selects the engine's bf16 compute mode (no Apex)
        m16 = tr.load_model(create_hparams(hpstr + ",fp16_run=True"))
        assert m16.precision == "bf16" and m16.decoder.attention_layer.score_mask_value == -65504.0
    finally:
        native.set_validate_only(False)
    with pytest.raises(native.NativeError):                  # without validate-only there is no CPU path
        tr.load_model(create_hparams(gu.TINY_HP)) if not torch.cuda.is_available() else (_ for _ in ()).throw(
            native.NativeError("gpu box"))


def test_launcher_child_commands():
    from tacotron2_amd.multiproc import child_commands
    cmds = child_commands(["-m", "tacotron2_amd.train", "-o", "out"], 4, "S", python="py")
    assert len(cmds) == 4
    for i, c in enumerate(cmds):
        assert c[:5] == ["py", "-m", "tacotron2_amd.train", "-o", "out"]
        assert c[5:] == ["--n_gpus=4", "--group_name=group_S", "--rank=%d" % i]


def test_notebook_half_call_keeps_master_weights(native_lib):
    """inference.ipynb cell 7 runs ``model.cuda().eval().half()``: parameters stay f32 (the engine's master
    weights), the compute mode becomes bf16 and inference hands back float16 tensors (cells 13/15 feed them
    to a half-precision WaveGlow); ``.float()`` restores the parity mode."""
    from tacotron2_amd.model import Tacotron2
    m = Tacotron2(create_hparams(gu.TINY_HP + ",max_decoder_steps=5"))
    assert m.eval().half() is m and m.precision == "bf16"
    assert all(p.dtype == torch.float32 for p in m.parameters())
    native.set_validate_only(True)
    try:
        out = m.inference(torch.randint(1, 148, (1, 9)))
        assert len(out) == 4 and all(o.dtype == torch.float16 for o in out)
        assert out[0].shape[:2] == (1, 80) and out[2].shape[2] == 1 and out[3].shape[2] == 9
        assert m.float() is m and m.precision == "fp32"
        assert all(o.dtype == torch.float32 for o in m.inference(torch.randint(1, 148, (1, 9))))
    finally:
        native.set_validate_only(False)


# ---- optimiser step -----------------------------------------------------------------------------
def test_fused_adam_is_a_torch_adam_on_the_host_side(native_lib):
    """State layout, state_dict interchange with torch.optim.Adam, step counting and refusals — kernels off."""
    from tacotron2_amd.optim import FusedAdam
    torch.manual_seed(3)
    flat = torch.randn(5000)
    params = [torch.nn.Parameter(flat[1:4097 + 1].clone()), torch.nn.Parameter(torch.randn(1)),
              torch.nn.Parameter(torch.randn(7, 13))]
    native.set_validate_only(True)
    try:
        opt = FusedAdam(params, lr=1e-3, weight_decay=1e-6)
        assert isinstance(opt, torch.optim.Adam)
        assert opt.step() is None                                # no gradients yet
        for p in params:
            p.grad = torch.randn_like(p)
        n = opt.step(clip_norm=1.0)
        assert n.shape == () and n.dtype == torch.float32
        opt.step()
        st = opt.state[params[0]]
        assert set(st) == {"step", "exp_avg", "exp_avg_sq"} and float(st["step"]) == 2.0
        assert st["exp_avg"].shape == params[0].shape
        ref = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-6)
        ref.load_state_dict(opt.state_dict())                    # fused -> torch
        assert float(ref.state[params[2]]["step"]) == 2.0
        for p in params:
            p.grad = torch.zeros_like(p)
        ref.step()
        opt2 = FusedAdam(params, lr=1e-3, weight_decay=1e-6)
        opt2.load_state_dict(ref.state_dict())                   # torch -> fused
        opt2.step(clip_norm=1.0)
        assert float(opt2.state[params[1]]["step"]) == 4.0
        L, blocks = native.tensor_list([p.grad for p in params])
        assert blocks == 2 + 1 + 1 and list(L.first_block[:3]) == [0, 2, 3] and L.count == 3
        with pytest.raises(native.NativeError):
            native.tensor_list([torch.zeros(3)] * 65)
        bad = FusedAdam([torch.nn.Parameter(torch.zeros(3))])
        bad.param_groups[0]["amsgrad"] = True
        bad.param_groups[0]["params"][0].grad = torch.zeros(3)
        with pytest.raises(native.NativeError):
            bad.step()
        half = FusedAdam([torch.nn.Parameter(torch.zeros(3, dtype=torch.float64))])
        half.param_groups[0]["params"][0].grad = torch.zeros(3, dtype=torch.float64)
        with pytest.raises(native.NativeError):
            half.step()
    finally:
        native.set_validate_only(False)
    # argument validation of the C entry points
    L = native.TensorList()
    rc = native_lib.t2amd_grad_norm_f32(ctypes_byref(L), 1.0, None, None, None)
    assert rc == 1 and b"tensor count" in native_lib.t2amd_last_error()


def ctypes_byref(x):
    import ctypes
    return ctypes.byref(x)


def test_train_driver_with_fused_optimizer_validate_only(native_lib, tmp_path, monkeypatch):
    from tacotron2_amd import train as tr
    monkeypatch.setattr(tr, "Tacotron2Loss", _FiniteLoss)
    native.set_validate_only(True)
    try:
        hpstr = gu.TINY_HP + ",batch_size=2,iters_per_checkpoint=1,epochs=1,training_files=synthetic:4:3:60," \
                             "validation_files=synthetic:2:4:60"
        out = tmp_path / "run"
        # gradients are uninitialised memory in this mode: only the plumbing (60 tensors -> one list, state, checkpoint)
        tr.train(str(out), "logs", None, False, 1, 0, "g", create_hparams(hpstr), max_iterations=2, fused_optimizer=True)
        files = sorted(os.listdir(out))
        assert "logs" in files
        ck = [f for f in files if f.startswith("checkpoint_")]
        if ck:                                                   # written only when the garbage norm happened to be finite
            st = torch.load(out / ck[0], weights_only=False)["optimizer"]["state"]
            assert len(st) == 60 and set(st[0]) == {"step", "exp_avg", "exp_avg_sq"}
    finally:
        native.set_validate_only(False)


def test_bench_inference_leg_plumbing(native_lib, monkeypatch, capsys):
    """bench.py's decode-steps/s leg (BASELINE configs 4/5) with the kernels off: shapes, forced step counts and the
    JSON fields; the numbers are meaningless here.  (In bench.py the leg sits inside a try/except so that it can never
    take the headline line down — which is exactly why its plumbing is pinned by a test.)"""
    import importlib
    import sys as _sys
    _sys.path.insert(0, gu.ROOT)
    bench = importlib.import_module("bench")
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    native.set_validate_only(True)
    try:
        out = bench.inference_leg(torch.device("cpu"))
    finally:
        native.set_validate_only(False)
    assert capsys.readouterr().out == ""                              # stdout belongs to the one JSON line
    assert {"config4_B1_fp32", "config4_B1_bf16", "config4_B1_bf16_launch_chain", "config5_B256_bf16",
            "config5_B256_bf16_2000"} <= set(out)
    # (the stop bookkeeping lives in device memory the kernels never wrote here: the step count itself is not checked)
    assert 1 <= out["config4_B1_fp32"]["steps"] <= 1000 and 1 <= out["config5_B256_bf16"]["steps"] <= 400
    b1 = out["config4_B1_bf16"]["hbm_roofline"]["algorithmic_bytes_per_step"]
    assert b1 == 2.0 * (18189969 + 640 * 100)                          # 36.5 MB: SURVEY 8d's figure
    assert all(v["decode_path"].startswith("launch chain") for v in out.values())   # kernels off: never the persistent path
    for v in out.values():
        assert v["decode_steps_per_s"] > 0 and 0 < v["hbm_roofline"]["frac"]
        assert v["utterance_steps_per_s"] == pytest.approx(v["B"] * v["decode_steps_per_s"])


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` outside torch.distributed.run starts the N ranks itself (the driver's SCALE
    run must not die on "launch through torch.distributed.run"; reference multiproc.py:1-23 is the same idea).  Here, on a
    box without GPUs, every rank must get as far as the GPU check with its own RANK / WORLD_SIZE / MASTER_* environment."""
    import subprocess
    import sys
    from tacotron2_amd import multiproc
    envs = multiproc.rank_environments(4, 2950, base={})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] and all(e["WORLD_SIZE"] == "4" for e in envs)
    assert all(e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "2950" and e["LOCAL_RANK"] == e["RANK"] for e in envs)
    if torch.cuda.is_available():
        pytest.skip("CPU-box check of the spawn path")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, T2AMD_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "needs an MI355X" in r.stderr                       # rank 0 reached the device check ...
    with open(os.path.join(root, "gpurun_out", "rank1.log")) as f:
        assert "needs an MI355X" in f.read()                   # ... and so did rank 1, in its own process
    # without the gloo override the launcher itself refuses: RCCL needs one device per rank
    env.pop("T2AMD_DIST_BACKEND")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr


def test_bench_stdout_line_is_compact_and_complete():
    """bench.py prints ONE line that must carry every key of the driver's contract plus `roofline` / `cpu_baseline`, and stay
    well inside what a bounded tail of stdout keeps (the verbose record goes to gpurun_out/bench_full.json)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    # round 4: the forward loop is one persistent launch -- the chain of the line is that launch + the two backward kernels
    with open(os.path.join(root, "profiles", "r04_g_bench_full.json")) as f:
        full4 = json.load(f)
    line4 = bench.compact_line(full4)
    assert len(json.dumps(line4)) < 5200
    r4 = line4["roofline"]
    assert set(r4["chain"]) == {"decoder_forward_persistent", "attention_backward", "dgrad_pair"}
    assert r4["kernel"] == max(r4["chain"].values(), key=lambda v: v["avg_launch_us"] * v["launches"])["kernel"]
    assert abs(r4["chain"]["decoder_forward_persistent"]["us_per_time_step"] * r4["whole_step"]["time_steps"]
               - r4["chain"]["decoder_forward_persistent"]["avg_launch_us"]) < 1.0
    assert line4["timed_loop"]["device_allocs"] <= 1 and line4["timed_loop"]["full_gc_collections_ms"] == [] and "build" in line4
    assert line4["cpu_baseline"]["calibration"]["port_over_reference"] > 0
    with open(os.path.join(root, "profiles", "r03_r_bench_full.json")) as f:
        full = json.load(f)
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < 5000, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    r = line["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert set(r["chain"]) == {"lstm_pair", "attention_forward", "attention_backward", "dgrad_pair"}
    assert r["kernel"] == max(r["chain"].values(), key=lambda v: v["avg_launch_us"] * v["launches"])["kernel"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert "workload" in line["config"] and "model" not in line["config"]


def test_splitk_model_of_the_bf16_resident_products():
    """engine._choose_splitk16 (a time model fitted to rocprofv3 dispatch times): a product that fills the chip
    takes no split, one of a few tiles over a 55 k-deep K takes dozens (the old whole-rounds heuristic gave the 512 x 400
    weight gradient of the first postnet layer ONE split: 1.2 ms for 23 GFLOP), every slab keeps at least 8 tile steps."""
    from tacotron2_amd import engine
    K = 55680
    assert engine._choose_splitk16(55680, 4096, 256) == 1                 # 3488 tiles: nothing to gain
    assert engine._choose_splitk16(8192, 8192, 8192) == 1
    few = engine._choose_splitk16(512, 400, K)                            # 4 tiles
    assert 32 <= few <= 64
    assert 6 <= engine._choose_splitk16(512, 2560, K) <= 24               # 20 tiles
    for M, N in ((4096, 1792), (4096, 2560), (512, 2560), (80, 2560), (512, 400), (4096, 256)):
        s = engine._choose_splitk16(M, N, K)
        assert 1 <= s <= 64 and ((K + 63) // 64) // s >= 8, (M, N, s)
    assert engine._choose_splitk16(512, 2560, 640) == 1                   # a shallow K cannot be split at all


def test_encoder_launch_routing_and_fallback_host_logic(monkeypatch):
    """engine._encoder_lstm_fwd without a GPU (the native calls are replaced): inference reads the status back and, on a
    give-up, recomputes the pre-activations, runs the launch chain and stays on it for 4, 8, ... 256 calls (exponential back-off); the training forward takes
    the persistent launch only when asked to, hands it the poison word and never reads the status; B == 1 and a refused
    geometry go to the chain."""
    import types
    import torch
    from tacotron2_amd import engine
    calls = []

    class FakeStatus(object):
        value = 0

    def fake_persistent(d0, d1, flags, status, poison=None):
        calls.append(('persistent', poison is not None))
        status.fill_(FakeStatus.value)

    monkeypatch.setattr(engine.nv, 'validate_only', lambda: False)
    monkeypatch.setattr(engine.nv, 'lstm_seq_batch_persistent_supported', lambda d, n, cus: None)
    monkeypatch.setattr(engine.nv, 'lstm_seq_batch_persistent_flag_words', lambda B, H, n=2: 8)
    monkeypatch.setattr(engine.nv, 'lstm_seq_fwd2_batch_persistent', fake_persistent)
    monkeypatch.setattr(engine.nv, 'lstm_seq_fwd2', lambda d0, d1, reads=None, writes=None: calls.append(('chain',)))
    monkeypatch.setattr(torch.cuda, 'get_device_properties', lambda dev: types.SimpleNamespace(multi_processor_count=256))
    monkeypatch.setattr(engine, 'ENCODER_BATCH_PERSISTENT', True)
    monkeypatch.setattr(engine, 'ENCODER_BATCH_PERSISTENT_TRAIN', False)
    model = types.SimpleNamespace()
    d = types.SimpleNamespace(B=64, H=256)
    regen = lambda: calls.append(('regen',))                                   # noqa: E731
    cpu = torch.device('cpu')
    assert engine._encoder_lstm_fwd(model, cpu, d, d, regen, [], []) == 'persistent'
    assert calls == [('persistent', False)]
    # a give-up in inference: pre-activations recomputed, chain, back-off
    del calls[:]
    FakeStatus.value = 2
    assert engine._encoder_lstm_fwd(model, cpu, d, d, regen, [], []) == 'launch chain'
    assert calls == [('persistent', False), ('regen',), ('chain',)] and model._enc_batch_backoff == 4
    del calls[:]
    assert engine._encoder_lstm_fwd(model, cpu, d, d, regen, [], []) == 'launch chain'
    assert calls == [('chain',)] and model._enc_batch_backoff == 3
    # training: the chain unless asked; with the switch the launch gets the poison word and the status is never read
    model2 = types.SimpleNamespace()
    del calls[:]
    word = torch.zeros(4)
    assert engine._encoder_lstm_fwd(model2, cpu, d, d, regen, [], [], poison=word) == 'launch chain'
    monkeypatch.setattr(engine, 'ENCODER_BATCH_PERSISTENT_TRAIN', True)
    assert engine._encoder_lstm_fwd(model2, cpu, d, d, regen, [], [], poison=word) == 'persistent'     # status 2 is not looked at
    assert calls == [('chain',), ('persistent', True)]
    # one utterance / a refused geometry: the chain
    del calls[:]
    FakeStatus.value = 0
    assert engine._encoder_lstm_fwd(model2, cpu, types.SimpleNamespace(B=1, H=256), d, regen, [], []) == 'launch chain'
    monkeypatch.setattr(engine.nv, 'lstm_seq_batch_persistent_supported', lambda d, n, cus: "too many workgroups")
    assert engine._encoder_lstm_fwd(model2, cpu, d, d, regen, [], []) == 'launch chain'
    assert calls == [('chain',), ('chain',)]
