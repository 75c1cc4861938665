"""Host-side orchestration of the HIP kernels for the Tacotron 2 hot path.

One ``torch.autograd.Function`` covers the whole of ``Tacotron2.forward``: the forward pass
enqueues the native encoder / decoder-loop / postnet kernels and keeps the activation slabs
the backward pass needs; the backward pass enqueues the BPTT loop and the deferred dense
weight-gradient GEMMs and returns one gradient per parameter.  Nothing here computes:
PyTorch supplies device buffers (caching allocator) and the current stream.

Data layout in HBM (fp32): activations are channel-last.  Encoder / postnet rows are
batch-major ``(b, t)``; decoder slabs are time-major ``[T][B][F]`` so that one step's slice is
contiguous.  Reference call sites are cited next to each stage.
"""
import os

import math
import torch

from . import native as nv
from .native import NativeError

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def check_hparams(hp):
    """Geometry the kernels are compiled / tiled for.  Anything else fails loudly."""
    if hp.n_frames_per_step != 1:
        raise ValueError("n_frames_per_step != 1 is not supported (nor by the reference, hparams.py:56)")
    K = hp.attention_location_kernel_size
    if not (1 <= hp.attention_dim <= nv.ATT_DIM and 1 <= hp.attention_location_n_filters <= nv.LOC_FILTERS
            and 1 <= K <= nv.LOC_KERNEL and K % 2 == 1):
        raise ValueError("the attention kernels hold attention_dim <= %d, <= %d location filters and an odd location "
                         "kernel <= %d (smaller geometries run zero-embedded in the compiled one; an even kernel size "
                         "does not run in the reference either: ConvNorm pads (k-1)//2, model.py:18-20)"
                         % (nv.ATT_DIM, nv.LOC_FILTERS, nv.LOC_KERNEL))
    for name in ('encoder_embedding_dim', 'attention_rnn_dim', 'decoder_rnn_dim', 'prenet_dim'):
        if getattr(hp, name) % 64 != 0:
            raise ValueError("%s must be a multiple of 64 for the MFMA recurrent kernels" % name)
    if (hp.encoder_embedding_dim // 2) % 64 != 0:
        raise ValueError("encoder_embedding_dim/2 must be a multiple of 64")
    if hp.symbols_embedding_dim != hp.encoder_embedding_dim:
        raise ValueError("symbols_embedding_dim must equal encoder_embedding_dim")
    for name in ('encoder_embedding_dim', 'postnet_embedding_dim', 'n_mel_channels'):
        if getattr(hp, name) % 16 != 0:
            raise ValueError("%s must be a multiple of 16 (implicit-GEMM convolution tiles)" % name)
    if hp.encoder_kernel_size % 2 != 1 or hp.postnet_kernel_size % 2 != 1:
        raise ValueError("convolution kernel sizes must be odd")


# ----------------------------------------------------------------------------
# dropout masks
# ----------------------------------------------------------------------------
class MaskSource(object):
    """Keep-masks (uint8, 1 = keep) in ENGINE layout.  Injected masks (tests: the oracle's masks,
    transposed to channel-last) win; otherwise a Philox4x32-10 stream seeded from torch's RNG.

    engine layouts:  enc[i] (B,Ti,E)   prenet[i] (To,B,P)   att (To,B,Ha)   dec (To,B,Hd)
                     post[i] (B,To,C)  prenet_infer (steps,2,B,P)"""

    def __init__(self, injected, device, run=None):
        self.injected = injected or {}
        self.device = device
        self.run = run                   # the step's allocator (engine._Run): masks live exactly as long as the step
        self.seed = None
        self.offset = 0

    def get(self, name, index, shape, p):
        src = self.injected.get(name)
        if src is not None:
            m = src[index] if index is not None else src
            if tuple(m.shape) != tuple(shape) or m.dtype != torch.uint8 or not m.is_contiguous():
                raise NativeError("injected mask %s[%s] must be contiguous uint8 of shape %s, got %s %s"
                                  % (name, index, tuple(shape), tuple(m.shape), m.dtype))
            return m
        if self.seed is None:
            self.seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        out = self.run.empty8(*shape) if self.run is not None else torch.empty(shape, dtype=torch.uint8, device=self.device)
        nv.philox_keep_mask(out, p, self.seed, self.offset)
        self.offset += (out.numel() + 3) // 4 * 4
        return out


# ----------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------
def _choose_splitk(M, N, K, batch=1, precision=0, a_km=False, b_kn=False):
    """Split count for a plain-epilogue GEMM.  The library tiles 256 x 256 (one workgroup per CU) for plain-bf16
    products when that still yields >= 192 workgroups, else 128 x 128 (three to five per CU); the split is chosen
    against whichever applies (native.gemm_tile_size is the same rule the launch uses)."""
    t256 = ((M + 255) // 256) * ((N + 255) // 256) * batch
    if precision >= 1 and M >= 512 and N >= 512 and nv.gemm_tile_size(M, N, precision, 1 << 20, a_km, b_kn) == 256:
        if t256 >= 192:
            return 1
        # fill whole rounds of 256 workgroups: efficiency = rounds / ceil(rounds), mild preference for fewer slabs
        best, best_score = 0, -1.0
        for s in range(1, 33):
            if s > 1 and K // s < 512:
                break
            if t256 * s < 192:
                continue
            w = t256 * s / 256.0
            score = w / math.ceil(w) * min(1.0, w / 0.85) - 0.02 * (s - 1)
            if score > best_score:
                best, best_score = s, score
        if best:
            return best
    tiles = ((M + 127) // 128) * ((N + 127) // 128) * batch
    if tiles >= 512:        # >= 2 workgroups per CU already: one's loads hide behind the other's MFMAs
        return 1
    s = (768 + tiles - 1) // tiles
    s = min(s, 64, max(1, K // 256))
    return max(1, s)


def _choose_splitk16(M, N, K):
    """Split count for the bf16-resident products (csrc/gemm16.hip: 256 x 256 x 64 tile steps, one workgroup per CU) by a time
    model fitted to rocprofv3 dispatch times on MI355X (profiles/r03_v_gemm16_trace.csv): rounds of 256 workgroups, 1.9 us
    per tile step, ~4 us of prologue / epilogue per round, and the partial slabs read back at ~5 TB/s by the reduction.  A
    product of a few tiles over a 55 k-deep K (the weight gradient of an 80-channel convolution: 4 tiles) takes dozens of
    splits; one that already fills the chip takes none."""
    t256 = ((M + 255) // 256) * ((N + 255) // 256)
    nkt = (K + 63) // 64
    best, best_t = 1, None
    for s in range(1, 65):
        if s > 1 and nkt // s < 8:
            break
        rounds = math.ceil(t256 * s / 256.0)
        t = rounds * (math.ceil(nkt / s) * 1.9 + 4.0)
        if s > 1:
            t += (s + 1) * M * N * 4 / 5.0e6 + 3.0
        if best_t is None or t < best_t - 1e-9:
            best, best_t = s, t
    return best


# T2AMD_WGRAD16=0 keeps the decoder-LSTM weight gradients on the f32-source GEMM in the bf16 mode (A/B runs)
WGRAD16 = os.environ.get('T2AMD_WGRAD16', '1') != '0'


# T2AMD_WGRAD_KK=0 keeps the round-2 route of the bf16 weight gradients (transposed K-contiguous copies + gemm16_tn, and the
# f32-source GEMM for the convolutions) instead of the K-major product on the slabs / halo images themselves (A/B runs)
WGRAD_KK = os.environ.get('T2AMD_WGRAD_KK', '1') != '0'


def _kk_to(run, out, A16, B16, K, M, N, lda=None, ldb=None, perm=None):
    """out (+ optional (taps, Ci) permutation of a packed conv-weight gradient) = sum_k A16[k, :M]^T B16[k, :N] on the K-major
    bf16 product, split along K to fill the chip; ``out`` may be a column block of a wider matrix."""
    sk = _choose_splitk16(M, N, K)
    if sk == 1 and perm is None:
        nv.gemm16_kk(out, A16, B16, K, M=M, N=N, lda=lda, ldb=ldb)
        return
    part = run.empty(sk, M * N)
    nv.gemm16_kk(part[0].view(M, N), A16, B16, K, M=M, N=N, lda=lda, ldb=ldb, splitk=sk, partials=part)
    if perm is not None:
        nv.splitk_reduce(part, sk, out, perm_taps=perm[0], perm_ci=perm[1])
    else:
        nv.splitk_reduce2d(part, sk, out)


def _lstm_wgrad_kk(run, dG2, B, parts, outs):
    """The same weight gradients from the slabs as they are (round 3): dG [To.B][4H] and every input slab [To.B][.] are
    K-major bf16 operands of native.gemm16_kk; an input read from the PREVIOUS time step is dG from row B on against the
    slab from row 0 -- no transposed copies.  The input blocks are separate products of ONE launch (gemm16_kk_group): their
    workgroups share the rows of dG, which is read once; each writes its own column block of dW_ih / dW_hh."""
    rowsD, G4 = dG2.shape
    widths = [src.shape[1] for src, _ in parts]
    offs = [0]
    for w in widths:
        offs.append(offs[-1] + w)
    jobs = []                                       # (destination block, A, B, K, width)
    for dW, i0, i1 in outs:
        base = offs[i0]
        for i in range(i0, i1):
            src, shifted = parts[i]
            if src.dtype != torch.bfloat16:
                src = run.cast16(src)
            blk = dW[:, offs[i] - base:offs[i + 1] - base]
            if shifted and rowsD <= B:
                blk.zero_()
            elif shifted:
                jobs.append((blk, dG2[B:], src, rowsD - B, widths[i]))
            else:
                jobs.append((blk, dG2, src, rowsD, widths[i]))
    if not jobs:
        return
    sk = _choose_splitk16(G4, sum((j[4] + 255) // 256 * 256 for j in jobs), rowsD)
    if sk == 1:
        nv.gemm16_kk_group([dict(Cm=blk, A16=A, B16=Bm, K=K, M=G4, N=w) for blk, A, Bm, K, w in jobs])
        return
    parts_ = [run.empty(sk, G4 * w) for _, _, _, _, w in jobs]
    nv.gemm16_kk_group([dict(Cm=pt[0].view(G4, w), A16=A, B16=Bm, K=K, M=G4, N=w, splitk=sk, partials=pt)
                        for (blk, A, Bm, K, w), pt in zip(jobs, parts_)])
    for (blk, _, _, _, _), pt in zip(jobs, parts_):
        nv.splitk_reduce2d(pt, sk, blk)


def _lstm_wgrad16(run, dG2, B, parts, outs):
    """Weight gradients of one decoder LSTM in the bf16 mode: [dW_ih | dW_hh] = dG^T . [x_0 | x_1 | ...] on the
    bf16-resident product (native.gemm16_tn).  Both operands are held [To.B][.] (K-major) by the time loops, so K-contiguous
    bf16 images are made first (transposing cast, K padded to the tile depth with zeros); an input the step reads from
    the PREVIOUS time step (ctx_{t-1}, h_{t-1}) is written B columns to the right, which turns its
    dG[B:]^T . x[:(To-1).B] into the same full-K product (the first B columns are zero).
    ``parts``: (slab2d, shifted) per input block, in weight-column order; ``outs``: (dW tensor, first block, last block + 1).
    (Round 2's route; since round 3 the K-major product above is the default, T2AMD_WGRAD_KK=0 selects this one.)"""
    rowsD, G4 = dG2.shape
    if WGRAD_KK and G4 % 8 == 0 and all(src.shape[1] % 8 == 0 for src, _ in parts):
        return _lstm_wgrad_kk(run, dG2, B, parts, outs)
    Kp = ((rowsD + B + 63) // 64) * 64
    GT = run.empty16(G4, Kp)
    nv.transpose_cast_bf16(dG2, GT)
    widths = [src.shape[1] for src, _ in parts]
    XT = run.empty16(sum(widths), Kp)
    n0, offs = 0, []
    for (src, shifted), w in zip(parts, widths):
        blk = XT[n0:n0 + w]
        if shifted:
            blk[:, :B].zero_()
            if rowsD > B:
                nv.transpose_cast_bf16(src[:rowsD - B], blk[:, B:])
            else:
                blk[:, B:].zero_()
        else:
            nv.transpose_cast_bf16(src, blk)
        offs.append(n0)
        n0 += w
    offs.append(n0)
    for dW, i0, i1 in outs:
        Bop = XT[offs[i0]:offs[i1]]
        M, N = dW.shape
        sk = _choose_splitk16(M, N, Kp)
        if sk == 1:
            nv.gemm16_tn(dW, GT, Bop)
        else:
            part = run.empty(sk, M * N)
            nv.gemm16_tn(dW, GT, Bop, splitk=sk, partials=part)
            nv.splitk_reduce(part, sk, dW)


class _Ctx(object):
    """What a training forward keeps for its backward.  Owns the lease of the step arena its slabs live in."""
    arena = None

    def release(self):
        a, self.arena = self.arena, None
        _arena_release(a)

    def __del__(self):
        try:
            self.release()
        except Exception:                # interpreter shutdown
            pass


class _EvalCtx(object):
    training = False


# Gradient GEMMs (dgrad / wgrad, tolerance 1e-3 of the gradient's max in the parity tests) run on the
# split-bf16 MFMA kernel (include/tacotron2_amd.h: t2amd_gemm_desc.precision = 1, ~2^-17 relative per
# product).  Every forward GEMM stays on the exact-f32 MFMA.  Set to False for bit-faithful f32 gradients.
FAST_GRAD_GEMM = True


def _dgrad_split():
    """Split-K factor of the two BPTT dgrad products (T2AMD_DGRAD_SPLIT, tuning knob): validated here because it
    sizes the dXd / dXa slabs the kernels write."""
    raw = os.environ.get('T2AMD_DGRAD_SPLIT', '2')
    try:
        ns = int(raw)
    except ValueError:
        ns = -1
    if not 1 <= ns <= 8:
        raise NativeError("T2AMD_DGRAD_SPLIT must be an integer in 1..8, got %r" % (raw,))
    return ns


DGRAD_SPLIT = _dgrad_split()


def _enc_dgrad_split():
    """Split-K factor of the encoder bi-LSTM's recurrent dgrad product (T2AMD_ENC_DGRAD_SPLIT, tuning knob; 1, 2 or 4 --
    it must divide the 4H/64 k tiles).  At H = 256 the unsplit product is 16 workgroups per direction walking 16 k tiles;
    measured on MI355X (profiles/r02_ad_enc_split.txt): 63.2 / 63.1 / 62.3 ms per training step at 1 / 2 / 4."""
    raw = os.environ.get('T2AMD_ENC_DGRAD_SPLIT', '4')
    if raw not in ('1', '2', '4'):
        raise NativeError("T2AMD_ENC_DGRAD_SPLIT must be 1, 2 or 4, got %r" % (raw,))
    return int(raw)


ENC_DGRAD_SPLIT = _enc_dgrad_split()

# bf16 mode: the BatchNorm backward of a convolution layer writes its output as the bf16 halo image + the bias gradient directly
# (T2AMD_BN_BWD_IMAGE=0: the f32 slab and the two separate passes, for A/B runs and the bit-identity test)
BN_BWD_IMAGE = os.environ.get('T2AMD_BN_BWD_IMAGE', '1') != '0'
BN_FWD_IMAGE = os.environ.get('T2AMD_BN_FWD_IMAGE', '1') != '0'   # the same fold in the forward: BatchNorm apply writes the next layer's image
# bf16 mode: the two LSTM bias gradients as column sums of the bf16 gate-gradient slabs (T2AMD_BIAS_GRAD16=0: of the f32 slabs)
BIAS_GRAD16 = os.environ.get('T2AMD_BIAS_GRAD16', '1') != '0'
# ... and then nothing reads the f32 gate-gradient slabs: not allocated, not written (T2AMD_GATE_GRADS_BF16_ONLY=0 keeps them)
GATE_GRADS_BF16_ONLY = os.environ.get('T2AMD_GATE_GRADS_BF16_ONLY', '1') != '0'
# slabs of the decoder-LSTM input gradients kept by the BPTT loop: a ring of this many (>= 3), 0 = one per time step
DXD_RING = int(os.environ.get('T2AMD_DXD_RING', '3'))


def _rg(run, *a, **k):
    return run.gemm(*a, fast=run.gradp, **k)


def _ng(run, *a, **k):
    return nv.gemm(*a, fast=run.gradp, **k)


def _fg(run, *a, exact=False, **k):
    """Forward GEMM: exact f32 (fp32 mode), plain bf16 (bf16 mode), split-bf16 x3 ('bf16x3' mode).  ``exact``: a product in front
    of a ReLU -- the encoder convolutions, the prenet -- stays on the exact-f32 MFMA in the 'bf16x3' mode: a pre-activation that
    lands on the other side of zero flips relu' for a whole row of the weight gradient (the ReLU-kink rows of the full-size
    parity test), and 2^-17-relative products do that ten times as often as f32 rounding; these products are 3 % of the step."""
    return nv.gemm(*a, fast=(0 if (exact and run.x3) else run.fwdp), **k)


# Packed / transposed / bf16 weight images are rebuilt only when a weight changed: the key is the
# parameters' storage addresses and torch version counters (every in-place optimiser update, copy_ and
# load_state_dict bumps them) plus a generation counter that raw-pointer updates bump (optim.FusedAdam.step).
# Writes through ``param.data`` bypass version counters by design of torch: call
# ``engine.bump_weight_generation()`` after such an edit.
_PACK_GEN = [0]


def bump_weight_generation():
    _PACK_GEN[0] += 1


# ----------------------------------------------------------------------------
# step arena (SURVEY 8b "Ownership": nothing on the hot path calls hipMalloc)
# ----------------------------------------------------------------------------
# Every slab a training step needs between `forward` and the end of `backward` (activations saved for BPTT, keep-masks,
# halo images, split-K partials, gradient slabs: ~11 GB at B = 64, To = 870) used to be its own ``torch.empty``: ~270 calls
# per step whose sizes follow the batch's (Ti_max, To_max).  The caching allocator serves most of them from its pool, but
# every change of shape splits / regroups its blocks and now and then ends in a ``hipMalloc`` of a GB in the middle of the
# step (profiles/r04_a_diag_before.json: two in 20 fresh batches, 13 ms each; the driver's loop read 65.3 ms per step where
# a replay of the same batches reads 61.6).  The engine now owns ONE region per in-flight step and bumps a pointer through
# it: the region is leased by `_forward`, handed back when `backward` has enqueued its last kernel (or when the saved context
# is dropped without one: eval-mode forwards, ``torch.no_grad()``), and reused by the next step as it is -- the kernels of
# consecutive steps are ordered by the stream they share, which is part of the pool key.  What OUTLIVES a step never comes
# from it: the four outputs, every parameter gradient (autograd may keep the very tensor as ``p.grad``), the cached weight
# images (`_Run.cached`).  The region grows in chunks during the first step(s) and is then replaced by a single chunk of
# 1.25 x the largest step seen, so a steady-state loop makes no allocator call for slabs at all.  T2AMD_ARENA=0 restores
# per-slab ``torch.empty``.
ARENA = os.environ.get('T2AMD_ARENA', '1') != '0'
ARENA_CHUNK = int(float(os.environ.get('T2AMD_ARENA_CHUNK_GB', '2')) * 2 ** 30)
ARENA_HEADROOM = float(os.environ.get('T2AMD_ARENA_HEADROOM', '1.25'))


class _Arena(object):
    ALIGN = 256

    def __init__(self, device):
        self.dev = device
        self.chunks = []                 # uint8 tensors
        self.ci, self.off = 0, 0         # bump position: chunk index, byte offset inside it
        self.used = 0                    # aligned bytes handed out since the lease began
        self.peak = 0                    # largest `used` of any lease
        self.busy = False
        self.growths = 0                 # torch.empty calls this arena has made (chunks + consolidations)
        self.leases = 0

    def capacity(self):
        return sum(c.numel() for c in self.chunks)

    def alloc(self, nbytes):
        """`nbytes` of device memory as a uint8 view, 256-byte aligned (what hipMalloc / the caching allocator give)."""
        n = max(int(nbytes), 1)
        na = (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        while True:
            if self.ci < len(self.chunks):
                ch = self.chunks[self.ci]
                if self.off + na <= ch.numel():
                    v = ch[self.off:self.off + n]
                    self.off += na
                    self.used += na
                    return v
                self.ci += 1
                self.off = 0
                continue
            self.chunks.append(torch.empty(max(na, ARENA_CHUNK), dtype=torch.uint8, device=self.dev))
            self.growths += 1

    def release(self):
        """End of a lease.  A step that needed more than one chunk leaves ONE chunk with headroom for the next -- made by
        `consolidate` at the NEXT acquire (`backward`'s finally block calls this while the dying context still
        holds views into every old chunk, so nothing could go back to the pool before the new request: a transient 2.25 x)."""
        self.peak = max(self.peak, self.used)
        self.ci, self.off, self.used = 0, 0, 0
        self.busy = False

    def consolidate(self):
        """Called with the lock held when the arena is leased again: the previous lease's context is dead by now (or about
        to be: its views keep their chunks alive on their own), so the old chunks go back to torch's pool first."""
        if len(self.chunks) > 1:
            want = int(self.peak * ARENA_HEADROOM)
            want = (want + (1 << 28) - 1) >> 28 << 28                      # whole 256 MiB
            self.chunks = []                                               # back to torch's pool before the new request
            self.chunks.append(torch.empty(want, dtype=torch.uint8, device=self.dev))
            self.growths += 1


import threading

_ARENAS = {}                      # (device, stream id) -> [arena, ...]; more than one only while steps overlap in time
# backward runs on autograd's device thread.  Re-entrant: regions that hold it allocate container objects, the
# cyclic collector may then finalise an unreachable _Ctx on the same thread, and its __del__ releases its lease through here.
_ARENA_LOCK = threading.RLock()


def _arena_key(device):
    if device.type == 'cuda':
        idx = device.index if device.index is not None else torch.cuda.current_device()
        return ('cuda', idx, torch.cuda.current_stream(idx).cuda_stream)
    return (device.type, 0, 0)


def _arena_acquire(device):
    if not ARENA:
        return None
    key = _arena_key(device)
    with _ARENA_LOCK:
        pool = _ARENAS.setdefault(key, [])
        for a in pool:
            if not a.busy:
                break
        else:
            a = _Arena(device)
            pool.append(a)
        a.busy = True
        a.leases += 1
        a.consolidate()
        return a


def _arena_release(a):
    if a is None:
        return
    with _ARENA_LOCK:
        if a.busy:
            a.release()


def arena_stats(model=None):
    """Counters of the step arenas of this process (tools / tests): capacity and peak in bytes, allocator calls made."""
    with _ARENA_LOCK:
        return [dict(key=str(k), chunks=len(a.chunks), capacity=a.capacity(), peak=a.peak, growths=a.growths,
                     leases=a.leases, busy=a.busy) for k, pool in _ARENAS.items() for a in pool]


def release_arenas():
    """Give every idle arena's memory back to torch's pool (e.g. before switching to inference on a small device)."""
    with _ARENA_LOCK:
        for k in list(_ARENAS):
            _ARENAS[k] = [a for a in _ARENAS[k] if a.busy]
            if not _ARENAS[k]:
                del _ARENAS[k]


def reserve_outputs(device, B, Ti_max, To_max, n_mel=80, copies=2):
    """What outlives a step -- the four outputs of `Tacotron2.forward` (mel, mel_postnet, gate, alignments) -- is a fresh
    ``torch.empty`` every step (SURVEY 8b Ownership: outputs freshly allocated), and the first batch with a new largest
    (To, Ti) sends the caching allocator to ``hipMalloc`` in the middle of a step.  A loop that knows its dataset's longest
    text and clip calls this ONCE: ``copies`` sets of maximum-size outputs are allocated and handed straight back to torch's
    pool, which then serves every later request of that size or smaller from cache -- no device allocation on the hot path
    from the second step on (bench.py: `timed_loop.device_allocs`).  Returns the bytes reserved."""
    shapes = [(B, n_mel, To_max), (B, n_mel, To_max), (B, To_max), (B, To_max, Ti_max)]
    held = [torch.empty(s, dtype=torch.float32, device=device) for _ in range(int(copies)) for s in shapes]
    n = sum(t.numel() * 4 for t in held)
    del held
    return n


class _Run(object):
    """Allocation + kernel helpers bound to one device."""

    def __init__(self, device, precision='fp32', cache=None, arena=None):
        self.dev = device
        self._ws = None
        self.arena = arena
        self._persist = 0                # > 0 inside `cached`: what is made there outlives the step
        self.cache = cache if cache is not None else {}
        if precision not in ('fp32', 'bf16', 'bf16x3'):
            raise NativeError("precision must be 'fp32', 'bf16' or 'bf16x3', got %r" % (precision,))
        self.bf16 = precision == 'bf16'
        # 'bf16x3' (round 6): the ACCURATE-FAST mode.  Everything is the fp32 parity mode -- f32 state, slabs,
        # attention kernels in their exact-f32 forms -- except the matrix products that were bound by the exact-f32 MFMA rate
        # (157 TF, 1/16 of bf16): the LSTM tiles of both time loops multiply SPLIT-bf16 operand images (x = hi + lo in bf16;
        # hi.hi + lo.hi + hi.lo on the bf16 MFMA, f32 accumulation: ~2^-17 relative per product, csrc/skinny_wide.h SW_X3) and the
        # dense forward products use the split-bf16 GEMM the gradient products already use (t2amd_gemm_desc.precision = 1).
        self.x3 = precision == 'bf16x3'
        # t2amd_gemm_desc.precision: 0 exact f32 MFMA, 1 split-bf16 x3 (f32-class), 2 plain bf16
        self.fwdp = 2 if self.bf16 else (1 if self.x3 else 0)
        self.gradp = 2 if self.bf16 else (1 if FAST_GRAD_GEMM else 0)

    def cached(self, tag, deps, fn):
        """fn() once per weight version: ``deps`` are the parameters the image is derived from."""
        key = (_PACK_GEN[0], str(self.dev)) + tuple((d.data_ptr(), d._version) for d in deps)
        hit = self.cache.get(tag)
        if hit is not None and hit[0] == key:
            return hit[1]
        self._persist += 1               # weight images live across steps: never in the step arena
        try:
            val = fn()
        finally:
            self._persist -= 1
        self.cache[tag] = (key, val)
        return val

    def _alloc(self, shape, dtype, esize):
        if self.arena is None or self._persist:
            return torch.empty(shape, dtype=dtype, device=self.dev)
        n = 1
        for v in shape:
            n *= int(v)
        if n == 0:                       # (a 1-byte arena view cannot be viewed as a wider dtype; torch.empty takes empty shapes)
            return torch.empty(shape, dtype=dtype, device=self.dev)
        return self.arena.alloc(n * esize).view(dtype).view(*shape)

    def out_empty(self, *shape):
        """f32 tensor that OUTLIVES the step (outputs, parameter gradients): always torch's allocator."""
        return torch.empty(shape, dtype=torch.float32, device=self.dev)

    def empty16(self, *shape):
        return self._alloc(shape, torch.bfloat16, 2)

    def empty8(self, *shape):
        return self._alloc(shape, torch.uint8, 1)

    def empty_i32(self, *shape):
        return self._alloc(shape, torch.int32, 4)

    def cast16(self, t):
        out = self.empty16(*t.shape)
        nv.cast_bf16(t.contiguous(), out)
        return out

    def split16(self, t):
        """The split-bf16 operand image of an f32 matrix ([..., K] -> bfloat16 [..., 2 K]; native.split_bf16x3)."""
        out = self.empty16(*(tuple(t.shape[:-1]) + (2 * t.shape[-1],)))
        nv.split_bf16x3(t.contiguous(), out)
        return out

    def empty(self, *shape):
        return self._alloc(shape, torch.float32, 4)

    def zeros(self, *shape):
        t = self.empty(*shape)
        nv.fill(t, 0.0)
        return t

    def ws(self, n):
        need = 2 * 64 * n
        if self._ws is None or self._ws.numel() < need:
            self._ws = self._alloc((need,), torch.float64, 8)
        return self._ws

    # C[M,N] (+)= A.B with automatic split-K for skinny outputs over a long K
    def gemm(self, Cm, A, B, a_km=False, b_kn=False, accumulate=False, convB=None, perm=None,
             batch=1, strides=(0, 0, 0), fast=False, **kw):
        M, N = Cm.shape
        K = A.shape[0] if a_km else A.shape[1]
        plain = kw.get('bias') is None and kw.get('act', 0) == 0 and kw.get('keep') is None \
            and kw.get('convA') is None
        sk = _choose_splitk(M, N, K, batch, int(fast), a_km, b_kn) if (plain and batch == 1) else 1
        if perm is not None and sk == 1:
            sk = 2 if K >= 512 else 1
        if sk == 1 and perm is None:
            nv.gemm(Cm, A, B, a_km=a_km, b_kn=b_kn, accumulate=accumulate, convB=convB, batch=batch,
                    strides=strides, fast=fast, **kw)
            return
        if not Cm.is_contiguous():
            raise NativeError("split-K / permuted GEMM output must be contiguous")
        part = self.empty(max(sk, 1), M * N)
        if sk == 1:
            nv.gemm(part[0].view(M, N), A, B, a_km=a_km, b_kn=b_kn, convB=convB, fast=fast)
        else:
            nv.gemm(part[0].view(M, N), A, B, a_km=a_km, b_kn=b_kn, convB=convB, splitk=sk, partials=part, fast=fast)
        pt, pc = perm if perm is not None else (0, 0)
        nv.splitk_reduce(part, max(sk, 1), Cm, accumulate=accumulate, perm_taps=pt, perm_ci=pc)

    def colsum(self, x, out, accumulate=False):
        nv.colsum(x, self.ws(x.shape[1]), out, accumulate)

    # ---- conv helpers: weights (Co, Ci, k) ------------------------------------------------
    def pack_conv_fwd(self, W):
        Co, Ci, k = W.shape
        Wp = self.empty(Co, k * Ci)
        nv.transpose(Wp.view(Co, k, Ci)[0], W[0], batch=Co, sstride=Ci * k, dstride=k * Ci)
        return Wp

    def pack_conv_dgrad(self, W):
        Co, Ci, k = W.shape
        Wd = self.empty(Ci, k * Co)
        nv.transpose(Wd.view(Ci * k, Co), W.view(Co, Ci * k))
        return Wd


def _bias_sum(run, b1, b2):
    out = run.empty(b1.numel())
    nv.copy2d(out.view(1, -1), b1.view(1, -1), b2.view(1, -1))
    return out


# ----------------------------------------------------------------------------
# conv + BN (+act +dropout) stack, shared by encoder and postnet
# ----------------------------------------------------------------------------
# T2AMD_CONV16=0 keeps the convolutions of the bf16 mode on the f32-source GEMM (A/B runs)
CONV16 = os.environ.get('T2AMD_CONV16', '1') != '0'


def _conv16_ok(run, rows, T, C, k=5):
    """The window form needs whole utterances of T rows, image rows of whole 16-byte chunks (C % 8), a window length k C
    whose round-up to the 64-deep k-steps stays inside the trailing halo (the weight image is zero there: an 80-channel
    layer's 400-long window runs as 448) and enough rows for 256-row tiles to fill the chip."""
    klen = k * C
    return run.bf16 and CONV16 and C % 8 == 0 and T > 0 and rows % T == 0 and rows >= 4096 and not nv.validate_only() \
        and ((klen + 63) // 64 * 64 - klen) <= (k - 1) * C


def _conv_wgrad_kk_ok(run, rows, T, Ci, Co):
    """The K-major weight-gradient product needs whole utterances of T rows, channel counts of whole 16-byte chunks and a K
    long enough to be worth two image casts."""
    return run.bf16 and CONV16 and WGRAD_KK and Ci % 8 == 0 and Co % 8 == 0 and T > 0 and rows % T == 0 and rows >= 4096 \
        and not nv.validate_only()


def _halo_image(run, x, T, pad):
    """bf16 image of the channel-last rows x (rows = B T) with `pad` zero rows around every utterance."""
    rows, C = x.shape
    img = run.empty16((rows // T) * (T + 2 * pad) + 2 * pad, C)
    nv.cast_halo_bf16(x, img, T, pad)             # writes every row of the image, halos included
    return img

def _conv_stack_fwd(run, P, bufs, prefix, n_layers, x, T, acts, masks, training, lens=None, exact=False):
    """x: (rows, C0) channel-last rows (b, t).  Returns the last activation and the saved slabs.
    reference model.py:141-146 (Postnet.forward), :174-175 (Encoder.forward)."""
    saved = []
    rows = x.shape[0]
    if training and rows == 1:
        # same refusal, same exception type as torch.nn.functional.batch_norm under the reference
        raise ValueError("Expected more than 1 value per channel when training, got input size %s"
                         % (torch.Size([1, x.shape[1], 1]),))
    next_img = None                 # the bf16 halo image of this layer's input, written by the previous layer's BatchNorm apply
    for i in range(n_layers):
        ximg = None
        W = P['%s.%d.0.conv.weight' % (prefix, i)]
        bias = P['%s.%d.0.conv.bias' % (prefix, i)]
        gamma = P['%s.%d.1.weight' % (prefix, i)]
        beta = P['%s.%d.1.bias' % (prefix, i)]
        rm = bufs['%s.%d.1.running_mean' % (prefix, i)]
        rv = bufs['%s.%d.1.running_var' % (prefix, i)]
        Co, Ci, k = W.shape
        pad = (k - 1) // 2
        packed = lambda W=W, i=i: run.cached('convfwd.%s.%d' % (prefix, i), [W], lambda: run.pack_conv_fwd(W))   # noqa: E731
        y = run.empty(rows, Co)
        invstd = run.empty(Co)
        K = Ci * k
        small = (not training) and ((rows + 127) // 128) * ((Co + 127) // 128) < 128 and K >= 1024
        if small:
            # Inference on a few hundred rows (one utterance): a rows x Co output is a few dozen 128-tiles on 256 CUs
            # (120 us per layer).  The bias moves into the BatchNorm shift -- (conv + b - mean) = conv - (mean - b), exact
            # in eval mode where mean is the running mean -- so the product has a plain epilogue and can be split along K.
            sk = max(1, min(10, K // 256))
            part = run.empty(sk, rows * Co)
            nv.gemm(part[0].view(rows, Co), x, packed(), convA=(T, Ci, pad, 1), splitk=sk, partials=part,
                    fast=(0 if (exact and run.x3) else run.fwdp))
            nv.splitk_reduce(part, sk, y)
        elif _conv16_ok(run, rows, T, Ci, k):
            # bf16 mode: the convolution as a product of sliding windows of a bf16 image with zero halo rows (csrc/gemm16.hip)
            W16 = run.cached('convfwd16.%s.%d' % (prefix, i), [W], lambda W=W: nv.pack_conv_bf16(W))
            ximg = next_img if next_img is not None else _halo_image(run, x, T, pad)
            nv.conv16(y, ximg, W16, rows // T, T, pad, bias=bias)
        else:
            _fg(run, y, x, packed(), bias=bias, convA=(T, Ci, pad, 1), exact=exact)
        if training:
            mean = run.empty(Co)
            nv.bn_stats(y, run.ws(Co), mean, invstd, rm, rv, BN_MOMENTUM, BN_EPS)
            bufs['%s.%d.1.num_batches_tracked' % (prefix, i)].add_(1)
        else:
            mean = rm
            if small:
                negb = run.cached('negbias.%s.%d' % (prefix, i), [bias], lambda bias=bias: bias.neg())
                mean = run.empty(Co)
                nv.copy2d(mean.view(1, Co), rm.view(1, Co), negb.view(1, Co))       # running mean - conv bias
            nv.bn_eval_invstd(rv, invstd, BN_EPS)
        z = run.empty(rows, Co)
        keep = masks[i] if masks is not None else None
        next_img = None
        if BN_FWD_IMAGE and training and lens is None and i + 1 < n_layers and Co % 4 == 0:
            Wn = P['%s.%d.0.conv.weight' % (prefix, i + 1)]
            kn = Wn.shape[2]
            padn = (kn - 1) // 2
            if _conv16_ok(run, rows, T, Co, kn) and T >= 2 * padn:
                # the next layer multiplies the bf16 halo image of z: written here, beside z, instead of by a cast pass over z
                next_img = run.empty16((rows // T) * (T + 2 * padn) + 2 * padn, Co)
        if next_img is not None:
            nv.bn_act_fwd_img(y, z, mean, invstd, gamma, beta, acts[i], keep.view(rows, Co) if keep is not None else None, 2.0,
                              next_img, T, padn)
        else:
            nv.bn_act_fwd(y, z, mean, invstd, gamma, beta, acts[i],
                          keep.view(rows, Co) if keep is not None else None, 2.0, lens, T if lens is not None else 0)
        saved.append(dict(x=x, y=y, z=z, mean=mean, invstd=invstd, keep=keep, W=W, act=acts[i], layer=i,
                          # the weight gradient multiplies the image again -- except the stack's own input: the reference
                          # masks mel_outputs IN PLACE after the postnet has run (model.py:491-495, `.data.masked_fill_`), so
                          # autograd's saved input of the first postnet convolution is the MASKED one; the backward therefore
                          # rebuilds layer 0's image from the slab as it is by then
                          ximg=ximg if (training and WGRAD_KK and i > 0) else None))
        x = z
    return x, saved


def _conv_stack_bwd(run, P, grads, prefix, saved, g, T, first_dx=None, first_dx_accumulate=False, G=None):
    """g: grad wrt the last activation (rows, C_last), overwritten.  Returns grad wrt the stack input
    (written to ``first_dx`` if given).  ``G(name, *shape)`` allocates a parameter gradient (a view of its
    data-parallel bucket when gradients are exchanged, distributed.GradSync.out)."""
    if G is None:
        G = lambda name, *shape: run.out_empty(*shape)                               # noqa: E731
    for i in range(len(saved) - 1, -1, -1):
        s = saved[i]
        W = s['W']
        Co, Ci, k = W.shape
        pad = (k - 1) // 2
        rows = g.shape[0]
        gamma = P['%s.%d.1.weight' % (prefix, i)]
        dgamma = G('%s.%d.1.weight' % (prefix, i), Co)
        dbeta = G('%s.%d.1.bias' % (prefix, i), Co)
        keep = s['keep']
        dbias = G('%s.%d.0.conv.bias' % (prefix, i), Co)
        gimg = None
        need_dx = i > 0 or first_dx is not None
        # bf16 mode, both followers of the BatchNorm backward on bf16 images (K-major weight gradient, window data gradient): its
        # output leaves as that image and as the bias gradient, never as an f32 slab (csrc/elementwise.hip
        # bn_act_bwd_stage2_img_kernel; bit-identical to the three separate passes)
        fused = (BN_BWD_IMAGE and _conv_wgrad_kk_ok(run, rows, T, Ci, Co) and (not need_dx or _conv16_ok(run, rows, T, Co, k))
                 and Co % 4 == 0 and T >= 2 * pad)
        if fused:
            gimg = run.empty16((rows // T) * (T + 2 * pad) + 2 * pad, Co)
            nv.bn_act_bwd_img(g, s['z'], s['y'], s['mean'], s['invstd'], gamma, s['act'],
                              keep.view(rows, Co) if keep is not None else None, 2.0, run.ws(Co), dgamma, dbeta, gimg, T, pad, dbias)
        else:
            nv.bn_act_bwd(g, s['z'], s['y'], s['mean'], s['invstd'], gamma, s['act'],
                          keep.view(rows, Co) if keep is not None else None, 2.0, run.ws(Co), dgamma, dbeta)
            run.colsum(g, dbias)
        grads['%s.%d.1.weight' % (prefix, i)] = dgamma
        grads['%s.%d.1.bias' % (prefix, i)] = dbeta
        grads['%s.%d.0.conv.bias' % (prefix, i)] = dbias
        # weight gradient: dW[co][(tap,ci)] = sum_r g[r][co] * x[r + tap - pad][ci]
        dW = G('%s.%d.0.conv.weight' % (prefix, i), Co, Ci, k)
        if _conv_wgrad_kk_ok(run, rows, T, Ci, Co):
            # bf16 mode: ONE K-major product over the two halo images (csrc/gemm16.hip, gemm16_kk): K runs over the image rows
            # (b, t) of pitch T + 2 pad, A = g's image from row `pad` on (zero where t >= T), B[k][(tap, ci)] = x's image
            # flat[k Ci + tap Ci + ci] -- overlapping rows, ldb = Ci -- i.e. the very windows the forward multiplied
            ximg = s.get('ximg')
            if ximg is None:
                ximg = _halo_image(run, s['x'], T, pad)
            if gimg is None:
                gimg = _halo_image(run, g, T, pad)
            _kk_to(run, dW, gimg[pad:], ximg, (rows // T) * (T + 2 * pad), Co, k * Ci, lda=Co, ldb=Ci, perm=(k, Ci))
        else:
            _rg(run, dW.view(Co, Ci * k), g, s['x'], a_km=True, b_kn=True, convB=(T, Ci, pad), perm=(k, Ci))
        grads['%s.%d.0.conv.weight' % (prefix, i)] = dW
        # data gradient
        if need_dx:
            if i == 0:
                dx = first_dx
                acc = first_dx_accumulate
            else:
                dx = run.empty(rows, Ci)
                acc = False
            if _conv16_ok(run, rows, T, Co, k):
                # dx[r][ci] = sum_{tap, co} g[r + pad - tap][co] W[co][ci][tap]: windows of g's halo image against the
                # tap-reversed weights [Ci][k Co]
                Wd16 = run.cached('convdgrad16.%s.%d' % (prefix, i), [W], lambda W=W: nv.pack_conv_bf16(W, reversed=True))
                nv.conv16(dx, gimg if gimg is not None else _halo_image(run, g, T, pad), Wd16, rows // T, T, pad, accumulate=acc)
            else:
                Wd = run.cached('convdgrad.%s.%d' % (prefix, i), [W], lambda W=W: run.pack_conv_dgrad(W))
                _ng(run, dx, g, Wd, accumulate=acc, convA=(T, Co, pad, -1))
            g = dx
    return g


# T2AMD_ENCODER_BATCH_PERSISTENT=0 keeps the encoder bi-LSTM of a batch on the launch chain (one launch per time step).
# The TRAINING forward takes the persistent launch only with T2AMD_ENCODER_BATCH_PERSISTENT_TRAIN=1: the launch itself is
# 0.5 ms shorter than its 170-launch chain (tools/ab_encoder_batch_persistent.py), but the training step gains from it only
# in a loop that reads the loss back every iteration (train.py, as the reference's: 61.35 vs 61.72 ms per step); in a loop
# that never synchronises (bench.py) the step measured 63.1 / 62.2 against 63.0 / 61.7 ms (profiles/r03_zd_*), so the
# default stays the chain there.
# Round 4: re-measured once the collector pause and the allocator calls were out of the step (tools/ab_engine_flag.py, four
# alternating blocks in one process): 60.93 vs 61.35 ms per training step with the persistent launch -- it is now the default
# for training too (T2AMD_ENCODER_BATCH_PERSISTENT_TRAIN=0 keeps the chain).
ENCODER_BATCH_PERSISTENT = os.environ.get('T2AMD_ENCODER_BATCH_PERSISTENT', '1') != '0'
ENCODER_BATCH_PERSISTENT_TRAIN = os.environ.get('T2AMD_ENCODER_BATCH_PERSISTENT_TRAIN', '1') != '0'


def _encoder_lstm_fwd(model, dev, d0, d1, regen_gx, reads, writes, poison=None, run=None):
    """Both directions of the encoder bi-LSTM of a batch (reference model.py:181-188): ONE persistent launch (csrc/
    decode_persist.hip, encoder_bilstm_batch_persistent_kernel: W_hh fragments in registers, h handed on through the output
    slab with write-through stores + step counters) instead of T dependent launches -- 4.3 vs 7.3 us per step at B = 64.  A
    bounded-spin give-up (its workgroups were not co-resident: a shared GPU) recomputes the pre-activations the kernel had
    begun to overwrite (``regen_gx``), runs the launch chain and stays on it for a while.  With ``poison`` (the training
    step) the status is NOT read back -- a host sync at the top of every step ties the step time to the host's enqueue speed
    (measured: 72 instead of 61 ms per step in the first process on a fresh box) -- a give-up turns ``poison[0]`` into NaN
    instead, the step goes non-finite and handle_nonfinite_step() finds the reason.  Returns the path taken."""
    import sys
    if ENCODER_BATCH_PERSISTENT and (poison is None or ENCODER_BATCH_PERSISTENT_TRAIN) and d0.B > 1 and not nv.validate_only():
        if getattr(model, '_enc_batch_backoff', 0) > 0:
            model._enc_batch_backoff -= 1
        elif nv.lstm_seq_batch_persistent_supported(d0, 2, torch.cuda.get_device_properties(dev).multi_processor_count) is None:
            ei32 = run.empty_i32 if run is not None else (lambda n: torch.empty(n, dtype=torch.int32, device=dev))
            flags = ei32(nv.lstm_seq_batch_persistent_flag_words(d0.B, d0.H, 2))
            status = ei32(1)
            nv.lstm_seq_fwd2_batch_persistent(d0, d1, flags, status, poison)
            if poison is not None or int(status.item()) == 0:
                model._enc_batch_timeouts = 0
                return 'persistent'
            print("tacotron2_amd: the persistent encoder kernel gave up (its workgroups were not co-resident within 30 ms -- is "
                  "the GPU shared?); running the launch chain", file=sys.stderr, flush=True)
            # exponential back-off, as _decode_persistent: 4, 8, ... up to 256 calls on the launch chain before the next try
            model._enc_batch_timeouts = getattr(model, '_enc_batch_timeouts', 0) + 1
            model._enc_batch_backoff = min(256, 2 << model._enc_batch_timeouts)
            regen_gx()
    nv.lstm_seq_fwd2(d0, d1, reads=reads, writes=writes)
    return 'launch chain'


# The teacher-forced decoder loop as ONE persistent launch (csrc/attention.hip dec_train_fwd_persistent_kernel<BF>; both precision
# modes since round 5 -- the fp32 parity mode runs it on the exact-f32 wide tile, T2AMD_TRAIN_FWD_PERSISTENT_FP32=0 keeps that mode
# on its chain; B <= 64, one workgroup per CU) -- the default since round 4; T2AMD_TRAIN_FWD_PERSISTENT=0 keeps the launch chain (two
# dependent launches per time step).  Bit-identical either way (tests/test_zz6_train_persistent_gpu.py); measured on one
# MI355X, alternating blocks in one process (tools/ab_train_fwd_persistent.py, profiles/r04_*_ab_train_fwd_persistent.json):
# forward 25.55 vs 25.86 ms, whole training step 61.8 vs 62.3 ms.
TRAIN_FWD_PERSISTENT = os.environ.get('T2AMD_TRAIN_FWD_PERSISTENT', '1') != '0'


# Give-ups of the persistent decoder loop seen by forwards that carry no poison word (eval mode: validation batches) -- counted
# here because nothing turns them into a NaN that a loss check would find; `give_up_counters()` reports them.
EVAL_GIVE_UPS = [0]


def _decoder_train_fwd(model, run, d, poison, reads, writes, regen_ga=None):
    """reference model.py:405-411.  The persistent launch when it is selected and the geometry fits this device.  TRAINING
    (``poison`` given): the status is not read back (no host sync in the training loop): a give-up -- its workgroups were not
    co-resident within 50 ms: a shared GPU -- turns ``poison[0]`` into NaN, the step goes non-finite and
    handle_nonfinite_step() switches back to the chain.  EVAL (``poison`` None: validation, reference train.py:133): nothing
    downstream would notice a NaN-free half-written slab, so the status IS read back (one sync per validation batch, as
    `_encoder_lstm_fwd` does): a give-up is counted, reported, the hoisted input projection the kernel had begun to overwrite is
    recomputed (``regen_ga``) and the launch chain runs -- with an exponential back-off before the next attempt."""
    if TRAIN_FWD_PERSISTENT and not nv.validate_only():          # (both precision modes since round 5)
        cus = torch.cuda.get_device_properties(run.dev).multi_processor_count
        if poison is None and getattr(model, '_dtp_eval_backoff', 0) > 0:
            model._dtp_eval_backoff -= 1
        elif nv.decoder_train_fwd_persistent_supported(d, cus) is None:
            flags = run.empty_i32(nv.decoder_train_fwd_persistent_flag_words(d.B, d.Ha))
            status = run.empty_i32(1)
            nv.decoder_train_fwd_persistent(d, flags, status, poison)
            if poison is not None:
                return 'persistent'
            if int(status.item()) == 0:
                model._dtp_eval_timeouts = 0
                return 'persistent'
            import sys
            EVAL_GIVE_UPS[0] += 1
            print("tacotron2_amd: the persistent decoder loop gave up in an eval-mode forward (its workgroups were not "
                  "co-resident -- is the GPU shared?); recomputing this batch on the launch chain", file=sys.stderr, flush=True)
            model._dtp_eval_timeouts = getattr(model, '_dtp_eval_timeouts', 0) + 1
            model._dtp_eval_backoff = min(256, 2 << model._dtp_eval_timeouts)
            if regen_ga is not None:
                regen_ga()               # GA is rewritten in place by the loop (pre-activations -> activated gates)
    nv.decoder_train_fwd_loop(d, reads=reads, writes=writes)
    return 'launch chain'


def _decoder_train_bwd(model, run, bw, poison, reads, writes):
    """reference model.py:405-411 under autograd: the launch chain (two dependent launches per time step).  The opt-in persistent
    launch of rounds 4-5 (TRAIN_BWD_PERSISTENT) was removed in round 6: 1 ms behind this chain for two rounds."""
    nv.decoder_train_bwd_loop(bw, reads=reads, writes=writes)
    return 'launch chain'


# BPTT of the encoder bi-LSTM as ONE persistent launch (csrc/decode_persist.hip, encoder_bilstm_batch_persistent_bwd_kernel)
# instead of 2 T dependent launches; T2AMD_ENCODER_BWD_PERSISTENT=0 keeps the chain.  Equal to the chain's gradients to the
# rounding of its split-bf16 recurrent product (~2^-17 per product).
ENCODER_BWD_PERSISTENT = os.environ.get('T2AMD_ENCODER_BWD_PERSISTENT', '1') != '0'


def _encoder_lstm_bwd(model, run, d0, d1, poison, reads, writes):
    """The persistent launch when selected and the geometry fits; the status is not read back (no host sync in the training
    loop): a give-up turns ``poison[0]`` into NaN, the gradient norm goes non-finite, the step is skipped and
    handle_nonfinite_step() selects the chain."""
    if ENCODER_BWD_PERSISTENT and ENCODER_BATCH_PERSISTENT and d0.B > 1 and not nv.validate_only():
        cus = torch.cuda.get_device_properties(run.dev).multi_processor_count
        if nv.lstm_seq_bwd2_batch_persistent_supported(d0, 2, cus) is None:
            flags = run.empty_i32(nv.lstm_seq_batch_persistent_flag_words(d0.B, d0.H, 2))
            status = run.empty_i32(1)
            nv.lstm_seq_bwd2_batch_persistent(d0, d1, flags, status, poison)
            return 'persistent'
    nv.lstm_seq_bwd2(d0, d1, reads=reads, writes=writes)
    return 'launch chain'


def _weight_cache(model):
    cache = getattr(model, '_weight_cache', None)
    if cache is None:
        cache = {}
        try:
            model._weight_cache = cache
        except Exception:                 # a model object that refuses attributes: no caching
            pass
    return cache


def invalidate_weight_cache(model):
    """Drop every packed / transposed / bf16 weight image of ``model``: the next call rebuilds them from the parameters.
    Needed only after writes the cache key cannot see (``param.data`` edits, raw-pointer optimisers); ``load_state_dict``,
    ``.to()`` / ``.cuda()`` / ``.half()`` and ``optim.FusedAdam`` invalidate by themselves."""
    cache = getattr(model, '_weight_cache', None)
    if cache is not None:
        cache.clear()


# Content guard of the weight-image cache.  A HEURISTIC, not a proof: the signature is the global L2 norm of
# all parameters, so an edit that preserves it (swapping two tensors of equal norm, a sign flip) is not seen -- the explicit
# remedies are model.invalidate_weight_cache() / engine.bump_weight_generation().  Writes through ``param.data`` (an EMA swap, ``p.data.copy_``) or
# by a raw-pointer optimiser change the weights without touching anything the cache key is made of, and the engine would
# go on multiplying by stale packed images while reading other operands live.  Every call therefore enqueues one pass over
# all parameters (the global-norm kernel of csrc/optim.hip: 113 MB, ~25 us on an MI355X) -- the weights' signature.
#   * key changed since the previous call (an optimiser step, load_state_dict, ...): the images are rebuilt anyway; the
#     signature is recorded with them, asynchronously -- a training loop never synchronises here;
#   * key unchanged (the images are about to be REUSED: inference calls, validation, a step after a ``.data`` edit): the
#     signature is read back (one event wait, ~tens of us) and compared with the recorded one BEFORE any image is used;
#     a difference drops the cache, so this very call already runs on fresh images, and a note on stderr names the
#     explicit remedy.  T2AMD_WEIGHT_GUARD=0 turns the guard off.
WEIGHT_GUARD = os.environ.get('T2AMD_WEIGHT_GUARD', '1') != '0'


def _guard_weights(model, run, P):
    if not WEIGHT_GUARD or nv.validate_only():
        return
    cache = run.cache
    params = [p for p in P.values() if p.dtype == torch.float32 and p.is_cuda]
    if not params or len(params) > nv.MAX_TENSORS:
        return
    key = (_PACK_GEN[0], str(run.dev)) + tuple((p.data_ptr(), p._version) for p in params)
    ptrs = tuple(p.data_ptr() for p in params)
    g = cache.get('__guard__')
    fresh = g is None or g['key'] != key
    if g is None or g['ptrs'] != ptrs:
        # buffers and the tensor list are made once per set of parameter storages (not per step: no allocation, no
        # pinned-memory call in the training loop)
        L, blocks = nv.tensor_list(params)
        g = dict(key=key, ptrs=ptrs, L=L, blocks=blocks, keep=params,
                 ws=torch.empty(blocks, dtype=torch.float64, device=run.dev),
                 dev=torch.empty(2, 2, dtype=torch.float32, device=run.dev),
                 host=torch.zeros(2, 2, dtype=torch.float32).pin_memory())
        cache['__guard__'] = g
    g['key'] = key
    slot = 0 if fresh else 1                       # 0: signature recorded with the images, 1: signature of this call
    nv.grad_norm(g['L'], g['blocks'], 0.0, g['ws'], g['dev'][slot])
    g['host'][slot].copy_(g['dev'][slot], non_blocking=True)
    if fresh:
        return
    ev = torch.cuda.Event()
    ev.record()
    ev.synchronize()                               # stream order: the slot-0 copy of an earlier call has landed too
    now, then = float(g['host'][1][0]), float(g['host'][0][0])
    if now != then and not (now != now and then != then):      # NaN weights: the same (non-)signature, not a change
        import sys
        print("tacotron2_amd: the parameters changed without their version counters changing (a write through "
              "param.data or a raw pointer); the packed weight images are rebuilt for this call.  Call "
              "model.invalidate_weight_cache() (or engine.bump_weight_generation()) after such an edit.",
              file=sys.stderr, flush=True)
        keep = dict(g)
        cache.clear()
        keep['host'][0].copy_(keep['host'][1])
        cache['__guard__'] = keep


# Re-promotion: a give-up demotes the process to the launch chains / separate-launch forms -- correct, but
# a single foreign kernel on the GPU should not cost a multi-day run its faster forms for good.  After REPROMOTE_AFTER clean
# training steps the forms that were selected before the demotion are selected again; if they give up again the interval
# doubles (up to 64 x).  0 = never re-promote.  T2AMD_REPROMOTE_AFTER sets it.
TRAIN_FWD_REPROMOTE_AFTER = int(os.environ.get('T2AMD_REPROMOTE_AFTER', '200'))
_DEMOTION = dict(active=False, count=0, clean=0, need=0, saved=None, repromotions=0, explicit=False, probation=0, given_up=False)
# After this many CONSECUTIVE failed re-promotions (each one sacrifices a training step: it is NaN-poisoned and skipped) the
# process stays on the launch chains for good: a GPU that is shared permanently is not going to stop being shared.
MAX_FAILED_REPROMOTIONS = 8


def _demote(saved_now):
    d = _DEMOTION
    if not d['active']:
        d['saved'] = saved_now
    d['active'] = True
    d['count'] += 1
    d['clean'] = 0
    d['probation'] = 0
    d['need'] = TRAIN_FWD_REPROMOTE_AFTER * (1 << min(d['count'] - 1, 6))
    if d['count'] > MAX_FAILED_REPROMOTIONS and not d['given_up']:
        d['given_up'] = True
        import sys
        print("tacotron2_amd: %d give-ups in a row, each right after a re-promotion: the launch chains stay selected for the rest "
              "of this process (the GPU is shared for good?)" % d['count'], file=sys.stderr, flush=True)


def note_clean_step():
    """Call from the training loop after a step whose loss / gradient norm was FINITE (tacotron2_amd/train.py does).  While the
    process is demoted this is what counts towards re-promotion -- not training forwards, which also count steps that turned out
    non-finite for other reasons and the extra forwards of gradient accumulation.  A loop that never calls it (the
    reference's unmodified train.py) keeps the old count by forwards."""
    d = _DEMOTION
    d['explicit'] = True
    if d['active']:
        d['clean'] += 1
    elif d['probation'] > 0:
        d['probation'] -= 1
        if d['probation'] == 0:
            d['count'] = 0                 # the re-promoted forms survived as long as they had been away: the streak is over


def _note_training_step(log=None):
    """One training forward is about to run.  Restores the demoted forms when enough clean steps have been seen (counted by
    note_clean_step(), or -- for a loop that does not call it -- by these forwards themselves)."""
    d = _DEMOTION
    if not d['active']:
        if not d['explicit'] and d['probation'] > 0:
            d['probation'] -= 1
            if d['probation'] == 0:
                d['count'] = 0
        return False
    if TRAIN_FWD_REPROMOTE_AFTER <= 0 or d['given_up']:
        return False
    if not d['explicit']:
        d['clean'] += 1
    if d['clean'] <= d['need']:
        return False
    global TRAIN_FWD_PERSISTENT, ENCODER_BATCH_PERSISTENT
    sv = d['saved']
    TRAIN_FWD_PERSISTENT, ENCODER_BATCH_PERSISTENT = sv['fwd'], sv['enc']
    nv.set_attn_fwd_fused(sv['attn_fwd_fused'])
    nv.set_attn_bwd_fused(sv['attn_bwd_fused'])
    nv.set_bptt_cell_fold(sv['cell_fold'])
    d['active'], d['saved'] = False, None
    d['repromotions'] += 1
    d['probation'] = d['need']
    msg = ("tacotron2_amd: %d clean training steps since the last abandoned hand-off: the one-launch / persistent forms are "
           "selected again (re-promotion %d; a further give-up doubles the interval, %d in a row end it)"
           % (d['clean'] - (0 if d['explicit'] else 1), d['repromotions'], MAX_FAILED_REPROMOTIONS))
    if log is not None:
        log(msg)
    else:
        import sys
        print(msg, file=sys.stderr, flush=True)
    return True


def give_up_counters():
    """What a bench line / run log needs to show that nobody fell back silently."""
    d = _DEMOTION
    return dict(demotions=d['count'], demoted_now=bool(d['active']), repromotions=d['repromotions'],
                repromotion_given_up=bool(d['given_up']), eval_give_ups=EVAL_GIVE_UPS[0])


def handle_nonfinite_step(log=None):
    """Call when a training step produced a non-finite loss / gradient norm.  If the reason is an ABANDONED in-launch
    hand-off -- of the one-launch attention forms (their four workgroups per utterance were not co-resident within 50 ms: a
    shared or partitioned GPU), of the persistent decoder loop (TRAIN_FWD_PERSISTENT: arrival census or
    a bounded spin) or of the persistent encoder launches; the kernels then poison the step with NaN rather than use
    half-exchanged data -- say so and select the launch chains and the separate-launch attention forms: this CLEARS
    ``TRAIN_FWD_PERSISTENT`` (and ``ENCODER_BATCH_PERSISTENT`` for an encoder give-up) and calls
    ``set_attn_fwd_fused(0)`` / ``set_attn_bwd_fused(0)`` / ``set_bptt_cell_fold(0)`` -- bit-identical results, no co-residency
    assumption -- until `_note_training_step` re-promotes them after TRAIN_FWD_REPROMOTE_AFTER clean steps.  Returns the number
    of abandoned hand-offs (0: the non-finite values have another cause)."""
    global ENCODER_BATCH_PERSISTENT, TRAIN_FWD_PERSISTENT
    ne = nv.encoder_handoff_timeouts(reset=True)
    n = nv.attn_handoff_timeouts(reset=True)
    if ne > 0 or n > 0:
        _demote(dict(fwd=TRAIN_FWD_PERSISTENT, enc=ENCODER_BATCH_PERSISTENT,
                     attn_fwd_fused=nv.get_attn_fwd_fused(), attn_bwd_fused=nv.get_attn_bwd_fused(),
                     cell_fold=nv.get_bptt_cell_fold()))
    if ne > 0:
        ENCODER_BATCH_PERSISTENT = False
        msg = ("tacotron2_amd: the persistent encoder launch gave up %d time(s) (its workgroups were not co-resident within "
               "30 ms -- is the GPU shared or partitioned?); that step was poisoned with NaN and is skipped; the encoder "
               "runs on the launch chain from here on" % ne)
        if log is not None:
            log(msg)
        else:
            import sys
            print(msg, file=sys.stderr, flush=True)
    if n > 0:
        TRAIN_FWD_PERSISTENT = False
        nv.set_attn_fwd_fused(0)
        nv.set_attn_bwd_fused(0)
        nv.set_bptt_cell_fold(0)
        msg = ("tacotron2_amd: %d in-launch attention hand-off(s) timed out (the workgroups of an utterance were not "
               "co-resident within 50 ms -- is the GPU shared or partitioned?); that step was poisoned with NaN and is "
               "skipped; the separate-launch forms are selected from here on" % n)
        if TRAIN_FWD_REPROMOTE_AFTER > 0:
            msg += " (and re-selected after %d clean steps)" % _DEMOTION['need']
        if log is not None:
            log(msg)
        else:
            import sys
            print(msg, file=sys.stderr, flush=True)
    return n + ne


def _cached_bias_sum(run, tag, b1, b2):
    return run.cached(tag, [b1, b2], lambda: _bias_sum(run, b1, b2))


_ATT_Q = 'decoder.attention_layer.query_layer.linear_layer.weight'
_ATT_M = 'decoder.attention_layer.memory_layer.linear_layer.weight'
_ATT_V = 'decoder.attention_layer.v.linear_layer.weight'
_ATT_D = 'decoder.attention_layer.location_layer.location_dense.linear_layer.weight'
_ATT_C = 'decoder.attention_layer.location_layer.location_conv.conv.weight'


def _embed_attention(run, P):
    """The attention kernels are compiled for attention_dim 128, 32 location filters, 31 taps (hparams.py:65-70
    defaults).  A smaller geometry is the same arithmetic with zeros in the missing rows / filters / outer taps --
    a zero row of the query, memory and location weights gives tanh(0) = 0 under a zero entry of v, a zero tap
    multiplies the same zero padding the shorter kernel would never have read -- so its five weights are embedded
    once per weight version into compiled-geometry images.  Returns (P with those five entries replaced, their names)
    or (P, ()) for the default geometry."""
    Wq, Wm, Wv, Wd, Wc = (P[n] for n in (_ATT_Q, _ATT_M, _ATT_V, _ATT_D, _ATT_C))
    A, F, K = Wq.shape[0], Wc.shape[0], Wc.shape[2]
    AD, FD, KD = nv.ATT_DIM, nv.LOC_FILTERS, nv.LOC_KERNEL
    if (A, F, K) == (AD, FD, KD):
        return P, ()

    def embed():
        o = (KD - K) // 2
        q = run.zeros(AD, Wq.shape[1]); q[:A] = Wq
        m = run.zeros(AD, Wm.shape[1]); m[:A] = Wm
        v = run.zeros(1, AD); v[:, :A] = Wv
        d = run.zeros(AD, FD); d[:A, :F] = Wd
        c = run.zeros(FD, 2, KD); c[:F, :, o:o + K] = Wc
        return q, m, v, d, c
    emb = run.cached('attention.embedded', [Wq, Wm, Wv, Wd, Wc], embed)
    P = dict(P)
    P.update(zip((_ATT_Q, _ATT_M, _ATT_V, _ATT_D, _ATT_C), emb))
    return P, (_ATT_Q, _ATT_M, _ATT_V, _ATT_D, _ATT_C)


def _crop_attention_grad(name, grad, shape):
    """The block of a compiled-geometry gradient that belongs to the model's own (smaller) weight."""
    if name == _ATT_C:
        o = (nv.LOC_KERNEL - shape[2]) // 2
        return grad[:shape[0], :, o:o + shape[2]]
    return grad[:shape[0], :shape[1]]


def _fold_U(run, Wdense, Wconv):
    U = run.empty(nv.ATT_DIM * nv.LOC_TAPS)
    nv.fold_location(Wdense, Wconv, U)
    return U


def _packed_projection(run, P, Cm, Hd, E):
    """[linear_projection ; gate_layer] as one (Cm+1, Hd+E) matrix and one bias (reference model.py:373-378)."""
    Wp = P['decoder.linear_projection.linear_layer.weight']      # (Cm, Hd+E)
    Wg = P['decoder.gate_layer.linear_layer.weight']             # (1, Hd+E)
    bp = P['decoder.linear_projection.linear_layer.bias']
    bg = P['decoder.gate_layer.linear_layer.bias']

    def pack():
        Wpg = run.empty(Cm + 1, Hd + E)
        nv.copy2d(Wpg[:Cm], Wp)
        nv.copy2d(Wpg[Cm:], Wg)
        bpg = run.empty(Cm + 1)
        nv.copy2d(bpg[:Cm].view(1, Cm), bp.view(1, Cm))
        nv.copy2d(bpg[Cm:].view(1, 1), bg.view(1, 1))
        return Wpg, bpg
    return run.cached('Wpg', [Wp, Wg, bp, bg], pack)


# ----------------------------------------------------------------------------
# forward (training / teacher-forced)
# ----------------------------------------------------------------------------
def _forward(model, P, bufs, text, in_lens, mels, max_len, out_lens, training):
    hp = model.hparams
    dev = text.device
    if not text.is_cuda and not nv.validate_only():
        raise NativeError("tacotron2_amd: the engine runs on the MI355X only (got %s tensors). "
                          "There is no CPU path; the CPU oracle lives in oracle/ for tests." % dev)
    nv.load()
    c = _Ctx()
    c.arena = _arena_acquire(dev)        # handed back by backward(), or by c's destructor when there is none
    run = _Run(dev, getattr(model, 'precision', 'fp32'), _weight_cache(model), c.arena)
    _guard_weights(model, run, P)
    P, _ = _embed_attention(run, P)
    ms = MaskSource(model.dropout_masks, dev, run)
    B = text.shape[0]
    Ti = int(max_len)
    To = mels.shape[2]
    E = hp.encoder_embedding_dim
    Ha, Hd, Pd = hp.attention_rnn_dim, hp.decoder_rnn_dim, hp.prenet_dim
    Cm = hp.n_mel_channels
    He = E // 2
    A = nv.ATT_DIM
    if text.shape[1] != Ti:
        text = text[:, :Ti]
    text = text.contiguous()
    mels = mels.contiguous().float()
    lens32 = in_lens.to(torch.int32).contiguous()
    olens32 = out_lens.to(torch.int32).contiguous() if (model.mask_padding and out_lens is not None) else None
    c.B, c.Ti, c.To, c.text, c.lens32, c.olens32 = B, Ti, To, text, lens32, olens32
    c.training = training

    # ---- encoder: embedding -> 3 x (conv k5 + BN + relu + dropout) -> bi-LSTM -----------------
    rowsE = B * Ti
    emb = run.empty(rowsE, E)
    nv.embedding_fwd(text, P['embedding.weight'], emb)                                   # model.py:503
    nconv = hp.encoder_n_convolutions
    enc_masks = [ms.get('enc', i, (B, Ti, E), 0.5) for i in range(nconv)] if training else None
    x3, c.enc_saved = _conv_stack_fwd(run, P, bufs, 'encoder.convolutions', nconv, emb, Ti,
                                      [1] * nconv, enc_masks, training, exact=True)      # model.py:174-175
    memory = run.empty(B, Ti, E)
    c.enc_lstm = []
    for d, sfx in enumerate(('', '_reverse')):                                           # model.py:181-188
        Wih = P['encoder.lstm.weight_ih_l0' + sfx]
        Whh = P['encoder.lstm.weight_hh_l0' + sfx]
        bsum = _cached_bias_sum(run, 'enc_bias' + sfx, P['encoder.lstm.bias_ih_l0' + sfx], P['encoder.lstm.bias_hh_l0' + sfx])
        GX = run.empty(rowsE, 4 * He)
        _fg(run, GX, x3, Wih, bias=bsum)
        Cst = run.empty(Ti, B, He)
        desc = nv.LstmSeq()
        desc.B, desc.T, desc.H, desc.reverse = B, Ti, He, d
        desc.Whh = nv.ptr(Whh)
        desc.GX = nv.ptr(GX)
        out_view = memory.view(rowsE, E)[:, d * He:(d + 1) * He]
        desc.out, desc.ld_out = nv.ptr(out_view), E
        desc.C = nv.ptr(Cst)
        desc.lens = nv.ptr(lens32, torch.int32)
        c.enc_lstm.append(dict(GX=GX, C=Cst, Whh=Whh, Wih=Wih, desc=desc, bsum=bsum))

    def regen_gx():
        for L_ in c.enc_lstm:
            _fg(run, L_['GX'], x3, L_['Wih'], bias=L_['bsum'])
    model.last_encoder_path = _encoder_lstm_fwd(
        model, text.device, c.enc_lstm[0]['desc'], c.enc_lstm[1]['desc'], regen_gx,
        reads=[L_[k_] for L_ in c.enc_lstm for k_ in ('Whh', 'GX')] + [lens32],
        writes=[memory] + [L_['C'] for L_ in c.enc_lstm], poison=memory if training else None, run=run)
    c.x3, c.memory = x3, memory

    # ---- decoder: hoisted dense parts ------------------------------------------------------
    rowsD = To * B
    x0 = run.empty(To, B, Cm)
    nv.frames_to_time_major(mels, x0)                                                    # model.py:396-398
    W1 = P['decoder.prenet.layers.0.linear_layer.weight']
    W2 = P['decoder.prenet.layers.1.linear_layer.weight']
    k0 = ms.get('prenet', 0, (To, B, Pd), 0.5)
    k1 = ms.get('prenet', 1, (To, B, Pd), 0.5)
    p1 = run.empty(rowsD, Pd)
    p2 = run.empty(rowsD, Pd)
    _fg(run, p1, x0.view(rowsD, Cm), W1, act=1, keep=k0.view(rowsD, Pd), keep_scale=2.0, exact=True)  # model.py:99, 399
    _fg(run, p2, p1, W2, act=1, keep=k1.view(rowsD, Pd), keep_scale=2.0, exact=True)
    Wmem = P['decoder.attention_layer.memory_layer.linear_layer.weight']
    pm = run.empty(B, Ti, A)
    _fg(run, pm.view(rowsE, A), memory.view(rowsE, E), Wmem)                              # model.py:288

    Wih_a, Whh_a = P['decoder.attention_rnn.weight_ih'], P['decoder.attention_rnn.weight_hh']
    Wih_d, Whh_d = P['decoder.decoder_rnn.weight_ih'], P['decoder.decoder_rnn.weight_hh']
    bias_a = _cached_bias_sum(run, 'bias_a', P['decoder.attention_rnn.bias_ih'], P['decoder.attention_rnn.bias_hh'])
    bias_d = _cached_bias_sum(run, 'bias_d', P['decoder.decoder_rnn.bias_ih'], P['decoder.decoder_rnn.bias_hh'])

    def pack_a_rec():
        W = run.empty(4 * Ha, E + Ha)
        nv.copy2d(W[:, :E], Wih_a[:, Pd:Pd + E])
        nv.copy2d(W[:, E:], Whh_a)
        return W

    def pack_d_cat():
        W = run.empty(4 * Hd, Ha + E + Hd)
        nv.copy2d(W[:, :Ha + E], Wih_d)
        nv.copy2d(W[:, Ha + E:], Whh_d)
        return W

    Wa_rec = run.cached('Wa_rec', [Wih_a, Whh_a], pack_a_rec)
    Wd_cat = run.cached('Wd_cat', [Wih_d, Whh_d], pack_d_cat)
    Wq = P['decoder.attention_layer.query_layer.linear_layer.weight'].contiguous()       # (A, Ha)
    Wdense = P['decoder.attention_layer.location_layer.location_dense.linear_layer.weight']
    Wconv = P['decoder.attention_layer.location_layer.location_conv.conv.weight']
    U = run.cached('U', [Wdense, Wconv], lambda: _fold_U(run, Wdense, Wconv))
    vvec = P['decoder.attention_layer.v.linear_layer.weight'].view(-1)

    GA = run.empty(To, B, 4 * Ha)

    def project_ga():
        if run.bf16 and WGRAD16 and Pd % 64 == 0 and (Pd + E) % 8 == 0 and not nv.validate_only():
            # bf16 mode: the hoisted input projection of the attention LSTM on the bf16-resident product (csrc/gemm16.hip)
            Wih_a16 = run.cached('Wih_a16', [Wih_a], lambda: run.cast16(Wih_a))
            nv.gemm16_tn(GA.view(rowsD, 4 * Ha), run.cast16(p2), Wih_a16[:, :Pd], bias=bias_a)
        else:
            _fg(run, GA.view(rowsD, 4 * Ha), p2, Wih_a[:, :Pd], bias=bias_a)
    project_ga()

    att_p, dec_p = hp.p_attention_dropout, hp.p_decoder_dropout
    keep_att = ms.get('att', None, (To, B, Ha), att_p) if training else None
    keep_dec = ms.get('dec', None, (To, B, Hd), dec_p) if training else None

    d = nv.DecTrain()
    d.B, d.Ti, d.To, d.E, d.Ha, d.Hd = B, Ti, To, E, Ha, Hd
    slabs = dict(HA=run.empty(To, B, Ha), CA=run.empty(To, B, Ha), GD=run.empty(To, B, 4 * Hd),
                 HD=run.empty(To, B, Hd), CD=run.empty(To, B, Hd), CTX=run.empty(To, B, E),
                 Q=run.empty(To, B, A), ALIGN=run.out_empty(B, To, Ti), CUM=run.empty(To, B, Ti),
                 cum_work=run.empty(B, Ti),
                 attn_ws=run.empty(nv.attn_fwd_ws_floats(B, Ti) + nv.attn_bwd_ws_floats(B, Ti)))
    d.Wa_rec, d.Wd_cat, d.bias_d = nv.ptr(Wa_rec), nv.ptr(Wd_cat), nv.ptr(bias_d)
    d.Wq, d.U, d.v = nv.ptr(Wq), nv.ptr(U), nv.ptr(vvec)
    d.GA, d.memory, d.pm = nv.ptr(GA), nv.ptr(memory), nv.ptr(pm)
    d.lens = nv.ptr(lens32, torch.int32)
    d.keep_att = nv.ptr(keep_att, torch.uint8)
    d.keep_dec = nv.ptr(keep_dec, torch.uint8)
    d.scale_att, d.scale_dec = nv.scale_for(att_p), nv.scale_for(dec_p)
    for k_, v_ in slabs.items():
        setattr(d, k_, nv.ptr(v_))
    if run.bf16:
        # bf16 compute mode: bf16 copies of the packed weights and of the recurrent operand slabs (the LSTM
        # products run on the bf16 MFMA; cell state, gates and every saved slab stay f32)
        c.bf16 = dict(Wa_rec16=run.cached('Wa_rec16', [Wih_a, Whh_a], lambda: run.cast16(Wa_rec)),
                      Wd_cat16=run.cached('Wd_cat16', [Wih_d, Whh_d], lambda: run.cast16(Wd_cat)),
                      HA16=run.empty16(To, B, Ha),
                      HD16=run.empty16(To, B, Hd), CTX16=run.empty16(To, B, E), memory16=run.cast16(memory),
                      Wq16=run.cached('Wq16', [Wq], lambda: run.cast16(Wq)))
        d.bf16 = 1
        for k_, v_ in c.bf16.items():
            setattr(d, k_, nv.ptr(v_, torch.bfloat16))
    elif run.x3:
        # 'bf16x3' mode: SPLIT-bf16 images (4 bytes per k: hi + lo) of the packed weights and of the three recurrent operand slabs;
        # the tile epilogues / K_c write the slabs' images next to the f32 values.  No bf16 memory / W_q: the attention stays f32.
        c.bf16 = dict(Wa_rec16=run.cached('Wa_rec16x3', [Wih_a, Whh_a], lambda: run.split16(Wa_rec)),
                      Wd_cat16=run.cached('Wd_cat16x3', [Wih_d, Whh_d], lambda: run.split16(Wd_cat)),
                      HA16=run.empty16(To, B, 2 * Ha), HD16=run.empty16(To, B, 2 * Hd), CTX16=run.empty16(To, B, 2 * E))
        d.bf16 = 3
        for k_, v_ in c.bf16.items():
            setattr(d, k_, nv.ptr(v_, torch.bfloat16))
    op16 = run.bf16 or run.x3
    model.last_train_decoder_path = _decoder_train_fwd(
        model, run, d, poison=slabs['CTX'] if training else None,                         # model.py:405-411
        reads=[Wa_rec, Wd_cat, bias_d, Wq, U, vvec, memory, pm, lens32, keep_att, keep_dec]
        + ([c.bf16[k_] for k_ in ('Wa_rec16', 'Wd_cat16')] if op16 else []) + ([c.bf16[k_] for k_ in ('memory16', 'Wq16')] if run.bf16 else []),
        writes=[GA] + list(slabs.values()) + ([c.bf16[k_] for k_ in ('HA16', 'HD16', 'CTX16')] if op16 else []),
        regen_ga=project_ga)

    # mel + gate projection over all steps (model.py:373-378)
    Wpg, bpg = _packed_projection(run, P, Cm, Hd, E)             # rows: Cm mel channels, then the gate
    PG = run.empty(rowsD, Cm + 1)
    _fg(run, PG, slabs['HD'].view(rowsD, Hd), Wpg[:, :Hd])
    _fg(run, PG, slabs['CTX'].view(rowsD, E), Wpg[:, Hd:], accumulate=True, bias=bpg)
    mel_cl = run.empty(B, To, Cm)
    gate = run.out_empty(B, To)
    nv.split_projection(PG, mel_cl, gate, olens32)                                       # model.py:326-336, 495

    # ---- postnet (model.py:141-146) -------------------------------------------------------
    npost = hp.postnet_n_convolutions
    rowsP = B * To
    post_chans = [hp.postnet_embedding_dim] * (npost - 1) + [Cm]
    post_masks = [ms.get('post', i, (B, To, post_chans[i]), 0.5) for i in range(npost)] if training else None
    post_cl, c.post_saved = _conv_stack_fwd(run, P, bufs, 'postnet.convolutions', npost,
                                            mel_cl.view(rowsP, Cm), To, [2] * (npost - 1) + [0],
                                            post_masks, training)
    mel = run.out_empty(B, Cm, To)
    mel_post = run.out_empty(B, Cm, To)
    nv.finalize_outputs(mel_cl, post_cl.view(B, To, Cm), mel, mel_post, olens32)         # model.py:511, 487-497

    c.run = run
    c.dec = d
    c.keep = dict(att=keep_att, dec=keep_dec, k0=k0, k1=k1)
    c.slabs = slabs
    c.tensors = dict(x0=x0, p1=p1, p2=p2, pm=pm, GA=GA, Wa_rec=Wa_rec, Wd_cat=Wd_cat, bias_d=bias_d,
                     Wq=Wq, U=U, Wpg=Wpg, mel_cl=mel_cl, vvec=vvec, lens32=lens32)
    return (mel, mel_post, gate, slabs['ALIGN']), c


# ----------------------------------------------------------------------------
# backward
# ----------------------------------------------------------------------------
def _backward(model, P, c, d_mel, d_post, d_gate, d_align):
    hp = model.hparams
    run = c.run
    g = {}
    B, Ti, To = c.B, c.Ti, c.To
    E = hp.encoder_embedding_dim
    Ha, Hd, Pd = hp.attention_rnn_dim, hp.decoder_rnn_dim, hp.prenet_dim
    Cm = hp.n_mel_channels
    He = E // 2
    A = nv.ATT_DIM
    rowsD, rowsE, rowsP = To * B, B * Ti, B * To
    S, T = c.slabs, c.tensors

    def cont(t):
        return t.contiguous() if t is not None else None

    # data parallel: buckets are all-reduced over RCCL as soon as they are complete (distributed.py)
    sync = getattr(model, '_grad_sync', None)
    if sync is not None:
        sync.start(run.dev)

    def G(name, *shape):
        """Where the kernels write the gradient of parameter ``name``: straight into its bucket when gradients
        are exchanged (no packing pass before the all-reduce), a fresh buffer otherwise."""
        return sync.out(name, shape) if sync is not None else run.out_empty(*shape)

    # a smaller attention geometry: the kernels produce compiled-geometry gradients, cropped into place further down
    own_shape = {n: tuple(P[n].shape) for n in (_ATT_Q, _ATT_M, _ATT_V, _ATT_D, _ATT_C)}
    P, embedded = _embed_attention(run, P)
    if embedded:
        G_own, staged = G, {}

        def G(name, *shape):
            if name in embedded:
                staged[name] = run.empty(*shape)
                return staged[name]
            return G_own(name, *shape)

    # ---- output boundary -> postnet backward ----------------------------------------------
    dmel_cl = run.empty(B, To, Cm)
    dpost_cl = run.empty(B, To, Cm)
    nv.grads_to_channel_last(cont(d_mel), cont(d_post), dmel_cl, dpost_cl)
    _conv_stack_bwd(run, P, g, 'postnet.convolutions', c.post_saved, dpost_cl.view(rowsP, Cm), To,
                    first_dx=dmel_cl.view(rowsP, Cm), first_dx_accumulate=True, G=G)
    if sync is not None:
        sync.bucket_ready('postnet')             # travels while the decoder BPTT below runs
    dout = run.empty(rowsD, Cm + 1)
    nv.gather_dout(dmel_cl, cont(d_gate), dout)

    # ---- projection backward --------------------------------------------------------------
    Wpg = T['Wpg']
    DHC = run.empty(rowsD, Hd + E)
    _ng(run, DHC, dout, Wpg, b_kn=True)
    dWpg_h = run.empty(Cm + 1, Hd)
    dWpg_c = run.empty(Cm + 1, E)
    _rg(run, dWpg_h, dout, S['HD'].view(rowsD, Hd), a_km=True, b_kn=True)
    _rg(run, dWpg_c, dout, S['CTX'].view(rowsD, E), a_km=True, b_kn=True)
    dWp = G('decoder.linear_projection.linear_layer.weight', Cm, Hd + E)
    dWg = G('decoder.gate_layer.linear_layer.weight', 1, Hd + E)
    nv.copy2d(dWp[:, :Hd], dWpg_h[:Cm]); nv.copy2d(dWp[:, Hd:], dWpg_c[:Cm])
    nv.copy2d(dWg[:, :Hd], dWpg_h[Cm:]); nv.copy2d(dWg[:, Hd:], dWpg_c[Cm:])
    dbpg = run.empty(Cm + 1)
    run.colsum(dout, dbpg)
    dbp = G('decoder.linear_projection.linear_layer.bias', Cm)
    dbg = G('decoder.gate_layer.linear_layer.bias', 1)
    nv.copy2d(dbp.view(1, Cm), dbpg[:Cm].view(1, Cm))
    nv.copy2d(dbg.view(1, 1), dbpg[Cm:].view(1, 1))
    g['decoder.linear_projection.linear_layer.weight'] = dWp
    g['decoder.gate_layer.linear_layer.weight'] = dWg
    g['decoder.linear_projection.linear_layer.bias'] = dbp
    g['decoder.gate_layer.linear_layer.bias'] = dbg

    # ---- BPTT through the decoder loop ----------------------------------------------------
    Wq = P['decoder.attention_layer.query_layer.linear_layer.weight']
    Wih_a, Whh_a = P['decoder.attention_rnn.weight_ih'], P['decoder.attention_rnn.weight_hh']
    Wih_d, Whh_d = P['decoder.decoder_rnn.weight_ih'], P['decoder.decoder_rnn.weight_hh']

    def transposed(src, rows, cols):
        Wt = run.empty(rows, cols)
        nv.transpose(Wt, src)
        return Wt

    Wa_recT = run.cached('Wa_recT', [Wih_a, Whh_a], lambda: transposed(T['Wa_rec'], E + Ha, 4 * Ha))
    Wd_catT = run.cached('Wd_catT', [Wih_d, Whh_d], lambda: transposed(T['Wd_cat'], Ha + E + Hd, 4 * Hd))
    ns = DGRAD_SPLIT                                        # split-K factor of the two BPTT dgrad GEMMs
    bw = nv.DecTrainBwd()
    bw.f = c.dec
    bw.Wa_recT, bw.Wd_catT = nv.ptr(Wa_recT), nv.ptr(Wd_catT)
    bw.DHC = nv.ptr(DHC)
    bw.d_align = nv.ptr(cont(d_align))
    bw.nsplit = ns
    # bf16 mode with whole-sequence bf16 gate-gradient slabs: weight gradients, bias gradients and the prenet's data gradient all
    # read THOSE, so the f32 slabs are neither allocated nor written (2 x 912 MB of stores per step at B = 64 / To = 870)
    # the decoder LSTM's input gradients of step t are read one step later and never again: a ring of slabs (stays in L2 / MALL)
    # instead of To of them, unless the two-stream loop lets that chain run ahead
    ring = DXD_RING if (DXD_RING >= 3 and nv.decoder_streams() == 1) else 0
    drop32 = GATE_GRADS_BF16_ONLY and BIAS_GRAD16 and run.bf16 and WGRAD16 and B % 8 == 0 and not nv.validate_only()
    out = dict(DGA=None if drop32 else run.empty(To, B, 4 * Ha), DGD=None if drop32 else run.empty(To, B, 4 * Hd), DCTX=run.empty(To, B, E),
               DQ=run.empty(To, B, A), d_pm=run.empty(B, Ti, A), dU_acc=run.empty(B, A, nv.LOC_TAPS),
               dv_acc=run.empty(B, A), dXd=run.empty(ring or To, ns, B, Ha + E + Hd), dXa=run.empty(ns, B, E + Ha),
               dc_a=run.empty(B, Ha), dc_d=run.empty(B, Hd), dwin_part=run.empty(nv.ATT_SLICES, B, 2, Ti),
               dcum_acc=run.empty(B, Ti), dq_h=run.empty(nv.ATT_SLICES, B, Ha))
    for k_, v_ in out.items():
        setattr(bw, k_, nv.ptr(v_))
    bw.dXd_ring = ring
    if run.bf16:
        # the bf16 gate gradients are kept for every step ([To][B][4H] slabs) when the weight gradients are formed from them
        slab16 = WGRAD16 and B % 8 == 0 and not nv.validate_only()
        nst = To if slab16 else 1
        b16 = dict(Wa_recT16=run.cached('Wa_recT16', [Wih_a, Whh_a], lambda: run.cast16(Wa_recT)),
                   Wd_catT16=run.cached('Wd_catT16', [Wih_d, Whh_d], lambda: run.cast16(Wd_catT)),
                   DGA16=run.empty16(nst, B, 4 * Ha), DGD16=run.empty16(nst, B, 4 * Hd))
        for k_, v_ in b16.items():
            setattr(bw, k_, nv.ptr(v_, torch.bfloat16))
        if slab16:
            bw.dg16_step_a, bw.dg16_step_d = B * 4 * Ha, B * 4 * Hd
    elif run.x3:
        # split-bf16 images of the transposed weights and of ONE step's gate gradients (written by the cell backward)
        b16 = dict(Wa_recT16=run.cached('Wa_recT16x3', [Wih_a, Whh_a], lambda: run.split16(Wa_recT)),
                   Wd_catT16=run.cached('Wd_catT16x3', [Wih_d, Whh_d], lambda: run.split16(Wd_catT)),
                   DGA16=run.empty16(1, B, 8 * Ha), DGD16=run.empty16(1, B, 8 * Hd))
        for k_, v_ in b16.items():
            setattr(bw, k_, nv.ptr(v_, torch.bfloat16))
    op16 = run.bf16 or run.x3
    model.last_train_decoder_bwd_path = _decoder_train_bwd(model, run, bw, out['dv_acc'],
                              reads=[Wa_recT, Wd_catT, DHC, cont(d_align), T['Wq'], T['U'], T['vvec'], T['pm'], c.memory,
                                     T['lens32'], T['GA'], c.keep['att'], c.keep['dec']]
                              + [S[k_] for k_ in ('HA', 'CA', 'GD', 'HD', 'CD', 'CTX', 'Q', 'ALIGN', 'CUM')]
                              + ([b16['Wa_recT16'], b16['Wd_catT16']] if op16 else [])
                              + ([c.bf16['memory16'], c.bf16['Wq16']] if run.bf16 else []),
                              writes=[v_ for v_ in out.values() if v_ is not None] + [S['attn_ws']] + ([b16['DGA16'], b16['DGD16']] if op16 else []))
    DGA, DGD, DCTX, DQ, d_pm = (out[k_] for k_ in ('DGA', 'DGD', 'DCTX', 'DQ', 'd_pm'))
    DGA2, DGD2 = (None, None) if drop32 else (DGA.view(rowsD, 4 * Ha), DGD.view(rowsD, 4 * Hd))

    # location layer + v
    Wdense = P['decoder.attention_layer.location_layer.location_dense.linear_layer.weight']
    Wconv = P['decoder.attention_layer.location_layer.location_conv.conv.weight']
    dWdense = G('decoder.attention_layer.location_layer.location_dense.linear_layer.weight', A, nv.LOC_FILTERS)
    dWconv = G('decoder.attention_layer.location_layer.location_conv.conv.weight', nv.LOC_FILTERS, 2, nv.LOC_KERNEL)
    dv = G('decoder.attention_layer.v.linear_layer.weight', 1, A)
    nv.unfold_location_grads(out['dU_acc'], out['dv_acc'], B, Wdense, Wconv, dWdense, dWconv, dv)
    g['decoder.attention_layer.location_layer.location_dense.linear_layer.weight'] = dWdense
    g['decoder.attention_layer.location_layer.location_conv.conv.weight'] = dWconv
    g['decoder.attention_layer.v.linear_layer.weight'] = dv
    # query layer: dWq = DQ^T . HA
    dWq = G('decoder.attention_layer.query_layer.linear_layer.weight', A, Ha)
    _rg(run, dWq, DQ.view(rowsD, A), S['HA'].view(rowsD, Ha), a_km=True, b_kn=True)
    g['decoder.attention_layer.query_layer.linear_layer.weight'] = dWq

    # attention LSTM weights: inputs [prenet_t | ctx_{t-1} | h_att_{t-1}]
    dWih_a = G('decoder.attention_rnn.weight_ih', 4 * Ha, Pd + E)
    dWhh_a = G('decoder.attention_rnn.weight_hh', 4 * Ha, Ha)
    use16 = run.bf16 and WGRAD16 and B % 8 == 0 and not nv.validate_only()
    Z = c.bf16 if use16 else None                           # bf16 images of the saved h_att / ctx / h_dec slabs
    if use16:
        _lstm_wgrad16(run, b16['DGA16'].view(rowsD, 4 * Ha), B, [(T['p2'], False), (Z['CTX16'].view(rowsD, E), True), (Z['HA16'].view(rowsD, Ha), True)],
                      [(dWih_a, 0, 2), (dWhh_a, 2, 3)])
    elif To > 1:
        tmp = run.empty(4 * Ha, Pd)
        _rg(run, tmp, DGA2, T['p2'], a_km=True, b_kn=True)
        nv.copy2d(dWih_a[:, :Pd], tmp)
        sh = (To - 1) * B
        tmp2 = run.empty(4 * Ha, E)
        _rg(run, tmp2, DGA2[B:], S['CTX'].view(rowsD, E)[:sh], a_km=True, b_kn=True)
        nv.copy2d(dWih_a[:, Pd:], tmp2)
        _rg(run, dWhh_a, DGA2[B:], S['HA'].view(rowsD, Ha)[:sh], a_km=True, b_kn=True)
    else:
        tmp = run.empty(4 * Ha, Pd)
        _rg(run, tmp, DGA2, T['p2'], a_km=True, b_kn=True)
        nv.copy2d(dWih_a[:, :Pd], tmp)
        nv.fill(dWhh_a, 0.0)
        z = run.zeros(4 * Ha, E)
        nv.copy2d(dWih_a[:, Pd:], z)
    db_a = G('decoder.attention_rnn.bias_ih', 4 * Ha)
    bias16 = use16 and BIAS_GRAD16                          # bf16 mode: the bias gradients from the bf16 slabs the weight gradients read
    if bias16:
        nv.colsum16(b16['DGA16'].view(rowsD, 4 * Ha), run.ws(4 * Ha), db_a)
    else:
        run.colsum(DGA2, db_a)
    db_a2 = G('decoder.attention_rnn.bias_hh', 4 * Ha)
    nv.copy2d(db_a2.view(1, -1), db_a.view(1, -1))
    g['decoder.attention_rnn.weight_ih'] = dWih_a
    g['decoder.attention_rnn.weight_hh'] = dWhh_a
    g['decoder.attention_rnn.bias_ih'] = db_a
    g['decoder.attention_rnn.bias_hh'] = db_a2

    # decoder LSTM weights: inputs [h_att_t | ctx_t | h_dec_{t-1}]
    dWih_d = G('decoder.decoder_rnn.weight_ih', 4 * Hd, Ha + E)
    dWhh_d = G('decoder.decoder_rnn.weight_hh', 4 * Hd, Hd)
    if use16:
        _lstm_wgrad16(run, b16['DGD16'].view(rowsD, 4 * Hd), B, [(Z['HA16'].view(rowsD, Ha), False), (Z['CTX16'].view(rowsD, E), False),
                                     (Z['HD16'].view(rowsD, Hd), True)], [(dWih_d, 0, 2), (dWhh_d, 2, 3)])
    else:
        tmp3 = run.empty(4 * Hd, Ha)
        _rg(run, tmp3, DGD2, S['HA'].view(rowsD, Ha), a_km=True, b_kn=True)
        nv.copy2d(dWih_d[:, :Ha], tmp3)
        tmp4 = run.empty(4 * Hd, E)
        _rg(run, tmp4, DGD2, S['CTX'].view(rowsD, E), a_km=True, b_kn=True)
        nv.copy2d(dWih_d[:, Ha:], tmp4)
        if To > 1:
            sh = (To - 1) * B
            _rg(run, dWhh_d, DGD2[B:], S['HD'].view(rowsD, Hd)[:sh], a_km=True, b_kn=True)
        else:
            nv.fill(dWhh_d, 0.0)
    db_d = G('decoder.decoder_rnn.bias_ih', 4 * Hd)
    if bias16:
        nv.colsum16(b16['DGD16'].view(rowsD, 4 * Hd), run.ws(4 * Hd), db_d)
    else:
        run.colsum(DGD2, db_d)
    db_d2 = G('decoder.decoder_rnn.bias_hh', 4 * Hd)
    nv.copy2d(db_d2.view(1, -1), db_d.view(1, -1))
    g['decoder.decoder_rnn.weight_ih'] = dWih_d
    g['decoder.decoder_rnn.weight_hh'] = dWhh_d
    g['decoder.decoder_rnn.bias_ih'] = db_d
    g['decoder.decoder_rnn.bias_hh'] = db_d2

    # prenet backward (model.py:99 under autograd)
    W2 = P['decoder.prenet.layers.1.linear_layer.weight']
    dp2 = run.empty(rowsD, Pd)
    if use16 and b16['DGA16'].shape[0] == To and (4 * Ha) % 64 == 0:
        # bf16 mode: dp2 = dG_a . W_ih_a[:, :P] on the bf16-resident product -- the gate gradients are already a bf16
        # K-contiguous slab ([To.B][4H]); the weight block is transposed once per weight version
        WpT16 = run.cached('Wih_a_pT16', [Wih_a], lambda: Wih_a[:, :Pd].t().contiguous().to(torch.bfloat16))
        nv.gemm16_tn(dp2, b16['DGA16'].view(rowsD, 4 * Ha), WpT16)
    else:
        _ng(run, dp2, DGA2, Wih_a[:, :Pd], b_kn=True)
    nv.relu_dropout_bwd(dp2, T['p2'], 2.0)
    dW2 = G('decoder.prenet.layers.1.linear_layer.weight', Pd, Pd)
    _rg(run, dW2, dp2, T['p1'], a_km=True, b_kn=True)
    dp1 = run.empty(rowsD, Pd)
    _ng(run, dp1, dp2, W2, b_kn=True)
    nv.relu_dropout_bwd(dp1, T['p1'], 2.0)
    dW1 = G('decoder.prenet.layers.0.linear_layer.weight', Pd, Cm)
    _rg(run, dW1, dp1, T['x0'].view(rowsD, Cm), a_km=True, b_kn=True)
    g['decoder.prenet.layers.0.linear_layer.weight'] = dW1
    g['decoder.prenet.layers.1.linear_layer.weight'] = dW2

    # memory gradient: d_mem[b] = ALIGN[b]^T . DCTX[:, b] + d_pm[b] . Wmem ; dWmem = d_pm^T . memory
    Wmem = P['decoder.attention_layer.memory_layer.linear_layer.weight']     # (A, E)
    dmem = run.empty(B, Ti, E)
    _ng(run, dmem[0], S['ALIGN'][0], DCTX[:, 0, :], a_km=True, b_kn=True, batch=B,
            strides=(To * Ti, E, Ti * E))
    _ng(run, dmem.view(rowsE, E), d_pm.view(rowsE, A), Wmem, b_kn=True, accumulate=True)
    dWmem = G('decoder.attention_layer.memory_layer.linear_layer.weight', A, E)
    _rg(run, dWmem, d_pm.view(rowsE, A), c.memory.view(rowsE, E), a_km=True, b_kn=True)
    g['decoder.attention_layer.memory_layer.linear_layer.weight'] = dWmem

    if embedded:
        for name, big in staged.items():
            g[name] = G_own(name, *own_shape[name])
            g[name].copy_(_crop_attention_grad(name, big, own_shape[name]))
    if sync is not None:
        sync.bucket_ready('decoder')             # travels while the encoder backward runs
    # ---- encoder backward -----------------------------------------------------------------
    dx3 = run.empty(rowsE, E)
    bdesc, DGs, keepalive = [], [], []
    for d, sfx in enumerate(('', '_reverse')):
        L = c.enc_lstm[d]
        WhhT = run.cached('enc_WhhT' + sfx, [L['Whh']], lambda L=L: transposed(L['Whh'], He, 4 * He))
        DG = run.empty(rowsE, 4 * He)
        desc = nv.LstmSeq()
        desc.B, desc.T, desc.H, desc.reverse = B, Ti, He, d
        desc.WhhT = nv.ptr(WhhT)
        desc.GX = nv.ptr(L['GX'])
        desc.C = nv.ptr(L['C'])
        desc.lens = nv.ptr(c.lens32, torch.int32)
        dout_view = dmem.view(rowsE, E)[:, d * He:(d + 1) * He]
        desc.dout, desc.ld_dout = nv.ptr(dout_view), E
        desc.DG = nv.ptr(DG)
        dX = run.empty(ENC_DGRAD_SPLIT, B, He)
        dc = run.empty(B, He)
        desc.dX, desc.dc, desc.dx_splits = nv.ptr(dX), nv.ptr(dc), ENC_DGRAD_SPLIT
        bdesc.append(desc)
        DGs.append(DG)
        keepalive.append((WhhT, dX, dc))
    model.last_encoder_bwd_path = _encoder_lstm_bwd(
        model, run, bdesc[0], bdesc[1], poison=DGs[0],                   # both directions (model.py:181-188 under autograd)
        reads=[k_[0] for k_ in keepalive] + [L_[k_] for L_ in c.enc_lstm for k_ in ('GX', 'C')] + [c.lens32, dmem],
        writes=DGs + [k_[i_] for k_ in keepalive for i_ in (1, 2)])
    for d, sfx in enumerate(('', '_reverse')):
        L, DG = c.enc_lstm[d], DGs[d]
        dWih = G('encoder.lstm.weight_ih_l0' + sfx, 4 * He, E)
        _rg(run, dWih, DG, c.x3, a_km=True, b_kn=True)
        # h_prev of row (b,t) is the output at (b, t-1) forward / (b, t+1) reverse, zero outside [0,T)
        dWhh = G('encoder.lstm.weight_hh_l0' + sfx, 4 * He, He)
        hview = c.memory.view(rowsE, E)[:, d * He:(d + 1) * He]
        _rg(run, dWhh, DG, hview, a_km=True, b_kn=True, convB=(Ti, He, 1 if d == 0 else -1))
        db = G('encoder.lstm.bias_ih_l0' + sfx, 4 * He)
        run.colsum(DG, db)
        db2 = G('encoder.lstm.bias_hh_l0' + sfx, 4 * He)
        nv.copy2d(db2.view(1, -1), db.view(1, -1))
        g['encoder.lstm.weight_ih_l0' + sfx] = dWih
        g['encoder.lstm.weight_hh_l0' + sfx] = dWhh
        g['encoder.lstm.bias_ih_l0' + sfx] = db
        g['encoder.lstm.bias_hh_l0' + sfx] = db2
        _ng(run, dx3, DG, L['Wih'], b_kn=True, accumulate=(d == 1))
    demb = run.empty(rowsE, E)
    _conv_stack_bwd(run, P, g, 'encoder.convolutions', c.enc_saved, dx3, Ti, first_dx=demb, G=G)
    dtable = G('embedding.weight', *P['embedding.weight'].shape)
    nv.embedding_bwd(c.text, demb, dtable)
    g['embedding.weight'] = dtable
    if sync is not None:
        sync.bucket_ready('encoder')
        sync.finish()
    return g


# The collector pause that hid in the headline (round 4; profiles/r04_b_diag_phases.json).  A training step makes a few
# thousand container objects (descriptors, dicts of slabs, closures); CPython counts them, and about once per dozen steps
# early in a run the count trips a FULL (generation-2) collection -- which walks every container object of the process,
# ~1.5 M of them once torch is imported: 60-130 ms on the host in the middle of `backward`, the GPU idle behind it.  In the
# driver's 20-step window exactly one lands: 65.3-67.2 ms per step where the same batches replayed read 61.5 (the whole gap
# VERDICT r03 "What's weak" #2 found, reproduced on step 9 of two differently ordered batch lists with gc.callbacks).  After
# the first complete training step everything that exists by then (modules, torch, the engine's caches) is collected once
# and moved to the permanent generation (`gc.freeze()`, what long-running servers do after start-up): later collections only
# walk what was made since -- a few thousand objects, well under a millisecond.  Reference counting is untouched, nothing
# is ever leaked but cyclic garbage that already existed at that moment.  T2AMD_GC_FREEZE=0 leaves the collector alone.
#
# Round 5: this is a policy of the whole interpreter, so a LIBRARY's forward pass no longer makes it by default.
# The loops that own their process do it explicitly -- `tacotron2_amd.train` and `bench.py` call engine.settle_gc() after their
# first complete step -- and T2AMD_GC_FREEZE=1 restores the implicit form (top of the second training forward) for a host
# loop that cannot be edited (the reference's own train.py); either way it is announced once on stderr.
GC_FREEZE = os.environ.get('T2AMD_GC_FREEZE', '0') == '1'
_gc_state = {'train_forwards': 0, 'frozen': False}


def settle_gc(force=False, quiet=False):
    """Collect once, then freeze the survivors out of later collections (`gc.freeze()`): later full collections walk only
    what was made since.  For training loops to call after their first complete step; see the note above."""
    if _gc_state['frozen'] and not force:
        return False
    import gc
    gc.collect()
    gc.freeze()
    _gc_state['frozen'] = True
    if not quiet:
        import sys
        print("tacotron2_amd: collected once and froze %d surviving objects out of later garbage collections (gc.freeze(); "
              "engine.settle_gc)" % gc.get_freeze_count(), file=sys.stderr, flush=True)
    return True


class Tacotron2TrainFunction(torch.autograd.Function):
    """forward(model, names, buffers, text, in_lens, mels, max_len, out_lens, *params)."""

    @staticmethod
    def forward(ctx, model, names, buffers, text, in_lens, mels, max_len, out_lens, *params):
        P = dict(zip(names, [p.detach() for p in params]))
        for n, p in P.items():
            if p.dtype != torch.float32:
                raise NativeError("parameter %s is %s: training keeps f32 master weights (select the bf16 compute mode "
                                  "with model.half() / hparams.fp16_run; reduced-precision parameter storage is accepted "
                                  "by inference only)" % (n, p.dtype))
        if model.training:
            _note_training_step()
        if GC_FREEZE and model.training and not _gc_state['frozen']:
            _gc_state['train_forwards'] += 1
            if _gc_state['train_forwards'] == 2:     # one whole step has run: its one-time objects exist by now
                settle_gc()
        outs, c = _forward(model, P, buffers, text, in_lens, mels, max_len, out_lens, model.training)
        # an eval-mode forward (validation, reference train.py:133) can never be differentiated: its activation
        # slabs are released right here instead of living until the outputs die
        ctx.model, ctx.names, ctx.P = model, names, P
        if model.training:
            ctx.c = c
        else:
            ctx.c = _EvalCtx()
            c.release()                  # nothing of an eval-mode forward is read again: the arena is free for the next call
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, d_mel, d_post, d_gate, d_align):
        if ctx.c is None:
            raise NativeError("the engine's saved activations were released by the first backward: "
                              "retain_graph / a second backward through the same forward is not supported")
        if not ctx.c.training:
            raise NativeError("backward through an eval-mode forward is not supported")
        c = ctx.c
        try:
            grads = _backward(ctx.model, ctx.P, c, d_mel, d_post, d_gate, d_align)
        finally:
            ctx.c = None
            c.release()                  # every kernel that reads the step's slabs is enqueued: the next step may reuse them
        out = []
        for n in ctx.names:
            if n not in grads:
                raise NativeError("internal error: no gradient produced for %s" % n)
            out.append(grads[n].view(ctx.P[n].shape))
        return (None,) * 8 + tuple(out)


# ----------------------------------------------------------------------------
# inference (reference model.py:517-529)
# ----------------------------------------------------------------------------
# T2AMD_COMPACT_BATCH=0 keeps finished utterances in the batch until the slowest one stops (A/B runs)
COMPACT_BATCH = os.environ.get('T2AMD_COMPACT_BATCH', '1') != '0'
# T2AMD_DECODE_PERSISTENT=0 keeps single-utterance bf16 decoding on the launch chain (A/B runs, shared GPUs)
PERSISTENT_DECODE = os.environ.get('T2AMD_DECODE_PERSISTENT', '1') != '0'
# T2AMD_ENCODER_PERSISTENT=0 keeps the single-utterance encoder bi-LSTM on the launch chain (A/B runs, shared GPUs)
PERSISTENT_ENCODER = os.environ.get('T2AMD_ENCODER_PERSISTENT', '1') != '0'


def _folded_projection(run, P, hp, Wpg, bpg):
    """[W1 . Wp ; Wp ; Wg] (P + C + 1 rows over Hd + E) and the matching bias: prenet layer 0 folded through the frame
    projection, p1 = relu(W1 (Wp hc + bp)) = relu((W1 Wp) hc + W1 bp) (reference model.py:99 after :373-375)."""
    E, H, Pd, Cm = hp.encoder_embedding_dim, hp.decoder_rnn_dim, hp.prenet_dim, hp.n_mel_channels
    W1 = P['decoder.prenet.layers.0.linear_layer.weight']
    Wp = P['decoder.linear_projection.linear_layer.weight']
    bp = P['decoder.linear_projection.linear_layer.bias']

    def fold():
        Wf = run.empty(Pd + Cm + 1, H + E)
        nv.gemm(Wf[:Pd], W1, Wp, b_kn=True)
        nv.copy2d(Wf[Pd:], Wpg)
        bf = run.empty(Pd + Cm + 1)
        nv.gemm(bf[:Pd].view(1, Pd), bp.view(1, Cm), W1)
        nv.copy2d(bf[Pd:].view(1, Cm + 1), bpg.view(1, Cm + 1))
        return Wf, bf
    return run.cached('Wf_fold', [W1, Wp, bp, P['decoder.gate_layer.linear_layer.weight'],
                                  P['decoder.gate_layer.linear_layer.bias']], fold)


# Batches of up to this many utterances are decoded one utterance after the other on the persistent single-utterance kernel
# (T2AMD_SMALL_BATCH_PERSISTENT=1 keeps them on the launch chain; measured: tools/bench_infer.py --small)
SMALL_BATCH_PERSISTENT = int(os.environ.get('T2AMD_SMALL_BATCH_PERSISTENT', '3'))


def _decode_persistent(model, run, P, hp, memory, pm, keep, st, out_lengths, Wa_cat, Wd_cat, bias_a, bias_d, Wq, U,
                       vvec, Wpg, bpg, i16, Ti):
    """reference model.py:435-449 (Decoder.inference loop) for B == 1 as one persistent launch.  Returns False when the
    kernel cannot run here (geometry, or a bounded-spin timeout): the caller then runs the launch chain from step 0."""
    import sys
    E, H, Pd, Cm = hp.encoder_embedding_dim, hp.attention_rnn_dim, hp.prenet_dim, hp.n_mel_channels
    dev = run.dev
    W2 = P['decoder.prenet.layers.1.linear_layer.weight']
    Wf, bf = _folded_projection(run, P, hp, Wpg, bpg)
    d = nv.DecPersist()
    d.Ti, d.E, d.H, d.P, d.C = Ti, E, H, Pd, Cm
    d.max_steps = hp.max_decoder_steps
    d.gate_threshold = float(hp.gate_threshold)
    d.weights_f32 = 0 if run.bf16 else 1            # fp32 parity mode: exact f32 rows (LDS + registers), same launch
    why = nv.decoder_persist_supported(d)
    if why is None and not nv.validate_only():
        # H/4 workgroups of ~137 KB LDS must all be co-resident: one per CU.  A partitioned / CU-masked device that shows
        # fewer CUs can never run it -- do not pay the 30 ms give-up for finding that out
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        if cus < H // 4:
            why = "the device shows %d CUs, the kernel needs %d co-resident workgroups" % (cus, H // 4)
    if why is None and getattr(model, '_persist_backoff', 0) > 0:
        # the kernel timed out recently on this model (shared GPU): straight to the launch chain for a while, then try again
        model._persist_backoff -= 1
        why = "persistent kernel timed out recently, %d more call(s) on the launch chain" % model._persist_backoff
    if why is not None:
        model.last_decode_path = 'launch chain (%s)' % why
        return False
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    steps_done = torch.zeros(1, dtype=torch.int32, device=dev)
    mailbox = torch.empty(nv.decoder_persist_mailbox_bytes(Ti, E, H, Pd) // 8, dtype=torch.int64, device=dev)
    trace = None
    if getattr(model, 'persist_trace', False):
        trace = run.zeros(hp.max_decoder_steps, 2 * H + E + 2 * Pd)
        model.last_persist_trace = trace
    if run.bf16:
        wa_, wd_ = i16['Wa_cat16'], i16['Wd_cat16']
        d.Wa16, d.Wd16 = nv.ptr(wa_, torch.bfloat16), nv.ptr(wd_, torch.bfloat16)
    else:
        wa_, wd_ = Wa_cat, Wd_cat
        d.Wa16, d.Wd16 = nv.ptr(wa_), nv.ptr(wd_)
    d.bias_a, d.bias_d = nv.ptr(bias_a), nv.ptr(bias_d)
    d.Wq, d.U, d.v, d.Wf, d.bias_f, d.W2 = nv.ptr(Wq), nv.ptr(U), nv.ptr(vvec), nv.ptr(Wf), nv.ptr(bf), nv.ptr(W2.contiguous())
    d.memory, d.pm = nv.ptr(memory), nv.ptr(pm)
    d.keep_prenet = nv.ptr(keep, torch.uint8)
    d.PG, d.ALIGN = nv.ptr(st['PG']), nv.ptr(st['ALIGN'])
    d.out_length = nv.ptr(out_lengths, torch.int32)
    d.status, d.steps_done = nv.ptr(status, torch.int32), nv.ptr(steps_done, torch.int32)
    d.mailbox = nv.ptr(mailbox, torch.int64)
    d.trace = nv.ptr(trace) if trace is not None else None
    timing = None
    if getattr(model, 'persist_timing', False):                 # tools/bench_decode_b1.py: per-phase clock
        timing = torch.zeros(32, dtype=torch.int64, device=dev)
        model.last_persist_timing = timing
        d.timing = nv.ptr(timing, torch.int64)
    nv.decoder_infer_persistent(d, reads=[wa_, wd_, bias_a, bias_d, Wq, U, vvec, Wf, bf, W2, memory, pm, keep],
                                writes=[st['PG'], st['ALIGN'], out_lengths, status, steps_done, mailbox]
                                + ([trace] if trace is not None else []) + ([timing] if timing is not None else []))
    code = int(status.item())
    if code != 0:
        print("tacotron2_amd: the persistent decode kernel gave up (status %d: %d workgroups were not co-resident "
              "within 30 ms -- is the GPU shared?); decoding again on the launch chain" % (code, H // 4),
              file=sys.stderr, flush=True)
        nv.fill(st['PG'], 0.0)
        nv.fill(st['ALIGN'], 0.0)
        out_lengths.zero_()
        model.last_decode_path = 'launch chain (persistent kernel timed out)'
        # exponential back-off: 4, 8, ... up to 256 calls on the launch chain before the next attempt
        model._persist_timeouts = getattr(model, '_persist_timeouts', 0) + 1
        model._persist_backoff = min(256, 2 << model._persist_timeouts)
        return False
    model.last_decode_path = 'persistent'
    model._persist_timeouts = 0
    return True



def infer(model, P, bufs, text, input_lengths=None, poll_steps=64):
    hp = model.hparams
    dev = text.device
    if not text.is_cuda and not nv.validate_only():
        raise NativeError("tacotron2_amd: the engine runs on the MI355X only (got %s tensors)" % dev)
    nv.load()
    P = {k: v.detach() for k, v in P.items()}
    precision = getattr(model, 'precision', 'fp32')
    run = _Run(dev, precision, _weight_cache(model))
    low = [n for n, p in P.items() if p.dtype in (torch.float16, torch.bfloat16)]
    if low:
        # Reduced-precision parameter STORAGE (the reference notebook's ``model.half()`` on a stock nn.Module, or
        # ``model.to(torch.bfloat16)``, inference.ipynb cell 7): accepted for inference.  The stored values are widened
        # once per weight version to the f32 images the kernels read (exact: every fp16 / bf16 value is an f32 value)
        # and the matrix products run in the bf16 compute mode -- there is no more precision in such weights to keep.
        if len(low) != len(P):
            raise NativeError("parameters mix reduced-precision and f32 storage (%s is %s): cast the whole model"
                              % (low[0], P[low[0]].dtype))
        P = {n: run.cached('widen.' + n, [p], lambda p=p: p.float()) for n, p in P.items()}
        bufs = {k: (v.float() if v.is_floating_point() and v.dtype != torch.float32 else v) for k, v in bufs.items()}
        if precision != 'bf16':
            run = _Run(dev, 'bf16', _weight_cache(model))
    for n, p in P.items():
        if p.dtype != torch.float32:
            raise NativeError("parameter %s is %s: the engine reads f32, fp16 or bf16 parameters" % (n, p.dtype))
    if not low:
        _guard_weights(model, run, P)
    P, _ = _embed_attention(run, P)
    ms = MaskSource(model.dropout_masks, dev)
    B, Ti = text.shape
    E = hp.encoder_embedding_dim
    Ha, Hd, Pd, Cm = hp.attention_rnn_dim, hp.decoder_rnn_dim, hp.prenet_dim, hp.n_mel_channels
    He, A = E // 2, nv.ATT_DIM
    max_steps = hp.max_decoder_steps
    if input_lengths is not None:
        lens32 = input_lengths.to(torch.int32).contiguous()
        Ti = int(input_lengths.max().item())
        text = text[:, :Ti]
    else:
        lens32 = torch.full((B,), Ti, dtype=torch.int32, device=dev)
    text = text.contiguous()
    rowsE = B * Ti
    ragged = input_lengths is not None

    # encoder, eval mode (model.py:192-201); ragged batches zero the activations beyond each length
    emb = run.empty(rowsE, E)
    nv.embedding_fwd(text, P['embedding.weight'], emb)
    x = emb
    if ragged:
        # zero padded positions of the embedding so the first conv sees exact zero padding
        ones = run.empty(E); nv.fill(ones, 1.0)
        zeros = run.zeros(E)
        xm = run.empty(rowsE, E)
        nv.bn_act_fwd(x, xm, zeros, ones, ones, zeros, 0, None, 1.0, lens32, Ti)
        x = xm
    x3, _ = _conv_stack_fwd(run, P, bufs, 'encoder.convolutions', hp.encoder_n_convolutions, x, Ti,
                            [1] * hp.encoder_n_convolutions, None, False, lens=lens32 if ragged else None, exact=True)
    memory = run.empty(B, Ti, E)
    idesc = []
    for d, sfx in enumerate(('', '_reverse')):
        Wih = P['encoder.lstm.weight_ih_l0' + sfx]
        Whh = P['encoder.lstm.weight_hh_l0' + sfx]
        bsum = _cached_bias_sum(run, 'enc_bias' + sfx, P['encoder.lstm.bias_ih_l0' + sfx], P['encoder.lstm.bias_hh_l0' + sfx])
        GX = run.empty(rowsE, 4 * He)
        nv.gemm(GX, x3, Wih, bias=bsum)
        Cst = run.empty(Ti, B, He)
        desc = nv.LstmSeq()
        desc.B, desc.T, desc.H, desc.reverse = B, Ti, He, d
        desc.Whh, desc.GX = nv.ptr(Whh), nv.ptr(GX)
        out_view = memory.view(rowsE, E)[:, d * He:(d + 1) * He]
        desc.out, desc.ld_out = nv.ptr(out_view), E
        desc.C = nv.ptr(Cst)
        desc.lens = nv.ptr(lens32, torch.int32)
        idesc.append((desc, GX, Cst))
    enc_persistent = False
    if (B == 1 and not ragged and PERSISTENT_ENCODER and not nv.validate_only()
            and getattr(model, '_enc_persist_backoff', 0) <= 0 and nv.lstm_seq_persistent_supported(idesc[0][0]) is None
            and torch.cuda.get_device_properties(dev).multi_processor_count >= He // 2):
        # one utterance: the whole bi-LSTM as ONE persistent launch (W_hh rows in registers, h as granules) instead of Ti
        # dependent launches; a timeout (shared GPU) falls back to the launch chain, with a back-off like the decoder's
        ebox = torch.empty(nv.load().t2amd_lstm_seq_persistent_mailbox_bytes(He, 2) // 8, dtype=torch.int64, device=dev)
        estat = torch.zeros(1, dtype=torch.int32, device=dev)
        nv.lstm_seq_fwd2_persistent(idesc[0][0], idesc[1][0], ebox, estat)
        enc_persistent = int(estat.item()) == 0
        if not enc_persistent:
            import sys
            print("tacotron2_amd: the persistent encoder kernel gave up (its %d workgroups were not co-resident within 30 ms); "
                  "running the launch chain" % (He // 2), file=sys.stderr, flush=True)
            model._enc_persist_backoff = 16
    elif getattr(model, '_enc_persist_backoff', 0) > 0:
        model._enc_persist_backoff -= 1
    model.last_encoder_path = 'persistent' if enc_persistent else 'launch chain'
    if not enc_persistent:
        def regen_gx():
            for d_, sfx_ in enumerate(('', '_reverse')):
                nv.gemm(idesc[d_][1], x3, P['encoder.lstm.weight_ih_l0' + sfx_],
                        bias=_cached_bias_sum(run, 'enc_bias' + sfx_, P['encoder.lstm.bias_ih_l0' + sfx_], P['encoder.lstm.bias_hh_l0' + sfx_]))
        model.last_encoder_path = _encoder_lstm_fwd(
            model, dev, idesc[0][0], idesc[1][0], regen_gx,
            reads=[P['encoder.lstm.weight_hh_l0'], P['encoder.lstm.weight_hh_l0_reverse'], idesc[0][1], idesc[1][1], lens32],
            writes=[memory, idesc[0][2], idesc[1][2]])

    Wmem = P['decoder.attention_layer.memory_layer.linear_layer.weight']
    pm = run.empty(B, Ti, A)
    nv.gemm(pm.view(rowsE, A), memory.view(rowsE, E), Wmem)

    Wih_a, Whh_a = P['decoder.attention_rnn.weight_ih'], P['decoder.attention_rnn.weight_hh']
    Wih_d, Whh_d = P['decoder.decoder_rnn.weight_ih'], P['decoder.decoder_rnn.weight_hh']
    bias_a = _cached_bias_sum(run, 'bias_a', P['decoder.attention_rnn.bias_ih'], P['decoder.attention_rnn.bias_hh'])
    bias_d = _cached_bias_sum(run, 'bias_d', P['decoder.decoder_rnn.bias_ih'], P['decoder.decoder_rnn.bias_hh'])

    def pack_a_cat():
        W = run.empty(4 * Ha, Pd + E + Ha)
        nv.copy2d(W[:, :Pd + E], Wih_a)
        nv.copy2d(W[:, Pd + E:], Whh_a)
        return W

    def pack_d_cat():
        W = run.empty(4 * Hd, Ha + E + Hd)
        nv.copy2d(W[:, :Ha + E], Wih_d)
        nv.copy2d(W[:, Ha + E:], Whh_d)
        return W

    Wa_cat = run.cached('Wa_cat', [Wih_a, Whh_a], pack_a_cat)
    Wd_cat = run.cached('Wd_cat', [Wih_d, Whh_d], pack_d_cat)
    Wq = P['decoder.attention_layer.query_layer.linear_layer.weight'].contiguous()
    Wdense = P['decoder.attention_layer.location_layer.location_dense.linear_layer.weight']
    Wconv = P['decoder.attention_layer.location_layer.location_conv.conv.weight']
    U = run.cached('U', [Wdense, Wconv], lambda: _fold_U(run, Wdense, Wconv))
    vvec = P['decoder.attention_layer.v.linear_layer.weight'].view(-1)
    Wpg, bpg = _packed_projection(run, P, Cm, Hd, E)

    keep = ms.get('prenet_infer', None, (max_steps, 2, B, Pd), 0.5)

    d = nv.DecInfer()
    d.B, d.Ti, d.E, d.Ha, d.Hd, d.P, d.C = B, Ti, E, Ha, Hd, Pd, Cm
    d.max_steps = max_steps
    d.gate_threshold = float(hp.gate_threshold)
    st = dict(h_a=run.zeros(2, B, Ha), c_a=run.zeros(2, B, Ha), c_d=run.zeros(2, B, Hd),
              hc=run.zeros(2, B, Hd + E), cum=run.zeros(B, Ti), x_prenet=run.empty(2, B, Pd),
              gates=run.empty(B, 4 * max(Ha, Hd)), zero_frame=run.zeros(B, Cm),
              attn_ws=run.zeros(nv.attn_fwd_ws_floats(B, Ti)),      # zeroed: the granule block of the one-launch step
              PG=run.zeros(max_steps, B, Cm + 1), ALIGN=run.zeros(B, max_steps, Ti))
    out_lengths = torch.zeros(B, dtype=torch.int32, device=dev)
    active = torch.ones(B, dtype=torch.uint8, device=dev)
    done = torch.zeros(1, dtype=torch.int32, device=dev)
    d.W1 = nv.ptr(P['decoder.prenet.layers.0.linear_layer.weight'])
    d.W2 = nv.ptr(P['decoder.prenet.layers.1.linear_layer.weight'])
    d.Wa_cat, d.bias_a, d.Wd_cat, d.bias_d = nv.ptr(Wa_cat), nv.ptr(bias_a), nv.ptr(Wd_cat), nv.ptr(bias_d)
    d.Wq, d.U, d.v, d.Wpg, d.bias_pg = nv.ptr(Wq), nv.ptr(U), nv.ptr(vvec), nv.ptr(Wpg), nv.ptr(bpg)
    d.memory, d.pm = nv.ptr(memory), nv.ptr(pm)
    d.lens = nv.ptr(lens32, torch.int32) if ragged else None
    d.keep_prenet = nv.ptr(keep, torch.uint8)
    for k_, v_ in st.items():
        setattr(d, k_, nv.ptr(v_))
    d.attn_ws_floats = st['attn_ws'].numel()
    d.out_lengths = nv.ptr(out_lengths, torch.int32)
    d.active = nv.ptr(active, torch.uint8)
    d.done_count = nv.ptr(done, torch.int32)
    # the launch chain's step: matrix-vector kernels up to nv.small_batch_max() rows, the 64-row MFMA tiles above (csrc/loops.hip)
    mode16 = 1 if run.bf16 else (3 if run.x3 else 0)
    tiles = nv.dec_infer_uses_tiles(B, mode16, E, Ha, Hd, Pd)
    if tiles:
        Wf_, bf_ = _folded_projection(run, P, hp, Wpg, bpg)     # prenet layer 0 rides in the projection launch
        d.Wf, d.bias_f = nv.ptr(Wf_), nv.ptr(bf_)

    if run.bf16:
        # bf16 operand mode of the two LSTM products: the tile path reads bf16 weights and bf16 copies of the recurrent
        # operands (wide MFMA kernel); the matrix-vector path reads bf16 weight rows against f32 inputs
        i16 = dict(Wa_cat16=run.cached('Wa_cat16', [Wih_a, Whh_a], lambda: run.cast16(Wa_cat)),
                   Wd_cat16=run.cached('Wd_cat16', [Wih_d, Whh_d], lambda: run.cast16(Wd_cat)),
                   x_prenet16=run.empty16(B, Pd),
                   h_a16=torch.zeros(2, B, Ha, dtype=torch.bfloat16, device=dev),
                   hc16=torch.zeros(2, B, Hd + E, dtype=torch.bfloat16, device=dev))
        if tiles:
            i16['memory16'] = run.cast16(memory)
            i16['Wq16'] = run.cached('Wq16', [Wq], lambda: run.cast16(Wq))
            # prenet + frame/gate projection on the bf16 MFMA path as well: bf16 images of [W1.Wp ; Wp ; Wg] and of W2
            W2_ = P['decoder.prenet.layers.1.linear_layer.weight']
            fold_deps = [P[n_] for n_ in ('decoder.prenet.layers.0.linear_layer.weight',
                                          'decoder.linear_projection.linear_layer.weight',
                                          'decoder.linear_projection.linear_layer.bias',
                                          'decoder.gate_layer.linear_layer.weight',
                                          'decoder.gate_layer.linear_layer.bias')]
            i16['Wf16'] = run.cached('Wf16', fold_deps, lambda: run.cast16(Wf_))
            i16['Wpg16'] = i16['Wf16'][Pd:]
            i16['W2_16'] = run.cached('W2_16', [W2_], lambda: run.cast16(W2_))
            i16['x_prenet1_16'] = torch.zeros(B, Pd, dtype=torch.bfloat16, device=dev)
        d.bf16 = 1
        for k_, v_ in i16.items():
            setattr(d, k_, nv.ptr(v_, torch.bfloat16))
    op16 = run.bf16
    if run.x3 and tiles:
        # 'bf16x3' mode, batched decode (BASELINE configs[4] in the accurate-fast mode): the two LSTM steps multiply split-bf16
        # operand images on the wide tile (csrc/skinny_wide.h SW_X3) -- weights split once per weight version, h_att / h_dec / ctx
        # written as images by the tile and K_c epilogues, the prenet output split by a small launch per step; everything else of
        # the step is the fp32 mode's (prenet, projection and the stop test on f32 operands; attention exact f32 but for its
        # split-form location conv).  Below the tile boundary the fp32 mode's kernels run (matrix-vector path / persistent single-utterance kernel).
        i16 = dict(Wa_cat16=run.cached('Wa_cat16x3', [Wih_a, Whh_a], lambda: run.split16(Wa_cat)),
                   Wd_cat16=run.cached('Wd_cat16x3', [Wih_d, Whh_d], lambda: run.split16(Wd_cat)),
                   x_prenet16=run.empty16(B, 2 * Pd),
                   h_a16=torch.zeros(2, B, 2 * Ha, dtype=torch.bfloat16, device=dev),
                   hc16=torch.zeros(2, B, 2 * (Hd + E), dtype=torch.bfloat16, device=dev))
        d.bf16 = 3
        for k_, v_ in i16.items():
            setattr(d, k_, nv.ptr(v_, torch.bfloat16))
        op16 = True
    infer_reads = [P['decoder.prenet.layers.0.linear_layer.weight'], P['decoder.prenet.layers.1.linear_layer.weight'],
                   Wa_cat, bias_a, Wd_cat, bias_d, Wq, U, vvec, Wpg, bpg] + ([Wf_, bf_] if tiles else [])
    # One utterance: the whole loop as ONE persistent launch, LSTM weights resident on the CUs -- bf16 rows in LDS in the
    # bf16 mode, exact f32 rows split between LDS and registers in the fp32 parity mode (csrc/decode_persist.hip).  Anything it cannot take -- or a timeout because the GPU is shared and H/4 workgroups
    # are not co-resident -- goes through the launch chain below.
    ran_persistent = False
    model.last_decode_path = 'launch chain'
    if B == 1 and not ragged and Ha == Hd and PERSISTENT_DECODE and not nv.validate_only():
        ran_persistent = _decode_persistent(model, run, P, hp, memory, pm, keep, st, out_lengths, Wa_cat, Wd_cat,
                                            bias_a, bias_d, Wq, U, vvec, Wpg, bpg, i16 if run.bf16 else None, Ti)
    elif 2 <= B <= SMALL_BATCH_PERSISTENT and Ha == Hd and PERSISTENT_DECODE and not nv.validate_only():
        # Two (or three) utterances: the launch chain's step costs ~38-47 us on either kind of kernel, the persistent
        # kernel 12 us per utterance and step -- so the utterances are decoded ONE AFTER THE OTHER on the persistent kernel,
        # each against its own rows of the encoder memory and of the dropout stream (rows of a batch never interact in
        # Decoder.inference, reference model.py:418-454), and their outputs land in the batch's arrays.
        lens_host = lens32.tolist()
        ran_persistent = True
        for b in range(B):
            Tb = int(lens_host[b])
            st_b = dict(PG=run.zeros(max_steps, 1, Cm + 1), ALIGN=run.zeros(1, max_steps, Tb))
            ol_b = torch.zeros(1, dtype=torch.int32, device=dev)
            if not _decode_persistent(model, run, P, hp, memory[b, :Tb], pm[b, :Tb], keep[:, :, b:b + 1].contiguous(), st_b, ol_b,
                                      Wa_cat, Wd_cat, bias_a, bias_d, Wq, U, vvec, Wpg, bpg, i16 if run.bf16 else None, Tb):
                ran_persistent = False
                break
            st['PG'][:, b] = st_b['PG'][:, 0]
            st['ALIGN'][b, :, :Tb] = st_b['ALIGN'][0]
            out_lengths[b:b + 1] = ol_b
        if ran_persistent:
            model.last_decode_path = 'persistent (%d utterances, one after the other)' % B
        else:
            nv.fill(st['PG'], 0.0)
            nv.fill(st['ALIGN'], 0.0)
            out_lengths.zero_()
    # ---- the launch chain, with early-exit compaction of the batch -----------------------------------
    # Finished utterances keep occupying every launch until the slowest one stops.  At a poll, once enough of them
    # have finished to free a 64-row tile of the LSTM kernels (or to fall below the tile boundary, nv.dec_infer_uses_tiles), the
    # rows still decoding are gathered into a smaller batch -- state, encoder memory, masks -- and the loop goes on with
    # B' rows; what a segment produced is scattered back to the utterances' own rows of the output arrays.
    t = 0
    on_tiles = tiles
    Bc = B                                      # rows of the current segment
    cur = None                                  # original utterance of each row (None: identity)
    seg_t0 = 0
    full_PG, full_ALIGN = st['PG'], st['ALIGN']
    final_len = torch.zeros(B, dtype=torch.int32, device=dev)
    min_rows = getattr(model, 'compact_min_rows', None)

    def close_segment(t_end):
        """Lengths and outputs of the current segment -> the utterances' own rows."""
        if cur is None:
            final_len.copy_(out_lengths)
            return
        fin = out_lengths > 0
        final_len[cur[fin]] = out_lengths[fin]
        full_PG[seg_t0:t_end, cur] = st['PG'][seg_t0:t_end]
        full_ALIGN[cur, seg_t0:t_end] = st['ALIGN'][:, seg_t0:t_end]

    while t < max_steps and not ran_persistent:
        n = min(poll_steps, max_steps - t)
        d.t0, d.n_steps = t, n
        nv.decoder_infer_steps(d, reads=infer_reads + [memory, pm, lens32, keep] + (list(i16.values()) if op16 else []),
                               writes=list(st.values()) + [out_lengths, active, done])
        t += n
        ndone = int(done.item())                # one device->host sync per poll_steps steps
        if ndone >= Bc:
            break
        left = Bc - ndone
        if min_rows is not None:
            shrink = ndone >= min_rows
        else:
            shrink = (left + 63) // 64 < (Bc + 63) // 64 or (on_tiles and not nv.dec_infer_uses_tiles(left, mode16, E, Ha, Hd, Pd))
        if not (COMPACT_BATCH and ragged and t < max_steps and ndone > 0 and shrink) or nv.validate_only():
            continue
        close_segment(t)
        rows = torch.nonzero(active, as_tuple=False).view(-1)              # rows still decoding
        cur = rows if cur is None else cur[rows]
        oldPG, oldALIGN = st['PG'], st['ALIGN']
        for k_ in ('h_a', 'c_a', 'c_d', 'hc', 'x_prenet'):
            st[k_] = st[k_][:, rows].contiguous()
        st['cum'] = st['cum'][rows].contiguous()
        st['zero_frame'] = st['zero_frame'][:left].contiguous()
        st['gates'] = run.empty(left, 4 * max(Ha, Hd))
        st['attn_ws'] = run.zeros(nv.attn_fwd_ws_floats(left, Ti))
        st['PG'] = run.zeros(max_steps, left, Cm + 1)
        st['PG'][t - 1] = oldPG[t - 1][rows]                               # the frame the next prenet input comes from
        st['ALIGN'] = run.zeros(left, max_steps, Ti)
        st['ALIGN'][:, t - 1] = oldALIGN[rows, t - 1]                      # the previous attention weights
        memory, pm, lens32 = memory[rows].contiguous(), pm[rows].contiguous(), lens32[rows].contiguous()
        keep = keep[:, :, rows].contiguous()
        out_lengths = torch.zeros(left, dtype=torch.int32, device=dev)
        active = torch.ones(left, dtype=torch.uint8, device=dev)
        done.zero_()
        d.B = left
        d.memory, d.pm, d.lens = nv.ptr(memory), nv.ptr(pm), nv.ptr(lens32, torch.int32)
        d.keep_prenet = nv.ptr(keep, torch.uint8)
        for k_, v_ in st.items():
            setattr(d, k_, nv.ptr(v_))
        d.attn_ws_floats = st['attn_ws'].numel()
        d.out_lengths, d.active = nv.ptr(out_lengths, torch.int32), nv.ptr(active, torch.uint8)
        on_tiles = nv.dec_infer_uses_tiles(left, mode16, E, Ha, Hd, Pd)
        if run.x3 and op16 and not on_tiles:
            # the split operand images belong to the wide tile; the matrix-vector kernels of the rows that are left run the fp32
            # mode's arithmetic on the f32 state the tiles kept beside the images
            d.bf16, op16 = 0, False
        if op16:
            i16['x_prenet16'] = i16['x_prenet16'][rows].contiguous()
            if 'x_prenet1_16' in i16:
                i16['x_prenet1_16'] = i16['x_prenet1_16'][rows].contiguous()
            i16['h_a16'] = i16['h_a16'][:, rows].contiguous()
            i16['hc16'] = i16['hc16'][:, rows].contiguous()
            if 'memory16' in i16:
                i16['memory16'] = i16['memory16'][rows].contiguous()
            for k_ in ('x_prenet16', 'x_prenet1_16', 'h_a16', 'hc16', 'memory16'):
                if k_ in i16:
                    setattr(d, k_, nv.ptr(i16[k_], torch.bfloat16))
        Bc, seg_t0 = left, t
        model.last_decode_path = 'launch chain (batch compacted to %d rows at step %d)' % (left, t)
    if not ran_persistent:
        close_segment(min(t, max_steps))
        out_lengths = final_len
    st['PG'], st['ALIGN'] = full_PG, full_ALIGN
    lengths = out_lengths.to(torch.long)
    Tout = int(lengths.max().item())
    hit_max = bool((lengths >= max_steps).any().item()) and Tout >= max_steps
    # if the stop fired exactly on the last allowed step the reference does not warn; it warns
    # only when the loop ends through the max_decoder_steps branch.
    if hit_max:
        PGl = st['PG'][max_steps - 1]
        sg = torch.sigmoid(PGl[:, Cm])
        hit_max = bool(((lengths >= max_steps) & ~(sg > hp.gate_threshold)).any().item())

    # outputs, padded to the longest utterance; frames past an utterance's own length are zeroed
    Tout = max(Tout, 1)
    PG = st['PG'][:Tout].contiguous()
    mel_cl = run.empty(B, Tout, Cm)
    gate_bt = run.empty(B, Tout)
    nv.split_projection(PG, mel_cl, gate_bt, None)
    olens32 = out_lengths if ragged else None
    xin = mel_cl.view(B * Tout, Cm)
    if ragged:
        # every utterance must see zero padding beyond its own last frame, as a B == 1 run would
        ones = run.empty(Cm); nv.fill(ones, 1.0)
        zeros = run.zeros(Cm)
        xm = run.empty(B * Tout, Cm)
        nv.bn_act_fwd(xin, xm, zeros, ones, ones, zeros, 0, None, 1.0, olens32, Tout)
        xin = xm
    post_cl, _ = _conv_stack_fwd(run, P, bufs, 'postnet.convolutions', hp.postnet_n_convolutions,
                                 xin, Tout, [2] * (hp.postnet_n_convolutions - 1) + [0], None, False,
                                 lens=olens32)
    mel = run.empty(B, Cm, Tout)
    mel_post = run.empty(B, Cm, Tout)
    nv.finalize_outputs(mel_cl, post_cl.view(B, Tout, Cm), mel, mel_post, olens32)
    align = st['ALIGN'][:, :Tout].contiguous()
    gate_out = gate_bt.unsqueeze(-1)                      # (B, T, 1) like model.py:440
    return [mel, mel_post, gate_out, align], lengths, hit_max
