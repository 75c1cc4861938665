"""Can the encoder bi-LSTM's BPTT chain (347 dependent small launches, ~2 ms) hide under the decoder's weight-gradient
products (~2 ms of 256-workgroup GEMMs)?  Both alone, back to back on one stream, side by side on two plain streams, and
side by side on two CU-MASKED streams (hipExtStreamCreateWithCUMask: chain on `--chain-cus` CUs, products on the rest).
    python tools/microbench_overlap.py [--chain-cus 64]"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd import native as nv      # noqa: E402

CH = int(sys.argv[sys.argv.index("--chain-cus") + 1]) if "--chain-cus" in sys.argv else 64
dev = torch.device("cuda")
lib = nv.load()
lib.t2amd_debug_stream_cu_range_.restype = C.c_void_p
lib.t2amd_debug_stream_cu_range_.argtypes = [C.c_int, C.c_int]
B, T, H, E = 64, 170, 256, 512
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.1).to(dev)        # noqa: E731
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
keep = []


def chain_descs():
    ds = []
    dmem = rnd(B * T, E)
    for d in range(2):
        desc = nv.LstmSeq()
        desc.B, desc.T, desc.H, desc.reverse = B, T, H, d
        WhhT, GX, Cst, DG = rnd(H, 4 * H), torch.sigmoid(rnd(B * T, 4 * H)), rnd(T, B, H), torch.empty(B * T, 4 * H, device=dev)
        dX, dc = torch.empty(4, B, H, device=dev), torch.empty(B, H, device=dev)
        desc.WhhT, desc.GX, desc.C, desc.lens = nv.ptr(WhhT), nv.ptr(GX), nv.ptr(Cst), nv.ptr(lens, torch.int32)
        dv = dmem[:, d * H:(d + 1) * H]
        desc.dout, desc.ld_dout, desc.DG = nv.ptr(dv), E, nv.ptr(DG)
        desc.dX, desc.dc, desc.dx_splits = nv.ptr(dX), nv.ptr(dc), 4
        keep.append((WhhT, GX, Cst, DG, dX, dc, dmem))
        ds.append(desc)
    return ds


cd = chain_descs()
K = 55680
dG = rnd(K, 4096).bfloat16()
X1, X2 = rnd(K, 2560).bfloat16(), rnd(K, 1792).bfloat16()
O1, O2 = torch.empty(4096, 2560, device=dev), torch.empty(4096, 1792, device=dev)
P1, P2 = torch.empty(8, 4096 * 2560, device=dev), torch.empty(9, 4096 * 1792, device=dev)


def chain():
    nv.lstm_seq_bwd2(cd[0], cd[1])


def products():
    nv.gemm16_kk(O1, dG, X1, K, splitk=8, partials=P1)
    nv.splitk_reduce(P1, 8, O1)
    nv.gemm16_kk(O2, dG, X2, K, splitk=9, partials=P2)
    nv.splitk_reduce(P2, 9, O2)


def timed(work, n=5):
    """work: list of (fn, stream); forked from and joined to the current stream"""
    cur = torch.cuda.current_stream()
    def once():
        for fn, s in work:
            if s is not None:
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    fn()
            else:
                fn()
        for _, s in work:
            if s is not None:
                cur.wait_stream(s)
    for _ in range(2):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        once()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


cus = torch.cuda.get_device_properties(dev).multi_processor_count
out = {"cus": cus, "chain_cus": CH}
out["chain_alone_ms"] = timed([(chain, None)])
out["products_alone_ms"] = timed([(products, None)])
out["back_to_back_ms"] = timed([(chain, None), (products, None)])
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
out["two_plain_streams_ms"] = timed([(chain, s1), (products, s2)])
pa, pb = lib.t2amd_debug_stream_cu_range_(0, CH), lib.t2amd_debug_stream_cu_range_(CH, cus - CH)
if pa and pb:
    ma, mb = torch.cuda.ExternalStream(pa), torch.cuda.ExternalStream(pb)
    out["chain_on_masked_stream_alone_ms"] = timed([(chain, ma)])
    out["products_on_masked_stream_alone_ms"] = timed([(products, mb)])
    out["two_masked_streams_ms"] = timed([(chain, ma), (products, mb)])
    out["chain_masked_products_plain_ms"] = timed([(chain, ma), (products, s2)])
else:
    out["masked"] = "hipExtStreamCreateWithCUMask failed"
print(json.dumps(out))
