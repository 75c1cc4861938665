"""Micro-benchmark of the LSTM step launches (tools only; not part of the product path).
    python tools/microbench_lstm.py
"""
import ctypes as C
import os
os.environ.setdefault('T2AMD_ATTN_TS_PICK', '100')
if os.environ.get('T2AMD_ATTN_TS') == '1':      # stamps exist only in the instrumented build (python -m tacotron2_amd.build --stamps)
    os.environ.setdefault('T2AMD_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tacotron2_amd', 'lib', 'libtacotron2_amd_stamps.so'))
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd import native as nv

dev = torch.device('cuda')
B, Ha, Hd, E = 64, 1024, 1024, 512
lib = nv.load()


def mk(K, H, nseg_w):
    st = nv.LstmStep()
    xs = [torch.randn(B, w, device=dev) for w in nseg_w]
    W = torch.randn(4 * H, K, device=dev) * 0.02
    gates = torch.empty(B, 4 * H, device=dev)
    c_prev = torch.randn(B, H, device=dev)
    c = torch.empty(B, H, device=dev)
    h = torch.empty(B, H, device=dev)
    bias = torch.randn(4 * H, device=dev)
    st.nseg = len(xs)
    for i, x in enumerate(xs):
        st.x[i] = nv._seg(x, x.shape[1])
    st.W, st.Ktot, st.H, st.B = nv.ptr(W), K, H, B
    st.bias = nv.ptr(bias)
    st.c_prev, st.ld_cprev = nv.ptr(c_prev), H
    st.gates_out, st.ld_gates = nv.ptr(gates), 4 * H
    st.c_out, st.ld_c = nv.ptr(c), H
    st.h_out, st.ld_h = nv.ptr(h), H
    st.keep_scale = 1.0
    keep = (xs, W, gates, c_prev, c, h, bias)
    return st, keep


def timeit(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


d, kd = mk(Ha + E + Hd, Hd, [Ha, E, Hd])
a, ka = mk(E + Ha, Ha, [E, Ha])
s = nv._stream()
print("LSTM_d alone  %.2f us" % timeit(lambda: lib.t2amd_lstm_step_fwd_f32(C.byref(d), s)))
print("LSTM_a alone  %.2f us" % timeit(lambda: lib.t2amd_lstm_step_fwd_f32(C.byref(a), s)))
print("fused d||a    %.2f us" % timeit(lambda: lib.t2amd_lstm_step_fwd2_f32(C.byref(d), C.byref(a), s)))
print("fused a||d    %.2f us" % timeit(lambda: lib.t2amd_lstm_step_fwd2_f32(C.byref(a), C.byref(d), s)))
print("fused d||d    %.2f us" % timeit(lambda: lib.t2amd_lstm_step_fwd2_f32(C.byref(d), C.byref(d), s)))
flops = 2.0 * B * 4 * Hd * (Ha + E + Hd)
t = timeit(lambda: lib.t2amd_lstm_step_fwd_f32(C.byref(d), s))
print("LSTM_d alone: %.1f TFLOP/s f32 MFMA, %.2f TB/s weights" % (flops / t / 1e6, 4 * 4 * Hd * (Ha + E + Hd) / t / 1e6))


# ---- bf16 operands (model.precision='bf16'): LSTM pair and the BPTT dgrad pair --------------------------------
def mk16(K, H, nseg_w):
    st, keep = mk(K, H, nseg_w)
    xs16 = [torch.randn(B, w, device=dev).to(torch.bfloat16) for w in nseg_w]
    W16 = (torch.randn(4 * H, K, device=dev) * 0.02).to(torch.bfloat16)
    for i, x in enumerate(xs16):
        st.x[i] = nv._seg(x, x.shape[1], torch.bfloat16)
    st.W = nv.ptr(W16, torch.bfloat16)
    st.bf16 = 1
    return st, (keep, xs16, W16)


def mkplain16(K, N, nsplit):
    a = nv.SkinnyGemm()
    x = torch.randn(B, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    Y = torch.empty(nsplit, B, N, device=dev)
    a.nseg = 1
    a.x[0] = nv._seg(x, K, torch.bfloat16)
    a.W = nv.ptr(W, torch.bfloat16)
    a.bf16 = 1
    a.Ktot, a.N, a.B = K, N, B
    a.Y, a.ldy, a.nsplit, a.split_stride = nv.ptr(Y), N, nsplit, B * N
    return a, (x, W, Y)


d16, kd16 = mk16(Ha + E + Hd, Hd, [Ha, E, Hd])
a16, ka16 = mk16(E + Ha, Ha, [E, Ha])
print("bf16 LSTM_d alone  %.2f us" % timeit(lambda: lib.t2amd_lstm_step_fwd_f32(C.byref(d16), s)))
print("bf16 LSTM_a alone  %.2f us" % timeit(lambda: lib.t2amd_lstm_step_fwd_f32(C.byref(a16), s)))
print("bf16 fused d||a    %.2f us" % timeit(lambda: lib.t2amd_lstm_step_fwd2_f32(C.byref(d16), C.byref(a16), s)))
for ns in (1, 2, 4):
    gd, kgd = mkplain16(4 * Hd, Ha + E + Hd, ns)
    ga, kga = mkplain16(4 * Ha, E + Ha, ns)
    print("bf16 dgrad d||a nsplit=%d  %.2f us" % (ns, timeit(lambda: lib.t2amd_skinny_gemm2_f32(C.byref(gd), C.byref(ga), s))))

# K sweep of the wide bf16 kernel (one problem per launch): slope = time per 128-k tile, intercept = fixed cost
for K in (256, 512, 1024, 2048, 4096):
    st, keep_ = mk16(K, Hd, [K])
    print("bf16 LSTM K=%4d (%2d tiles) alone  %.2f us" % (K, K // 128, timeit(lambda: lib.t2amd_lstm_step_fwd_f32(C.byref(st), s))))

# in-kernel phase stamps of the wide kernel (T2AMD_ATTN_TS=1): entry, prologue done, first tile landed, k loop done,
# partial sums exchanged, stores issued -- for workgroup 0 (a decoder-LSTM workgroup) of the fused launch
if os.environ.get("T2AMD_ATTN_TS") == "1":
    lib.t2amd_lstm_step_fwd2_f32(C.byref(d16), C.byref(a16), s); torch.cuda.synchronize()
    gd, kgd = mkplain16(4 * Hd, Ha + E + Hd, 2); ga, kga = mkplain16(4 * Ha, E + Ha, 2)
    lib.t2amd_skinny_gemm2_f32(C.byref(gd), C.byref(ga), s); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 128)()
    lib.t2amd_debug_attn_ts_.argtypes = [C.c_void_p]
    assert lib.t2amd_debug_attn_ts_(buf) == 0
    for base, nm in ((64, "LSTM pair "), (80, "dgrad pair")):
        ts = [buf[base + i] for i in range(6)]
        print(nm, "stamps (us from entry): " + " ".join("%.2f" % ((t - ts[0]) / 100.0) for t in ts))
