"""Micro-benchmark of the LSTM step launches (tools only; not part of the product path).
    python tools/microbench_lstm.py
"""
import ctypes as C
import os
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
