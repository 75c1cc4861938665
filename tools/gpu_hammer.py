"""A second process that keeps the GPU busy for <seconds>: `mm` = 2048^3 bf16 GEMMs (MFMA waves on every SIMD), `ew` = a memory-bound
elementwise kernel.  Beside tools/stress_attn_bwd.py / tools/stress_lds_poison.py.

    python tools/gpu_hammer.py 60 mm &
"""
import sys, time, torch
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30
kind = sys.argv[2] if len(sys.argv) > 2 else "mm"
a = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
b = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
x = torch.randn(1 << 22, device="cuda")
t0 = time.time()
n = 0
while time.time() - t0 < secs:
    for _ in range(50):
        if kind == "mm":
            c = a @ b
        else:
            x.mul_(1.0001)
    torch.cuda.synchronize()
    n += 50
print("hammer", kind, n)
