"""Times the tiled GEMM kernels (exact f32 / split-bf16 x3 / plain bf16) on the shapes one training step
issues: python tools/microbench_gemm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron2_amd import native

SHAPES = [
    # name, M, N, K, a_km, b_kn   (Cm[M,N] = A.B)
    ("gin_fwd   X[55k,256].W[4096,256]^T", 55000, 4096, 256, False, False),
    ("proj_fwd  X[55k,1536].W[80,1536]^T", 55000, 80, 1536, False, False),
    ("post_conv X[55k,2560].W[512,2560]^T", 55000, 512, 2560, False, False),
    ("dgrad     dG[55k,4096].W[4096,256]", 55000, 256, 4096, False, True),
    ("wgrad_d   dG^T[4096,55k].X[55k,2560]", 4096, 2560, 55000, True, True),
    ("wgrad_a   dG^T[4096,55k].X[55k,1792]", 4096, 1792, 55000, True, True),
    ("wgrad_pc  dY^T[512,55k].X[55k,2560]", 512, 2560, 55000, True, True),
]


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    for name, M, N, K, a_km, b_kn in SHAPES:
        A = torch.randn((K, M) if a_km else (M, K), generator=g).to(dev)
        B = torch.randn((K, N) if b_kn else (N, K), generator=g).to(dev)
        Cm = torch.empty(M, N, device=dev)
        line = "%-40s" % name
        for fast in (0, 1, 2):
            sk = 1
            part = None
            for _ in range(3):
                native.gemm(Cm, A, B, a_km=a_km, b_kn=b_kn, fast=fast)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            e0.record()
            for _ in range(n):
                native.gemm(Cm, A, B, a_km=a_km, b_kn=b_kn, fast=fast)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            line += "  p%d %7.3f ms %6.1f TF" % (fast, ms, 2.0 * M * N * K / ms * 1e-9)
        print(line, flush=True)


if __name__ == "__main__":
    main()
