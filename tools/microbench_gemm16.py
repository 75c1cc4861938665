"""bf16-resident GEMM (csrc/gemm16.hip) at the shapes of one training step, beside the f32-source bf16 kernel of gemm.hip:
python tools/microbench_gemm16.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron2_amd import native

SHAPES = [  # name, M, N, K, splitk
    ("wgrad_d  dG^T[4096,55k].X[55k,2560]", 4096, 2560, 55680, 3),
    ("wgrad_a  dG^T[4096,55k].X[55k,1792]", 4096, 1792, 55680, 4),
    ("wgrad_pc dY^T[512,55k].X[55k,2560]", 512, 2560, 55680, 12),
    ("post_conv X[55k,2560].W[512,2560]^T", 55680, 512, 2560, 1),
    ("gin_fwd  X[55k,256].W[4096,256]^T", 55680, 4096, 256, 1),
    ("square 8192", 8192, 8192, 8192, 1),
]


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = torch.device("cuda:0")
    native.load()
    g = torch.Generator().manual_seed(0)
    for name, M, N, K, sk in SHAPES:
        A = torch.randn(M, K, generator=g).to(dev)
        B = torch.randn(N, K, generator=g).to(dev)
        A16, B16 = A.bfloat16(), B.bfloat16()
        Cm = torch.empty(M, N, device=dev)
        part = torch.empty(sk, M * N, device=dev) if sk > 1 else None
        line = "%-40s" % name
        for s in sorted({1, sk}):
            pp = torch.empty(s, M * N, device=dev) if s > 1 else None
            ms = timeit(lambda: native.gemm16_tn(Cm, A16, B16, splitk=s, partials=pp))
            line += "  gemm16/sk%d %7.3f ms %6.0f TF" % (s, ms, 2.0 * M * N * K / ms * 1e-9)
        if M <= 4096 and M * K <= 4096 * 55680:
            # the same product from K-major operands ([K][M], [K][N]: the layout the slabs have), transposing LDS reads
            At16, Bt16 = A16.t().contiguous(), B16.t().contiguous()
            for s in sorted({1, sk}):
                pp = torch.empty(s, M * N, device=dev) if s > 1 else None
                ms = timeit(lambda: native.gemm16_kk(Cm, At16, Bt16, K, splitk=s, partials=pp))
                line += "  kk/sk%d %7.3f ms %6.0f TF" % (s, ms, 2.0 * M * N * K / ms * 1e-9)
            del At16, Bt16
        # the f32-source bf16 kernel on the same product (K-contiguous operands)
        ms = timeit(lambda: (native.gemm(part[0].view(M, N), A, B, fast=2, splitk=sk, partials=part) if sk > 1
                             else native.gemm(Cm, A, B, fast=2)))
        line += "  | f32-source/sk%d %7.3f ms %6.0f TF" % (sk, ms, 2.0 * M * N * K / ms * 1e-9)
        if M * K <= 4096 * 55680:
            At = torch.empty(K, M, device=dev)          # the slab as the engine holds it: [K][M]
            A16t = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
            ms = timeit(lambda: native.transpose_cast_bf16(At, A16t))
            line += "  | transpose-cast [K][M] f32 -> bf16 [M][K] %6.3f ms %5.0f GB/s" % (ms, (M * K * 6) / ms * 1e-6)
        print(line, flush=True)
        del A, B, A16, B16, Cm


if __name__ == "__main__":
    main()
