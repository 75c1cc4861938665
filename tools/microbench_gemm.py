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
    # probes: non power-of-two leading dimensions / square-ish tiles
    ("probe_m4224 dG^T[4224,55k].X[55k,2560]", 4224, 2560, 55000, True, True),
    ("probe_m3968 dG^T[3968,55k].X[55k,2688]", 3968, 2688, 55000, True, True),
]


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--prec", default="0,1,2")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--splitk", default="1", help="comma list; >1 writes partial slabs (reduce not timed)")
    args = ap.parse_args()
    precs = [int(x) for x in args.prec.split(",")]
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    for name, M, N, K, a_km, b_kn in SHAPES:
        if args.only and not name.startswith(args.only):
            continue
        A = torch.randn((K, M) if a_km else (M, K), generator=g).to(dev)
        B = torch.randn((K, N) if b_kn else (N, K), generator=g).to(dev)
        Cm = torch.empty(M, N, device=dev)
        line = "%-40s" % name
        for fast, sk in [(f, k) for f in precs for k in [int(x) for x in args.splitk.split(",")]]:
            part = torch.empty(sk, M * N, device=dev) if sk > 1 else None

            def run():
                if sk > 1:
                    native.gemm(part[0].view(M, N), A, B, a_km=a_km, b_kn=b_kn, fast=fast, splitk=sk, partials=part)
                else:
                    native.gemm(Cm, A, B, a_km=a_km, b_kn=b_kn, fast=fast)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = args.iters
            e0.record()
            for _ in range(n):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            line += "  p%d%s %7.3f ms %6.1f TF" % (fast, ("/sk%d" % sk) if sk > 1 else "", ms, 2.0 * M * N * K / ms * 1e-9)
        print(line, flush=True)


if __name__ == "__main__":
    main()
