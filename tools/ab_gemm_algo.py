#!/usr/bin/env python3
"""In-process A/B of the GEMM kernels selectable through ltxmi_gemm_args.algo on the hot shapes (alternating launches on
the same tensors): 0 = the product's choice (persistent 256x256), 256 = the non-persistent 256x256 kernel, 128 = 128x128
tiles (round 2 also had 4 / 5 = experimental K loops here, see profiles/r02_gemm_ring_w4.log).
    python tools/ab_gemm_algo.py [algo ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from ltxmi import ops  # noqa: E402


def main():
    algos = [int(a) for a in sys.argv[1:]] or [0, 256]
    dev = torch.device("cuda", 0)
    shapes = [(14976, 8192, 2048, ops.EPI_GELU_TANH, "ff1"), (14976, 2048, 8192, ops.EPI_GATE_RESIDUAL, "ff2"),
              (14976, 6144, 2048, ops.EPI_NONE, "qkv"), (14976, 2048, 2048, ops.EPI_GATE_RESIDUAL, "to_out"),
              (4992, 8192, 2048, ops.EPI_GELU_TANH, "ff1 B1"), (8192, 8192, 8192, ops.EPI_NONE, "8k^3")]
    for (M, N, K, epi, name) in shapes:
        a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev).to(torch.bfloat16)
        out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        res = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi == ops.EPI_GATE_RESIDUAL else None
        ref = None
        times = [[] for _ in algos]
        for rep in range(6):
            for i, al in enumerate(algos):
                for _ in range(3):
                    ops.gemm(a, w, b, out=out, epilogue=epi, residual=res, algo=al)
                if rep == 0:
                    if ref is None:
                        ref = out.clone()
                    elif not torch.equal(ref, out):
                        print(f"  !! algo {al} differs from algo {algos[0]}: max {float((ref.float() - out.float()).abs().max()):.4g}")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.gemm(a, w, b, out=out, epilogue=epi, residual=res, algo=al)
                e1.record()
                torch.cuda.synchronize()
                if rep > 0:
                    times[i].append(e0.elapsed_time(e1) / 10)
        med = [sorted(t)[len(t) // 2] for t in times]
        print(f"{name:8s} {M}x{N}x{K}: " + " | ".join(f"algo {al}: {m:.4f} ms {2.0 * M * N * K / m / 1e9:7.1f} TF" for al, m in zip(algos, med))
              + f" | x{med[0] / med[-1]:.3f}", flush=True)


if __name__ == "__main__":
    main()
