#!/usr/bin/env python3
"""Kernel micro-benchmarks on one MI355X (HIP-event timed, random data).
    python tools/microbench.py attn|gemm|all
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from ltxmi import ops  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def attn():
    for (B, H, N, dh, it) in [(3, 32, 4992, 64, 20), (1, 32, 13376, 64, 10), (1, 32, 32768, 64, 4),
                              (1, 32, 98304, 64, 2), (1, 12, 32760, 128, 4), (3, 32, 4992, 128, 10)]:
        qkv = torch.randn(B, N, 3, H, dh, device=DEV).to(BF)
        out = torch.empty(B, N, H, dh, device=DEV, dtype=BF)
        ms = timeit(lambda: ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], out=out), it)
        tf = 4.0 * B * H * N * N * dh / ms / 1e9
        print(f"attn B{B} H{H} N{N} dh{dh}: {ms:9.3f} ms  {tf:7.1f} TFLOP/s  {tf / 25:.1f}% of 2.5 PF", flush=True)
        del qkv, out


def gemm():
    for (M, N, K, epi, it) in [(14976, 8192, 2048, ops.EPI_GELU_TANH, 20), (14976, 2048, 8192, ops.EPI_NONE, 20),
                               (14976, 6144, 2048, ops.EPI_NONE, 20), (14976, 2048, 2048, ops.EPI_NONE, 20),
                               (4992, 8192, 2048, ops.EPI_GELU_TANH, 20), (4992, 2048, 2048, ops.EPI_NONE, 20),
                               (8192, 8192, 8192, ops.EPI_NONE, 5), (4096, 4096, 4096, ops.EPI_NONE, 20)]:
        a = torch.randn(M, K, device=DEV).to(BF)
        w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(BF)
        b = torch.randn(N, device=DEV).to(BF)
        out = torch.empty(M, N, device=DEV, dtype=BF)
        ms = timeit(lambda: ops.gemm(a, w, b, out=out, epilogue=epi), it)
        tf = 2.0 * M * N * K / ms / 1e9
        print(f"gemm {M}x{N}x{K} epi{epi}: {ms:8.4f} ms  {tf:7.1f} TFLOP/s  {tf / 25:.1f}% of 2.5 PF", flush=True)
        del a, w, b, out


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("attn", "all"):
        attn()
    if what in ("gemm", "all"):
        gemm()
