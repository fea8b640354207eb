#!/usr/bin/env python3
"""Calibration only (not part of the product path): what the vendor GEMM (torch.matmul -> hipBLASLt/rocBLAS)
reaches on the hot shapes of the denoise step, on the same box and clocks as tools/microbench.py."""
import torch

def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for rep in range(2):
    for (M, N, K) in [(14976, 8192, 2048), (14976, 2048, 8192), (14976, 6144, 2048), (14976, 2048, 2048), (8192, 8192, 8192)]:
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        ms = timeit(lambda: torch.matmul(a, w.t()), 20)
        print(f"vendor gemm {M}x{N}x{K}: {ms:8.4f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s", flush=True)
