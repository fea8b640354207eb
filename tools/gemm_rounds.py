#!/usr/bin/env python3
"""What does the last, partial round of the persistent GEMM cost?  Times N = 8192 / 2048, K = 2048 at M giving exactly
7 / 7.375 (the FF1 shape) / 7.5 / 8 rounds of 256 tiles (round 2: a partial round costs 58-92 % of a full one -- the CUs
that do run it are faster -- which is why a stream-K split of it did not pay; profiles/r02_gemm_streamk.log)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from ltxmi import ops  # noqa: E402

dev = torch.device("cuda", 0)


def t(M, N, K, epi):
    a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    res = out if epi == ops.EPI_GATE_RESIDUAL else None
    ts = []
    for rep in range(5):
        for _ in range(3):
            ops.gemm(a, w, b, out=out, epilogue=epi, residual=res)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm(a, w, b, out=out, epilogue=epi, residual=res)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    return sorted(ts)[2] * 1e3


for (N, K, epi, name) in ((8192, 2048, ops.EPI_GELU_TANH, "ff1"), (2048, 2048, ops.EPI_GATE_RESIDUAL, "to_out"), (2048, 8192, ops.EPI_GATE_RESIDUAL, "ff2")):
    tn = N // 256
    for tiles_m in (256 * 7 // tn, 59, 256 * 7 // tn + 256 // tn // 2, 256 * 8 // tn) if N == 8192 else (32, 48, 59, 64):
        M = tiles_m * 256 - (128 if tiles_m == 59 else 0)
        rounds = tiles_m * tn / 256
        print(f"{name} M={M} ({rounds:.3f} rounds): {t(M, N, K, epi):7.1f} us", flush=True)
