#!/usr/bin/env python3
"""Diagnostic: the implicit-GEMM convolution against the dense GEMM of the same M, N, K (tile kernel vs tile kernel)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from ltxmi import ops  # noqa: E402


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for (T, H, W, C, Co) in [(49, 64, 96, 256, 256), (97, 128, 192, 128, 128), (25, 32, 48, 512, 512), (49, 64, 96, 256, 1024)]:
    x = torch.randn(1, T, H, W, C, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Co, 27 * C, device="cuda") * (27 * C) ** -0.5).to(torch.bfloat16)
    b = torch.randn(Co, device="cuda").to(torch.bfloat16)
    M, K = T * H * W, 27 * C
    fl = 2.0 * M * Co * K
    for rep in (False, True):
        ms = timeit(lambda: ops.conv3d(x, w, b, False, rep, algo=1))        # implicit GEMM
        print(f"conv {C}->{Co} @ {T}x{H}x{W} replicate={rep}: {ms:8.3f} ms {fl / ms / 1e9:7.1f} TF", flush=True)
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    ms = timeit(lambda: ops.gemm(a, w, b, algo=256))
    print(f"dense gemm {M}x{Co}x{K}: {ms:8.3f} ms {fl / ms / 1e9:7.1f} TF (256x256 tile kernel)", flush=True)
    del a, x, w
