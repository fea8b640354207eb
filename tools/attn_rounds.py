#!/usr/bin/env python3
"""Does the workgroup-round quantisation of the pipelined attention kernel show?  N = 4992, H = 32, head_dim 64 at batch
sizes whose 256-row workgroups make 2.5 .. 5 rounds of the chip's 512 slots (2 per CU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ltx-video-gpupoor_amd"))
import torch  # noqa: E402
from ltxmi import ops  # noqa: E402

N, H, dh = 4992, 32, 64
for B in (2, 3, 4, 5, 8, 3):
    qkv = torch.randn(B, N, 3, H, dh, device="cuda").to(torch.bfloat16)
    out = torch.empty(B, N, H, dh, device="cuda", dtype=torch.bfloat16)
    f = lambda: ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], out=out)
    for _ in range(10):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    wgs = B * H * ((N + 255) // 256)
    print(f"B {B}: {wgs} workgroups = {wgs / 512:.2f} rounds: {us:7.1f} us  = {us / (wgs / 512):6.1f} us per round-equivalent, "
          f"{4.0 * B * H * N * N * dh / us / 1e6:7.1f} TF", flush=True)
