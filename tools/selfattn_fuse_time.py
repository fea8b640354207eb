#!/usr/bin/env python3
"""Self-attention of the DiT (B 3, H 32, N 4992, head_dim 64): q's RMSNorm + RoPE as a pass of its own (rmsnorm_rope_ on q,
then attention) against applied on load inside the attention kernel -- per-launch HIP-event times, alternating."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ltx-video-gpupoor_amd"))
import torch  # noqa: E402
from ltxmi import ops  # noqa: E402

dev = "cuda"
B, H, dh, N = 3, 32, 64, 4992
D = H * dh
qkv = torch.randn(B * N, 3 * D, device=dev).to(torch.bfloat16)
w = torch.ones(D, device=dev).to(torch.bfloat16)
ang = torch.rand(N, D // 2, device=dev) * 6.28
cos = ang.cos().repeat_interleave(2, dim=-1).to(torch.bfloat16)
sin = ang.sin().repeat_interleave(2, dim=-1).to(torch.bfloat16)
ss = qkv[:, :D].float().reshape(B * N, D // 64, 64).pow(2).sum(-1).contiguous()
v5 = qkv.view(B, N, 3, H, dh)
out = torch.empty(B, N, H, dh, device=dev, dtype=torch.bfloat16)
qtmp = qkv.clone()


def t(fn, n=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rep in range(3):
    a = t(lambda: ops.rmsnorm_rope_(qtmp[:, :D], w, 1e-6, cos, sin, N))
    b = t(lambda: ops.attention(v5[:, :, 0], v5[:, :, 1], v5[:, :, 2], out=out))
    c = t(lambda: ops.attention(v5[:, :, 0], v5[:, :, 1], v5[:, :, 2], out=out, q_norm=(ss, w, 1e-6), rope=(cos, sin, N)))
    print(f"q norm + RoPE pass {a:.1f} us + attention {b:.1f} us = {a + b:.1f} us   |   applied on load {c:.1f} us", flush=True)
