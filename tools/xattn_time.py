#!/usr/bin/env python3
"""Cross-attention of the DiT (B 3, H 32, N 4992 queries, 256 text keys with a key bias) with q's RMSNorm as a pass of its
own and applied on load: per-launch HIP-event times."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from ltxmi import ops  # noqa: E402

dev = "cuda"
B, H, dh, Lq, Lk = 3, 32, 64, 4992, 256
D = H * dh
q = torch.randn(B * Lq, D, device=dev).to(torch.bfloat16)
k = torch.randn(B, Lk, H, dh, device=dev).to(torch.bfloat16)
v = torch.randn(B, Lk, H, dh, device=dev).to(torch.bfloat16)
w = torch.ones(D, device=dev).to(torch.bfloat16)
kb = torch.zeros(B, Lk, device=dev)
kb[:, 200:] = -10000.0
ss = q.float().reshape(B * Lq, D // 64, 64).pow(2).sum(-1).contiguous()
out = torch.empty(B, Lq, H, dh, device=dev, dtype=torch.bfloat16)


def t(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


qq = q.clone()
print(f"rmsnorm pass {t(lambda: ops.rmsnorm_rope_(qq, w, 1e-6)):.1f} us + attention {t(lambda: ops.attention(q.view(B, Lq, H, dh), k, v, out=out, key_bias=kb)):.1f} us"
      f"   |   attention with q normalised on load {t(lambda: ops.attention(q.view(B, Lq, H, dh), k, v, out=out, key_bias=kb, q_norm=(ss, w, 1e-6))):.1f} us")
# round 3: one factor per row (ltxmi_rowsumsq_rstd_f32, a launch of its own) instead of 32 partials re-summed per workgroup
rstd = torch.empty(B * Lq, device=dev, dtype=torch.float32)
print(f"row-factor launch {t(lambda: ops.rowsumsq_rstd(ss, D, 1e-6, out=rstd)):.1f} us + attention with q normalised on load from it "
      f"{t(lambda: ops.attention(q.view(B, Lq, H, dh), k, v, out=out, key_bias=kb, q_norm=(rstd, w, 1e-6))):.1f} us")
