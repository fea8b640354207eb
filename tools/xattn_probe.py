#!/usr/bin/env python3
"""Where the short-key attention launch (B 3, H 32, 4992 queries x 256 keys) spends its time: the same launch with the Q reads
and / or the O writes collapsed onto one row (token stride 0: every access hits L2), i.e. without its HBM traffic."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from ltxmi import ops  # noqa: E402

dev = "cuda"
B, H, dh, Lq, Lk = 3, 32, 64, 4992, 256
q = torch.randn(B, Lq, H, dh, device=dev).to(torch.bfloat16)
kv = torch.randn(B, Lk, 2, H, dh, device=dev).to(torch.bfloat16)
k, v = kv[:, :, 0], kv[:, :, 1]
kb = torch.zeros(B, Lk, device=dev)
kb[:, 200:] = -10000.0
out = torch.empty(B, Lq, H, dh, device=dev, dtype=torch.bfloat16)
q1 = q[:, :1].expand(B, Lq, H, dh)
o1 = torch.empty(B, 1, H, dh, device=dev, dtype=torch.bfloat16).expand(B, Lq, H, dh)


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


print(f"kernel id {ops.attention_kernel_id(B, H, Lq, Lk, dh, True, 2 * H * dh, 2 * H * dh)}")
print(f"full launch                 {t(lambda: ops.attention(q, k, v, out=out, key_bias=kb)):.1f} us")
print(f"Q from one row (L2)         {t(lambda: ops.attention(q1, k, v, out=out, key_bias=kb)):.1f} us")
print(f"O to one row (L2)           {t(lambda: ops.attention(q, k, v, out=o1, key_bias=kb)):.1f} us")
print(f"both (no HBM traffic)       {t(lambda: ops.attention(q1, k, v, out=o1, key_bias=kb)):.1f} us")
