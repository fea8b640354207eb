#!/usr/bin/env python3
"""Per-row error of the head_dim-128 pipelined attention kernel on the 'redo' test input (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import torch
from ltxmi import ops

def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)

B, H, dh = 4, 32, 128
Lq, Lk = int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 1500
q, k, v = rnd(B, Lq, H, dh, seed=43), rnd(B, Lk, H, dh, seed=44), rnd(B, Lk, H, dh, seed=45)
k[:, Lk - 30] = q[:, 5] * 30.0
k[:, Lk // 2] = q[:, 40] * 25.0
out = ops.attention(q.cuda(), k.cuda(), v.cuda()).float().cpu()
qf, kf, vf = q.float().cuda(), k.float().cuda(), v.float().cuda()
s = torch.einsum("blhd,bkhd->bhlk", qf, kf) * dh ** -0.5
truth = torch.einsum("bhlk,bkhd->blhd", torch.softmax(s, -1), vf).cpu()
err = (out - truth).norm(dim=-1) / truth.norm(dim=-1).clamp_min(1e-9)      # [B, Lq, H]
print("overall rel L2", float((out - truth).norm() / truth.norm()))
worst = torch.topk(err.flatten(), 12)
for val, idx in zip(worst.values.tolist(), worst.indices.tolist()):
    b, rem = divmod(idx, Lq * H)
    l, h = divmod(rem, H)
    sc = s[b, h, l].cpu() * 1.4427
    top = torch.topk(sc, 3)
    first128 = float(sc[:128].max())
    print(f"b{b} h{h} row {l}: err {val:.3e}; top scores (bits) {[round(x, 1) for x in top.values.tolist()]} at {top.indices.tolist()}; max of first 128 keys {first128:.1f}")
print("rows with err > 1e-2:", int((err > 1e-2).sum()), "of", err.numel())
