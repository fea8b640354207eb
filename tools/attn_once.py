import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ltx-video-gpupoor_amd"))
import torch
from ltxmi import ops
B, H, N, dh = int(sys.argv[2]) if len(sys.argv) > 2 else 1, 32, int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 64   # attn_once.py [N [B]]
qkv = torch.randn(B, N, 3, H, dh, device="cuda").to(torch.bfloat16)
out = torch.empty(B, N, H, dh, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], out=out)
torch.cuda.synchronize()
