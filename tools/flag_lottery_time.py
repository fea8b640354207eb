#!/usr/bin/env python3
"""One line of timings for the library named by LTXMI_LIB: the workload's self-attention launch as the model makes it (q on load)
or the four GEMM shapes of a DiT block.  python tools/flag_lottery_time.py attention|gemm"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ltx-video-gpupoor_amd"))
import torch  # noqa: E402
from ltxmi import ops  # noqa: E402


def ms(fn, iters, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    return sorted(ts)[len(ts) // 2]


g = torch.Generator(device="cuda").manual_seed(5)
if sys.argv[1] == "attention":
    B, N, H, dh = 3, 4992, 32, 64
    D = H * dh
    raw = torch.randn(B, N, D, generator=g, device="cuda").to(torch.bfloat16)
    k = torch.randn(B, N, H, dh, generator=g, device="cuda").to(torch.bfloat16)
    v = torch.randn(B, N, H, dh, generator=g, device="cuda").to(torch.bfloat16)
    w = (1 + 0.1 * torch.randn(D, generator=g, device="cuda")).to(torch.bfloat16)
    ang = torch.rand(N, D // 2, generator=g, device="cuda") * 6.28
    cos = torch.cos(ang).repeat_interleave(2, dim=-1).to(torch.bfloat16).contiguous()
    sin = torch.sin(ang).repeat_interleave(2, dim=-1).to(torch.bfloat16).contiguous()
    rstd = torch.rsqrt(raw.float().pow(2).mean(-1) + 1e-6).reshape(-1).contiguous()
    out = torch.empty(B, N, H, dh, device="cuda", dtype=torch.bfloat16)
    t = ms(lambda: ops.attention(raw.view(B, N, H, dh), k, v, out=out, softmax_scale=dh ** -0.5, q_norm=(rstd, w, 1e-6), rope=(cos, sin, N)), 30)
    chk = float(out.float().abs().sum())
    print(f"attention B3 N4992 on load: {t:.4f} ms  {4.0 * B * H * N * N * dh / t / 1e9 / 2500:.4f} of peak  checksum {chk:.6e}", flush=True)
else:
    M = 14976
    line = []
    for name, N_, K_, epi in (("ff1", 8192, 2048, ops.EPI_GELU_TANH if hasattr(ops, "EPI_GELU_TANH") else 1), ("ff2", 2048, 8192, 0), ("qkv", 6144, 2048, 0), ("to_out", 2048, 2048, 0)):
        a = torch.randn(M, K_, generator=g, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N_, K_, generator=g, device="cuda") * K_ ** -0.5).to(torch.bfloat16)
        b = torch.randn(N_, generator=g, device="cuda").to(torch.bfloat16)
        out = torch.empty(M, N_, device="cuda", dtype=torch.bfloat16)
        t = ms(lambda: ops.gemm(a, w, b, out=out, epilogue=epi), 20)
        line.append(f"{name} {t:.4f} ms {2.0 * M * N_ * K_ / t / 1e9 / 2500:.4f}")
    print("gemm M14976: " + " | ".join(line), flush=True)
