#!/usr/bin/env python3
"""Where an iteration of the pipelined attention kernel spends its cycles: runs a DIAGNOSTIC build
(make ATTN_DEFS=-DLTXMI_ATTN_STAMPS OUT=...) whose waves sum s_memtime differences per section.
    python tools/attn_stamps.py path/to/libltxmi_stamps.so [B H N]
Sections: 0 barrier wait, 1 DMA issue, 2 seg1 head, 3 seg1 chunks, 4 seg2 head, 5 seg2 chunks, 6 vmcnt wait.
Read the SHARES, never this build's run time (the stamps forbid overlaps the real kernel has)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from ltxmi import _lib  # noqa: E402

lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
lib.ltxmi_attention_fwd_bf16.restype = ctypes.c_int32
lib.ltxmi_attention_fwd_bf16.argtypes = [ctypes.POINTER(_lib.AttnArgs), ctypes.c_void_p]
lib.ltxmi_debug_set_attn_stamps.argtypes = [ctypes.c_void_p]
B, H, N = (int(x) for x in sys.argv[2:5]) if len(sys.argv) >= 5 else (3, 32, 4992)
dh = 64
qkv = torch.randn(B, N, 3, H, dh, device="cuda").to(torch.bfloat16)
o = torch.empty(B, N, H, dh, device="cuda", dtype=torch.bfloat16)
dbg = torch.zeros(4096 * 4 * 8, device="cuda", dtype=torch.int64)
assert lib.ltxmi_debug_set_attn_stamps(dbg.data_ptr()) == 0
a = _lib.AttnArgs()
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
a.q, a.q_stride_b, a.q_stride_l = q.data_ptr(), q.stride(0), q.stride(1)
a.k, a.k_stride_b, a.k_stride_l = k.data_ptr(), k.stride(0), k.stride(1)
a.v, a.v_stride_b, a.v_stride_l = v.data_ptr(), v.stride(0), v.stride(1)
a.o, a.o_stride_b, a.o_stride_l = o.data_ptr(), o.stride(0), o.stride(1)
a.key_bias, a.bias_stride_b = None, 0
a.B, a.H, a.Lq, a.Lk, a.head_dim, a.softmax_scale = B, H, N, N, dh, dh ** -0.5
if os.environ.get("STAMPS_PLAIN_Q") != "1":
    # the launch form of the DiT: q finished on load (row factor + weight; no RoPE tables here), scale folded into q
    rstd = torch.ones(B * N, device="cuda", dtype=torch.float32)
    wq = torch.ones(H * dh, device="cuda", dtype=torch.bfloat16)
    a.q_rstd, a.q_rstd_stride_b, a.q_rstd_stride_l = rstd.data_ptr(), N, 1
    a.q_norm_weight, a.q_norm_eps = wq.data_ptr(), 1e-6
stream = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    assert lib.ltxmi_attention_fwd_bf16(ctypes.byref(a), stream) == 0
torch.cuda.synchronize()
d = dbg.view(-1, 8).cpu()
d = d[d[:, 7] > 0]
rt = (d[:, 7] >> 32).double()
nt = (d[:, 7:8] & 0xffffffff).double()
d = d.double()
clk = d[:, :7].sum(dim=1) / rt * 100e6
print(f"in-kernel clock (sum of s_memtime sections / s_memrealtime): median {float(clk.median()) / 1e9:.3f} GHz")
per = d[:, :7] / nt
names = ["barrier", "dma issue", "seg1 head", "seg1 chunks", "seg2 head", "seg2 chunks", "vmcnt wait"]
med = per.median(dim=0).values
tot = float(med.sum())
print(f"B{B} H{H} N{N}: cycles per iteration (median over {len(d)} waves), total {tot:.0f}")
for n_, m_ in zip(names, med.tolist()):
    print(f"  {n_:12s} {m_:8.0f}  {m_ / tot:6.1%}")
for w in range(4):
    sel = per[w::4]
    print(f"  wave {w}: " + " ".join(f"{x:7.0f}" for x in sel.median(dim=0).values.tolist()))
