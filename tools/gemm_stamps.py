#!/usr/bin/env python3
"""Diagnostic only: read the in-kernel s_memtime stamps of a stamped build of the persistent GEMM
(GEMM notes) after running the FF1 shape.  Usage: LTXMI_LIB=.../libltxmi_stamp.so python tools/gemm_stamps.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ltxmi import ops, _lib  # noqa: E402

for (M, N, K, epi) in [(14976, 8192, 2048, ops.EPI_GELU_TANH), (8192, 8192, 8192, ops.EPI_NONE)]:
    a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(30):
        ops.gemm(a, w, b, out=out, epilogue=epi)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (256 * 8 * 8))()
    _lib.lib.ltxmi_debug_read_stamps.restype = ctypes.c_int
    rc = _lib.lib.ltxmi_debug_read_stamps(buf)
    d = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 8).astype(np.float64)
    n = d[..., 5]
    print(f"shape {M}x{N}x{K}: rc {rc}; per middle K-tile, cycles (median over waves; wave 0 / wave 7 of the median workgroup)")
    for name, idx in (("slot 0 (see the stamped build)", 0), ("vmcnt wait", 1), ("barrier", 2), ("slot 3 (see the stamped build)", 3)):
        v = d[..., idx] / np.maximum(n, 1)
        print(f"  {name:42s} {np.median(v):8.0f}   w0 {np.median(v[:, 0]):8.0f}  w7 {np.median(v[:, 7]):8.0f}")
    tiles = d[..., 7]
    print(f"  epilogue per tile {np.median(d[..., 4] / np.maximum(tiles, 1)):8.0f};  whole kernel per wave {np.median(d[..., 6]):10.0f} cycles, tiles per WG {np.median(tiles):.1f}")
