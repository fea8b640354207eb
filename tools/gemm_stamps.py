#!/usr/bin/env python3
"""Diagnostic only: in-kernel s_memtime stamps of a stamped build of the persistent GEMM (-DLTXMI_GEMM_STAMPS, see
gemm.hip) on the hot shapes, for the product kernel (algo 0: 8 waves of 128x64; round 2 also stamped experimental K loops
through other algo values, see profiles/r02_gemm_ring_w4.log).
    make -C ltx-video-gpupoor_amd/csrc stamps && LTXMI_LIB=ltx-video-gpupoor_amd/ltxmi/libltxmi_stamp.so python tools/gemm_stamps.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from ltxmi import ops, _lib  # noqa: E402

dev = torch.device("cuda", 0)
buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
_lib.lib.ltxmi_debug_set_gemm_stamps.restype = ctypes.c_int
_lib.lib.ltxmi_debug_set_gemm_stamps.argtypes = [ctypes.c_void_p]
assert _lib.lib.ltxmi_debug_set_gemm_stamps(buf.data_ptr()) == 0
algos = [int(a) for a in sys.argv[1:]] or [0]
for (M, N, K, epi, name) in [(14976, 8192, 2048, ops.EPI_GELU_TANH, "ff1"), (14976, 2048, 8192, ops.EPI_GATE_RESIDUAL, "ff2"),
                             (14976, 2048, 2048, ops.EPI_GATE_RESIDUAL, "to_out"), (8192, 8192, 8192, ops.EPI_NONE, "8k^3")]:
    a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi == ops.EPI_GATE_RESIDUAL else None
    for al in algos:
        for _ in range(20):
            ops.gemm(a, w, b, out=out, epilogue=epi, residual=res, algo=al)
        buf.zero_()
        ops.gemm(a, w, b, out=out, epilogue=epi, residual=res, algo=al)
        torch.cuda.synchronize()
        nw = 4 if al == 4 else 8
        d = buf.cpu().numpy().reshape(256, 8, 8)[:, :nw].astype("float64")
        n = d[..., 5].clip(min=1)
        tiles = d[..., 7].clip(min=1)
        import numpy as np
        med = lambda x: float(np.median(x))
        pa, vm, bar, pb = (med(d[..., i] / n) for i in range(4))
        unit, bound = ("mini-tile (K 32)", 1024) if al == 4 else ("middle K-tile", 2048)
        extra = f"  other {med(d[..., 4] / n):5.0f}" if al == 4 else ""
        print(f"{name} {M}x{N}x{K} algo {al}: per {unit} (median over waves): MFMA phase A {pa:6.0f}  vmcnt {vm:5.0f}  barrier {bar:5.0f}  "
              f"phase B {pb:6.0f}{extra}  = {pa + vm + bar + pb + (med(d[..., 4] / n) if al == 4 else 0):6.0f} cycles (pipe-bound: {bound});  "
              f"epilogue/tile {med(d[..., 6] / tiles):7.0f};  rest/tile {med(d[..., 4] / tiles):7.0f};  tiles/WG {med(tiles):.1f}", flush=True)
        # per wave (round 3: waves 0-3 issue all LDS-DMA, waves 4-7 only compute)
        for wv in range(nw):
            x = d[:, wv]
            print(f"      wave {wv}: A {med(x[:, 0] / n[:, wv]):6.0f}  vmcnt {med(x[:, 1] / n[:, wv]):5.0f}  barrier {med(x[:, 2] / n[:, wv]):5.0f}  "
                  f"B {med(x[:, 3] / n[:, wv]):6.0f}  epilogue/tile {med(x[:, 6] / tiles[:, wv]):7.0f}")
