#!/usr/bin/env python3
"""Diagnostic only: in-kernel s_memtime stamps of the direct convolution (-DLTXMI_CONV_STAMPS build, `make stamps`) on the
VAE decoder's layer shapes.
    LTXMI_LIB=ltx-video-gpupoor_amd/ltxmi/libltxmi_stamp.so python tools/conv_stamps.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ltxmi import ops, _lib  # noqa: E402

dev = torch.device("cuda", 0)
buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
_lib.lib.ltxmi_debug_set_conv_stamps.restype = ctypes.c_int
_lib.lib.ltxmi_debug_set_conv_stamps.argtypes = [ctypes.c_void_p]
assert _lib.lib.ltxmi_debug_set_conv_stamps(buf.data_ptr()) == 0
for (T, Hh, W, cin, cout, name) in [(97, 128, 192, 128, 128, "128->128 full res"), (49, 64, 96, 256, 256, "256->256"),
                                    (25, 32, 48, 512, 512, "512->512"), (13, 16, 24, 1024, 1024, "1024->1024")]:
    x = torch.randn(1, T, Hh, W, cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(cout, 27 * cin, device=dev) * (27 * cin) ** -0.5).to(torch.bfloat16)
    b = torch.randn(cout, device=dev).to(torch.bfloat16)
    for _ in range(3):
        y = ops.conv3d(x, w, b, causal=False, pad_replicate=True, algo=2)
    buf.zero_()
    y = ops.conv3d(x, w, b, causal=False, pad_replicate=True, algo=2)
    torch.cuda.synchronize()
    d = buf.cpu().numpy().reshape(256, 8, 8).astype("float64")
    chunks = np.full(d[..., 0].shape, cin // 64, dtype='float64') * (2400 // 256 if False else 1)
    tiles = 1.0
    taps = chunks * 27
    med = lambda a: float(np.median(a))
    # [3] barrier release -> top of the tap body, [6] weight DMA issue, [7] MFMA blocks + next tap's fragment reads,
    # [0] table read + halo piece issue, [1] vmcnt wait, [2] barrier wait;  [4] / [5]: the once-per-tile load phase
    names = [(3, "loop"), (6, "w-issue"), (7, "mfma+reads"), (0, "halo-issue"), (1, "vmcnt"), (2, "barrier")]
    tot = sum(d[..., i] for i, _ in names) / taps
    print(f"{name:18s}: per tap, median over the first round's waves: " + "  ".join(f"{n} {med(d[..., i] / taps):5.0f}" for i, n in names)
          + f"  = {med(tot):6.0f} cycles (matrix pipe: 1024 per SIMD pair);  per tile: first loads issue {med(d[..., 4]):6.0f} wait {med(d[..., 5]):6.0f}", flush=True)
    for wv in range(8):
        print(f"      wave {wv}: " + "  ".join(f"{n} {med(d[:, wv, i] / taps[:, wv]):5.0f}" for i, n in names))
