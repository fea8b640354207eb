#!/usr/bin/env python3
"""A/B several builds of libltxmi.so on the VAE decoder's convolution shapes inside ONE process (alternating launches on
the same tensors); the first library is the reference of the ratios and of a bit-equality check.
    python tools/ab_conv.py libA.so libB.so [...]
An arm may be written lib.so:ALGO (ltxmi_conv3d_args.algo for that arm: 3 / 4 = the direct convolution's four- / eight-wave form)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from ltxmi import _lib  # noqa: E402


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    lib.ltxmi_conv3d_ndhwc_bf16.restype = ctypes.c_int32
    lib.ltxmi_conv3d_ndhwc_bf16.argtypes = [ctypes.POINTER(_lib.Conv3dArgs), ctypes.c_void_p]
    return lib


def main():
    arms = [a.rsplit(":", 1) if ":" in a else [a, "0"] for a in sys.argv[1:]]
    libs = [load(p) for p, _ in arms]
    algos = [int(al) for _, al in arms]
    names = [(os.path.basename(p).replace("libltxmi", "").replace(".so", "") or "base") + (f":{al}" if al != "0" else "") for p, al in arms]
    # (T, H, W, Cin, Cout, depth-to-space, skip add): the layers of a 768x512x97 decode, with their share of it
    shapes = [(97, 128, 192, 128, 128, 0, 1, "128->128 +skip (x4: 8.0 ms)"), (49, 64, 96, 256, 256, 0, 0, "256->256 (x4: 3.8 ms)"),
              (49, 64, 96, 256, 1024, 1, 0, "256->1024 d2s (4.1 ms)"), (25, 32, 48, 512, 512, 0, 1, "512->512 +skip (x4: 2.2 ms)"),
              (25, 32, 48, 512, 2048, 1, 0, "512->2048 d2s (2.0 ms)"), (13, 16, 24, 1024, 1024, 0, 0, "1024->1024 (x4: 1.5 ms)"),
              (13, 16, 24, 1024, 4096, 1, 0, "1024->4096 d2s (1.5 ms)"), (97, 128, 192, 128, 48, 0, 0, "conv_out 128->48 (1.25 ms)"),
              # outside the sum: the epilogues with the PixelNorm -> AdaLN -> SiLU (post_norm; "+ y_norm": the raw result too)
              (97, 128, 192, 128, 128, 0, 0, "128->128 post_norm"), (97, 128, 192, 128, 128, 0, 1, "128->128 +skip + y_norm"),
              (49, 64, 96, 256, 1024, 1, 0, "256->1024 d2s + y_norm")]
    stream = torch.cuda.current_stream().cuda_stream
    total = [0.0] * len(libs)
    for (T, H, W, cin, cout, d2s, add, name) in shapes:
        x = torch.randn(1, T, H, W, cin, device="cuda").to(torch.bfloat16)
        w = (torch.randn(cout, 27 * cin, device="cuda") * (27 * cin) ** -0.5).to(torch.bfloat16)
        b = torch.randn(cout, device="cuda").to(torch.bfloat16)
        y = torch.empty((1, 2 * T - 1, 2 * H, 2 * W, cout // 8) if d2s else (1, T, H, W, cout), device="cuda", dtype=torch.bfloat16)
        skip = torch.randn(1, T, H, W, cout, device="cuda").to(torch.bfloat16) if add else None
        a = _lib.Conv3dArgs()
        a.x, a.w, a.bias, a.y = x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr()
        a.B, a.T, a.H, a.W, a.Cin, a.Cout = 1, T, H, W, cin, cout
        a.causal, a.pad_replicate, a.d2s = 0, 1, d2s
        if d2s:
            a.residual, a.res_channels = x.data_ptr(), cin
        if add:
            a.add = skip.data_ptr()
        if "post_norm" in name or "y_norm" in name:
            cn = cout // 8 if d2s else cout
            sc, sh = torch.randn(1, cn, device="cuda") * 0.3, torch.randn(1, cn, device="cuda") * 0.3
            a.post_norm, a.post_scale, a.post_shift, a.post_eps = 1, sc.data_ptr(), sh.data_ptr(), 1e-8
            if "y_norm" in name:
                y2 = torch.empty_like(y)
                a.y_norm = y2.data_ptr()
        times = [[] for _ in libs]
        ref = None
        for rep in range(6):
            for i, lib in enumerate(libs):
                a.algo = algos[i]
                for _ in range(2):
                    assert lib.ltxmi_conv3d_ndhwc_bf16(ctypes.byref(a), stream) == 0
                if rep == 0:
                    if ref is None:
                        ref = y.clone()
                    elif not torch.equal(ref, y):
                        print(f"  !! {names[i]} differs from {names[0]}: max {float((ref.float() - y.float()).abs().max()):.4g}")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    lib.ltxmi_conv3d_ndhwc_bf16(ctypes.byref(a), stream)
                e1.record()
                torch.cuda.synchronize()
                if rep > 0:
                    times[i].append(e0.elapsed_time(e1) / 5)
        med = [sorted(t)[len(t) // 2] for t in times]
        mult = 4 if "x4" in name else (0 if "norm" in name else 1)
        for i, m in enumerate(med):
            total[i] += mult * m
        flop = 2.0 * T * H * W * cout * 27 * cin
        print(f"{name:30s}: " + " | ".join(f"{n} {m:.3f} ms {flop / m / 1e9:6.0f} TF x{med[0] / m:.3f}" for n, m in zip(names, med)), flush=True)
    print("sum over a decode's convolutions: " + " | ".join(f"{n} {t:.2f} ms x{total[0] / t:.3f}" for n, t in zip(names, total)))


if __name__ == "__main__":
    main()
