#!/usr/bin/env python3
"""Per-call timing of one VAE decode at the bench size: every ltxmi.ops call made by the decoder, in order, with tensor
shapes, HIP-event time and (convolutions) TFLOP/s.  Diagnostic; usage: python tools/vae_layers.py [grid T H W]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402
from ltxmi import ops  # noqa: E402
import ltxmi  # noqa: E402

calls = []


def wrap(name, fn):
    def inner(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        shapes = [tuple(t.shape) for t in list(a) + list(k.values()) if torch.is_tensor(t)][:3]
        calls.append((name, shapes, {kk: ("yes" if isinstance(vv, tuple) and any(torch.is_tensor(t) for t in vv) else vv)
                                     for kk, vv in k.items() if not torch.is_tensor(vv) and vv is not None}, e0, e1))
        return r
    return inner


def main():
    dev = torch.device("cuda", 0)
    vae, z, t = bench.make_vae(dev)
    torch.set_grad_enabled(False)
    for _ in range(2):
        ltxmi.vae_decode(z, vae, True, vae_per_channel_normalize=True, timestep=t)
    torch.cuda.synchronize()
    for n in dir(ops):
        f = getattr(ops, n)
        if callable(f) and not n.startswith("_") and getattr(f, "__module__", "") == ops.__name__ and n not in ("check", "watch_launches", "launch_times_ms"):
            setattr(ops, n, wrap(n, f))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ltxmi.vae_decode(z, vae, True, vae_per_channel_normalize=True, timestep=t)
    e1.record()
    torch.cuda.synchronize()
    tot = 0.0
    for name, shapes, kw, a, b in calls:
        ms = a.elapsed_time(b)
        tot += ms
        extra = ""
        if name == "conv3d" and len(shapes) >= 2:
            x, w = shapes[0], shapes[1]
            pos = x[0] * x[1] * x[2] * x[3]
            flop = 2.0 * pos * w[0] * w[1]
            extra = f"  {flop / ms / 1e9:7.1f} TF"
        print(f"{name:28s} {str(shapes):70s} {ms:8.3f} ms{extra}  {kw if kw else ''}")
    print(f"sum of calls {tot:.3f} ms; decode (events around it, with the per-call events inside) {e0.elapsed_time(e1):.3f} ms")


if __name__ == "__main__":
    main()
