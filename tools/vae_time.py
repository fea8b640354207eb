#!/usr/bin/env python3
"""VAE decode timing only (bench.py's time_vae), for A/B runs with LTXMI_LIB.
    python tools/vae_time.py [iters]
    python tools/vae_time.py --ab-post-norm [rounds]    in-process A/B: conv1's norm2 -> SiLU in the convolution's epilogue
                                                        (ops.CONV_POST_NORM_FUSE) against the second launch, alternating decodes
    python tools/vae_time.py --ab-second-output [rounds]   the same for the consumer's norm as a second output of conv2 + skip / of
                                                        the depth-to-space store (ops.CONV_SECOND_OUTPUT_FUSE)
    python tools/vae_time.py --ab-split [rounds]           the 1024-channel stage split over its input channels (ops.CONV_SPLIT)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import bench  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] in ("--ab-post-norm", "--ab-second-output", "--ab-split"):
    knob = {"--ab-post-norm": "CONV_POST_NORM_FUSE", "--ab-second-output": "CONV_SECOND_OUTPUT_FUSE", "--ab-split": "CONV_SPLIT"}[sys.argv[1]]
    import torch
    import ltxmi
    from ltxmi import ops
    vae, z, ts = bench.make_vae("cuda")
    outs, times = {}, {True: [], False: []}
    with torch.no_grad():
        for rnd_ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
            for fuse in (True, False):
                setattr(ops, knob, fuse)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = ltxmi.vae_decode(z, vae, True, vae_per_channel_normalize=True, timestep=ts)
                e1.record()
                torch.cuda.synchronize()
                if rnd_ > 0:
                    times[fuse].append(e0.elapsed_time(e1))
                outs[fuse] = out
    med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    d = float((outs[True].float() - outs[False].float()).norm() / outs[False].float().norm())
    print(f"decode 768x512x97: {knob} on {med[True]:.3f} ms | off (the norm as a launch of its own) {med[False]:.3f} ms "
          f"(x{med[False] / med[True]:.4f}); the two renderings differ by {d:.2e} relative L2", flush=True)
else:
    # --no-check (last argument): without the implicit-GEMM rendering bench.py compares the decode with (for profiler runs)
    nocheck = sys.argv[-1] == "--no-check"
    print(bench.time_vae("cuda", int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5, cross_check=not nocheck), flush=True)
