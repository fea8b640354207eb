#!/usr/bin/env python3
"""Does a HIP graph of one denoise step pay?  Captures bench.StepRunner.step() (model forward + fused guidance/Euler: ~460
kernel launches issued from Python through ctypes) with torch.cuda.graph and times replays against eager launches."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402
from ltxmi import ops  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ops.set_step_invariant_caching(False)
r = bench.StepRunner(dev)
for _ in range(3):
    r.step()
torch.cuda.synchronize()


def timed(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager = timed(r.step)
lat0 = r.latents.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    r.step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
r.latents.copy_(lat0)
with torch.cuda.graph(g):
    r.step()
r.latents.copy_(lat0)
g.replay()
torch.cuda.synchronize()
a = r.latents.clone()
r.latents.copy_(lat0)
r.step()
torch.cuda.synchronize()
print("graph replay == eager step:", torch.equal(a, r.latents))
graph = timed(g.replay)
eager2 = timed(r.step)
print(f"eager {eager:.3f} / {eager2:.3f} ms per step, graph replay {graph:.3f} ms per step ({eager2 / graph:.4f}x)")
