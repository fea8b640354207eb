#!/usr/bin/env python3
"""One GPU, the bench's denoise step (B_eff 3): the block loop as one batch against the same loop over micro-batches of batch
rows on side streams (Transformer3DModel.forward(_microbatches=...), the machinery of the Ulysses mode) -- does running the
conditions' kernels concurrently fill the tile rounds the batched launches leave empty?  Diagnostic."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402

r = bench.StepRunner(torch.device("cuda", 0))
NC = bench.NUM_CONDS


@torch.no_grad()
def step(mb):
    x = r.latents.to(torch.bfloat16).expand(NC, -1, -1)
    out = r.m(x, freqs_cis=r.freqs, encoder_hidden_states=r.embeds, encoder_attention_mask=r.mask, timestep=r.t_dev,
              skip_layer_mask=r.skip, skip_layer_strategy=r.ltxmi.SkipLayerStrategy.AttentionValues, latent_shape=r.grid,
              ltxv_model=r.holder, return_dict=False, _microbatches=mb)[0]
    return out


forms = {"one batch of 3": None, "rows [0,1] + [2]": [slice(0, 2), slice(2, 3)], "rows [0] + [1,2]": [slice(0, 1), slice(1, 3)],
         "rows [0] + [1] + [2]": [slice(0, 1), slice(1, 2), slice(2, 3)]}
ref = step(None).clone()
times = {k: [] for k in forms}
for rep in range(4):
    for name, mb in forms.items():
        step(mb)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            out = step(mb)
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 5)
        if rep == 0:
            print(f"{name}: bit-identical to the batched forward: {torch.equal(out, ref)}", flush=True)
for name, ts in times.items():
    print(f"{name:24s} {sorted(ts)[len(ts) // 2]:.3f} ms per forward  (all: {' '.join(f'{t:.2f}' for t in ts)})", flush=True)
