"""Oracle restatement of the denoise loop's host-side control (schedule slicing, per-step guidance
tables, initial-latent noising).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows ltx_video/pipelines/pipeline_ltx_video.py:
  retrieve_timesteps                  :125-198
  guidance tables in __call__         :959-1013  (guidance_timesteps -> guidance_mapping -> per-step lists)
  prepare_latents                     :632-710   (latents given: t0 * noise + (1 - t0) * latents)
"""
import torch

from . import sched


def retrieve_timesteps(num_inference_steps, samples_shape, timesteps=None, max_timestep=1.0,
                       skip_initial_inference_steps=0, skip_final_inference_steps=0):
    """Returns the schedule the loop iterates over (the scheduler is re-set to exactly this list, :196)."""
    if timesteps is not None:
        ts = torch.tensor(timesteps, dtype=torch.float32)
        num_inference_steps = len(ts)
    else:
        ts = sched.set_timesteps(num_inference_steps, samples_shape)
    if (skip_initial_inference_steps < 0 or skip_final_inference_steps < 0
            or skip_initial_inference_steps + skip_final_inference_steps >= num_inference_steps):
        raise ValueError("invalid skip inference step values")
    ts = ts[skip_initial_inference_steps: len(ts) - skip_final_inference_steps]
    if max_timestep < 1.0:
        if max_timestep < ts.min():
            raise ValueError(f"max_timestep {max_timestep} is smaller than the minimum timestep {ts.min()}")
        ts = ts[ts <= max_timestep]
    return ts


def guidance_tables(timesteps, guidance_scale, stg_scale, rescaling_scale, skip_block_list, guidance_timesteps=None):
    """:959-1013.  Returns per-step lists (guidance_scale zeroed where <= 1) and the three global flags."""
    n = len(timesteps)
    mapping = None
    if guidance_timesteps:
        mapping = []
        for t in timesteps:
            idx = [i for i, v in enumerate(guidance_timesteps) if v <= t]
            mapping.append(idx[0] if len(idx) > 0 else len(guidance_timesteps) - 1)

    def table(v):
        if not isinstance(v, list):
            return [v] * n
        return [v[mapping[i]] for i in range(n)]

    gs = [x if x > 1.0 else 0.0 for x in table(guidance_scale)]
    stg = table(stg_scale)
    rs = table(rescaling_scale)
    if skip_block_list is not None:
        if len(skip_block_list) == 0 or not isinstance(skip_block_list[0], list):
            skip_block_list = [skip_block_list] * n
        else:
            skip_block_list = [skip_block_list[mapping[i]] for i in range(n)]
    return gs, stg, rs, skip_block_list, any(x > 1.0 for x in gs), any(x > 0.0 for x in stg), any(x != 1.0 for x in rs)


def prepare_latents(latents, t0, noise_patchified, latent_shape, init_noise_sigma=1.0):
    """:632-710 without media items: noise drawn as (b, f*h*w, c), rearranged to (b, c, f, h, w)."""
    b, c, f, h, w = latent_shape
    noise = noise_patchified.reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3) * init_noise_sigma
    if latents is None:
        return noise
    return t0 * noise + (1 - t0) * latents
