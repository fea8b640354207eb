"""Oracle restatement of the scheduler, patchifier and per-step guidance math.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows:
  RectifiedFlowScheduler (Uniform sampler, SD3 shifting)  ltx_video/schedulers/rf.py:69-149, 201-261, 311-392
  SymmetricPatchifier                                      ltx_video/models/transformers/symmetric_patchifier.py:33-84
  latent_to_pixel_coords_from_factors                      ltx_video/models/autoencoders/vae_encode.py:214-225
  guidance math (CFG-star, STG, std-rescale)               ltx_video/pipelines/pipeline_ltx_video.py:1183-1222  [parity unpinned]
  denoising_step                                           pipeline_ltx_video.py:1309-1342
"""
import math

import torch


# ------------------------------------------------------------------ scheduler
def sd3_shift(samples_shape, timesteps, target_shift_terminal=None):
    """sd3_resolution_dependent_timestep_shift, rf.py:112-149 (+ :69-109)."""
    if len(samples_shape) == 3:
        m = samples_shape[1]
    else:
        m = math.prod(samples_shape[2:])
    slope = (2.05 - 0.95) / (4096 - 1024)
    mu = slope * m + (0.95 - slope * 1024)
    ts = math.exp(mu) / (math.exp(mu) + (1 / timesteps - 1) ** 1)
    if target_shift_terminal is not None:
        one_minus = 1 - ts
        ts = 1 - one_minus / (one_minus[-1] / (1 - target_shift_terminal))
    return ts


def set_timesteps(num_inference_steps, samples_shape, shifting="SD3", target_shift_terminal=0.1,
                  num_train_timesteps=1000):
    """RectifiedFlowScheduler.set_timesteps for sampler="Uniform", rf.py:201-205, 227-261."""
    n = min(num_train_timesteps, num_inference_steps)
    ts = torch.linspace(1, 1 / n, n)
    if shifting == "SD3":
        ts = sd3_shift(samples_shape, ts, target_shift_terminal)
    return ts


def scheduler_step(timesteps, model_output, timestep, sample, stochastic_noise=None):
    """RectifiedFlowScheduler.step, rf.py:344-380.  ``stochastic_noise``: the torch.randn_like(sample) draw of the
    stochastic_sampling branch (:368-373, per-token timestep form), or None for the deterministic Euler step."""
    t_eps = 1e-6
    padded = torch.cat([timesteps, torch.zeros(1)])
    if timestep.ndim == 0:
        lower = padded[padded < timestep - t_eps][0]
        dt = timestep - lower
    else:
        assert timestep.ndim == 2
        mask = padded[:, None, None] < timestep[None] - t_eps
        lower, _ = (mask * padded[:, None, None]).max(dim=0)
        dt = (timestep - lower)[..., None]
    if stochastic_noise is not None:
        x0 = sample - timestep[..., None] * model_output
        sigma = timestep[..., None] - dt                                   # add_noise (:382-392): alphas = 1 - sigmas
        return (1 - sigma) * x0 + sigma * stochastic_noise
    return sample - dt * model_output


def denoising_step(timesteps, latents, noise_pred, current_timestep, conditioning_mask, t, t_eps=1e-6):
    """LTXVideoPipeline.denoising_step, pipeline_ltx_video.py:1309-1342."""
    den = scheduler_step(timesteps, noise_pred, t if current_timestep is None else current_timestep, latents)
    if conditioning_mask is None:
        return den
    mask = (t - t_eps < (1.0 - conditioning_mask)).unsqueeze(-1)
    return torch.where(mask, den, latents)


# ------------------------------------------------------------------- guidance
def guidance(noise_pred, num_conds, guidance_scale, stg_scale, rescaling_scale,
             do_cfg, do_stg, do_rescaling, cfg_star_rescale=True):
    """pipeline_ltx_video.py:1183-1222.  ``noise_pred`` is [num_conds*B, N, C] ordered
    (uncond, text, text_perturbed) as built at :1035-1051.  Pinned by golden G7 (the reference's own __call__) at
    B = 1; for B > 1 the reference's ``alpha * noise_pred_uncond`` ([B,1] x [B,N,C], :1199) does not broadcast per
    sample -- the per-sample form below is this restatement's reading, identical at B = 1."""
    chunks = noise_pred.chunk(num_conds)
    if do_stg:
        text, perturb = chunks[-2:]
    batch_size = chunks[0].shape[0]
    if do_cfg and guidance_scale != 0 and guidance_scale != 1:
        uncond, text = chunks[:2]
        if cfg_star_rescale:
            pos = text.reshape(batch_size, -1)
            neg = uncond.reshape(batch_size, -1)
            dot = torch.sum(pos * neg, dim=1, keepdim=True)
            sq = torch.sum(neg ** 2, dim=1, keepdim=True) + 1e-8
            alpha = dot / sq
            uncond = alpha.view(batch_size, 1, 1) * uncond if uncond.ndim == 3 else alpha * uncond
        out = uncond + guidance_scale * (text - uncond)
    elif do_stg:
        out = text
    else:
        out = chunks[0]
    if do_stg:
        out = out + stg_scale * (text - perturb)
        if do_rescaling and stg_scale > 0.0:
            s_text = text.reshape(batch_size, -1).std(dim=1, keepdim=True)
            s_out = out.reshape(batch_size, -1).std(dim=1, keepdim=True)
            factor = rescaling_scale * (s_text / s_out) + (1 - rescaling_scale)
            out = out * factor.view(batch_size, 1, 1)
    return out


# ----------------------------------------------------------------- patchifier
def get_latent_coords(f, h, w, batch_size):
    """Patchifier.get_latent_coords with patch_size 1, symmetric_patchifier.py:33-51."""
    grid = torch.meshgrid(torch.arange(f), torch.arange(h), torch.arange(w), indexing="ij")
    coords = torch.stack(grid, dim=0).unsqueeze(0).repeat(batch_size, 1, 1, 1, 1)
    return coords.reshape(batch_size, 3, -1)


def patchify(latents):
    """SymmetricPatchifier.patchify (patch 1): b c f h w -> b (f h w) c, :55-65."""
    b, c, f, h, w = latents.shape
    return latents.permute(0, 2, 3, 4, 1).reshape(b, f * h * w, c), get_latent_coords(f, h, w, b)


def unpatchify(latents, f, h, w):
    """SymmetricPatchifier.unpatchify (patch 1): b (f h w) c -> b c f h w, :67-84."""
    b, n, c = latents.shape
    return latents.reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3)


def latent_to_pixel_coords(latent_coords, scale_factors=(8, 32, 32), causal_fix=True):
    """vae_encode.py:214-225."""
    pc = latent_coords * torch.tensor(scale_factors)[None, :, None]
    if causal_fix:
        pc[:, 0] = (pc[:, 0] + 1 - scale_factors[0]).clamp(min=0)
    return pc


def fractional_coords(f, h, w, batch_size, frame_rate=25.0):
    """pipeline_ltx_video.py:1086-1088: seconds on the time axis, pixels on y/x."""
    pc = latent_to_pixel_coords(get_latent_coords(f, h, w, batch_size)).to(torch.float32)
    pc[:, 0] = pc[:, 0] * (1.0 / frame_rate)
    return pc
