"""Denoise loop of ``LTXVideoPipeline.__call__`` (text-to-video path) with everything on device.

Mirrors ltx_video/pipelines/pipeline_ltx_video.py:919-1307 for the inputs the hot path sees:
pre-computed prompt embeddings (the T5 encoder is outside this path), no conditioning items,
``joint_pass=True``.  Per step (pipeline_ltx_video.py:1104-1256):

    latent_model_input = cat([latents] * num_conds)            (:1115)
    noise_pred = transformer(...)                               (:1160-1179)   <- libltxmi
    CFG-star / STG / std-rescale + scheduler.step (Euler)       (:1183-1241)   <- one fused
                                                                   ltxmi_guidance_step_bf16

The reference's per-step host work (``skip_layer_mask.min()`` per block, ``.item()`` calls,
Python float timesteps) is replaced by host-side schedule scalars computed once; no
host<->device synchronisation happens inside the loop.
"""
import math
from typing import List, Optional

import torch

from . import ops
from .attention import SkipLayerStrategy
from .autoencoder import vae_decode
from .patchifier import SymmetricPatchifier, latent_to_pixel_coords_from_factors


class LTXVideoPipeline:
    def __init__(self, transformer, scheduler, vae=None, patchifier=None):
        self.transformer = transformer
        self.scheduler = scheduler
        self.vae = vae
        self.patchifier = patchifier or SymmetricPatchifier(patch_size=1)
        self.vae_scale_factor = 32
        self.video_scale_factor = 8
        self._interrupt = False

    def prepare_latents(self, latent_shape, dtype, device, generator=None):
        """pipeline_ltx_video.py:632-710 (no media items): noise is drawn in PATCHIFIED order
        (b, f*h*w, c) (:696-699) and scaled by init_noise_sigma."""
        b, c, f, h, w = latent_shape
        noise = torch.randn((b, f * h * w, c), generator=generator, device=device, dtype=dtype)
        return noise * self.scheduler.init_noise_sigma

    @torch.no_grad()
    def __call__(self, height: int, width: int, num_frames: int, prompt_embeds, prompt_attention_mask,
                 negative_prompt_embeds=None, negative_prompt_attention_mask=None, frame_rate: float = 25.0,
                 num_inference_steps: int = 40, guidance_scale: float = 3.0, stg_scale: float = 1.0,
                 rescaling_scale: float = 0.7, skip_block_list: Optional[List[int]] = None,
                 skip_layer_strategy: Optional[SkipLayerStrategy] = SkipLayerStrategy.AttentionValues,
                 generator=None, latents=None, output_type: str = "latent", decode_timestep: float = 0.05,
                 decode_noise_scale: Optional[float] = 0.025, vae_per_channel_normalize: bool = True,
                 callback_on_step_end=None, latents_dtype=torch.float32):
        tr = self.transformer
        device = tr.device
        batch_size = prompt_embeds.shape[0]
        if batch_size != 1:
            raise NotImplementedError("one prompt per call on this path (the reference's CFG-star "
                                      "broadcast at pipeline_ltx_video.py:1199 is only well-formed for batch 1)")
        latent_height = height // self.vae_scale_factor
        latent_width = width // self.vae_scale_factor
        latent_num_frames = num_frames // self.video_scale_factor + 1                    # :921-923
        C = tr.config.in_channels
        latent_shape = (batch_size, C, latent_num_frames, latent_height, latent_width)

        self.scheduler.set_timesteps(num_inference_steps, samples_shape=latent_shape, device=device)   # :943-952
        timesteps = self.scheduler.host_timesteps

        guidance_scale = guidance_scale if guidance_scale > 1.0 else 0.0                # :980
        do_cfg = guidance_scale > 1.0
        do_stg = stg_scale > 0.0
        do_rescale = rescaling_scale != 1.0
        num_conds = 1 + int(do_cfg) + int(do_stg)

        skip_mask = None
        if do_stg and skip_block_list:
            skip_mask = tr.create_skip_layer_mask(batch_size, num_conds, num_conds - 1, skip_block_list)   # :1021-1026

        embeds, mask = prompt_embeds, prompt_attention_mask                              # :1035-1051
        if do_cfg:
            embeds = torch.cat([negative_prompt_embeds, embeds], dim=0)
            mask = torch.cat([negative_prompt_attention_mask, mask], dim=0)
        if do_stg:
            embeds = torch.cat([embeds, prompt_embeds], dim=0)
            mask = torch.cat([mask, prompt_attention_mask], dim=0)
        embeds = embeds.to(device=device, dtype=tr.dtype)
        mask = mask.to(device)

        if latents is None:
            latents = self.prepare_latents(latent_shape, latents_dtype, device, generator)
        else:
            latents = latents.to(device=device, dtype=latents_dtype).clone()

        coords = self.patchifier.get_latent_coords(latent_num_frames, latent_height, latent_width, batch_size, device)
        pixel_coords = latent_to_pixel_coords_from_factors(
            coords, (self.video_scale_factor, self.vae_scale_factor, self.vae_scale_factor), causal_fix=True)
        frac = pixel_coords.to(torch.float32)
        frac[:, 0] = frac[:, 0] * (1.0 / frame_rate)                                     # :1086-1087
        freqs_cis = tr.precompute_freqs_cis(frac)

        workspace = torch.zeros(8, dtype=torch.float32, device=device)
        t_dev = torch.tensor(timesteps, dtype=torch.float32, device=device)
        for i, t in enumerate(timesteps):
            model_in = latents.to(tr.dtype)
            if num_conds > 1:
                model_in = model_in.expand(num_conds, -1, -1)
            current_timestep = t_dev[i].expand(num_conds).unsqueeze(-1)                  # [B_eff, 1]
            noise_pred = tr(model_in, freqs_cis=freqs_cis, encoder_hidden_states=embeds,
                            encoder_attention_mask=mask, timestep=current_timestep,
                            skip_layer_mask=skip_mask, skip_layer_strategy=skip_layer_strategy,
                            latent_shape=latent_shape[2:], joint_pass=True, ltxv_model=self, return_dict=False)[0]
            if noise_pred is None:
                return None
            dt = self.scheduler.host_dt(t)
            ops.guidance_step_(noise_pred, latents, dt, guidance_scale, stg_scale, rescaling_scale,
                               do_cfg, do_stg, do_rescale, workspace)
            if callback_on_step_end is not None:
                callback_on_step_end(self, i, t, {})

        latents = self.patchifier.unpatchify(latents, latent_height, latent_width, C)    # :1262-1268
        if output_type == "latent":
            return latents
        ts = None
        if self.vae.decoder.timestep_conditioning:                                       # :1270-1286
            noise = torch.randn(latents.shape, generator=generator, device=device, dtype=latents.dtype)
            s = decode_timestep if decode_noise_scale is None else decode_noise_scale
            latents = latents * (1 - s) + noise * s
            ts = torch.tensor([decode_timestep] * latents.shape[0], device=device)
        return vae_decode(latents.to(self.vae.dtype), self.vae, True,
                          vae_per_channel_normalize=vae_per_channel_normalize, timestep=ts)
