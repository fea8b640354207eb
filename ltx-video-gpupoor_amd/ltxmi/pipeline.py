"""Denoise loop of ``LTXVideoPipeline.__call__`` (text-/image-/video-to-video) with everything on device.

Mirrors ltx_video/pipelines/pipeline_ltx_video.py:919-1307 for the inputs the hot path sees:
pre-computed prompt embeddings (the T5 encoder is outside this path), optional conditioning
items (``ConditioningItem`` :203-219, ``prepare_conditioning`` :1344-1548), ``joint_pass=True``.
Per step (pipeline_ltx_video.py:1104-1256):

    latent_model_input = cat([latents] * num_conds)            (:1115)
    noise_pred = transformer(...)                               (:1160-1179)   <- libltxmi
    CFG-star / STG / std-rescale + scheduler.step (Euler)       (:1183-1241)   <- one fused
                                                                   ltxmi_guidance_step_bf16

The reference's per-step host work (``skip_layer_mask.min()`` per block, ``.item()`` calls,
Python float timesteps) is replaced by host-side schedule scalars computed once; no
host<->device synchronisation happens inside the loop.
"""
import copy
import math
from dataclasses import dataclass
from typing import List, Optional

import torch

from . import ops
from .attention import SkipLayerStrategy
from .autoencoder import vae_decode, vae_encode
from .patchifier import SymmetricPatchifier, latent_to_pixel_coords_from_factors


@dataclass
class ConditioningItem:
    """pipeline_ltx_video.py:203-219: a frame (f = 1) or frame sequence (f = 8k+1) to condition on."""
    media_item: torch.Tensor                 # (b, 3, f, h, w) in [-1, 1]
    media_frame_number: int
    conditioning_strength: float
    media_x: Optional[int] = None
    media_y: Optional[int] = None


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

    # ---- conditioning (pipeline_ltx_video.py:1344-1690) ---------------------------------------
    # Setup-time token assembly: slicing / lerp on small latent tensors, once per call (not per step);
    # the encoder it feeds from and everything inside the loop run on libltxmi kernels.
    @staticmethod
    def resize_tensor(media_items, height, width):                                       # :748-760
        if media_items.shape[-2:] != (height, width):
            raise NotImplementedError("ltxmi: provide conditioning media at the target size "
                                      "(the bilinear resize of :752-759 is host preprocessing outside this path)")
        return media_items

    @staticmethod
    def _resize_conditioning_item(item, height, width):                                  # :1550-1563
        if item.media_x or item.media_y:
            raise ValueError("Provide media_item in the target size for spatial conditioning.")
        new = copy.copy(item)
        new.media_item = LTXVideoPipeline.resize_tensor(item.media_item, height, width)
        return new

    def _get_latent_spatial_position(self, latents, item, height, width, strip_latent_border):   # :1566-1611
        scale = self.vae_scale_factor
        h, w = item.media_item.shape[-2:]
        assert h <= height and w <= width, f"Conditioning item size {h}x{w} is larger than target size {height}x{width}"
        assert h % scale == 0 and w % scale == 0
        x_start, y_start = item.media_x, item.media_y
        x_start = (width - w) // 2 if x_start is None else x_start
        y_start = (height - h) // 2 if y_start is None else y_start
        x_end, y_end = x_start + w, y_start + h
        assert x_end <= width and y_end <= height, \
            f"Conditioning item {x_start}:{x_end}x{y_start}:{y_end} is out of bounds for target size {width}x{height}"
        if strip_latent_border:
            if x_start > 0:
                x_start += scale
                latents = latents[:, :, :, :, 1:]
            if y_start > 0:
                y_start += scale
                latents = latents[:, :, :, 1:, :]
            if x_end < width:
                latents = latents[:, :, :, :, :-1]
            if y_end < height:
                latents = latents[:, :, :, :-1, :]
        return latents, x_start // scale, y_start // scale

    @staticmethod
    def _handle_non_first_conditioning_sequence(init_latents, init_conditioning_mask, latents, media_frame_number,
                                                strength, num_prefix_latent_frames=2, prefix_latents_mode="concat",
                                                prefix_soft_conditioning_strength=0.15):           # :1614-1690
        f_l = latents.shape[2]
        f_l_p = num_prefix_latent_frames
        assert f_l >= f_l_p
        assert media_frame_number % 8 == 0
        if f_l > f_l_p:
            s = media_frame_number // 8 + f_l_p
            e = s + f_l - f_l_p
            init_latents[:, :, s:e] = torch.lerp(init_latents[:, :, s:e], latents[:, :, f_l_p:], strength)
            init_conditioning_mask[:, s:e] = strength
        if prefix_latents_mode == "soft":
            if f_l_p > 1:
                s = media_frame_number // 8 + 1
                e = s + f_l_p - 1
                strength = min(prefix_soft_conditioning_strength, strength)
                init_latents[:, :, s:e] = torch.lerp(init_latents[:, :, s:e], latents[:, :, 1:f_l_p], strength)
                init_conditioning_mask[:, s:e] = strength
            latents = None
        elif prefix_latents_mode == "drop":
            latents = None
        elif prefix_latents_mode == "concat":
            latents = latents[:, :, :f_l_p]
        else:
            raise ValueError(f"Invalid prefix_latents_mode: {prefix_latents_mode}")
        return init_latents, init_conditioning_mask, latents

    def _pixel_coords(self, latent_coords, causal_fix=True):
        return latent_to_pixel_coords_from_factors(
            latent_coords, (self.video_scale_factor, self.vae_scale_factor, self.vae_scale_factor), causal_fix=causal_fix)

    def prepare_conditioning(self, conditioning_items, init_latents, num_frames, height, width,
                             vae_per_channel_normalize=False, generator=None, sample_posterior=True):
        """:1344-1548.  init_latents (b, c, f_l, h_l, w_l) -> (latents (b, N, c), pixel_coords (b, 3, N),
        conditioning_mask (b, N) fp32 or None, number of extra conditioning tokens in front)."""
        extra_latents, extra_coords, extra_mask, n_extra = [], [], [], 0
        causal_fix = bool(getattr(self.transformer.config, "causal_temporal_positioning", True))
        if conditioning_items:
            init_mask = torch.zeros(init_latents[:, 0].shape, dtype=torch.float32, device=init_latents.device)
            for item in conditioning_items:
                item = self._resize_conditioning_item(item, height, width)
                media, frame_no, strength = item.media_item, item.media_frame_number, item.conditioning_strength
                assert media.ndim == 5
                b, c, n_frames, h, w = media.shape
                assert (height == h and width == w) or frame_no == 0, \
                    f"Dimensions do not match: {height}x{width} != {h}x{w} - allowed only when media_frame_number == 0"
                assert n_frames % 8 == 1
                assert frame_no >= 0 and frame_no + n_frames <= num_frames
                lat = vae_encode(media.to(dtype=self.vae.dtype, device=self.vae.device), self.vae,
                                 vae_per_channel_normalize=vae_per_channel_normalize, generator=generator,
                                 sample_posterior=sample_posterior).to(dtype=init_latents.dtype)
                if frame_no == 0:
                    lat, l_x, l_y = self._get_latent_spatial_position(lat, item, height, width, strip_latent_border=True)
                    _, _, f_l, h_l, w_l = lat.shape
                    region = init_latents[:, :, :f_l, l_y:l_y + h_l, l_x:l_x + w_l]
                    init_latents[:, :, :f_l, l_y:l_y + h_l, l_x:l_x + w_l] = torch.lerp(region, lat, strength)
                    init_mask[:, :f_l, l_y:l_y + h_l, l_x:l_x + w_l] = strength
                else:
                    if n_frames > 1:
                        init_latents, init_mask, lat = self._handle_non_first_conditioning_sequence(
                            init_latents, init_mask, lat, frame_no, strength)
                    if lat is not None:
                        noise = torch.randn(lat.shape, generator=generator, device=lat.device, dtype=lat.dtype)
                        lat = torch.lerp(noise, lat, strength)
                        lat, coords = self.patchifier.patchify(lat)
                        pc = self._pixel_coords(coords, causal_fix)
                        pc[:, 0] += frame_no
                        n_extra += lat.shape[1]
                        extra_latents.append(lat)
                        extra_coords.append(pc)
                        extra_mask.append(torch.full(lat.shape[:2], strength, dtype=torch.float32,
                                                     device=init_latents.device))
        latents, coords = self.patchifier.patchify(init_latents)
        pixel_coords = self._pixel_coords(coords, causal_fix)
        if not conditioning_items:
            return latents, pixel_coords, None, 0
        mask, _ = self.patchifier.patchify(init_mask.unsqueeze(1))
        mask = mask.squeeze(-1)
        if extra_latents:
            latents = torch.cat([*extra_latents, latents], dim=1)
            pixel_coords = torch.cat([*extra_coords, pixel_coords], dim=2)
            mask = torch.cat([*extra_mask, mask], dim=1)
        return latents, pixel_coords, mask, n_extra

    @torch.no_grad()
    def __call__(self, height: int, width: int, num_frames: int, prompt_embeds, prompt_attention_mask,
                 negative_prompt_embeds=None, negative_prompt_attention_mask=None, frame_rate: float = 25.0,
                 num_inference_steps: int = 40, guidance_scale: float = 3.0, stg_scale: float = 1.0,
                 rescaling_scale: float = 0.7, skip_block_list: Optional[List[int]] = None,
                 skip_layer_strategy: Optional[SkipLayerStrategy] = SkipLayerStrategy.AttentionValues,
                 generator=None, latents=None, output_type: str = "latent", decode_timestep: float = 0.05,
                 decode_noise_scale: Optional[float] = 0.025, vae_per_channel_normalize: bool = True,
                 callback_on_step_end=None, latents_dtype=torch.float32,
                 conditioning_items: Optional[List[ConditioningItem]] = None, image_cond_noise_scale: float = 0.0,
                 sample_conditioning_posterior: bool = True):
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

        # conditioning items -> latents / coords / mask (+ extra tokens in front)           :1067-1085
        grid5 = self.patchifier.unpatchify(latents, latent_height, latent_width, C).contiguous()
        latents, pixel_coords, cond_mask, num_cond_latents = self.prepare_conditioning(
            conditioning_items, grid5, num_frames, height, width, vae_per_channel_normalize, generator,
            sample_posterior=sample_conditioning_posterior)
        latents = latents.contiguous()
        init_latents = latents.clone() if cond_mask is not None else None
        if cond_mask is not None:
            cond_mask = cond_mask.contiguous()
            one_minus_mask = (1.0 - cond_mask).expand(num_conds, -1) if num_conds > 1 else (1.0 - cond_mask)
        frac = pixel_coords.to(torch.float32)
        frac[:, 0] = frac[:, 0] * (1.0 / frame_rate)                                     # :1086-1087
        freqs_cis = tr.precompute_freqs_cis(frac)

        workspace = torch.zeros(8, dtype=torch.float32, device=device)
        t_dev = torch.tensor(timesteps, dtype=torch.float32, device=device)
        for i, t in enumerate(timesteps):
            if cond_mask is not None and image_cond_noise_scale > 0.0:                   # :1105-1113
                noise = torch.randn(latents.shape, generator=generator, device=device, dtype=latents.dtype)
                ops.image_cond_noise_(latents, init_latents, noise, cond_mask, image_cond_noise_scale, t)
            model_in = latents.to(tr.dtype)
            if num_conds > 1:
                model_in = model_in.expand(num_conds, -1, -1)
            current_timestep = t_dev[i].expand(num_conds).unsqueeze(-1)                  # [B_eff, 1]
            if cond_mask is not None:                                                    # :1145-1150, [B_eff, N]
                current_timestep = torch.minimum(current_timestep, one_minus_mask)
            noise_pred = tr(model_in, freqs_cis=freqs_cis, encoder_hidden_states=embeds,
                            encoder_attention_mask=mask, timestep=current_timestep,
                            skip_layer_mask=skip_mask, skip_layer_strategy=skip_layer_strategy,
                            latent_shape=latent_shape[2:], joint_pass=True, ltxv_model=self, return_dict=False)[0]
            if noise_pred is None:
                return None
            dt = self.scheduler.host_dt(t)
            ops.guidance_step_(noise_pred, latents, dt, guidance_scale, stg_scale, rescaling_scale,
                               do_cfg, do_stg, do_rescale, workspace, cond_mask=cond_mask, t=t)   # :1183-1241, 1309-1342
            if callback_on_step_end is not None:
                callback_on_step_end(self, i, t, {})

        latents = latents[:, num_cond_latents:]                                          # :1258-1259
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
