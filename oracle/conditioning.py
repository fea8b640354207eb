"""Oracle restatement of the reference's conditioning-token assembly (image-/video-to-video).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows ltx_video/pipelines/pipeline_ltx_video.py:
  ConditioningItem                              :203-219
  add_noise_to_image_conditioning_latents       :606-629
  per-token timestep in the loop                :1145-1150   (min(t, 1 - conditioning_mask))
  prepare_conditioning                          :1344-1548
  _get_latent_spatial_position                  :1566-1611
  _handle_non_first_conditioning_sequence       :1614-1690
``encode`` is the caller's ``vae_encode`` (media [b,3,f,h,w] -> normalised latents [b,C,f_l,h_l,w_l]);
``noise_fn(shape)`` supplies the randn draws in call order (the reference uses diffusers' randn_tensor).
"""
from dataclasses import dataclass
from typing import Optional

import torch

from . import sched


@dataclass
class ConditioningItem:
    media_item: torch.Tensor
    media_frame_number: int
    conditioning_strength: float
    media_x: Optional[int] = None
    media_y: Optional[int] = None


def add_noise_to_image_conditioning_latents(t, init_latents, latents, noise_scale, conditioning_mask, noise, eps=1e-6):
    need = (conditioning_mask > 1.0 - eps).unsqueeze(-1)
    return torch.where(need, init_latents + noise_scale * noise * (t ** 2), latents)


def per_token_timestep(t, conditioning_mask, num_conds):
    """:1139-1150: [num_conds, N] per-token timesteps."""
    cur = torch.as_tensor(t, dtype=torch.float32).reshape(1).expand(num_conds).unsqueeze(-1)
    return torch.min(cur, 1.0 - torch.cat([conditioning_mask] * num_conds))


def get_latent_spatial_position(latents, item, height, width, strip_latent_border, scale=32):
    h, w = item.media_item.shape[-2:]
    assert h <= height and w <= width and h % scale == 0 and w % scale == 0
    x_start, y_start = item.media_x, item.media_y
    x_start = (width - w) // 2 if x_start is None else x_start
    y_start = (height - h) // 2 if y_start is None else y_start
    x_end, y_end = x_start + w, y_start + h
    assert x_end <= width and y_end <= height
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


def handle_non_first_conditioning_sequence(init_latents, init_mask, latents, media_frame_number, strength,
                                           num_prefix_latent_frames=2, prefix_latents_mode="concat",
                                           prefix_soft_conditioning_strength=0.15):
    f_l = latents.shape[2]
    f_l_p = num_prefix_latent_frames
    assert f_l >= f_l_p and media_frame_number % 8 == 0
    if f_l > f_l_p:
        s = media_frame_number // 8 + f_l_p
        e = s + f_l - f_l_p
        init_latents[:, :, s:e] = torch.lerp(init_latents[:, :, s:e], latents[:, :, f_l_p:], strength)
        init_mask[:, s:e] = strength
    if prefix_latents_mode == "soft":
        if f_l_p > 1:
            s = media_frame_number // 8 + 1
            e = s + f_l_p - 1
            strength = min(prefix_soft_conditioning_strength, strength)
            init_latents[:, :, s:e] = torch.lerp(init_latents[:, :, s:e], latents[:, :, 1:f_l_p], strength)
            init_mask[:, s:e] = strength
        latents = None
    elif prefix_latents_mode == "drop":
        latents = None
    elif prefix_latents_mode == "concat":
        latents = latents[:, :, :f_l_p]
    else:
        raise ValueError(f"Invalid prefix_latents_mode: {prefix_latents_mode}")
    return init_latents, init_mask, latents


def prepare_conditioning(items, init_latents, num_frames, height, width, encode, noise_fn, causal_fix=True,
                         scale_factors=(8, 32, 32)):
    """Returns (latents [b,N,C], pixel_coords [b,3,N], conditioning_mask [b,N] or None, num_extra_tokens).
    ``init_latents`` [b,C,f,h,w] is updated in place like the reference's."""
    extra_lat, extra_pc, extra_mask, n_extra = [], [], [], 0
    if items:
        init_mask = torch.zeros(init_latents[:, 0].shape, dtype=torch.float32)
        for item in items:
            if item.media_x or item.media_y:                    # _resize_conditioning_item :1551-1563
                raise ValueError("Provide media_item in the target size for spatial conditioning.")
            media = item.media_item
            assert media.shape[-2:] == (height, width), "oracle: resize (bilinear) is outside the restated path"
            frame, strength = item.media_frame_number, item.conditioning_strength
            b, c, n_frames, h, w = media.shape
            assert n_frames % 8 == 1 and frame >= 0 and frame + n_frames <= num_frames
            lat = encode(media).to(init_latents.dtype)
            if frame == 0:
                lat, l_x, l_y = get_latent_spatial_position(lat, item, height, width, True, scale_factors[1])
                _, _, f_l, h_l, w_l = lat.shape
                reg = init_latents[:, :, :f_l, l_y:l_y + h_l, l_x:l_x + w_l]
                init_latents[:, :, :f_l, l_y:l_y + h_l, l_x:l_x + w_l] = torch.lerp(reg, lat, strength)
                init_mask[:, :f_l, l_y:l_y + h_l, l_x:l_x + w_l] = strength
            else:
                if n_frames > 1:
                    init_latents, init_mask, lat = handle_non_first_conditioning_sequence(
                        init_latents, init_mask, lat, frame, strength)
                if lat is not None:
                    lat = torch.lerp(noise_fn(lat.shape), lat, strength)
                    lat, coords = sched.patchify(lat)
                    pc = sched.latent_to_pixel_coords(coords, scale_factors, causal_fix)
                    pc[:, 0] += frame
                    n_extra += lat.shape[1]
                    extra_lat.append(lat)
                    extra_pc.append(pc)
                    extra_mask.append(torch.full(lat.shape[:2], strength, dtype=torch.float32))
    latents, coords = sched.patchify(init_latents)
    pixel_coords = sched.latent_to_pixel_coords(coords, scale_factors, causal_fix)
    if not items:
        return latents, pixel_coords, None, 0
    mask, _ = sched.patchify(init_mask.unsqueeze(1))
    mask = mask.squeeze(-1)
    if extra_lat:
        latents = torch.cat([*extra_lat, latents], dim=1)
        pixel_coords = torch.cat([*extra_pc, pixel_coords], dim=2)
        mask = torch.cat([*extra_mask, mask], dim=1)
    return latents, pixel_coords, mask, n_extra
