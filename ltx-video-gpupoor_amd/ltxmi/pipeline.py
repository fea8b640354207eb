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
        if x_start + w > width or y_start + h > height:
            raise AssertionError(f"Conditioning item {x_start}:{x_start + w}x{y_start}:{y_start + h} is out of bounds for "
                                 f"target size {width}x{height}")
        if strip_latent_border:
            # one latent row / column is dropped on every side of the item that does not touch the frame's border
            # (pipeline_ltx_video.py:1598-1611); a cut on the left / top moves the item's origin by one latent
            cut_l, cut_t = int(x_start > 0), int(y_start > 0)
            cut_r, cut_b = int(x_start + w < width), int(y_start + h < height)
            hl, wl = latents.shape[-2], latents.shape[-1]
            latents = latents[..., cut_t:hl - cut_b, cut_l:wl - cut_r]
            x_start += cut_l * scale
            y_start += cut_t * scale
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

    @staticmethod
    def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, max_timestep=1.0,
                           skip_initial_inference_steps=0, skip_final_inference_steps=0, **kwargs):
        """pipeline_ltx_video.py:125-198: the scheduler's (or the given) schedule, minus skipped head/tail
        steps, truncated to ``max_timestep``; the scheduler is re-set to exactly the returned list."""
        if timesteps is not None:
            scheduler.set_timesteps(timesteps=timesteps, device=device, **kwargs)
        else:
            scheduler.set_timesteps(num_inference_steps, device=device, **kwargs)
        ts = list(scheduler.host_timesteps)
        n = len(ts)
        if (skip_initial_inference_steps < 0 or skip_final_inference_steps < 0
                or skip_initial_inference_steps + skip_final_inference_steps >= n):
            raise ValueError("invalid skip inference step values: must be non-negative and the sum of "
                             "skip_initial_inference_steps and skip_final_inference_steps must be less than the "
                             "number of inference steps")
        ts = ts[skip_initial_inference_steps: n - skip_final_inference_steps]
        if max_timestep < 1.0:
            if max_timestep < min(ts):
                raise ValueError(f"max_timestep {max_timestep} is smaller than the minimum timestep {min(ts)}")
            ts = [t for t in ts if t <= max_timestep]
        scheduler.set_timesteps(timesteps=ts, device=device, **kwargs)
        return list(scheduler.host_timesteps), len(ts)

    @staticmethod
    def _guidance_tables(timesteps, guidance_scale, stg_scale, rescaling_scale, skip_block_list, guidance_timesteps):
        """:959-1013: per-step guidance / STG / rescale / skip-block tables (lists are indexed through
        ``guidance_timesteps``; scalars are broadcast)."""
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
            if mapping is None:
                raise ValueError("list-valued guidance parameters need `guidance_timesteps`")
            return [v[mapping[i]] for i in range(n)]

        gs = [x if x > 1.0 else 0.0 for x in table(guidance_scale)]
        stg, rs = table(stg_scale), table(rescaling_scale)
        if skip_block_list is not None:
            if len(skip_block_list) == 0 or not isinstance(skip_block_list[0], list):
                skip_block_list = [skip_block_list] * n
            else:
                skip_block_list = [skip_block_list[mapping[i]] for i in range(n)]
        return gs, stg, rs, skip_block_list

    @torch.no_grad()
    def __call__(self, height: int, width: int, num_frames: int, prompt_embeds, prompt_attention_mask,
                 negative_prompt_embeds=None, negative_prompt_attention_mask=None, frame_rate: float = 25.0,
                 num_inference_steps: int = 40, guidance_scale=3.0, stg_scale=1.0, rescaling_scale=0.7,
                 skip_block_list=None,
                 skip_layer_strategy: Optional[SkipLayerStrategy] = SkipLayerStrategy.AttentionValues,
                 generator=None, latents=None, output_type: str = "latent", decode_timestep: float = 0.05,
                 decode_noise_scale: Optional[float] = 0.025, vae_per_channel_normalize: bool = True,
                 callback_on_step_end=None, latents_dtype=torch.float32,
                 conditioning_items: Optional[List[ConditioningItem]] = None, image_cond_noise_scale: float = 0.0,
                 sample_conditioning_posterior: bool = True, timesteps: Optional[List[float]] = None,
                 guidance_timesteps: Optional[List[float]] = None, skip_initial_inference_steps: int = 0,
                 skip_final_inference_steps: int = 0, strength: float = 1.0, joint_pass: bool = True,
                 stg_row_dedup: bool = True, dead_row_elimination: bool = True):
        """``latents``: (b, c, f, h, w) as in the reference -- re-noised to the first timestep
        (t0 * noise + (1 - t0) * latents, :688-707) -- or, as an extension for tests, (b, N, c) patchified
        initial noise used as is.  ``stg_row_dedup``: the STG "perturbed" row has the text row's inputs, so it
        is the text row until the step's first skipped block; those blocks run on one row less and the row is
        filled in by a copy (bit-identical results, see Transformer3DModel.forward)."""
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

        assert strength == 1.0 or latents is not None, \
            "strength < 1 is used for image-to-image/video-to-video - media_item or latents should be provided."
        timesteps, num_inference_steps = self.retrieve_timesteps(                       # :943-952
            self.scheduler, None if timesteps is not None else num_inference_steps, device, timesteps,
            max_timestep=strength, skip_initial_inference_steps=skip_initial_inference_steps,
            skip_final_inference_steps=skip_final_inference_steps, samples_shape=latent_shape)

        gs_tab, stg_tab, rs_tab, skip_tab = self._guidance_tables(                      # :959-1013
            timesteps, guidance_scale, stg_scale, rescaling_scale, skip_block_list, guidance_timesteps)
        do_cfg = any(x > 1.0 for x in gs_tab)
        do_stg = any(x > 0.0 for x in stg_tab)
        do_rescale = any(x != 1.0 for x in rs_tab)

        # Rows of the batch per step.  The reference keeps num_conds constant and zeroes the scales of the
        # steps that should not use a guidance (:980-983); a row whose scale is zero at a step does not reach
        # that step's result (:1183-1222), so it is not computed here (``dead_row_elimination``; bit-identical:
        # every kernel computes a row independently of the others).
        def rows_for(i):
            if not dead_row_elimination:
                return do_cfg, do_stg
            return (do_cfg and gs_tab[i] > 1.0), (do_stg and stg_tab[i] > 0.0)

        batches = {}                             # (use_cfg, use_stg) -> (embeds, mask, num_conds)  :1035-1051

        def batch_for(use_cfg, use_stg):
            key = (use_cfg, use_stg)
            if key not in batches:
                e, m = prompt_embeds, prompt_attention_mask
                if use_cfg:
                    e = torch.cat([negative_prompt_embeds, e], dim=0)
                    m = torch.cat([negative_prompt_attention_mask, m], dim=0)
                if use_stg:
                    e = torch.cat([e, prompt_embeds], dim=0)
                    m = torch.cat([m, prompt_attention_mask], dim=0)
                batches[key] = (e.to(device=device, dtype=tr.dtype), m.to(device), 1 + int(use_cfg) + int(use_stg))
            return batches[key]

        mask_cache = {}

        def skip_mask_for(i, use_stg, nconds):                                           # :1016-1026
            if not use_stg or skip_tab is None:
                return None
            key = (tuple(skip_tab[i]), nconds)
            if key not in mask_cache:
                mask_cache[key] = tr.create_skip_layer_mask(batch_size, nconds, nconds - 1, list(skip_tab[i]))
            return mask_cache[key]

        if latents is not None and latents.dim() == 3:                                   # test hook: given noise
            latents = latents.to(device=device, dtype=latents_dtype).clone()
        else:                                                                            # prepare_latents :632-710
            noise = self.prepare_latents(latent_shape, latents_dtype, device, generator)
            if latents is not None:
                assert tuple(latents.shape) == latent_shape, \
                    f"Latents have to be of shape {latent_shape} but are {tuple(latents.shape)}."
                assert timesteps[0] < 1.0, \
                    "Input media_item or latents are provided, but they will be replaced with noise."    # :679-681
                given, _ = self.patchifier.patchify(latents.to(device=device, dtype=latents_dtype))
                noise = timesteps[0] * noise + (1 - timesteps[0]) * given
            latents = noise

        # conditioning items -> latents / coords / mask (+ extra tokens in front)           :1067-1085
        grid5 = self.patchifier.unpatchify(latents, latent_height, latent_width, C).contiguous()
        latents, pixel_coords, cond_mask, num_cond_latents = self.prepare_conditioning(
            conditioning_items, grid5, num_frames, height, width, vae_per_channel_normalize, generator,
            sample_posterior=sample_conditioning_posterior)
        latents = latents.contiguous()
        init_latents = latents.clone() if cond_mask is not None else None
        if cond_mask is not None:
            cond_mask = cond_mask.contiguous()
            one_minus_mask = 1.0 - cond_mask
        frac = pixel_coords.to(torch.float32)
        frac[:, 0] = frac[:, 0] * (1.0 / frame_rate)                                     # :1086-1087
        freqs_cis = tr.precompute_freqs_cis(frac)

        workspace = torch.empty(ops.GUIDANCE_WORKSPACE_FLOATS, dtype=torch.float32, device=device)
        t_dev = torch.tensor(timesteps, dtype=torch.float32, device=device)
        for i, t in enumerate(timesteps):
            if cond_mask is not None and image_cond_noise_scale > 0.0:                   # :1105-1113
                noise = torch.randn(latents.shape, generator=generator, device=device, dtype=latents.dtype)
                ops.image_cond_noise_(latents, init_latents, noise, cond_mask, image_cond_noise_scale, t)
            use_cfg, use_stg = rows_for(i)
            embeds, mask, nconds = batch_for(use_cfg, use_stg)
            model_in = latents.to(tr.dtype)
            if nconds > 1:
                model_in = model_in.expand(nconds, -1, -1)
            current_timestep = t_dev[i].expand(nconds).unsqueeze(-1)                     # [B_eff, 1]
            if cond_mask is not None:                                                    # :1145-1150, [B_eff, N]
                current_timestep = torch.minimum(current_timestep, one_minus_mask.expand(nconds, -1))
            alias = 0
            if stg_row_dedup and use_stg and joint_pass:
                blocks = skip_tab[i] if skip_tab is not None else []
                alias = min(blocks) if len(blocks) > 0 else len(tr.transformer_blocks)
            noise_pred = tr(model_in, freqs_cis=freqs_cis, encoder_hidden_states=embeds,
                            encoder_attention_mask=mask, timestep=current_timestep, stg_alias_blocks=alias,
                            skip_layer_mask=skip_mask_for(i, use_stg, nconds),
                            skip_layer_strategy=skip_layer_strategy,
                            latent_shape=latent_shape[2:], joint_pass=joint_pass, ltxv_model=self, return_dict=False)[0]
            if noise_pred is None:
                return None
            dt = self.scheduler.host_dt(t)
            ops.guidance_step_(noise_pred, latents, dt, gs_tab[i], stg_tab[i], rs_tab[i],
                               use_cfg, use_stg, do_rescale, workspace, cond_mask=cond_mask, t=t)   # :1183-1241, 1309-1342
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


class LTXMultiScalePipeline:
    """pipeline_ltx_video.py:1741-1905: pass 1 at the downscaled size -> latent upsampler (x2) ->
    AdaIN against the pass-1 latents -> pass 2 from the re-noised upsampled latents.  Prompt embeddings
    are inputs (the T5 encoder is outside this path); the final pixel-space bilinear resize to the
    requested size (:1893-1903) is left to the caller -- this returns pass 2's output as is."""

    def __init__(self, video_pipeline: LTXVideoPipeline, latent_upsampler):
        self.video_pipeline = video_pipeline
        self.vae = video_pipeline.vae
        self.latent_upsampler = latent_upsampler

    def _upsample_latents(self, latent_upsampler, latents):                              # :1760-1772
        from .latent_upsampler import upsample_latents
        return upsample_latents(latent_upsampler, latents, self.vae)

    def __call__(self, downscale_factor: float, first_pass: dict, second_pass: dict, **kwargs):
        from .latent_upsampler import adain_filter_latent
        vp = self.video_pipeline
        original_output_type = kwargs.get("output_type", "latent")
        x_width = int(kwargs["width"] * downscale_factor)                                # :1797-1800
        downscaled_width = x_width - (x_width % vp.vae_scale_factor)
        x_height = int(kwargs["height"] * downscale_factor)
        downscaled_height = x_height - (x_height % vp.vae_scale_factor)
        original_kwargs = dict(kwargs)

        kw = dict(original_kwargs, output_type="latent", width=downscaled_width, height=downscaled_height,
                  joint_pass=True)
        kw.update(first_pass)
        if "num_inference_steps1" in kw:
            kw["num_inference_steps"] = kw.pop("num_inference_steps1")
        kw.pop("num_inference_steps2", None)
        latents = vp(**kw)
        if latents is None:
            return None

        upsampled = self._upsample_latents(self.latent_upsampler, latents)               # :1867-1872
        upsampled = adain_filter_latent(latents=upsampled, reference_latents=latents)

        kw = dict(original_kwargs, latents=upsampled, output_type=original_output_type,
                  width=downscaled_width * 2, height=downscaled_height * 2, joint_pass=False)
        kw.update(second_pass)
        if "num_inference_steps2" in kw:
            kw["num_inference_steps"] = kw.pop("num_inference_steps2")
        kw.pop("num_inference_steps1", None)
        return vp(**kw)
