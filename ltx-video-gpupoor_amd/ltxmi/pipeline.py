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
from typing import Any, Callable, Dict, List, Optional, Union

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


class LTXVideoPipeline:
    """Same constructor keywords, attributes and ``__call__`` signature as the reference's pipeline
    (pipeline_ltx_video.py:222-304, 762-807), so ``ltxv.py:420-445`` calls through unchanged.  The text encoder
    is outside this path: ``tokenizer`` / ``text_encoder`` are whatever the caller has (the HF T5 objects the
    reference loads, ``ltxv.py:186-192``) and are only touched by ``encode_prompt``; the prompt enhancer models
    are accepted and never used (``enhance_prompt=True`` raises)."""

    def __init__(self, tokenizer=None, text_encoder=None, vae=None, transformer=None, scheduler=None, patchifier=None,
                 prompt_enhancer_image_caption_model=None, prompt_enhancer_image_caption_processor=None,
                 prompt_enhancer_llm_model=None, prompt_enhancer_llm_tokenizer=None,
                 allowed_inference_steps: Optional[List[float]] = None):
        self.tokenizer = tokenizer
        self.text_encoder = text_encoder
        self.vae = vae
        self.transformer = transformer
        self.scheduler = scheduler
        self.patchifier = patchifier or SymmetricPatchifier(patch_size=1)
        self.prompt_enhancer_image_caption_model = prompt_enhancer_image_caption_model
        self.prompt_enhancer_image_caption_processor = prompt_enhancer_image_caption_processor
        self.prompt_enhancer_llm_model = prompt_enhancer_llm_model
        self.prompt_enhancer_llm_tokenizer = prompt_enhancer_llm_tokenizer
        self.allowed_inference_steps = allowed_inference_steps
        self.video_scale_factor, self.vae_scale_factor = 8, 32         # get_vae_size_scale_factor of the LTX VAEs (:300-302)
        self._interrupt = False

    @property
    def _execution_device(self):
        return self.transformer.device

    # ---- prompt side (pipeline_ltx_video.py:315-485, 513-606): host checks and the hand-over to the caller's T5 ----
    def check_inputs(self, prompt, height, width, negative_prompt, prompt_embeds=None, negative_prompt_embeds=None,
                     prompt_attention_mask=None, negative_prompt_attention_mask=None, enhance_prompt=False):
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`. Please make sure to only forward one of the two.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        if prompt is not None and not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if prompt is not None and negative_prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `negative_prompt_embeds`. Please make sure to only forward one of the two.")
        if negative_prompt is not None and negative_prompt_embeds is not None:
            raise ValueError("Cannot forward both `negative_prompt` and `negative_prompt_embeds`. Please make sure to only forward one of the two.")
        if prompt_embeds is not None and prompt_attention_mask is None:
            raise ValueError("Must provide `prompt_attention_mask` when specifying `prompt_embeds`.")
        if negative_prompt_embeds is not None and negative_prompt_attention_mask is None:
            raise ValueError("Must provide `negative_prompt_attention_mask` when specifying `negative_prompt_embeds`.")
        if prompt_embeds is not None and negative_prompt_embeds is not None:
            if prompt_embeds.shape != negative_prompt_embeds.shape:
                raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed directly, but"
                                 f" got: `prompt_embeds` {prompt_embeds.shape} != `negative_prompt_embeds` {negative_prompt_embeds.shape}.")
            if prompt_attention_mask.shape != negative_prompt_attention_mask.shape:
                raise ValueError("`prompt_attention_mask` and `negative_prompt_attention_mask` must have the same shape when passed "
                                 f"directly, but got: {prompt_attention_mask.shape} != {negative_prompt_attention_mask.shape}.")
        if enhance_prompt:
            raise NotImplementedError("ltxmi: enhance_prompt (Florence / LLM prompt rewriting) is outside this path")

    @staticmethod
    def _text_preprocessing(text):                                                       # :592-600
        if not isinstance(text, (tuple, list)):
            text = [text]
        return [t.strip() for t in text]

    def encode_prompt(self, prompt, do_classifier_free_guidance: bool = True, negative_prompt: str = "",
                      num_images_per_prompt: int = 1, device=None, prompt_embeds=None, negative_prompt_embeds=None,
                      prompt_attention_mask=None, negative_prompt_attention_mask=None,
                      text_encoder_max_tokens: int = 256, **kwargs):
        """:315-485.  Tokenise, run the caller's text encoder, repeat per image; the T5 itself is not part of
        this library -- without ``text_encoder`` / ``tokenizer`` a string prompt is an explicit error."""
        if device is None:
            device = self._execution_device
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None and isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        max_length = text_encoder_max_tokens
        text_enc_device = None
        if prompt_embeds is None:
            if self.text_encoder is None or self.tokenizer is None:
                raise RuntimeError("ltxmi.LTXVideoPipeline.encode_prompt: a string prompt needs the caller's T5 "
                                   "(`tokenizer=` and `text_encoder=` at construction) -- the text encoder is outside "
                                   "this library; alternatively pass prompt_embeds / prompt_attention_mask")
            text_enc_device = next(self.text_encoder.parameters()).device
            prompt = self._text_preprocessing(prompt)
            text_inputs = self.tokenizer(prompt, padding="max_length", max_length=max_length, truncation=True,
                                         add_special_tokens=True, return_tensors="pt")
            prompt_attention_mask = text_inputs.attention_mask.to(text_enc_device).to(device)
            prompt_embeds = self.text_encoder(text_inputs.input_ids.to(text_enc_device),
                                              attention_mask=prompt_attention_mask)[0]
        if self.text_encoder is not None:
            dtype = self.text_encoder.dtype
        elif self.transformer is not None:
            dtype = self.transformer.dtype
        else:
            dtype = None
        prompt_embeds = prompt_embeds.to(dtype=dtype, device=device)
        bs_embed, seq_len, _ = prompt_embeds.shape
        prompt_embeds = prompt_embeds.repeat(1, num_images_per_prompt, 1).view(bs_embed * num_images_per_prompt, seq_len, -1)
        prompt_attention_mask = prompt_attention_mask.repeat(1, num_images_per_prompt).view(bs_embed * num_images_per_prompt, -1)
        if do_classifier_free_guidance and negative_prompt_embeds is None:
            if self.text_encoder is None or self.tokenizer is None:
                raise RuntimeError("ltxmi.LTXVideoPipeline.encode_prompt: the negative prompt needs the caller's T5 too")
            text_enc_device = next(self.text_encoder.parameters()).device
            uncond_tokens = self._text_preprocessing(negative_prompt) * batch_size
            uncond_input = self.tokenizer(uncond_tokens, padding="max_length", max_length=prompt_embeds.shape[1],
                                          truncation=True, return_attention_mask=True, add_special_tokens=True,
                                          return_tensors="pt")
            negative_prompt_attention_mask = uncond_input.attention_mask.to(text_enc_device)
            negative_prompt_embeds = self.text_encoder(uncond_input.input_ids.to(text_enc_device),
                                                       attention_mask=negative_prompt_attention_mask)[0]
        if do_classifier_free_guidance:
            seq_len = negative_prompt_embeds.shape[1]
            negative_prompt_embeds = negative_prompt_embeds.to(dtype=dtype, device=device)
            negative_prompt_embeds = negative_prompt_embeds.repeat(1, num_images_per_prompt, 1).view(
                batch_size * num_images_per_prompt, seq_len, -1)
            negative_prompt_attention_mask = negative_prompt_attention_mask.repeat(1, num_images_per_prompt).view(
                bs_embed * num_images_per_prompt, -1)
        else:
            negative_prompt_embeds = None
            negative_prompt_attention_mask = None
        return prompt_embeds, prompt_attention_mask, negative_prompt_embeds, negative_prompt_attention_mask

    @staticmethod
    def postprocess(image, output_type):
        """``self.image_processor.postprocess`` (:1299; diffusers' VaeImageProcessor, restated: PARITY UNPINNED like
        the other diffusers leaves): "pt" = (x / 2 + 0.5).clamp(0, 1), which is what ``ltxv.py:462`` undoes.  The
        numpy / PIL forms of that class do not take 5-D video tensors in the reference either."""
        if output_type == "latent":
            return image
        if output_type == "pt":
            return (image / 2 + 0.5).clamp(0, 1)
        raise ValueError(f"ltxmi.LTXVideoPipeline: output_type {output_type!r} is not available for video tensors "
                         "(use 'pt' or 'latent')")

    def prepare_latents(self, latents, media_items, timestep, latent_shape, dtype, device, generator,
                        vae_per_channel_normalize: bool = True):
        """pipeline_ltx_video.py:632-710: (b, c, f, h, w) latents = pure noise, or the given latents / the encoded
        ``media_items`` noised to ``timestep``.  The noise is drawn in PATCHIFIED order (b, f*h*w, c) (:696-699)."""
        if isinstance(generator, list) and len(generator) != latent_shape[0]:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective "
                             f"batch size of {latent_shape[0]}. Make sure the batch size matches the length of the generators.")
        assert latents is None or media_items is None, \
            "Cannot provide both latents and media_items. Please provide only one of the two."
        assert (latents is None and media_items is None) or timestep < 1.0, \
            "Input media_item or latents are provided, but they will be replaced with noise."
        if media_items is not None:
            latents = vae_encode(media_items.to(dtype=self.vae.dtype, device=self.vae.device), self.vae,
                                 vae_per_channel_normalize=vae_per_channel_normalize)
        if latents is not None:
            assert tuple(latents.shape) == tuple(latent_shape), \
                f"Latents have to be of shape {tuple(latent_shape)} but are {tuple(latents.shape)}."
            latents = latents.to(device=device, dtype=dtype)
        b, c, f, h, w = latent_shape
        noise = torch.randn((b, f * h * w, c), generator=generator, device=device, dtype=dtype)
        noise = self.patchifier.unpatchify(noise, h, w, c)
        noise = noise * self.scheduler.init_noise_sigma
        if latents is None:
            return noise
        return timestep * noise + (1 - timestep) * latents

    # ---- conditioning (pipeline_ltx_video.py:1344-1690) ---------------------------------------
    # Setup-time token assembly: slicing / lerp on small latent tensors, once per call (not per step);
    # the encoder it feeds from and everything inside the loop run on libltxmi kernels.
    @staticmethod
    def resize_tensor(media_items, height, width):                                       # :748-760
        """Host pre-processing (a torch bilinear resize per frame), only taken when the media is not at the target size."""
        n_frames = media_items.shape[2]
        if media_items.shape[-2:] != (height, width):
            flat = media_items.permute(0, 2, 1, 3, 4).flatten(0, 1)
            flat = torch.nn.functional.interpolate(flat, size=(height, width), mode="bilinear", align_corners=False)
            media_items = flat.unflatten(0, (-1, n_frames)).permute(0, 2, 1, 3, 4)
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

    retrieve_timesteps = staticmethod(retrieve_timesteps)        # (a module-level function in the reference, :125)

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
    def __call__(
        self,
        height: int,
        width: int,
        num_frames: int,
        frame_rate: float,
        prompt: Union[str, List[str]] = None,
        negative_prompt: str = None,
        num_inference_steps: int = 20,
        timesteps: List[int] = None,
        guidance_scale: Union[float, List[float]] = 4.5,
        skip_layer_strategy: Optional[SkipLayerStrategy] = None,
        skip_block_list: Optional[Union[List[List[int]], List[int]]] = None,
        stg_scale: Union[float, List[float]] = 1.0,
        rescaling_scale: Union[float, List[float]] = 0.7,
        guidance_timesteps: Optional[List[int]] = None,
        num_images_per_prompt: Optional[int] = 1,
        eta: float = 0.0,
        generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
        latents: Optional[torch.FloatTensor] = None,
        prompt_embeds: Optional[torch.FloatTensor] = None,
        prompt_attention_mask: Optional[torch.FloatTensor] = None,
        negative_prompt_embeds: Optional[torch.FloatTensor] = None,
        negative_prompt_attention_mask: Optional[torch.FloatTensor] = None,
        output_type: Optional[str] = "pil",
        return_dict: bool = True,
        callback_on_step_end: Optional[Callable[[int, int, Dict], None]] = None,
        conditioning_items: Optional[List[ConditioningItem]] = None,
        decode_timestep: Union[List[float], float] = 0.0,
        decode_noise_scale: Optional[List[float]] = None,
        mixed_precision: bool = False,
        offload_to_cpu: bool = False,
        enhance_prompt: bool = False,
        text_encoder_max_tokens: int = 256,
        stochastic_sampling: bool = False,
        media_items: Optional[torch.Tensor] = None,
        strength: Optional[float] = 1.0,
        skip_initial_inference_steps: int = 0,
        skip_final_inference_steps: int = 0,
        joint_pass: bool = False,
        pass_no: int = -1,
        ltxv_model=None,
        callback=None,
        *,
        stg_row_dedup: bool = True,
        dead_row_elimination: bool = True,
        latents_dtype: Optional[torch.dtype] = None,
        sample_conditioning_posterior: bool = True,
        **kwargs,
    ):
        """The reference's signature, parameter for parameter (pipeline_ltx_video.py:762-807; defaults included), and its
        behaviour for every argument the video path uses.  Read from ``**kwargs`` as the reference does (:901, :918-919):
        ``is_video``, ``vae_per_channel_normalize``, ``image_cond_noise_scale``; anything else in ``**kwargs`` (the YAML keys
        ``ltxv.py:420`` spreads into the call, ``VAE_tile_size``, ``device``, ``num_inference_steps1/2``,
        ``cfg_star_rescale`` ...) is accepted and ignored, as there.

        Returns what the reference returns: ``None`` when the transformer was interrupted, the tensor itself for
        ``return_dict=True`` (:1306), ``(tensor,)`` otherwise; ``output_type`` "latent" = (b, c, f, h, w) latents,
        "pt" = decoded video in [0, 1].

        Explicit refusals (outside this path): a string ``prompt`` (the reference's ``__call__`` does not encode it
        either: it reads ``prompt_embeds``, :1029-1051 -- ``LTXMultiScalePipeline`` / ``encode_prompt`` do), ``is_video=False``,
        ``mixed_precision``, ``offload_to_cpu``, ``enhance_prompt``, more than one prompt / image per prompt.

        Keyword-only extensions behind the reference's parameters: ``stg_row_dedup`` (the STG "perturbed" row has the text
        row's inputs, so it is the text row until the step's first skipped block; those blocks run on one row less and the
        row is filled in by a copy), ``dead_row_elimination`` (rows whose guidance scale is zero at a step are not
        computed) -- both bit-identical to the plain loop; ``latents_dtype`` (default = the reference's: the dtype of
        ``prompt_embeds``, :1062); ``sample_conditioning_posterior``; and a (b, N, c) ``latents`` tensor is taken as the
        patchified initial noise as is (the reference rejects 3-D latents)."""
        tr = self.transformer
        is_video = kwargs.get("is_video", False)
        self.check_inputs(prompt, height, width, negative_prompt, prompt_embeds, negative_prompt_embeds,
                          prompt_attention_mask, negative_prompt_attention_mask, enhance_prompt)
        if prompt is not None or prompt_embeds is None:
            raise NotImplementedError("ltxmi.LTXVideoPipeline.__call__ takes prompt_embeds / prompt_attention_mask (as the "
                                      "reference's loop does, :1029-1051); string prompts go through encode_prompt / "
                                      "LTXMultiScalePipeline with the caller's T5")
        if not is_video:
            raise NotImplementedError("ltxmi: is_video=False (single images, video_scale_factor 1) is outside this path")
        if mixed_precision:
            raise NotImplementedError("ltxmi: mixed_precision=True (fp32 latents under autocast, :1153-1156) is not on this path")
        if offload_to_cpu:
            raise NotImplementedError("ltxmi: offload_to_cpu is not on this path (everything is resident in HBM)")
        batch_size = prompt_embeds.shape[0]
        if batch_size * num_images_per_prompt != 1:
            raise NotImplementedError("one prompt and one video per call on this path (the reference's CFG-star "
                                      "broadcast at pipeline_ltx_video.py:1199 is only well-formed for batch 1)")
        device = self._execution_device
        vae_per_channel_normalize = kwargs.get("vae_per_channel_normalize", True)
        image_cond_noise_scale = kwargs.get("image_cond_noise_scale", 0.0)
        if ltxv_model is None:
            ltxv_model = self                                                            # holder of ``_interrupt``

        latent_height = height // self.vae_scale_factor
        latent_width = width // self.vae_scale_factor
        latent_num_frames = num_frames // self.video_scale_factor + 1                    # :921-923
        C = tr.config.in_channels
        latent_shape = (batch_size * num_images_per_prompt, C, latent_num_frames, latent_height, latent_width)

        assert strength == 1.0 or latents is not None or media_items is not None, \
            "strength < 1 is used for image-to-image/video-to-video - media_item or latents should be provided."
        timesteps, num_inference_steps = self.retrieve_timesteps(                       # :943-952
            self.scheduler, None if timesteps is not None else num_inference_steps, device, timesteps,
            max_timestep=strength, skip_initial_inference_steps=skip_initial_inference_steps,
            skip_final_inference_steps=skip_final_inference_steps, samples_shape=latent_shape)
        if self.allowed_inference_steps is not None:                                     # :953-957
            for t in [round(x, 4) for x in timesteps]:
                assert t in self.allowed_inference_steps, \
                    f"Invalid inference timestep {t}. Allowed timesteps are {self.allowed_inference_steps}."

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
                e, m = prompt_embeds, prompt_attention_mask.to(device)
                if use_cfg:
                    e = torch.cat([negative_prompt_embeds, e], dim=0)
                    m = torch.cat([negative_prompt_attention_mask.to(device), m], dim=0)
                if use_stg:
                    e = torch.cat([e, prompt_embeds], dim=0)
                    m = torch.cat([m, prompt_attention_mask.to(device)], dim=0)
                batches[key] = (e.to(device=device, dtype=tr.dtype), m, 1 + int(use_cfg) + int(use_stg))
            return batches[key]

        mask_cache = {}

        def skip_mask_for(i, use_stg, nconds):                                           # :1016-1026
            if not use_stg or skip_tab is None:
                return None
            key = (tuple(skip_tab[i]), nconds)
            if key not in mask_cache:
                mask_cache[key] = tr.create_skip_layer_mask(batch_size, nconds, nconds - 1, list(skip_tab[i]))
            return mask_cache[key]

        if latents_dtype is None:
            latents_dtype = prompt_embeds.dtype                                          # :1062 (mixed_precision is refused)
        if latents_dtype not in (torch.float32, torch.bfloat16):
            raise TypeError("ltxmi.LTXVideoPipeline: latents are kept in fp32 or bf16 (prompt_embeds' dtype by default)")
        if latents is not None and latents.dim() == 3:                                   # extension: given patchified noise
            grid5 = self.patchifier.unpatchify(latents.to(device=device, dtype=latents_dtype), latent_height, latent_width, C)
        else:                                                                            # prepare_latents :1056-1065
            grid5 = self.prepare_latents(latents=latents, media_items=media_items, timestep=timesteps[0],
                                         latent_shape=latent_shape, dtype=latents_dtype, device=device, generator=generator,
                                         vae_per_channel_normalize=vae_per_channel_normalize)

        # conditioning items -> latents / coords / mask (+ extra tokens in front)           :1067-1085
        latents, pixel_coords, cond_mask, num_cond_latents = self.prepare_conditioning(
            conditioning_items, grid5.contiguous(), num_frames, height, width, vae_per_channel_normalize, generator,
            sample_posterior=sample_conditioning_posterior)
        latents = latents.contiguous()
        init_latents = latents.clone() if cond_mask is not None else None
        if cond_mask is not None:
            cond_mask = cond_mask.contiguous()
            one_minus_mask = 1.0 - cond_mask
        frac = pixel_coords.to(torch.float32)
        frac[:, 0] = frac[:, 0] * (1.0 / frame_rate)                                     # :1086-1087
        freqs_cis = tr.precompute_freqs_cis(frac)

        if getattr(tr, "_sp_interrupt", None) is not None:          # sequence parallelism: see distributed.begin_generation
            tr._sp_interrupt.reset()
        if callback is not None:                                                         # :1100-1101
            callback(-1, None, True, override_num_inference_steps=num_inference_steps, pass_no=pass_no)

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
                            skip_layer_strategy=skip_layer_strategy, latent_shape=latent_shape[2:], joint_pass=joint_pass,
                            ltxv_model=ltxv_model, mixed=mixed_precision, return_dict=False)[0]
            if noise_pred is None:                                                       # :1180-1181
                return None
            dt = self.scheduler.host_dt(t)
            if not stochastic_sampling:
                ops.guidance_step_(noise_pred, latents, dt, gs_tab[i], stg_tab[i], rs_tab[i],
                                   use_cfg, use_stg, do_rescale, workspace, cond_mask=cond_mask, t=t)   # :1183-1241, 1309-1342
            else:
                # rf.py:368-373 behind the same guidance: x0 = x - t v by the fused kernel with dt = t on a copy (tokens
                # the mask holds back stay as they are), then the re-noising to t - dt; the draw comes from the global
                # RNG, as the reference's torch.randn_like does
                x0 = latents.clone()
                ops.guidance_step_(noise_pred, x0, t, gs_tab[i], stg_tab[i], rs_tab[i],
                                   use_cfg, use_stg, do_rescale, workspace, cond_mask=cond_mask, t=t)
                t_next = t - dt
                renoised = (1 - t_next) * x0 + t_next * torch.randn_like(latents)
                if cond_mask is None:
                    latents.copy_(renoised)
                else:
                    latents.copy_(torch.where((t - 1e-6 < one_minus_mask).unsqueeze(-1), renoised, latents))
            if callback is not None:                                                     # :1243-1247
                preview = latents[:, num_cond_latents:].squeeze(0).transpose(0, 1)
                callback(i, preview.reshape(preview.shape[0], latent_num_frames, latent_height, latent_width), False,
                         pass_no=pass_no)
            if callback_on_step_end is not None:
                callback_on_step_end(self, i, t, {})

        latents = latents[:, num_cond_latents:]                                          # :1258-1259
        latents = self.patchifier.unpatchify(latents, latent_height, latent_width, C)    # :1262-1268
        if output_type != "latent":
            ts = None
            if self.vae.decoder.timestep_conditioning:                                   # :1270-1288
                noise = torch.randn_like(latents)
                if not isinstance(decode_timestep, list):
                    decode_timestep = [decode_timestep] * latents.shape[0]
                if decode_noise_scale is None:
                    decode_noise_scale = decode_timestep
                elif not isinstance(decode_noise_scale, list):
                    decode_noise_scale = [decode_noise_scale] * latents.shape[0]
                ts = torch.tensor(decode_timestep).to(latents.device)
                scale = torch.tensor(decode_noise_scale).to(latents.device)[:, None, None, None, None]
                latents = latents * (1 - scale) + noise * scale
            image = vae_decode(latents.to(self.vae.dtype), self.vae, is_video,
                               vae_per_channel_normalize=vae_per_channel_normalize, timestep=ts)
            image = self.postprocess(image, output_type)
        else:
            image = latents
        if not return_dict:
            return (image,)
        return image                                                                     # :1306: the bare tensor


class LTXMultiScalePipeline:
    """pipeline_ltx_video.py:1741-1905: pass 1 at the downscaled size -> latent upsampler (x2) -> AdaIN against the
    pass-1 latents -> pass 2 from the re-noised upsampled latents -> bilinear resize to the requested size.  Same call
    contract as the reference (``ltxv.py:420-445`` calls through): ``prompt`` / ``negative_prompt`` strings are encoded by
    ``video_pipeline.encode_prompt`` with the caller's T5; everything else travels in ``**kwargs``."""

    def __init__(self, video_pipeline: LTXVideoPipeline, latent_upsampler):
        self.video_pipeline = video_pipeline
        self.vae = video_pipeline.vae
        self.latent_upsampler = latent_upsampler

    def _upsample_latents(self, latest_upsampler, latents):                              # :1760-1772
        from .latent_upsampler import upsample_latents
        return upsample_latents(latest_upsampler, latents, self.vae)

    def __call__(self, downscale_factor: float, first_pass: dict, second_pass: dict, *args: Any, **kwargs: Any) -> Any:
        from .latent_upsampler import adain_filter_latent
        vp = self.video_pipeline
        original_output_type = kwargs["output_type"]
        original_width, original_height = kwargs["width"], kwargs["height"]
        x_width = int(kwargs["width"] * downscale_factor)                                # :1797-1800
        downscaled_width = x_width - (x_width % vp.vae_scale_factor)
        x_height = int(kwargs["height"] * downscale_factor)
        downscaled_height = x_height - (x_height % vp.vae_scale_factor)
        kwargs["output_type"] = "latent"
        kwargs["width"] = downscaled_width
        kwargs["height"] = downscaled_height

        # extension: VAE_tile_size / ltxv_model / device / prompt may be absent (the reference raises KeyError)
        z_tile, hw_tile = kwargs.get("VAE_tile_size") or (0, 0)                          # :1806-1814
        if z_tile > 0:
            self.vae.enable_z_tiling(z_tile)
        if hw_tile > 0:
            self.vae.enable_hw_tiling()
            self.vae.set_tiling_params(hw_tile)

        ltxv_model = kwargs.get("ltxv_model")
        prompt = kwargs.pop("prompt", None)
        negative_prompt = kwargs.pop("negative_prompt", None)
        if prompt is not None or kwargs.get("prompt_embeds") is None:                    # :1833-1852
            (kwargs["prompt_embeds"], kwargs["prompt_attention_mask"], kwargs["negative_prompt_embeds"],
             kwargs["negative_prompt_attention_mask"]) = vp.encode_prompt(
                prompt, True, negative_prompt=negative_prompt, device=kwargs.get("device"), text_encoder_max_tokens=256)
        if ltxv_model is not None and ltxv_model._interrupt:
            return None
        original_kwargs = kwargs.copy()

        kwargs["joint_pass"] = True
        kwargs["pass_no"] = 1
        kwargs.update(**first_pass)
        if "num_inference_steps1" in kwargs:                                             # :1862 (required there)
            kwargs["num_inference_steps"] = kwargs["num_inference_steps1"]
        latents = vp(*args, **kwargs)
        if latents is None:
            return None

        upsampled = self._upsample_latents(self.latent_upsampler, latents)               # :1869-1873
        upsampled = adain_filter_latent(latents=upsampled, reference_latents=latents)

        kwargs = original_kwargs
        kwargs["latents"] = upsampled
        kwargs["output_type"] = original_output_type
        kwargs["width"] = downscaled_width * 2
        kwargs["height"] = downscaled_height * 2
        kwargs["joint_pass"] = False
        kwargs["pass_no"] = 2
        kwargs.update(**second_pass)
        if "num_inference_steps2" in kwargs:
            kwargs["num_inference_steps"] = kwargs["num_inference_steps2"]
        result = vp(*args, **kwargs)
        if result is None:
            return None
        if original_output_type != "latent":                                             # :1891-1903 (host post-processing)
            num_frames = result.shape[2]
            videos = result.permute(0, 2, 1, 3, 4).flatten(0, 1)
            videos = torch.nn.functional.interpolate(videos, size=(original_height, original_width), mode="bilinear",
                                                     align_corners=False)
            result = videos.unflatten(0, (-1, num_frames)).permute(0, 2, 1, 3, 4)
        return result
