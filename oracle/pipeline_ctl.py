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


def denoise_pass(sd, cfg, latents_tokens, grid, timesteps, pos, neg, pmask, nmask, guidance, dtype, frame_rate=25.0,
                 on_step=None):
    """The loop of LTXVideoPipeline.__call__ for one pass without conditioning items (:959-1051, 1086-1256): per-step
    guidance tables, batch assembly [uncond | text | perturbed], transformer, CFG-star / STG / rescale, Euler step.
    ``latents_tokens`` (1, N, C) fp32; ``guidance`` = dict(guidance_scale, stg_scale, rescaling_scale, skip_block_list,
    guidance_timesteps); the DiT runs in ``dtype`` (bf16 = the reference's eager rendering), latents stay fp32.
    Returns (b, c, f, h, w)."""
    from . import dit
    f, h, w = grid
    gs, stg, rs, skips, do_cfg, do_stg, do_rs = guidance_tables(
        [float(x) for x in timesteps], guidance["guidance_scale"], guidance["stg_scale"], guidance["rescaling_scale"],
        guidance.get("skip_block_list"), guidance.get("guidance_timesteps"))
    nc = 1 + int(do_cfg) + int(do_stg)
    sdd = {k: v.to(dtype) for k, v in sd.items()}
    emb = torch.cat(([neg] if do_cfg else []) + [pos] + ([pos] if do_stg else [])).to(dtype)
    msk = torch.cat(([nmask] if do_cfg else []) + [pmask] + ([pmask] if do_stg else []))
    pix = sched.latent_to_pixel_coords(sched.get_latent_coords(f, h, w, 1),
                                       causal_fix=cfg.get("causal_temporal_positioning", False)).to(torch.float32)
    pix[:, 0] = pix[:, 0] * (1.0 / frame_rate)
    fc = dit.precompute_freqs_cis(pix, cfg, dtype)
    lat = latents_tokens.clone().float()
    for i, t in enumerate(timesteps):
        skip = None
        if do_stg and skips is not None:
            skip = dit.create_skip_layer_mask(cfg["num_layers"], 1, nc, nc - 1, skips[i], dtype)
        npred = dit.transformer3d_forward(sdd, cfg, torch.cat([lat.to(dtype)] * nc), fc, emb, t.expand(nc).unsqueeze(-1),
                                          encoder_attention_mask=msk, latent_shape=(f, h, w), skip_layer_mask=skip,
                                          skip_layer_strategy=dit.ATTENTION_VALUES)
        v = sched.guidance(npred.float(), nc, gs[i], stg[i], rs[i], do_cfg, do_stg, do_rs)
        lat = sched.denoising_step(timesteps, lat, v, t.expand(1).unsqueeze(-1), None, t)
        if on_step is not None:
            on_step(i, lat)
    return sched.unpatchify(lat, f, h, w)


def multiscale_call(sd, cfg, vae_sd, vae_cfg, up_sd, up_cfg, pos, neg, pmask, nmask, height, width, num_frames, frame_rate,
                    downscale_factor, first_pass, second_pass, steps1, steps2, noise1, noise2, decode_noise,
                    decode_timestep, decode_noise_scale, dtype=torch.float32, vae_dtype=None, stats=None):
    """LTXMultiScalePipeline.__call__ (:1741-1905) with output_type "pt": pass 1 at the downscaled size (:1797-1866) ->
    _upsample_latents + adain_filter_latent (:1869-1873) -> pass 2 from the re-noised upsampled latents with the second
    pass's sliced schedule (:1877-1889) -> decode-noise mix + vae_decode + de-normalise (:1270-1299) -> bilinear resize
    to the requested size (:1891-1903).  ``steps1`` / ``steps2`` = num_inference_steps1 / 2 (override the passes' own
    counts, :1862, :1887).  Returns (video, pass-1 latents, upsampled latents before AdaIN)."""
    from . import upsampler as ou, vae as ov
    vsf, tsf = 32, 8
    x_w = int(width * downscale_factor)
    dw = x_w - (x_w % vsf)
    x_h = int(height * downscale_factor)
    dh = x_h - (x_h % vsf)
    f = num_frames // tsf + 1
    C = cfg["in_channels"]

    def schedule(kw, steps, shape):
        return retrieve_timesteps(steps, shape, skip_initial_inference_steps=kw.get("skip_initial_inference_steps", 0),
                                  skip_final_inference_steps=kw.get("skip_final_inference_steps", 0))

    h1, w1 = dh // vsf, dw // vsf
    shape1 = (1, C, f, h1, w1)
    ts1 = schedule(first_pass, steps1, shape1)
    lat1 = denoise_pass(sd, cfg, noise1, (f, h1, w1), ts1, pos, neg, pmask, nmask, first_pass, dtype, frame_rate)
    up_raw = ou.upsample_latents(up_sd, up_cfg, lat1, stats)
    up = ou.adain_filter_latent(up_raw, lat1)
    h2, w2 = 2 * h1, 2 * w1
    shape2 = (1, C, f, h2, w2)
    ts2 = schedule(second_pass, steps2, shape2)
    start = prepare_latents(up, float(ts2[0]), noise2, shape2)
    lat2 = denoise_pass(sd, cfg, sched.patchify(start)[0], (f, h2, w2), ts2, pos, neg, pmask, nmask, second_pass, dtype,
                        frame_rate)
    s = decode_timestep if decode_noise_scale is None else decode_noise_scale
    z = lat2 * (1 - s) + decode_noise * s
    vdt = vae_dtype or torch.float32
    vsd = {k: (v.to(vdt) if v.is_floating_point() and v.dim() > 0 else v) for k, v in vae_sd.items()}
    img = ov.vae_decode(vsd, vae_cfg, z.to(vdt), torch.tensor([decode_timestep])).float()
    img = (img / 2 + 0.5).clamp(0, 1)
    n = img.shape[2]
    flat = img.permute(0, 2, 1, 3, 4).flatten(0, 1)
    flat = torch.nn.functional.interpolate(flat, size=(height, width), mode="bilinear", align_corners=False)
    return flat.unflatten(0, (-1, n)).permute(0, 2, 1, 3, 4), lat1, up_raw
