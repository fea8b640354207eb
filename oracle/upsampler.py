"""Oracle restatement of the multi-scale bridge (pass 1 -> pass 2).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows:
  ResBlock.forward / LatentUpsampler.forward    ltx_video/models/autoencoders/latent_upsampler.py:30-39, 109-149
  PixelShuffleND                                ltx_video/models/autoencoders/pixel_shuffle.py:12-33
  adain_filter_latent                           ltx_video/pipelines/pipeline_ltx_video.py:1709-1737
  LTXMultiScalePipeline._upsample_latents       pipeline_ltx_video.py:1760-1772
``sd`` carries the reference's key names (initial_conv.weight, res_blocks.0.conv1.weight, upsampler.0.weight ...).
"""
import math

import torch
import torch.nn.functional as F


def _conv(x, sd, p, dims):
    f = F.conv2d if dims == 2 else F.conv3d
    return f(x, sd[p + ".weight"], sd[p + ".bias"], padding=1)


def res_block(x, sd, p, dims):
    r = x
    x = _conv(x, sd, p + ".conv1", dims)
    x = F.silu(F.group_norm(x, 32, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5))
    x = _conv(x, sd, p + ".conv2", dims)
    x = F.group_norm(x, 32, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    return F.silu(x + r)


def pixel_shuffle(x, dims):
    """PixelShuffleND(dims) with upscale factors (2,2,2)."""
    if dims == 3:
        B, C, D, H, W = x.shape
        c = C // 8
        return x.view(B, c, 2, 2, 2, D, H, W).permute(0, 1, 5, 2, 6, 3, 7, 4).reshape(B, c, D * 2, H * 2, W * 2)
    if dims == 2:
        B, C, H, W = x.shape
        c = C // 4
        return x.view(B, c, 2, 2, H, W).permute(0, 1, 4, 2, 5, 3).reshape(B, c, H * 2, W * 2)
    B, C, D, H, W = x.shape            # dims == 1: time only
    c = C // 2
    return x.view(B, c, 2, D, H, W).permute(0, 1, 3, 2, 4, 5).reshape(B, c, D * 2, H, W)


def latent_upsampler_forward(sd, cfg, latent):
    """LatentUpsampler.forward (:109-149)."""
    dims = cfg.get("dims", 2)
    nb = cfg.get("num_blocks_per_stage", 4)
    spatial, temporal = cfg.get("spatial_upsample", True), cfg.get("temporal_upsample", False)
    b, c, f, h, w = latent.shape

    def trunk(x, d):
        x = _conv(x, sd, "initial_conv", d)
        x = F.silu(F.group_norm(x, 32, sd["initial_norm.weight"], sd["initial_norm.bias"], 1e-5))
        for i in range(nb):
            x = res_block(x, sd, f"res_blocks.{i}", d)
        return x

    def tail(x, d):
        for i in range(nb):
            x = res_block(x, sd, f"post_upsample_res_blocks.{i}", d)
        return _conv(x, sd, "final_conv", d)

    def fold(x):
        return x.permute(0, 2, 1, 3, 4).reshape(b * x.shape[2], x.shape[1], x.shape[3], x.shape[4])

    def unfold(x):
        return x.view(b, f, x.shape[1], x.shape[2], x.shape[3]).permute(0, 2, 1, 3, 4)

    if dims == 2:
        x = trunk(fold(latent), 2)
        x = pixel_shuffle(F.conv2d(x, sd["upsampler.0.weight"], sd["upsampler.0.bias"], padding=1), 2)
        return unfold(tail(x, 2))
    x = trunk(latent, 3)
    if temporal:
        x = F.conv3d(x, sd["upsampler.0.weight"], sd["upsampler.0.bias"], padding=1)
        x = pixel_shuffle(x, 3 if spatial else 1)[:, :, 1:]
    else:
        x = fold(x)
        x = pixel_shuffle(F.conv2d(x, sd["upsampler.0.weight"], sd["upsampler.0.bias"], padding=1), 2)
        x = unfold(x)
    return tail(x, 3)


def adain_filter_latent(latents, reference_latents, factor=1.0):
    """:1709-1737."""
    result = latents.clone()
    for i in range(latents.size(0)):
        for c in range(latents.size(1)):
            r_sd, r_mean = torch.std_mean(reference_latents[i, c], dim=None)
            i_sd, i_mean = torch.std_mean(result[i, c], dim=None)
            result[i, c] = ((result[i, c] - i_mean) / i_sd) * r_sd + r_mean
    return torch.lerp(latents, result, factor)


def upsample_latents(sd, cfg, latents, stats_sd):
    """_upsample_latents (:1760-1772): un_normalize -> upsampler -> normalize (per-channel statistics)."""
    std = stats_sd["per_channel_statistics.std-of-means"].to(latents.dtype).view(1, -1, 1, 1, 1)
    mean = stats_sd["per_channel_statistics.mean-of-means"].to(latents.dtype).view(1, -1, 1, 1, 1)
    up = latent_upsampler_forward(sd, cfg, latents * std + mean)
    return (up - mean) / std


def init_state_dict(cfg, seed=3, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    dims = cfg.get("dims", 2)
    cin, mid, nb = cfg.get("in_channels", 4), cfg.get("mid_channels", 128), cfg.get("num_blocks_per_stage", 4)
    spatial, temporal = cfg.get("spatial_upsample", True), cfg.get("temporal_upsample", False)
    sd = {}

    def conv(name, ci, co, d):
        bound = 1.0 / math.sqrt(ci * 3 ** d)
        sd[name + ".weight"] = (torch.rand(co, ci, *([3] * d), generator=g) * 2 - 1) * bound
        sd[name + ".bias"] = (torch.rand(co, generator=g) * 2 - 1) * bound

    def norm(name, ch):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(ch, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(ch, generator=g)

    conv("initial_conv", cin, mid, dims)
    norm("initial_norm", mid)
    for stage in ("res_blocks", "post_upsample_res_blocks"):
        for i in range(nb):
            conv(f"{stage}.{i}.conv1", mid, mid, dims)
            norm(f"{stage}.{i}.norm1", mid)
            conv(f"{stage}.{i}.conv2", mid, mid, dims)
            norm(f"{stage}.{i}.norm2", mid)
    if spatial and temporal:
        conv("upsampler.0", mid, 8 * mid, 3)
    elif spatial:
        conv("upsampler.0", mid, 4 * mid, 2)
    else:
        conv("upsampler.0", mid, 2 * mid, 3)
    conv("final_conv", mid, cin, dims)
    return {k: v.to(dtype) for k, v in sd.items()}
