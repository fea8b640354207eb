"""Oracle restatement of the reference's causal 3-D VAE ENCODE path (image/video conditioning).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows:
  Encoder.__init__/forward             ltx_video/models/autoencoders/causal_video_autoencoder.py:344-557
  SpaceToDepthDownsample.forward       causal_video_autoencoder.py:976-1020
  CausalConv3d (strided)               ltx_video/models/autoencoders/causal_conv3d.py:7-59
  patchify                             causal_video_autoencoder.py:1261-1279
  AutoencoderKLWrapper.encode/_encode  ltx_video/models/autoencoders/vae.py:156-191, 265-340
  vae_encode / normalize_latents       ltx_video/models/autoencoders/vae_encode.py:22-91, 228-236
  DiagonalGaussianDistribution         diffusers (absent here, "parity unpinned" leaf): mean/logvar =
                                       chunk(moments, 2, dim=1), logvar clamped to [-30, 20],
                                       sample = mean + exp(0.5 logvar) * randn, mode = mean.
"""
import math

import torch
import torch.nn.functional as F

from . import vae as V

_STRIDES = {"compress_time": (2, 1, 1), "compress_space": (1, 2, 2), "compress_all": (2, 2, 2),
            "compress_all_x_y": (2, 2, 2), "compress_all_res": (2, 2, 2), "compress_space_res": (1, 2, 2),
            "compress_time_res": (2, 1, 1)}


def demo_encoder_blocks():
    """create_video_autoencoder_demo_config, causal_video_autoencoder.py:1303-1313."""
    return [("res_x", {"num_layers": 2}), ("compress_space_res", {"multiplier": 2}),
            ("res_x", {"num_layers": 2}), ("compress_time_res", {"multiplier": 2}),
            ("res_x", {"num_layers": 1}), ("compress_all_res", {"multiplier": 2}),
            ("res_x", {"num_layers": 1}), ("compress_all_res", {"multiplier": 2}),
            ("res_x", {"num_layers": 1})]


def encoder_plan(cfg):
    """Channel bookkeeping of Encoder.__init__ (:362-476).  Returns (conv_in_in, base, plan, final)."""
    blocks = cfg.get("encoder_blocks", cfg.get("blocks"))
    patch = cfg.get("patch_size", 1)
    base = cfg.get("encoder_base_channels", 128)
    ch = base
    plan = []
    for name, params in blocks:
        cin = ch
        if isinstance(params, int):
            params = {"num_layers": params}
        if name == "res_x":
            plan.append(dict(kind="mid", channels=cin, num_layers=params["num_layers"]))
        elif name == "res_x_y":
            ch = params.get("multiplier", 2) * ch
            plan.append(dict(kind="res", cin=cin, cout=ch))
        elif name in ("compress_time", "compress_space", "compress_all", "compress_all_x_y"):
            if name == "compress_all_x_y":
                ch = params.get("multiplier", 2) * ch
            plan.append(dict(kind="down", cin=cin, cout=ch, stride=_STRIDES[name]))
        elif name in ("compress_all_res", "compress_space_res", "compress_time_res"):
            ch = params.get("multiplier", 2) * ch
            stride = _STRIDES[name]
            plan.append(dict(kind="s2d", cin=cin, cout=ch, stride=stride,
                             conv_out=ch // math.prod(stride), group=cin * math.prod(stride) // ch))
        else:
            raise ValueError(f"unknown block: {name}")
    return cfg.get("in_channels", 3) * patch ** 2, base, plan, ch


def conv_out_channels(cfg):
    """Encoder.__init__ :489-497."""
    c = cfg["latent_channels"]
    llv = cfg.get("latent_log_var", "per_channel" if cfg.get("double_z", True) else "none")
    return {"per_channel": 2 * c, "uniform": c + 1, "constant": c + 1, "none": c}[llv], llv


def strided_causal_conv3d(x, sd, p, stride, pad_mode):
    """CausalConv3d.forward(causal=True) with nn.Conv3d(stride=stride, padding=(0,1,1)) (causal_conv3d.py:33-57)."""
    w, b = sd[p + "conv.weight"], sd.get(p + "conv.bias")
    x = torch.cat([x[:, :, :1].repeat(1, 1, w.shape[2] - 1, 1, 1), x], dim=2)
    if pad_mode == "zeros":
        return F.conv3d(x, w, b, stride=stride, padding=(0, 1, 1))
    x = F.pad(x, (1, 1, 1, 1, 0, 0), mode=pad_mode)
    return F.conv3d(x, w, b, stride=stride)


def space_to_depth(x, stride):
    """rearrange 'b c (d p1) (h p2) (w p3) -> b (c p1 p2 p3) d h w' (:998-1004)."""
    B, C, D, H, W = x.shape
    p1, p2, p3 = stride
    x = x.view(B, C, D // p1, p1, H // p2, p2, W // p3, p3)
    return x.permute(0, 1, 3, 5, 7, 2, 4, 6).reshape(B, C * p1 * p2 * p3, D // p1, H // p2, W // p3)


def space_to_depth_downsample(x, sd, p, blk, pad_mode):
    """SpaceToDepthDownsample.forward (:991-1020)."""
    stride = blk["stride"]
    if stride[0] == 2:
        x = torch.cat([x[:, :, :1], x], dim=2)
    x_in = space_to_depth(x, stride)
    B, C, D, H, W = x_in.shape
    x_in = x_in.view(B, C // blk["group"], blk["group"], D, H, W).mean(dim=2)
    x = V.causal_conv3d(x, sd, p + "conv.", True, pad_mode)
    return space_to_depth(x, stride) + x_in


def encoder_forward(sd, cfg, sample, prefix="encoder."):
    """Encoder.forward (:514-557): pixels [B,3,F,H,W] -> moments [B, 2C (or as latent_log_var says), f, h, w]."""
    pad_mode = cfg.get("spatial_padding_mode", "zeros")
    assert cfg.get("norm_layer", "group_norm") == "pixel_norm"
    _, _, plan, _ = encoder_plan(cfg)
    _, llv = conv_out_channels(cfg)
    x = V.patchify(sample, patch_size_hw=cfg.get("patch_size", 1), patch_size_t=1)
    x = V.causal_conv3d(x, sd, prefix + "conv_in.", True, pad_mode)
    for i, blk in enumerate(plan):
        p = f"{prefix}down_blocks.{i}."
        if blk["kind"] == "mid":
            for j in range(blk["num_layers"]):
                x = V.resnet_block(x, sd, f"{p}res_blocks.{j}.", True, pad_mode, None)
        elif blk["kind"] == "res":
            x = V.resnet_block(x, sd, p, True, pad_mode, None)
        elif blk["kind"] == "down":
            x = strided_causal_conv3d(x, sd, p, blk["stride"], pad_mode)
        else:
            x = space_to_depth_downsample(x, sd, p, blk, pad_mode)
    x = F.silu(V.pixel_norm(x))
    x = V.causal_conv3d(x, sd, prefix + "conv_out.", True, pad_mode)
    if llv == "uniform":
        x = torch.cat([x, x[:, -1:].repeat(1, x.shape[1] - 2, 1, 1, 1)], dim=1)
    elif llv == "constant":
        x = x[:, :-1]
        x = torch.cat([x, torch.ones_like(x) * -30], dim=1)
    return x


def hw_tiled_encode(sd, cfg, x, tile_sample_min_size=512, overlap=0.25):
    """AutoencoderKLWrapper._hw_tiled_encode (vae.py:156-191)."""
    tile_latent_min_size = int(tile_sample_min_size / 32)
    overlap_size = int(tile_sample_min_size * (1 - overlap))
    blend_extent = int(tile_latent_min_size * overlap)
    row_limit = tile_latent_min_size - blend_extent
    rows = []
    for i in range(0, x.shape[3], overlap_size):
        row = []
        for j in range(0, x.shape[4], overlap_size):
            row.append(encoder_forward(sd, cfg, x[:, :, :, i:i + tile_sample_min_size, j:j + tile_sample_min_size]))
        rows.append(row)
    result_rows = []
    for i, row in enumerate(rows):
        result_row = []
        for j, tile in enumerate(row):
            if i > 0:
                tile = V._blend(rows[i - 1][j], tile, blend_extent, 3)
            if j > 0:
                tile = V._blend(row[j - 1], tile, blend_extent, 4)
            result_row.append(tile[:, :, :, :row_limit, :row_limit])
        result_rows.append(torch.cat(result_row, dim=4))
    return torch.cat(result_rows, dim=3)


def encode(sd, cfg, x, use_z_tiling=False, z_sample_size=4, use_hw_tiling=False, tile_sample_min_size=512):
    """AutoencoderKLWrapper.encode (vae.py:265-312), use_quant_conv=False: returns the moments."""
    def _enc(t, hw):
        return hw_tiled_encode(sd, cfg, t, tile_sample_min_size) if hw else encoder_forward(sd, cfg, t)

    if use_z_tiling and x.shape[2] > (z_sample_size + 1) > 1:
        tl = z_sample_size
        ts = tl * 8
        overlap_size = int(ts * 0.75)
        blend_extent = int(tl * 0.25)
        t_limit = tl - blend_extent
        row = []
        for i in range(0, x.shape[2], overlap_size):
            tile = _enc(x[:, :, i:i + ts + 1], use_hw_tiling)
            if i > 0:
                tile = tile[:, :, 1:]
            row.append(tile)
        result = []
        for i, tile in enumerate(row):
            if i > 0:
                tile = V._blend(row[i - 1], tile, blend_extent, 2)
                result.append(tile[:, :, :t_limit])
            else:
                result.append(tile[:, :, :t_limit + 1])
        return torch.cat(result, dim=2)
    return _enc(x, use_hw_tiling and x.shape[2] > 1)


def gaussian_params(moments):
    """DiagonalGaussianDistribution.__init__ (diffusers, restated): (mean, logvar clamped, std)."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return mean, logvar, torch.exp(0.5 * logvar)


def normalize_latents(latents, sd, per_channel=True, scaling_factor=1.0):
    """vae_encode.py:228-236."""
    if per_channel:
        std = sd["per_channel_statistics.std-of-means"].to(latents.dtype).view(1, -1, 1, 1, 1)
        mean = sd["per_channel_statistics.mean-of-means"].to(latents.dtype).view(1, -1, 1, 1, 1)
        return (latents - mean) / std
    return latents * scaling_factor


def vae_encode(sd, cfg, media, noise=None, per_channel_normalize=True, **tiling):
    """vae_encode (vae_encode.py:22-91) with latent_dist.sample() = mean + std * noise
    (noise=None -> the distribution's mode)."""
    mean, _, std = gaussian_params(encode(sd, cfg, media, **tiling))
    z = mean if noise is None else mean + std * noise
    return normalize_latents(z, sd, per_channel_normalize)


def init_state_dict(cfg, seed=1, dtype=torch.float32, prefix="encoder."):
    """Random encoder weights under the reference's key names."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, cin, cout, k=3):
        bound = 1.0 / math.sqrt(cin * k ** 3)
        sd[name + ".weight"] = (torch.rand(cout, cin, k, k, k, generator=g) * 2 - 1) * bound
        sd[name + ".bias"] = (torch.rand(cout, generator=g) * 2 - 1) * bound

    cin0, base, plan, cfinal = encoder_plan(cfg)
    conv(prefix + "conv_in.conv", cin0, base)
    for i, blk in enumerate(plan):
        p = f"{prefix}down_blocks.{i}"
        if blk["kind"] == "mid":
            c = blk["channels"]
            for j in range(blk["num_layers"]):
                conv(f"{p}.res_blocks.{j}.conv1.conv", c, c)
                conv(f"{p}.res_blocks.{j}.conv2.conv", c, c)
        elif blk["kind"] == "res":
            conv(p + ".conv1.conv", blk["cin"], blk["cout"])
            conv(p + ".conv2.conv", blk["cout"], blk["cout"])
            if blk["cin"] != blk["cout"]:
                conv(p + ".conv_shortcut", blk["cin"], blk["cout"], k=1)
                sd[p + ".norm3.norm.weight"] = 1.0 + 0.1 * torch.randn(blk["cin"], generator=g)
                sd[p + ".norm3.norm.bias"] = 0.1 * torch.randn(blk["cin"], generator=g)
        elif blk["kind"] == "down":
            conv(p + ".conv", blk["cin"], blk["cout"])
        else:
            conv(p + ".conv.conv", blk["cin"], blk["conv_out"])
    conv(prefix + "conv_out.conv", cfinal, conv_out_channels(cfg)[0])
    C = cfg["latent_channels"]
    sd["per_channel_statistics.std-of-means"] = 0.5 + torch.rand(C, generator=g)
    sd["per_channel_statistics.mean-of-means"] = 0.2 * torch.randn(C, generator=g)
    return {k: v.to(dtype) for k, v in sd.items()}
