"""Oracle restatement of the reference's causal 3-D VAE decode path.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows:
  CausalConv3d.forward                 ltx_video/models/autoencoders/causal_conv3d.py:44-59
  PixelNorm                            ltx_video/models/autoencoders/pixel_norm.py:5-12
  PixelShuffleND                       ltx_video/models/autoencoders/pixel_shuffle.py:12-21
  Decoder.__init__/forward             ltx_video/models/autoencoders/causal_video_autoencoder.py:585-802
  UNetMidBlock3D.forward               causal_video_autoencoder.py:897-973   (no attention blocks)
  DepthToSpaceUpsample.forward         causal_video_autoencoder.py:1051-1065
  ResnetBlock3D.forward                causal_video_autoencoder.py:1197-1258 (inject_noise: the draws are passed in)
  unpatchify                           causal_video_autoencoder.py:1282-1299
  AutoencoderKLWrapper.decode/_decode  ltx_video/models/autoencoders/vae.py:343-413 (+tiling :193-263)
  vae_decode/_run_decoder/un_normalize_latents  ltx_video/models/autoencoders/vae_encode.py:94-165,239-247

Functional: ``sd`` is a flat state dict with the reference's key names
(``decoder.up_blocks.0.res_blocks.1.conv1.conv.weight`` ...), ``cfg`` the VAE config
dict (same keys as CausalVideoAutoencoder.from_config reads, :123-177).
"""
import math

import torch
import torch.nn.functional as F

from . import leaves


def demo_config(latent_channels=128):
    """create_video_autoencoder_demo_config, causal_video_autoencoder.py:1302-1338
    (decoder side): the 0.9.5+-style timestep-conditioned decoder."""
    return {
        "_class_name": "CausalVideoAutoencoder", "dims": 3,
        "decoder_blocks": [
            ("res_x", {"num_layers": 2, "inject_noise": False}),
            ("compress_all", {"residual": True, "multiplier": 2}),
            ("res_x", {"num_layers": 2, "inject_noise": False}),
            ("compress_all", {"residual": True, "multiplier": 2}),
            ("res_x", {"num_layers": 2, "inject_noise": False}),
            ("compress_all", {"residual": True, "multiplier": 2}),
            ("res_x", {"num_layers": 2, "inject_noise": False}),
        ],
        "latent_channels": latent_channels, "norm_layer": "pixel_norm", "patch_size": 4,
        "latent_log_var": "uniform", "use_quant_conv": False, "causal_decoder": False,
        "timestep_conditioning": True, "spatial_padding_mode": "replicate",
    }


def decoder_plan(cfg):
    """Channel bookkeeping of Decoder.__init__ (causal_video_autoencoder.py:607-698).
    Returns (conv_in_out_channels, [block dicts], final_channels)."""
    blocks = cfg.get("decoder_blocks", cfg.get("blocks"))
    base = cfg.get("decoder_base_channels", 128)
    ch = base
    for name, params in reversed(blocks):
        params = params if isinstance(params, dict) else {}
        if name == "res_x_y":
            ch *= params.get("multiplier", 2)
        if name == "compress_all":
            ch *= params.get("multiplier", 1)
    conv_in_out = ch
    plan = []
    for name, params in reversed(blocks):
        cin = ch
        if isinstance(params, int):
            params = {"num_layers": params}
        if name == "res_x":
            plan.append(dict(kind="mid", channels=cin, num_layers=params["num_layers"],
                             inject_noise=params.get("inject_noise", False)))
        elif name == "attn_res_x":
            # UNetMidBlock3D with attention blocks (:646-657): in the reference as shipped this block cannot run --
            # UNetMidBlock3D.forward hands a bare tensor to Attention (:951-956) whose AttnProcessor2_0 treats it as the
            # 1-element list of the DiT (``hidden_states_wrapper.clear()``, attention.py:1002-1003) and raises
            # AttributeError: 'Tensor' object has no attribute 'clear'.  There is nothing to restate.
            raise NotImplementedError("attn_res_x: unreachable in the reference (its own forward raises AttributeError)")
        elif name == "res_x_y":
            ch = ch // params.get("multiplier", 2)
            plan.append(dict(kind="res", cin=cin, cout=ch, inject_noise=params.get("inject_noise", False)))
        elif name in ("compress_time", "compress_space", "compress_all"):
            stride = {"compress_time": (2, 1, 1), "compress_space": (1, 2, 2),
                      "compress_all": (2, 2, 2)}[name]
            red = params.get("multiplier", 1) if name == "compress_all" else 1
            if name == "compress_all":
                ch = ch // red
            plan.append(dict(kind="up", cin=cin, stride=stride, reduction=red,
                             residual=params.get("residual", False) if name == "compress_all" else False,
                             conv_out=math.prod(stride) * cin // red))
        else:
            raise ValueError(f"unknown layer: {name}")
    return conv_in_out, plan, ch


# ------------------------------------------------------------------ primitives
def causal_conv3d(x, sd, p, causal, spatial_padding_mode="zeros"):
    """causal_conv3d.py:44-59 + nn.Conv3d(k, stride 1, padding (0,1,1), padding_mode)."""
    w, b = sd[p + "conv.weight"], sd.get(p + "conv.bias")
    kt = w.shape[2]
    if causal:
        x = torch.cat([x[:, :, :1].repeat(1, 1, kt - 1, 1, 1), x], dim=2)
    else:
        n = (kt - 1) // 2
        x = torch.cat([x[:, :, :1].repeat(1, 1, n, 1, 1), x, x[:, :, -1:].repeat(1, 1, n, 1, 1)], dim=2)
    ph, pw = w.shape[3] // 2, w.shape[4] // 2
    if spatial_padding_mode == "zeros":
        return F.conv3d(x, w, b, padding=(0, ph, pw))
    x = F.pad(x, (pw, pw, ph, ph, 0, 0), mode=spatial_padding_mode)
    return F.conv3d(x, w, b)


def pixel_norm(x, eps=1e-8):
    """pixel_norm.py:11."""
    return x / torch.sqrt(torch.mean(x ** 2, dim=1, keepdim=True) + eps)


def pixel_shuffle_3d(x, stride):
    """pixel_shuffle.py:14-21: b (c p1 p2 p3) d h w -> b c (d p1) (h p2) (w p3)."""
    p1, p2, p3 = stride
    b, cc, d, h, w = x.shape
    c = cc // (p1 * p2 * p3)
    x = x.view(b, c, p1, p2, p3, d, h, w).permute(0, 1, 5, 2, 6, 3, 7, 4)
    return x.reshape(b, c, d * p1, h * p2, w * p3)


def unpatchify(x, patch_size_hw, patch_size_t=1):
    """causal_video_autoencoder.py:1282-1299: b (c p r q) f h w -> b c (f p) (h q) (w r)."""
    if patch_size_hw == 1 and patch_size_t == 1:
        return x
    b, cc, f, h, w = x.shape
    p, q, r = patch_size_t, patch_size_hw, patch_size_hw
    c = cc // (p * q * r)
    x = x.view(b, c, p, r, q, f, h, w).permute(0, 1, 5, 2, 6, 4, 7, 3)
    return x.reshape(b, c, f * p, h * q, w * r)


def patchify(x, patch_size_hw, patch_size_t=1):
    """causal_video_autoencoder.py:1261-1279: b c (f p) (h q) (w r) -> b (c p r q) f h w."""
    if patch_size_hw == 1 and patch_size_t == 1:
        return x
    b, c, F_, H, W = x.shape
    p, q, r = patch_size_t, patch_size_hw, patch_size_hw
    x = x.view(b, c, F_ // p, p, H // q, q, W // r, r).permute(0, 1, 3, 7, 5, 2, 4, 6)
    return x.reshape(b, c * p * r * q, F_ // p, H // q, W // r)


def _layer_norm_ch(x, sd, p, eps):
    """LayerNorm wrapper over the channel axis, causal_video_autoencoder.py:1068-1077."""
    x = x.permute(0, 2, 3, 4, 1)
    x = F.layer_norm(x, (x.shape[-1],), sd[p + "norm.weight"], sd[p + "norm.bias"], eps)
    return x.permute(0, 4, 1, 2, 3)


def _feed_spatial_noise(h, scale, noise):
    """ResnetBlock3D._feed_spatial_noise (:1183-1195): ``noise`` is the [H, W] draw the reference takes from
    torch.randn there (passed in, so that oracle and product can be fed the same draw)."""
    return h + (noise[None].to(h.dtype) * scale.to(h.dtype))[None, :, None]


def resnet_block(x, sd, p, causal, pad_mode, timestep_embed=None, cin=None, cout=None, noise=None):
    """ResnetBlock3D.forward with norm_layer="pixel_norm", eps=1e-6 (causal_video_autoencoder.py:1197-1258).
    noise = (n1, n2): the two [H, W] draws of an inject_noise block, or None."""
    B = x.shape[0]
    h = pixel_norm(x)
    if timestep_embed is not None:
        ada = sd[p + "scale_shift_table"][None, ..., None, None, None] + timestep_embed.reshape(
            B, 4, -1, timestep_embed.shape[-3], timestep_embed.shape[-2], timestep_embed.shape[-1])
        shift1, scale1, shift2, scale2 = ada.unbind(dim=1)
        h = h * (1 + scale1) + shift1
    h = F.silu(h)
    h = causal_conv3d(h, sd, p + "conv1.", causal, pad_mode)
    if noise is not None:
        h = _feed_spatial_noise(h, sd[p + "per_channel_scale1"], noise[0])
    h = pixel_norm(h)
    if timestep_embed is not None:
        h = h * (1 + scale2) + shift2
    h = F.silu(h)
    h = causal_conv3d(h, sd, p + "conv2.", causal, pad_mode)
    if noise is not None:
        h = _feed_spatial_noise(h, sd[p + "per_channel_scale2"], noise[1])
    if (p + "conv_shortcut.weight") in sd:
        x = _layer_norm_ch(x, sd, p + "norm3.", 1e-6)
        x = F.conv3d(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"])
    return x + h


def depth_to_space_upsample(x, sd, p, blk, causal, pad_mode):
    """DepthToSpaceUpsample.forward (causal_video_autoencoder.py:1051-1065)."""
    stride = blk["stride"]
    if blk["residual"]:
        x_in = pixel_shuffle_3d(x, stride)
        x_in = x_in.repeat(1, math.prod(stride) // blk["reduction"], 1, 1, 1)
        if stride[0] == 2:
            x_in = x_in[:, :, 1:]
    x = causal_conv3d(x, sd, p + "conv.", causal, pad_mode)
    x = pixel_shuffle_3d(x, stride)
    if stride[0] == 2:
        x = x[:, :, 1:]
    if blk["residual"]:
        x = x + x_in
    return x


def decoder_forward(sd, cfg, sample, timestep=None, prefix="decoder.", noises=None):
    """Decoder.forward (causal_video_autoencoder.py:735-802).  ``noises``: the [H, W] draws of the inject_noise blocks
    in the order the reference takes them (two per ResnetBlock3D: after conv1, after conv2)."""
    noises = list(noises) if noises is not None else None

    def take(blk):
        if not blk.get("inject_noise"):
            return None
        return (noises.pop(0), noises.pop(0))

    causal = cfg.get("causal_decoder", False)
    pad_mode = cfg.get("spatial_padding_mode", "zeros")
    tcond = cfg.get("timestep_conditioning", False)
    patch = cfg.get("patch_size", 1)
    assert cfg.get("norm_layer", "group_norm") == "pixel_norm"
    _, plan, _ = decoder_plan(cfg)
    B = sample.shape[0]
    x = causal_conv3d(sample, sd, prefix + "conv_in.", causal, pad_mode)
    if tcond:
        assert timestep is not None
        scaled_t = timestep * sd[prefix + "timestep_scale_multiplier"]
    for i, blk in enumerate(plan):
        p = f"{prefix}up_blocks.{i}."
        if blk["kind"] == "mid":
            temb = None
            if tcond:
                temb = leaves.combined_timestep_size_embeddings(
                    scaled_t.flatten(), sd, p + "time_embedder.", x.dtype)
                temb = temb.view(B, temb.shape[-1], 1, 1, 1)
            for j in range(blk["num_layers"]):
                x = resnet_block(x, sd, f"{p}res_blocks.{j}.", causal, pad_mode, temb, noise=take(blk))
        elif blk["kind"] == "res":
            x = resnet_block(x, sd, p, causal, pad_mode, None, noise=take(blk))
        else:
            x = depth_to_space_upsample(x, sd, p, blk, causal, pad_mode)
    x = pixel_norm(x)
    if tcond:
        emb = leaves.combined_timestep_size_embeddings(
            scaled_t.flatten(), sd, prefix + "last_time_embedder.", x.dtype)
        emb = emb.view(B, emb.shape[-1], 1, 1, 1)
        ada = sd[prefix + "last_scale_shift_table"][None, ..., None, None, None] + emb.reshape(
            B, 2, -1, emb.shape[-3], emb.shape[-2], emb.shape[-1])
        shift, scale = ada.unbind(dim=1)
        x = x * (1 + scale) + shift
    x = F.silu(x)
    x = causal_conv3d(x, sd, prefix + "conv_out.", causal, pad_mode)
    return unpatchify(x, patch_size_hw=patch, patch_size_t=1)


# ------------------------------------------------------------- wrapper / tiling
def _blend(a, b, extent, dim):
    """blend_z / blend_v / blend_h, vae.py:193-221 (b is modified in place there)."""
    extent = min(a.shape[dim], b.shape[dim], extent)
    for i in range(extent):
        ia = [slice(None)] * 5
        ib = [slice(None)] * 5
        ia[dim] = -extent + i
        ib[dim] = i
        b[tuple(ib)] = a[tuple(ia)] * (1 - i / extent) + b[tuple(ib)] * (i / extent)
    return b


def hw_tiled_decode(sd, cfg, z, timestep, tile_sample_min_size=512, overlap=0.25):
    """AutoencoderKLWrapper._hw_tiled_decode, vae.py:223-263."""
    tile_latent = int(tile_sample_min_size / 32)
    overlap_size = int(tile_latent * (1 - overlap))
    blend_extent = int(tile_sample_min_size * overlap)
    row_limit = tile_sample_min_size - blend_extent
    rows = []
    for i in range(0, z.shape[3], overlap_size):
        row = []
        for j in range(0, z.shape[4], overlap_size):
            tile = z[:, :, :, i:i + tile_latent, j:j + tile_latent]
            row.append(decoder_forward(sd, cfg, tile, timestep))
        rows.append(row)
    result_rows = []
    for i, row in enumerate(rows):
        result_row = []
        for j, tile in enumerate(row):
            if i > 0:
                tile = _blend(rows[i - 1][j], tile, blend_extent, 3)
            if j > 0:
                tile = _blend(row[j - 1], tile, blend_extent, 4)
            result_row.append(tile[:, :, :, :row_limit, :row_limit])
        result_rows.append(torch.cat(result_row, dim=4))
    return torch.cat(result_rows, dim=3)


def decode(sd, cfg, z, timestep=None, use_z_tiling=False, z_sample_size=4, use_hw_tiling=False,
           tile_sample_min_size=512, noises=None):
    """AutoencoderKLWrapper.decode, vae.py:357-413 (use_quant_conv=False,
    normalize_latent_channels=False).  The z-tiled branch returns fp16 like the
    reference (:388).  ``noises``: the draws of inject_noise blocks (untiled decode only)."""
    def _dec(t):
        if use_hw_tiling:
            return hw_tiled_decode(sd, cfg, t, timestep, tile_sample_min_size)
        return decoder_forward(sd, cfg, t, timestep, noises=noises)

    if use_z_tiling and z.shape[2] > (z_sample_size + 1) > 1:
        tl = z_sample_size
        ts = tl * 8
        overlap_size = int(tl * 0.75)
        blend_extent = int(ts * 0.25)
        t_limit = ts - blend_extent
        row = []
        for i in range(0, z.shape[2], overlap_size):
            d = _dec(z[:, :, i:i + tl + 1])
            if i > 0:
                d = d[:, :, 1:]
            row.append(d.to(torch.float16))
        out = []
        for i, tile in enumerate(row):
            if i > 0:
                tile = _blend(row[i - 1], tile, blend_extent, 2)
                out.append(tile[:, :, :t_limit])
            else:
                out.append(tile[:, :, :t_limit + 1])
        return torch.cat(out, dim=2)
    return _dec(z)


def un_normalize_latents(latents, sd, per_channel=True, scaling_factor=1.0):
    """vae_encode.py:239-247."""
    if per_channel:
        std = sd["per_channel_statistics.std-of-means"].to(latents.dtype).view(1, -1, 1, 1, 1)
        mean = sd["per_channel_statistics.mean-of-means"].to(latents.dtype).view(1, -1, 1, 1, 1)
        return latents * std + mean
    return latents / scaling_factor


def vae_decode(sd, cfg, latents, timestep=None, per_channel_normalize=True, **tiling):
    """vae_decode -> _run_decoder, vae_encode.py:94-165."""
    return decode(sd, cfg, un_normalize_latents(latents, sd, per_channel_normalize), timestep, **tiling)


def init_state_dict(cfg, seed=0, dtype=torch.float32, prefix="decoder."):
    """Random decoder weights with the reference's key names (default Conv3d /
    Linear init; scale_shift_table ~ N(0,1)/sqrt(C))."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, cin, cout, k=3):
        bound = 1.0 / math.sqrt(cin * k ** 3)
        sd[name + ".weight"] = (torch.rand(cout, cin, k, k, k, generator=g) * 2 - 1) * bound
        sd[name + ".bias"] = (torch.rand(cout, generator=g) * 2 - 1) * bound

    def lin(name, fin, fout):
        bound = 1.0 / math.sqrt(fin)
        sd[name + ".weight"] = (torch.rand(fout, fin, generator=g) * 2 - 1) * bound
        sd[name + ".bias"] = (torch.rand(fout, generator=g) * 2 - 1) * bound

    def temb(name, dim):
        lin(name + ".timestep_embedder.linear_1", 256, dim)
        lin(name + ".timestep_embedder.linear_2", dim, dim)

    tcond = cfg.get("timestep_conditioning", False)
    cin0, plan, cfinal = decoder_plan(cfg)
    conv(prefix + "conv_in.conv", cfg["latent_channels"], cin0)
    for i, blk in enumerate(plan):
        p = f"{prefix}up_blocks.{i}"
        if blk["kind"] == "mid":
            c = blk["channels"]
            if tcond:
                temb(p + ".time_embedder", 4 * c)
            for j in range(blk["num_layers"]):
                conv(f"{p}.res_blocks.{j}.conv1.conv", c, c)
                conv(f"{p}.res_blocks.{j}.conv2.conv", c, c)
                if blk.get("inject_noise"):                 # (the reference initialises these to zero)
                    sd[f"{p}.res_blocks.{j}.per_channel_scale1"] = 0.3 * torch.randn(c, 1, 1, generator=g)
                    sd[f"{p}.res_blocks.{j}.per_channel_scale2"] = 0.3 * torch.randn(c, 1, 1, generator=g)
                if tcond:
                    sd[f"{p}.res_blocks.{j}.scale_shift_table"] = torch.randn(4, c, generator=g) / c ** 0.5
        elif blk["kind"] == "res":
            conv(p + ".conv1.conv", blk["cin"], blk["cout"])
            conv(p + ".conv2.conv", blk["cout"], blk["cout"])
            if blk.get("inject_noise"):
                # (in the reference BOTH scales have in_channels entries, :1131,1160: such a block only works with
                # cin == cout or broadcastable sizes; kept as the reference has it)
                sd[p + ".per_channel_scale1"] = 0.3 * torch.randn(blk["cin"], 1, 1, generator=g)
                sd[p + ".per_channel_scale2"] = 0.3 * torch.randn(blk["cin"], 1, 1, generator=g)
            if blk["cin"] != blk["cout"]:
                conv(p + ".conv_shortcut", blk["cin"], blk["cout"], k=1)
                sd[p + ".norm3.norm.weight"] = 1.0 + 0.1 * torch.randn(blk["cin"], generator=g)
                sd[p + ".norm3.norm.bias"] = 0.1 * torch.randn(blk["cin"], generator=g)
        else:
            conv(p + ".conv.conv", blk["cin"], blk["conv_out"])
    patch = cfg.get("patch_size", 1)
    conv(prefix + "conv_out.conv", cfinal, cfg.get("out_channels", 3) * patch ** 2)
    if tcond:
        sd[prefix + "timestep_scale_multiplier"] = torch.tensor(1000.0)
        temb(prefix + "last_time_embedder", 2 * cfinal)
        sd[prefix + "last_scale_shift_table"] = torch.randn(2, cfinal, generator=g) / cfinal ** 0.5
    C = cfg["latent_channels"]
    sd["per_channel_statistics.std-of-means"] = 0.5 + torch.rand(C, generator=g)
    sd["per_channel_statistics.mean-of-means"] = 0.2 * torch.randn(C, generator=g)
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
