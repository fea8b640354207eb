"""``LatentUpsampler`` + ``adain_filter_latent`` + the multi-scale bridge on libltxmi kernels.

Drop-in for:
  ResBlock / LatentUpsampler (from_config, from_pretrained, forward)   ltx_video/models/autoencoders/latent_upsampler.py:15-203
  adain_filter_latent                                                  ltx_video/pipelines/pipeline_ltx_video.py:1709-1737
  LTXMultiScalePipeline._upsample_latents                              pipeline_ltx_video.py:1760-1772
Same config keys and parameter names (checkpoint-compatible with ltxv-spatial-upscaler-0.9.7).

Activations are channels-last bf16.  nn.Conv3d(padding=1) / nn.Conv2d(padding=1) are the implicit GEMM of
``ltxmi_conv3d_ndhwc_bf16`` with zero padding in time (``time_pad_zeros``) or a per-frame 3x3 kernel
(``kernel_t = 1``); GroupNorm(32) + SiLU (+ the ResBlock residual) is one statistics pass + one apply pass;
the 2-D pixel shuffle is a 16-byte-vector copy because the upsampling conv's rows are packed (p1 p2 c);
un_normalize / normalize ride on the NCDHW<->NDHWC layout kernels.  The temporal upsampler variants are
not shipped by the reference (ltxv.py:194 loads the spatial one) and are rejected.
"""
import json
import math
from typing import Optional

import torch
from torch import nn

from . import ops

BF16 = torch.bfloat16


class _ConvParams(nn.Module):
    """nn.Conv2d / nn.Conv3d(kernel 3, padding 1) parameters under the reference's names."""

    def __init__(self, cin, cout, dims):
        super().__init__()
        self.dims = dims
        bound = 1.0 / math.sqrt(cin * 3 ** dims)
        self.weight = nn.Parameter((torch.rand(cout, cin, *([3] * dims)) * 2 - 1) * bound)
        self.bias = nn.Parameter((torch.rand(cout) * 2 - 1) * bound)
        self._packed = None

    def packed(self, shuffle2d=False):
        key = (self.weight.data_ptr(), shuffle2d)
        if self._packed is None or self._packed[0] != key:
            with torch.no_grad():
                cout = self.weight.shape[0]
                perm = (0, 2, 3, 1) if self.dims == 2 else (0, 2, 3, 4, 1)
                w = self.weight.permute(*perm).reshape(cout, -1)             # tap-major, cin fastest
                b = self.bias
                if shuffle2d:                                                # rows (c p1 p2) -> (p1 p2 c)
                    c = cout // 4
                    w = w.view(c, 4, -1).transpose(0, 1).reshape(cout, -1)
                    b = b.view(c, 4).transpose(0, 1).reshape(-1)
                self._packed = (key, w.contiguous().to(BF16), b.contiguous().to(BF16))
        return self._packed[1], self._packed[2]

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._packed = None
        return super()._load_from_state_dict(*a, **k)

    def forward(self, x, shuffle2d=False):
        """x NDHWC bf16."""
        w, b = self.packed(shuffle2d)
        return ops.conv3d(x, w, b, causal=False, pad_replicate=False, kernel_t=1 if self.dims == 2 else 3,
                          time_pad_zeros=self.dims == 3)


class _GroupNormParams(nn.Module):
    def __init__(self, groups, ch, eps=1e-5):
        super().__init__()
        self.num_groups, self.eps = groups, eps
        self.weight = nn.Parameter(torch.ones(ch))
        self.bias = nn.Parameter(torch.zeros(ch))


class ResBlock(nn.Module):                                                   # latent_upsampler.py:15-39
    def __init__(self, channels: int, mid_channels: Optional[int] = None, dims: int = 3):
        super().__init__()
        mid_channels = channels if mid_channels is None else mid_channels
        self.dims = dims
        self.conv1 = _ConvParams(channels, mid_channels, dims)
        self.norm1 = _GroupNormParams(32, mid_channels)
        self.conv2 = _ConvParams(mid_channels, channels, dims)
        self.norm2 = _GroupNormParams(32, channels)

    def forward(self, x):
        samples = x.shape[0] * (x.shape[1] if self.dims == 2 else 1)
        h = self.conv1(x)
        h = ops.groupnorm_silu(h, self.norm1.weight, self.norm1.bias, 32, self.norm1.eps, samples=samples, out=h)
        h = self.conv2(h)
        return ops.groupnorm_silu(h, self.norm2.weight, self.norm2.bias, 32, self.norm2.eps, residual=x,
                                  samples=samples, out=h)


class _Upsampler(nn.Module):
    """nn.Sequential(Conv2d(mid, 4 mid, 3, padding=1), PixelShuffleND(2)) with the reference's key ``upsampler.0``."""

    def __init__(self, mid):
        super().__init__()
        setattr(self, "0", _ConvParams(mid, 4 * mid, 2))

    def forward(self, x):
        return ops.pixel_shuffle2d(getattr(self, "0")(x, shuffle2d=True))


class LatentUpsampler(nn.Module):
    def __init__(self, in_channels: int = 128, mid_channels: int = 512, num_blocks_per_stage: int = 4, dims: int = 3,
                 spatial_upsample: bool = True, temporal_upsample: bool = False):
        super().__init__()
        if not spatial_upsample or temporal_upsample:
            if not (spatial_upsample or temporal_upsample):
                raise ValueError("Either spatial_upsample or temporal_upsample must be True")
            raise NotImplementedError("ltxmi.LatentUpsampler: only the spatial upsampler is on this path")
        if dims not in (2, 3):
            raise NotImplementedError("ltxmi.LatentUpsampler: dims must be 2 or 3")
        self.in_channels, self.mid_channels = in_channels, mid_channels
        self.num_blocks_per_stage, self.dims = num_blocks_per_stage, dims
        self.spatial_upsample, self.temporal_upsample = spatial_upsample, temporal_upsample
        self.initial_conv = _ConvParams(in_channels, mid_channels, dims)
        self.initial_norm = _GroupNormParams(32, mid_channels)
        self.res_blocks = nn.ModuleList([ResBlock(mid_channels, dims=dims) for _ in range(num_blocks_per_stage)])
        self.upsampler = _Upsampler(mid_channels)
        self.post_upsample_res_blocks = nn.ModuleList(
            [ResBlock(mid_channels, dims=dims) for _ in range(num_blocks_per_stage)])
        self.final_conv = _ConvParams(mid_channels, in_channels, dims)

    @property
    def dtype(self):
        return self.initial_conv.weight.dtype

    @property
    def device(self):
        return self.initial_conv.weight.device

    def forward_ndhwc(self, x):
        samples = x.shape[0] * (x.shape[1] if self.dims == 2 else 1)
        x = self.initial_conv(x)
        x = ops.groupnorm_silu(x, self.initial_norm.weight, self.initial_norm.bias, 32, self.initial_norm.eps,
                               samples=samples, out=x)
        for blk in self.res_blocks:
            x = blk(x)
        x = self.upsampler(x)
        for blk in self.post_upsample_res_blocks:
            x = blk(x)
        return self.final_conv(x)

    def forward(self, latent, _stats=None):
        """latent [b, c, f, h, w] -> [b, c, f, 2h, 2w] (latent_upsampler.py:109-149).  ``_stats`` = (std, mean)
        fp32 [c]: un_normalize on the way in and normalize on the way out (``_upsample_latents``)."""
        std, mean = _stats if _stats is not None else (None, None)
        x = ops.ncdhw_to_ndhwc(latent.to(BF16), std, mean)
        y = self.forward_ndhwc(x)
        return ops.ndhwc_to_ncdhw(y, 0, self.in_channels, std, mean)

    @classmethod
    def from_config(cls, config):                                            # :151-160
        return cls(in_channels=config.get("in_channels", 4), mid_channels=config.get("mid_channels", 128),
                   num_blocks_per_stage=config.get("num_blocks_per_stage", 4), dims=config.get("dims", 2),
                   spatial_upsample=config.get("spatial_upsample", True),
                   temporal_upsample=config.get("temporal_upsample", False))

    def config(self):                                                        # :162-171
        return {"_class_name": "LatentUpsampler", "in_channels": self.in_channels, "mid_channels": self.mid_channels,
                "num_blocks_per_stage": self.num_blocks_per_stage, "dims": self.dims,
                "spatial_upsample": self.spatial_upsample, "temporal_upsample": self.temporal_upsample}

    @classmethod
    def from_pretrained(cls, pretrained_model_path, *args, device="cuda", dtype=BF16, **kwargs):   # :173-191
        from safetensors import safe_open
        path = str(pretrained_model_path)
        if not path.endswith(".safetensors"):
            raise ValueError(f"unrecognised checkpoint path: {path}")
        sd = {}
        with safe_open(path, framework="pt", device="cpu") as f:
            meta = f.metadata()
            for k in f.keys():
                sd[k] = f.get_tensor(k)
        m = cls.from_config(json.loads(meta["config"]))
        m.load_state_dict(sd, strict=True)
        return m.to(device=device, dtype=dtype).eval()


def adain_filter_latent(latents, reference_latents, factor=1.0):
    """pipeline_ltx_video.py:1709-1737, one kernel (a block per (b, c) plane)."""
    return ops.adain_filter(latents, reference_latents.to(latents.dtype), factor)


def upsample_latents(latent_upsampler: LatentUpsampler, latents, vae):
    """LTXMultiScalePipeline._upsample_latents (:1760-1772): un_normalize_latents -> upsampler ->
    normalize_latents with the VAE's per-channel statistics, both folded into the layout passes."""
    stats = (vae.std_of_means.float().contiguous(), vae.mean_of_means.float().contiguous())
    return latent_upsampler(latents, _stats=stats).to(latents.dtype)
