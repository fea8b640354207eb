"""``CausalVideoAutoencoder`` (decode and encode) on libltxmi kernels.

Drop-in for the reference's decode path:
  CausalVideoAutoencoder / from_config     ltx_video/models/autoencoders/causal_video_autoencoder.py:33-177
  AutoencoderKLWrapper.decode (+tiling)    ltx_video/models/autoencoders/vae.py:193-263, 343-413
  Decoder / UNetMidBlock3D / ResnetBlock3D / DepthToSpaceUpsample   causal_video_autoencoder.py:560-1258
  CausalConv3d                             ltx_video/models/autoencoders/causal_conv3d.py
  vae_decode / un_normalize_latents        ltx_video/models/autoencoders/vae_encode.py:94-165, 239-247
Same config keys, same parameter names (checkpoint-compatible), same ``decode(z, return_dict,
target_shape, timestep)`` signature and return convention.

Inside, activations are channels-last (NDHWC) bf16 so that every 3x3x3 convolution is an
implicit GEMM whose K axis (tap, cin) is contiguous in HBM; the reference's temporal
replicate-pad ``torch.concatenate`` (causal_conv3d.py:46-57) and spatial padding are folded into
the kernel's address computation, PixelNorm + AdaLN + SiLU are one pass, and the
depth-to-space rearranges + residual of DepthToSpaceUpsample are the conv's store pattern.
The encode side (image/video conditioning) reuses the same kernels:
  Encoder / SpaceToDepthDownsample        causal_video_autoencoder.py:317-557, 976-1020
  AutoencoderKLWrapper.encode (+tiling)   vae.py:156-191, 265-341
  vae_encode / normalize_latents          vae_encode.py:22-91, 228-236
strided CausalConv3d is the implicit GEMM with a strided row->position map, patchify writes the
NDHWC layout directly, and SpaceToDepthDownsample's duplicated first frame is a padding parameter.
"""
import math
from dataclasses import dataclass
from typing import Optional

import torch
from torch import nn

from . import ops
from .transformer3d import _CombinedTimestepEmbeddings

BF16 = torch.bfloat16


@dataclass
class DecoderOutput:
    sample: torch.Tensor


class DiagonalGaussianDistribution:
    """The diffusers leaf the reference wraps the encoder moments in (vae.py:308): mean / logvar
    (clamped to [-30, 20]) with ``sample()`` = mean + std * randn and ``mode()`` = mean.  Built from
    the two halves directly (the reference's ``parameters`` tensor is their concatenation)."""

    def __init__(self, mean, logvar):
        self.mean = mean
        if logvar is None:                                   # latent_log_var == "none"
            self.logvar = torch.full_like(mean, -30.0)
            self.deterministic = True
        else:
            self.logvar = torch.clamp(logvar.float(), -30.0, 20.0)
            self.deterministic = False

    @property
    def parameters(self):
        return torch.cat([self.mean, self.logvar.to(self.mean.dtype)], dim=1)

    @property
    def std(self):
        return torch.exp(0.5 * self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None):
        if self.deterministic:
            return self.mean
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=torch.float32)
        return (self.mean.float() + self.std * noise).to(self.mean.dtype)

    def mode(self):
        return self.mean


@dataclass
class AutoencoderKLOutput:
    latent_dist: DiagonalGaussianDistribution


class _Conv3dParams(nn.Module):
    """Holds ``weight [Cout,Cin,k,k,k]`` / ``bias`` under the name ``conv`` like nn.Conv3d."""

    def __init__(self, cin, cout, k):
        super().__init__()
        bound = 1.0 / math.sqrt(cin * k ** 3)
        self.weight = nn.Parameter((torch.rand(cout, cin, k, k, k) * 2 - 1) * bound)
        self.bias = nn.Parameter((torch.rand(cout) * 2 - 1) * bound)


class CausalConv3d(nn.Module):
    """causal_conv3d.py:7-63.  Kernel 3, stride 1 or 2 per axis; time padding by frame replication
    (2 in front when causal, 1+1 otherwise), spatial padding 1 in ``spatial_padding_mode``.
    The packed weight pads Cin up to a multiple of 64 and Cout up to a multiple of 8 with zeros
    (conv_in of the encoder has 48 input channels, its conv_out 129 outputs); the result then has
    ``cout_padded`` channels and the caller reads the first ``out_channels``."""

    def __init__(self, in_channels, out_channels, kernel_size: int = 3, stride=1, dilation=1, groups=1,
                 spatial_padding_mode: str = "zeros", **kwargs):
        super().__init__()
        stride = (stride,) * 3 if isinstance(stride, int) else tuple(stride)
        if kernel_size != 3 or dilation != 1 or groups != 1 or any(s not in (1, 2) for s in stride) \
                or stride[1] != stride[2]:
            raise NotImplementedError("ltxmi.CausalConv3d: only dense 3x3x3 with strides 1/2 is on this path")
        if spatial_padding_mode not in ("zeros", "replicate"):
            raise NotImplementedError(f"spatial_padding_mode {spatial_padding_mode}")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.cin_padded, self.cout_padded = -(-in_channels // 64) * 64, -(-out_channels // 8) * 8
        self.stride = stride
        self.time_kernel_size = 3
        self.pad_replicate = spatial_padding_mode == "replicate"
        self.conv = _Conv3dParams(in_channels, out_channels, 3)
        self._packed = None

    @property
    def weight(self):
        return self.conv.weight

    def packed(self, d2s=False):
        """[Cout, 27*Cin] tap-major bf16 (+ rows re-ordered (p1 p2 p3, c') for the depth-to-space store)."""
        # storage AND version counter of the sources: an in-place weight edit rebuilds the pack, as a reload does
        # (edits through ``.data`` or of inference tensors bump no counter: call invalidate_packed() after those)
        wt, bs = self.conv.weight, self.conv.bias
        key = (wt.data_ptr(), ops.tensor_version(wt), None if bs is None else (bs.data_ptr(), ops.tensor_version(bs)), d2s)
        if self._packed is None or self._packed[0] != key:
            with torch.no_grad():
                w = self.conv.weight.permute(0, 2, 3, 4, 1)                  # [Cout, 3,3,3, Cin]
                b = self.conv.bias
                if self.cin_padded != self.in_channels or self.cout_padded != self.out_channels:
                    w = torch.nn.functional.pad(w, (0, self.cin_padded - self.in_channels, 0, 0, 0, 0, 0, 0,
                                                    0, self.cout_padded - self.out_channels))
                    b = torch.nn.functional.pad(b, (0, self.cout_padded - self.out_channels))
                w = w.reshape(self.cout_padded, -1)
                if d2s:
                    cp = self.out_channels // 8
                    w = w.view(cp, 8, -1).transpose(0, 1).reshape(self.out_channels, -1)
                    b = b.view(cp, 8).transpose(0, 1).reshape(-1)
                self._packed = (key, w.contiguous().to(BF16), b.contiguous().to(BF16))
        return self._packed[1], self._packed[2]

    def invalidate_packed(self):
        """Drop the packed copy (rebuilt on the next forward); needed after edits the key cannot see (``.data``)."""
        self._packed = None

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._packed = None
        return super()._load_from_state_dict(*a, **k)

    def forward(self, x, causal: bool = True, d2s=False, residual=None, add=None, tpad=0, out_T=0, post_norm=None,
                keep_raw=False):
        """x: NDHWC bf16.  post_norm = (scale, shift, eps): the PixelNorm -> AdaLN -> SiLU that follows this convolution
        (ops.conv3d: in its epilogue where the kernel holds all channels of a position, a second launch otherwise);
        keep_raw: return (raw, activated)."""
        w, b = self.packed(d2s)
        if post_norm is not None and self.cout_padded != self.out_channels:
            raise ValueError("ltxmi.CausalConv3d: post_norm on a convolution with padded output channels")
        return ops.conv3d(x, w, b, causal, self.pad_replicate, d2s=d2s, residual=residual, add=add,
                          stride=self.stride, tpad=tpad, out_T=out_T, post_norm=post_norm, keep_raw=keep_raw)


def make_conv_nd(dims, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, causal=False, spatial_padding_mode="zeros", temporal_padding_mode="zeros"):
    """conv_nd_factory.py:9-72, the (dims=3, causal=True) branch -- the only one the VAE uses."""
    if dims != 3 or not causal:
        raise NotImplementedError("ltxmi.make_conv_nd: only dims=3, causal=True is on this path")
    return CausalConv3d(in_channels, out_channels, kernel_size, stride=stride, dilation=dilation, groups=groups,
                        spatial_padding_mode=spatial_padding_mode)


class _ChannelLayerNorm(nn.Module):            # causal_video_autoencoder.py:1068-1077, key ``norm``
    def __init__(self, dim, eps):
        super().__init__()
        self.norm = nn.LayerNorm(dim, eps=eps, elementwise_affine=True)

    def forward(self, x):
        return ops.layernorm_affine(x, self.norm.weight, self.norm.bias, self.norm.eps)


class ResnetBlock3D(nn.Module):
    """causal_video_autoencoder.py:1080-1258 with norm_layer="pixel_norm".
    inject_noise (:1183-1195): the [H, W] noise is drawn on the host side of the kernels (``torch.randn`` on the device,
    from ``noise_generator`` if one is set -- the reference uses the global RNG) and added as a broadcast."""

    noise_generator = None      # class-wide default: the global RNG, as in the reference

    def __init__(self, dims, in_channels, out_channels=None, dropout=0.0, groups=32, eps=1e-6,
                 norm_layer="pixel_norm", inject_noise=False, timestep_conditioning=False,
                 spatial_padding_mode="zeros"):
        super().__init__()
        if norm_layer != "pixel_norm":
            raise NotImplementedError("ltxmi.ResnetBlock3D: pixel_norm only")
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.inject_noise = inject_noise
        if inject_noise:
            self.per_channel_scale1 = nn.Parameter(torch.zeros((in_channels, 1, 1)))      # :1131
            self.per_channel_scale2 = nn.Parameter(torch.zeros((in_channels, 1, 1)))      # :1160
        self.conv1 = CausalConv3d(in_channels, out_channels, 3, spatial_padding_mode=spatial_padding_mode)
        self.conv2 = CausalConv3d(out_channels, out_channels, 3, spatial_padding_mode=spatial_padding_mode)
        if in_channels != out_channels:
            self.conv_shortcut = _Conv3dParams(in_channels, out_channels, 1)
            self.norm3 = _ChannelLayerNorm(in_channels, eps)
        else:
            self.conv_shortcut = None
            self.norm3 = None
        self.timestep_conditioning = timestep_conditioning
        if timestep_conditioning:
            self.scale_shift_table = nn.Parameter(torch.randn(4, in_channels) / in_channels ** 0.5)

    def ada_values(self, B, timestep=None):
        """(shift1, scale1, shift2, scale2), fp32 [B, C] each (None without timestep conditioning):
        ada_values = table[None] + timestep.reshape(B, 4, C)  (:1211-1221)."""
        if not self.timestep_conditioning:
            return (None, None, None, None)
        assert timestep is not None, "should pass timestep with timestep_conditioning=True"
        ada = (self.scale_shift_table.float()[None] + timestep.float().reshape(B, 4, -1))
        return tuple(t.contiguous() for t in ada.unbind(dim=1))

    def forward(self, x, causal: bool = True, timestep=None, ada=None, x_act=None, next_norm=None):
        """x NDHWC; timestep: the mid-block's embedding [B, 4C] (bf16) or None.
        ada: ``ada_values`` worked out by the caller.  x_act: this block's norm1 -> AdaLN -> SiLU of x, already applied by the
        producer of x (in its convolution's epilogue).  next_norm = (scale, shift, eps) of the norm that CONSUMES this block's
        result (the next block's norm1, the decoder's tail): the block then returns (y, activated y), the second riding on
        conv2's epilogue where the kernel can."""
        B = x.shape[0]
        sh1, sc1, sh2, sc2 = self.ada_values(B, timestep) if ada is None else ada
        h = x_act if x_act is not None else ops.pixelnorm_ada_silu(x, sc1, sh1, apply_silu=True)
        if self.inject_noise:
            h = self.conv1(h, causal=causal)
            h = h + self._spatial_noise(h, self.per_channel_scale1)
            h = ops.pixelnorm_ada_silu(h, sc2, sh2, apply_silu=True, out=h)
        else:
            # norm2 -> AdaLN -> SiLU (:1226-1243) rides on conv1: in its epilogue at the 128-channel stage (one wave holds all
            # channels of a position there), as a launch on conv1's result elsewhere
            h = self.conv1(h, causal=causal, post_norm=(sc2, sh2, 1e-8))
        if self.conv_shortcut is not None:
            s = self.norm3(x)
            Bx, T, H, W, C = s.shape
            w1 = self.conv_shortcut.weight.reshape(self.out_channels, C)
            s = ops.gemm(s.view(-1, C), w1, self.conv_shortcut.bias).view(Bx, T, H, W, self.out_channels)
        else:
            s = x
        if self.inject_noise:
            # conv2's noise (:1246) joins the skip tensor: conv2 + noise + skip is one sum, taken in conv2's epilogue
            s = s + self._spatial_noise(s, self.per_channel_scale2)
        if next_norm is not None:
            return self.conv2(h, causal=causal, add=s, post_norm=next_norm, keep_raw=True)
        return self.conv2(h, causal=causal, add=s)            # conv2 + skip add in one epilogue

    def _spatial_noise(self, like, per_channel_scale):
        """_feed_spatial_noise (:1183-1195) in NDHWC: noise[H, W] * scale[C], broadcast over batch and frames."""
        _, _, H, W, C = like.shape
        noise = torch.randn((H, W), device=like.device, dtype=like.dtype, generator=self.noise_generator)
        return noise.view(1, 1, H, W, 1) * per_channel_scale.to(like.dtype).view(1, 1, 1, 1, C)


class UNetMidBlock3D(nn.Module):
    """causal_video_autoencoder.py:805-973 without attention blocks ("res_x")."""

    def __init__(self, dims, in_channels, dropout=0.0, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 norm_layer="pixel_norm", inject_noise=False, timestep_conditioning=False,
                 attention_head_dim=-1, spatial_padding_mode="zeros"):
        super().__init__()
        if attention_head_dim > 0:
            # The reference as shipped cannot run this block either: UNetMidBlock3D.forward hands Attention a bare
            # tensor (:951-956) and AttnProcessor2_0 treats it as the DiT's 1-element list
            # (``hidden_states_wrapper.clear()``, attention.py:1002-1003): AttributeError.  No behaviour to reproduce.
            raise NotImplementedError("ltxmi.UNetMidBlock3D: attn_res_x is unreachable in the reference "
                                      "(its own forward raises AttributeError: 'Tensor' object has no attribute 'clear')")
        self.timestep_conditioning = timestep_conditioning
        if timestep_conditioning:
            self.time_embedder = _CombinedTimestepEmbeddings(in_channels * 4)
        self.res_blocks = nn.ModuleList([
            ResnetBlock3D(dims, in_channels, in_channels, eps=resnet_eps, norm_layer=norm_layer,
                          inject_noise=inject_noise, timestep_conditioning=timestep_conditioning,
                          spatial_padding_mode=spatial_padding_mode) for _ in range(num_layers)])

    def plan(self, B, timestep=None):
        """The blocks' AdaLN values for one forward (a list of ResnetBlock3D.ada_values): worked out ahead of the forward so
        that the producer of this block's input can apply the first norm1 -> AdaLN -> SiLU itself."""
        temb = None
        if self.timestep_conditioning:
            assert timestep is not None, "should pass timestep with timestep_conditioning=True"
            temb = self.time_embedder(timestep.flatten().float())          # [B, 4C]
        return [blk.ada_values(B, temb) for blk in self.res_blocks]

    def forward(self, x, causal: bool = True, timestep=None, plan=None, x_act=None, next_norm=None):
        """plan / x_act / next_norm: see ResnetBlock3D.forward (x_act belongs to the first block, next_norm to the consumer of
        the last block's result; with next_norm the result is (y, activated y))."""
        adas = self.plan(x.shape[0], timestep) if plan is None else plan
        n = len(self.res_blocks)
        for i, blk in enumerate(self.res_blocks):
            nn_ = (adas[i + 1][1], adas[i + 1][0], 1e-8) if i + 1 < n else next_norm
            r = blk(x, causal=causal, ada=adas[i], x_act=x_act, next_norm=nn_)
            x, x_act = r if nn_ is not None else (r, None)
        return (x, x_act) if next_norm is not None else x


class DepthToSpaceUpsample(nn.Module):
    """causal_video_autoencoder.py:1023-1065.  stride (2,2,2) ("compress_all", every shipped config): convolution,
    pixel shuffle, first-frame drop and residual in ONE kernel epilogue.  (2,1,1) / (1,2,2) ("compress_time" /
    "compress_space", :671-684, in no shipped config): the convolution kernel, then the shuffle as a layout copy."""

    def __init__(self, dims, in_channels, stride, residual=False, out_channels_reduction_factor=1,
                 spatial_padding_mode="zeros"):
        super().__init__()
        self.stride = tuple(stride)
        if self.stride not in ((2, 2, 2), (2, 1, 1), (1, 2, 2)):
            raise NotImplementedError(f"ltxmi.DepthToSpaceUpsample: stride {self.stride}")
        prod = self.stride[0] * self.stride[1] * self.stride[2]
        self.out_channels = prod * in_channels // out_channels_reduction_factor
        self.conv = CausalConv3d(in_channels, self.out_channels, 3, spatial_padding_mode=spatial_padding_mode)
        self.residual = residual
        self.out_channels_reduction_factor = out_channels_reduction_factor

    def _shuffle(self, y):
        """PixelShuffleND (pixel_shuffle.py:14-21) on NDHWC: channels (c p1 p2 p3) -> [B, T p1, H p2, W p3, c]."""
        B, T, H, W, Cf = y.shape
        p1, p2, p3 = self.stride
        c = Cf // (p1 * p2 * p3)
        y = y.view(B, T, H, W, c, p1, p2, p3).permute(0, 1, 5, 2, 6, 3, 7, 4)
        return y.reshape(B, T * p1, H * p2, W * p3, c)

    def forward(self, x, causal: bool = True, next_norm=None):
        """next_norm = (scale, shift, eps) of the norm that consumes the result: returns (y, activated y) then."""
        if self.stride == (2, 2, 2):
            if next_norm is not None:
                return self.conv(x, causal=causal, d2s=True, residual=x if self.residual else None, post_norm=next_norm,
                                 keep_raw=True)
            return self.conv(x, causal=causal, d2s=True, residual=x if self.residual else None)
        if next_norm is not None:
            y = self.forward(x, causal=causal)
            return y, ops.pixelnorm_ada_silu(y, next_norm[0], next_norm[1], True, next_norm[2])
        y = self._shuffle(self.conv(x, causal=causal))
        if self.stride[0] == 2:
            y = y[:, 1:]
        if self.residual:
            prod = self.stride[0] * self.stride[1] * self.stride[2]
            x_in = self._shuffle(x).repeat(1, 1, 1, 1, prod // self.out_channels_reduction_factor)
            if self.stride[0] == 2:
                x_in = x_in[:, 1:]
            y = y + x_in
        return y.contiguous()


class SpaceToDepthDownsample(nn.Module):
    """causal_video_autoencoder.py:976-1020.  The block's stride-1 convolution runs over the input
    with its first frame duplicated (stride_t == 2): here that is the same convolution with three
    replicated frames in front and T+1 output frames -- no concatenated copy -- and the two
    rearranges + group-mean skip + add are one pass over the result."""

    def __init__(self, dims, in_channels, out_channels, stride, spatial_padding_mode="zeros"):
        super().__init__()
        self.stride = tuple(stride)
        prod = self.stride[0] * self.stride[1] * self.stride[2]
        self.group_size = in_channels * prod // out_channels
        self.conv = CausalConv3d(in_channels, out_channels // prod, 3, spatial_padding_mode=spatial_padding_mode)

    def forward(self, x, causal: bool = True):
        T = x.shape[1]
        if self.stride[0] == 2:
            y = self.conv(x, causal=causal, tpad=3 if causal else 2, out_T=T + 1)
        else:
            y = self.conv(x, causal=causal)
        return ops.space_to_depth_skip(y, x, self.stride)


class Encoder(nn.Module):
    """causal_video_autoencoder.py:317-557 (dims=3, pixel_norm)."""

    _STRIDES = {"compress_time": (2, 1, 1), "compress_space": (1, 2, 2), "compress_all": (2, 2, 2),
                "compress_all_x_y": (2, 2, 2), "compress_all_res": (2, 2, 2), "compress_space_res": (1, 2, 2),
                "compress_time_res": (2, 1, 1)}

    def __init__(self, dims=3, in_channels=3, out_channels=3, blocks=(("res_x", 1),), base_channels=128,
                 norm_num_groups=32, patch_size=1, norm_layer="group_norm", latent_log_var="per_channel",
                 spatial_padding_mode="zeros"):
        super().__init__()
        if dims != 3 or norm_layer != "pixel_norm":
            raise NotImplementedError("ltxmi.Encoder: dims=3 with pixel_norm only")
        self.patch_size = patch_size
        self.norm_layer = norm_layer
        self.latent_channels = out_channels
        self.latent_log_var = latent_log_var
        self.blocks_desc = blocks
        ch = base_channels
        self.conv_in = make_conv_nd(dims, in_channels * patch_size ** 2, ch, 3, 1, 1, causal=True,
                                    spatial_padding_mode=spatial_padding_mode)
        self.down_blocks = nn.ModuleList([])
        for name, params in blocks:
            cin = ch
            if isinstance(params, int):
                params = {"num_layers": params}
            if name == "res_x":
                blk = UNetMidBlock3D(dims, cin, num_layers=params["num_layers"], resnet_eps=1e-6,
                                     norm_layer=norm_layer, spatial_padding_mode=spatial_padding_mode)
            elif name == "res_x_y":
                ch = params.get("multiplier", 2) * ch
                blk = ResnetBlock3D(dims, cin, ch, eps=1e-6, norm_layer=norm_layer,
                                    spatial_padding_mode=spatial_padding_mode)
            elif name in ("compress_time", "compress_space", "compress_all", "compress_all_x_y"):
                if name == "compress_all_x_y":
                    ch = params.get("multiplier", 2) * ch
                blk = make_conv_nd(dims, cin, ch, 3, stride=self._STRIDES[name], causal=True,
                                   spatial_padding_mode=spatial_padding_mode)
            elif name in ("compress_all_res", "compress_space_res", "compress_time_res"):
                ch = params.get("multiplier", 2) * ch
                blk = SpaceToDepthDownsample(dims, cin, ch, self._STRIDES[name], spatial_padding_mode)
            else:
                raise ValueError(f"unknown block: {name}")
            self.down_blocks.append(blk)
        conv_out_channels = out_channels
        if latent_log_var == "per_channel":
            conv_out_channels *= 2
        elif latent_log_var in ("uniform", "constant"):
            conv_out_channels += 1
        elif latent_log_var != "none":
            raise ValueError(f"Invalid latent_log_var: {latent_log_var}")
        self.conv_out = make_conv_nd(dims, ch, conv_out_channels, 3, padding=1, causal=True,
                                     spatial_padding_mode=spatial_padding_mode)

    def forward(self, sample):
        """sample: pixels NCDHW bf16.  Returns conv_out's result NDHWC [B,f,h,w,cout_padded]; the
        latent_log_var expansion of :531-555 is a view decision left to ``moments_ncdhw``."""
        x = ops.patchify_to_ndhwc(sample, self.patch_size, self.conv_in.cin_padded)
        x = self.conv_in(x, causal=True)
        for blk in self.down_blocks:
            x = blk(x, causal=True)
        x = ops.pixelnorm_ada_silu(x, None, None, apply_silu=True, out=x)
        return self.conv_out(x, causal=True)

    def moments_ncdhw(self, y, std=None, mean=None):
        """(mean, logvar) NCDHW from conv_out's NDHWC result, per ``latent_log_var`` (:531-555);
        ``std``/``mean`` (fp32 [C]) fold normalize_latents (vae_encode.py:228-236) into the mean's layout pass."""
        C = self.latent_channels
        mu = ops.ndhwc_to_ncdhw(y, 0, C, std, mean)
        if self.latent_log_var == "per_channel":
            logvar = ops.ndhwc_to_ncdhw(y, C, C)
        elif self.latent_log_var == "uniform":
            logvar = ops.ndhwc_to_ncdhw(y, C, 1).expand(-1, C, -1, -1, -1)
        elif self.latent_log_var == "constant":
            logvar = torch.full_like(mu, -30.0)
        else:
            logvar = None
        return mu, logvar


class Decoder(nn.Module):
    """causal_video_autoencoder.py:560-802."""

    def __init__(self, dims, in_channels=3, out_channels=3, blocks=(("res_x", 1),), base_channels=128,
                 layers_per_block=2, norm_num_groups=32, patch_size=1, norm_layer="group_norm", causal=True,
                 timestep_conditioning=False, spatial_padding_mode="zeros"):
        super().__init__()
        if dims != 3 or norm_layer != "pixel_norm":
            raise NotImplementedError("ltxmi.Decoder: dims=3 with pixel_norm only")
        self.patch_size = patch_size
        self.out_channels_rgb = out_channels
        out_channels = out_channels * patch_size ** 2
        self.causal = causal
        self.blocks_desc = blocks
        ch = base_channels
        for name, params in reversed(list(blocks)):
            params = params if isinstance(params, dict) else {}
            if name == "res_x_y":
                ch *= params.get("multiplier", 2)
            if name == "compress_all":
                ch *= params.get("multiplier", 1)
        self.conv_in = make_conv_nd(dims, in_channels, ch, 3, 1, 1, causal=True,
                                    spatial_padding_mode=spatial_padding_mode)
        self.up_blocks = nn.ModuleList([])
        for name, params in reversed(list(blocks)):
            cin = ch
            if isinstance(params, int):
                params = {"num_layers": params}
            if name == "res_x":
                blk = UNetMidBlock3D(dims, cin, num_layers=params["num_layers"], resnet_eps=1e-6,
                                     norm_layer=norm_layer, inject_noise=params.get("inject_noise", False),
                                     timestep_conditioning=timestep_conditioning,
                                     spatial_padding_mode=spatial_padding_mode)
            elif name == "res_x_y":
                ch = ch // params.get("multiplier", 2)
                blk = ResnetBlock3D(dims, cin, ch, eps=1e-6, norm_layer=norm_layer,
                                    inject_noise=params.get("inject_noise", False), timestep_conditioning=False,
                                    spatial_padding_mode=spatial_padding_mode)
            elif name == "compress_all":
                ch = ch // params.get("multiplier", 1)
                blk = DepthToSpaceUpsample(dims, cin, (2, 2, 2), residual=params.get("residual", False),
                                           out_channels_reduction_factor=params.get("multiplier", 1),
                                           spatial_padding_mode=spatial_padding_mode)
            elif name == "compress_time":                                                # :671-677
                blk = DepthToSpaceUpsample(dims, cin, (2, 1, 1), spatial_padding_mode=spatial_padding_mode)
            elif name == "compress_space":                                               # :678-684
                blk = DepthToSpaceUpsample(dims, cin, (1, 2, 2), spatial_padding_mode=spatial_padding_mode)
            elif name == "attn_res_x":
                blk = UNetMidBlock3D(dims, cin, num_layers=params["num_layers"], norm_layer=norm_layer,
                                     inject_noise=params.get("inject_noise", False),
                                     timestep_conditioning=timestep_conditioning,
                                     attention_head_dim=params["attention_head_dim"],
                                     spatial_padding_mode=spatial_padding_mode)          # raises: see UNetMidBlock3D
            else:
                raise ValueError(f"unknown layer: {name}")
            self.up_blocks.append(blk)
        self.conv_out = make_conv_nd(dims, ch, out_channels, 3, padding=1, causal=True,
                                     spatial_padding_mode=spatial_padding_mode)
        self.timestep_conditioning = timestep_conditioning
        if timestep_conditioning:
            self.timestep_scale_multiplier = nn.Parameter(torch.tensor(1000.0, dtype=torch.float32))
            self.last_time_embedder = _CombinedTimestepEmbeddings(ch * 2)
            self.last_scale_shift_table = nn.Parameter(torch.randn(2, ch) / ch ** 0.5)

    def forward(self, sample, target_shape, timestep: Optional[torch.Tensor] = None):
        """sample: NDHWC bf16 latents (already un-normalised).  Returns pixels NCDHW bf16."""
        assert target_shape is not None, "target_shape must be provided"
        B = sample.shape[0]
        x = self.conv_in(sample, causal=self.causal)
        scaled_t = None
        if self.timestep_conditioning:
            assert timestep is not None, "should pass timestep with timestep_conditioning=True"
            scaled_t = timestep.to(torch.float32) * self.timestep_scale_multiplier.float()
        sc = sh = None
        if self.timestep_conditioning:
            emb = self.last_time_embedder(scaled_t.flatten())                    # [B, 2C]
            ada = self.last_scale_shift_table.float()[None] + emb.float().reshape(B, 2, -1)
            sh, sc = [t.contiguous() for t in ada.unbind(dim=1)]
        # Every PixelNorm -> AdaLN -> SiLU in front of a convolution is handed to the PRODUCER of its input as `next_norm`: the
        # producer returns (raw, activated), from its own epilogue where its kernel holds all channels of a position (the
        # 128-channel stage: 3 of the passes over the largest tensors of a decode), from a launch of its own otherwise.
        blocks = list(self.up_blocks)
        plans = {i: blk.plan(B, scaled_t if self.timestep_conditioning else None)
                 for i, blk in enumerate(blocks) if isinstance(blk, UNetMidBlock3D)}
        x_act = None
        for i, blk in enumerate(blocks):
            nxt = blocks[i + 1] if i + 1 < len(blocks) else None
            if nxt is None:
                nn_ = (sc, sh, 1e-8)
            elif isinstance(nxt, UNetMidBlock3D):
                nn_ = (plans[i + 1][0][1], plans[i + 1][0][0], 1e-8)
            elif isinstance(nxt, ResnetBlock3D):
                nn_ = (None, None, 1e-8)
            else:
                nn_ = None
            if isinstance(blk, UNetMidBlock3D):
                r = blk(x, causal=self.causal, plan=plans[i], x_act=x_act, next_norm=nn_)
            elif isinstance(blk, ResnetBlock3D):
                r = blk(x, causal=self.causal, x_act=x_act, next_norm=nn_)
            else:
                r = blk(x, causal=self.causal, next_norm=nn_)
            x, x_act = r if nn_ is not None else (r, None)
        x = self.conv_out(x_act, causal=self.causal)
        return ops.unpatchify_to_ncdhw(x, self.out_channels_rgb, self.patch_size)


class _Stats(nn.Module):
    pass


class CausalVideoAutoencoder(nn.Module):
    def __init__(self, decoder: Decoder, latent_channels=128, dims=3, config=None, encoder: Optional[Encoder] = None):
        super().__init__()
        if encoder is not None:
            self.encoder = encoder
        self.decoder = decoder
        self.dims = dims
        self._config = dict(config or {})
        self.per_channel_statistics = _Stats()
        self.per_channel_statistics.register_buffer("std-of-means", torch.ones(latent_channels))
        self.per_channel_statistics.register_buffer("mean-of-means", torch.zeros(latent_channels))
        self.use_z_tiling = False
        self.use_hw_tiling = False
        self.z_sample_size = 1
        self.set_tiling_params(sample_size=512, overlap_factor=0.25)

    # ---- construction ------------------------------------------------------------------
    @staticmethod
    def from_config(config):                                            # causal_video_autoencoder.py:123-177
        assert config["_class_name"] == "CausalVideoAutoencoder", "config must have _class_name=CausalVideoAutoencoder"
        dims = tuple(config["dims"]) if isinstance(config["dims"], list) else config["dims"]
        if config.get("use_quant_conv", True):
            raise NotImplementedError("use_quant_conv=True is not on this path (LTX-Video VAEs use False)")
        if config.get("normalize_latent_channels", False):
            raise NotImplementedError("normalize_latent_channels is not on this path")
        decoder = Decoder(dims=dims, in_channels=config["latent_channels"], out_channels=config.get("out_channels", 3),
                          blocks=config.get("decoder_blocks", config.get("blocks")),
                          patch_size=config.get("patch_size", 1), norm_layer=config.get("norm_layer", "group_norm"),
                          causal=config.get("causal_decoder", False),
                          timestep_conditioning=config.get("timestep_conditioning", False),
                          base_channels=config.get("decoder_base_channels", 128),
                          spatial_padding_mode=config.get("spatial_padding_mode", "zeros"))
        encoder = None
        if config.get("encoder_blocks", config.get("blocks")) is not None and config.get("build_encoder", True):
            double_z = config.get("double_z", True)
            encoder = Encoder(dims=dims, in_channels=config.get("in_channels", 3), out_channels=config["latent_channels"],
                              blocks=config.get("encoder_blocks", config.get("blocks")),
                              patch_size=config.get("patch_size", 1),
                              latent_log_var=config.get("latent_log_var", "per_channel" if double_z else "none"),
                              norm_layer=config.get("norm_layer", "group_norm"),
                              base_channels=config.get("encoder_base_channels", 128),
                              spatial_padding_mode=config.get("spatial_padding_mode", "zeros"))
        return CausalVideoAutoencoder(decoder, latent_channels=config["latent_channels"], dims=dims, config=config,
                                      encoder=encoder)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *args, device="cuda", dtype=BF16, **kwargs):
        """causal_video_autoencoder.py:34-120 (decode side): diffusers directory or single file."""
        from .loading import load_vae
        return load_vae(pretrained_model_name_or_path, device=device, dtype=dtype)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):   # :248-298
        if any(k.startswith("vae.") for k in state_dict.keys()):
            state_dict = {k.replace("vae.", ""): v for k, v in state_dict.items() if k.startswith("vae.")}
        remap = {".resnets.": ".res_blocks.", "downsamplers.0": "downsample", "upsamplers.0": "upsample"}
        out = {}
        for k, v in state_dict.items():
            for a, b in remap.items():
                k = k.replace(a, b)
            if k.startswith("encoder.") and not hasattr(self, "encoder"):
                continue                                   # decode-only instance
            out[k] = v
        return super().load_state_dict(out, strict=strict, assign=assign)

    @property
    def std_of_means(self):
        return getattr(self.per_channel_statistics, "std-of-means")

    @property
    def mean_of_means(self):
        return getattr(self.per_channel_statistics, "mean-of-means")

    @property
    def dtype(self):
        return self.decoder.conv_in.conv.weight.dtype

    @property
    def device(self):
        return self.decoder.conv_in.conv.weight.device

    @property
    def spatial_downscale_factor(self):                                 # :207-225 (decoder blocks mirror the encoder's)
        n = len([b for b in self.decoder.blocks_desc if b[0] in ("compress_space", "compress_all")])
        return 2 ** n * self.decoder.patch_size

    @property
    def temporal_downscale_factor(self):                                # :227-241
        return 2 ** len([b for b in self.decoder.blocks_desc if b[0] in ("compress_time", "compress_all")])

    # ---- tiling knobs (vae.py:91-154) --------------------------------------------------
    @staticmethod
    def get_VAE_tile_size(vae_config, device_mem_capacity, mixed_precision):     # vae.py:91-115
        z_tile = 4
        if vae_config == 0:
            if mixed_precision:
                device_mem_capacity = device_mem_capacity / 1.5
            use = 1 if device_mem_capacity >= 24000 else (2 if device_mem_capacity >= 8000 else 3)
        else:
            use = vae_config
        return (z_tile, 0 if use == 1 else (512 if use == 2 else 256))

    def set_tiling_params(self, sample_size: int = 512, overlap_factor: float = 0.25):
        self.tile_sample_min_size = sample_size
        self.tile_latent_min_size = int(sample_size / 32)
        self.tile_overlap_factor = overlap_factor

    def enable_z_tiling(self, z_sample_size: int = 4):
        self.use_z_tiling = z_sample_size > 1
        self.z_sample_size = z_sample_size
        assert z_sample_size % 4 == 0 or z_sample_size == 1, \
            f"z_sample_size must be a multiple of 4 or 1. Got {z_sample_size}."

    def disable_z_tiling(self):
        self.use_z_tiling = False

    def enable_hw_tiling(self):
        self.use_hw_tiling = True

    def disable_hw_tiling(self):
        self.use_hw_tiling = False

    # ---- encode ------------------------------------------------------------------------
    def _encode(self, x, stats=None):                                            # vae.py:337-341
        """pixels NCDHW -> (mean, logvar) NCDHW; ``stats`` folds normalize_latents into the mean."""
        if not hasattr(self, "encoder"):
            raise RuntimeError("ltxmi.CausalVideoAutoencoder: this instance was built without an encoder")
        std, mean = stats if stats is not None else (None, None)
        return self.encoder.moments_ncdhw(self.encoder(x.to(BF16)), std, mean)

    @staticmethod
    def _cat_moments(mv):
        mu, lv = mv
        return mu if lv is None else torch.cat([mu, lv], dim=1)

    def _hw_tile_grid(self, fn, x, tile, stride, blend_extent, keep):
        """The spatial tiling shared by the tiled encode and decode (vae.py:156-191, 223-263): ``fn`` is applied to
        ``tile`` x ``tile`` windows of the last two axes taken every ``stride`` positions; each result is cross-faded over
        ``blend_extent`` with its upper and left neighbour (blend_v / blend_h, in this order), cropped to ``keep`` and the
        crops are stitched back together."""
        H, W = x.shape[3], x.shape[4]
        grid = [[fn(x[:, :, :, top:top + tile, left:left + tile]) for left in range(0, W, stride)]
                for top in range(0, H, stride)]
        bands = []
        for r, row in enumerate(grid):
            done = []
            for c, t in enumerate(row):
                if r:
                    t = self._blend(grid[r - 1][c], t, blend_extent, 3)
                if c:
                    t = self._blend(row[c - 1], t, blend_extent, 4)
                row[c] = t                       # (the right / lower neighbours blend against the blended tile, as there)
                done.append(t[:, :, :, :keep, :keep])
            bands.append(torch.cat(done, dim=4))
        return torch.cat(bands, dim=3)

    def _hw_tiled_encode(self, x):                                               # vae.py:156-191
        blend_extent = int(self.tile_latent_min_size * self.tile_overlap_factor)
        return self._hw_tile_grid(lambda t: self._cat_moments(self._encode(t)), x, self.tile_sample_min_size,
                                  int(self.tile_sample_min_size * (1 - self.tile_overlap_factor)), blend_extent,
                                  self.tile_latent_min_size - blend_extent)

    def encode(self, z, return_dict: bool = True, _stats=None):
        """vae.py:265-312: pixels [B,3,F,H,W] -> AutoencoderKLOutput(latent_dist).  ``_stats`` (std, mean)
        is vae_encode's per-channel normalisation, applied to the mean in the untiled case."""
        if self.use_z_tiling and z.shape[2] > (self.z_sample_size + 1) > 1:
            tl = self.z_sample_size
            ts = tl * 8
            overlap_size = int(ts * 0.75)
            blend_extent = int(tl * 0.25)
            t_limit = tl - blend_extent
            row = []
            for i in range(0, z.shape[2], overlap_size):
                tile = z[:, :, i:i + ts + 1]
                tile = self._hw_tiled_encode(tile) if self.use_hw_tiling else self._cat_moments(self._encode(tile))
                if i > 0:
                    tile = tile[:, :, 1:].contiguous()          # (the blend kernel writes into it in place)
                row.append(tile)
            result = []
            for i, tile in enumerate(row):
                if i > 0:
                    tile = self._blend(row[i - 1], tile, blend_extent, 2)
                    result.append(tile[:, :, :t_limit])
                else:
                    result.append(tile[:, :, :t_limit + 1])
            moments = torch.cat(result, dim=2)
        elif self.use_hw_tiling and z.shape[2] > 1:
            moments = self._hw_tiled_encode(z)
        else:
            moments = None
            mu, logvar = self._encode(z, stats=_stats)
        if moments is not None:
            C = self.encoder.latent_channels
            mu, logvar = moments[:, :C], (moments[:, C:] if moments.shape[1] > C else None)
            if _stats is not None:
                std, mean = _stats
                mu = ((mu.float() - mean.view(1, -1, 1, 1, 1)) / std.view(1, -1, 1, 1, 1)).to(mu.dtype)
        posterior = DiagonalGaussianDistribution(mu, logvar)
        if not return_dict:
            return (posterior,)
        return AutoencoderKLOutput(latent_dist=posterior)

    # ---- decode ------------------------------------------------------------------------
    def _decode(self, z, target_shape=None, timestep=None, stats=None):          # vae.py:343-355
        std, mean = stats if stats is not None else (None, None)
        x = ops.ncdhw_to_ndhwc(z.to(BF16), std, mean)
        return self.decoder(x, target_shape=target_shape, timestep=timestep)

    @staticmethod
    def _blend(a, b, extent, dim):
        """blend_z / blend_v / blend_h (vae.py:193-221): linear cross-fade of the overlap, written into ``b`` -- one kernel
        over the overlap (``ltxmi_tile_blend``) instead of the reference's per-slice loop."""
        if a.dtype != b.dtype:
            a = a.to(b.dtype)
        if not b.is_contiguous():
            raise ValueError("tile blend: the tile written to must be contiguous")
        return ops.tile_blend_(a.contiguous(), b, extent, dim)

    def _hw_tiled_decode(self, z, target_shape, timestep=None, stats=None):      # vae.py:223-263
        blend_extent = int(self.tile_sample_min_size * self.tile_overlap_factor)
        return self._hw_tile_grid(lambda t: self._decode(t, target_shape=target_shape, timestep=timestep, stats=stats), z,
                                  self.tile_latent_min_size, int(self.tile_latent_min_size * (1 - self.tile_overlap_factor)),
                                  blend_extent, self.tile_sample_min_size - blend_extent)

    def _hw_tiled_extent(self, n_latent):
        """Pixels ``_hw_tiled_decode`` returns along an axis of ``n_latent`` latent positions (every tile cropped to
        ``row_limit`` before the concatenation, vae.py:249-262)."""
        ss = self.spatial_downscale_factor
        stride = int(self.tile_latent_min_size * (1 - self.tile_overlap_factor))
        row_limit = self.tile_sample_min_size - int(self.tile_sample_min_size * self.tile_overlap_factor)
        return sum(min(min(self.tile_latent_min_size, n_latent - i) * ss, row_limit) for i in range(0, n_latent, stride))

    def _decoded_tile_shape(self, zshape, drop_first):
        """Shape of one decoded z-tile, from its latent slice alone (so that ranks that did not decode it can allocate
        it without being told): [B, 3, 8 (L - 1) + 1 (- 1 for the dropped frame), 32 H, 32 W]."""
        B, _, L, H, W = zshape
        frames = (L - 1) * self.temporal_downscale_factor + 1 - (1 if drop_first else 0)
        ss = self.spatial_downscale_factor
        hw = (self._hw_tiled_extent(H), self._hw_tiled_extent(W)) if self.use_hw_tiling else (H * ss, W * ss)
        return (B, self.decoder.out_channels_rgb, frames) + hw

    def decode(self, z, return_dict: bool = True, target_shape=None, timestep: Optional[torch.Tensor] = None,
               _stats=None, _tile_exchange=None):
        """vae.py:357-413.  With z-tiling the tiles are kept on the device (the reference's
        ``.to(float16).cpu()`` per tile at :388 was a VRAM workaround); the result is fp16 as there.
        ``_tile_exchange`` (extension, ltxmi.distributed.tile_parallel_vae_decode): ``(decoders, shapes) -> tiles`` -- the
        z-tiles are independent until the blends, so ranks can decode different tiles and exchange them;
        ``decoders[n](out)`` writes tile n (fp16, ``shapes[n]``) into ``out``."""
        assert target_shape is not None, "target_shape must be provided for decoding"

        def dec(t):
            if self.use_hw_tiling:
                return self._hw_tiled_decode(t, target_shape, timestep, _stats)
            return self._decode(t, target_shape=target_shape, timestep=timestep, stats=_stats)

        if self.use_z_tiling and z.shape[2] > (self.z_sample_size + 1) > 1:
            tl = self.z_sample_size
            ts = tl * 8
            overlap_size = int(tl * 0.75)
            blend_extent = int(ts * 0.25)
            t_limit = ts - blend_extent
            starts = list(range(0, z.shape[2], overlap_size))

            def tile(i, out=None):
                d = dec(z[:, :, i:i + tl + 1])
                if i > 0:
                    d = d[:, :, 1:]
                if out is None:
                    return d.to(torch.float16).contiguous()
                if tuple(out.shape) != tuple(d.shape):
                    raise RuntimeError(f"tiled decode: tile at latent frame {i} is {tuple(d.shape)}, predicted {tuple(out.shape)}")
                out.copy_(d)
                return out

            if _tile_exchange is None:
                row = [tile(i) for i in starts]
            else:
                shapes = [self._decoded_tile_shape(z[:, :, i:i + tl + 1].shape, i > 0) for i in starts]
                row = _tile_exchange([(lambda out, i=i: tile(i, out)) for i in starts], shapes)
            result = []
            for i, t in enumerate(row):
                if i > 0:
                    t = self._blend(row[i - 1], t, blend_extent, 2)
                    result.append(t[:, :, :t_limit])
                else:
                    result.append(t[:, :, :t_limit + 1])
            decoded = torch.cat(result, dim=2)
        else:
            decoded = dec(z)
        if not return_dict:
            return (decoded,)
        return DecoderOutput(sample=decoded)


def _scaling_factor(vae):
    cfg = getattr(vae, "_config", None)
    return float(cfg.get("scaling_factor", 1.0)) if cfg is not None else float(vae.config.scaling_factor)


def normalize_latents(latents, vae, vae_per_channel_normalize: bool = False):
    """vae_encode.py:228-236, for callers that hold latents outside the encode / decode calls (inside them the
    statistics ride on the layout kernels)."""
    if vae_per_channel_normalize:
        mean = vae.mean_of_means.to(latents.dtype).to(latents.device).view(1, -1, 1, 1, 1)
        std = vae.std_of_means.to(latents.dtype).to(latents.device).view(1, -1, 1, 1, 1)
        return (latents - mean) / std
    return latents * _scaling_factor(vae)


def un_normalize_latents(latents, vae, vae_per_channel_normalize: bool = False):
    """vae_encode.py:239-247."""
    if vae_per_channel_normalize:
        mean = vae.mean_of_means.to(latents.dtype).to(latents.device).view(1, -1, 1, 1, 1)
        std = vae.std_of_means.to(latents.dtype).to(latents.device).view(1, -1, 1, 1, 1)
        return latents * std + mean
    return latents / _scaling_factor(vae)


def vae_encode(media_items, vae: CausalVideoAutoencoder, split_size: int = 1, vae_per_channel_normalize=False,
               generator: Optional[torch.Generator] = None, sample_posterior: bool = True):
    """vae_encode.py:22-91: pixels [B,3,F,H,W] (or [B,3,H,W]) in [-1,1] -> normalised latents [B,C,f,h,w].
    ``latent_dist.sample()`` as there (``sample_posterior=False`` takes the mode).  The per-channel
    normalisation commutes with the sampling as (mean - m)/s + (std/s) * noise; it is folded into the
    encoder's layout pass for the mean and applied to the noise term here."""
    if media_items.dim() == 4:
        media_items = media_items.unsqueeze(2)
    if media_items.shape[1] != 3:
        raise ValueError(f"Expects tensors with 3 channels, got {media_items.shape[1]}.")
    if split_size != 1:
        raise NotImplementedError("split_size > 1 is not on this path")
    stats = None
    if vae_per_channel_normalize:
        stats = (vae.std_of_means.float().contiguous(), vae.mean_of_means.float().contiguous())
    dist = vae.encode(media_items.to(vae.dtype), _stats=stats).latent_dist
    scale = 1.0 if vae_per_channel_normalize else float(vae._config.get("scaling_factor", 1.0))
    z = dist.mean.float() * scale
    if sample_posterior and not dist.deterministic:
        noise = torch.randn(z.shape, generator=generator, device=z.device, dtype=torch.float32)
        s = dist.std * scale
        if stats is not None:
            s = s / stats[0].view(1, -1, 1, 1, 1)
        z = z + s * noise
    return z.to(vae.dtype)


def vae_decode(latents, vae: CausalVideoAutoencoder, is_video: bool = True, split_size: int = 1,
               vae_per_channel_normalize=False, timestep=None, _tile_exchange=None):
    """vae_encode.py:94-165: un-normalise (per-channel std/mean, fused into the layout kernel)
    and decode.  latents [B,C,F,H,W]."""
    if split_size != 1:
        raise NotImplementedError("split_size > 1 is not on this path")
    *_, fl, hl, wl = latents.shape
    ts, ss = vae.temporal_downscale_factor, vae.spatial_downscale_factor
    stats = None
    if vae_per_channel_normalize:
        stats = (vae.std_of_means.float().contiguous(), vae.mean_of_means.float().contiguous())
    else:
        sf = float(vae._config.get("scaling_factor", 1.0))
        if sf != 1.0:                                    # un_normalize_latents: latents / scaling_factor (vae_encode.py:246)
            latents = latents / sf
    return vae.decode(latents.to(vae.dtype), return_dict=False,
                      target_shape=(1, 3, fl * ts if is_video else 1, hl * ss, wl * ss),
                      timestep=timestep, _stats=stats, _tile_exchange=_tile_exchange)[0]
