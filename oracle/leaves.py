"""Oracle restatement of the third-party (diffusers) leaves the reference calls.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED for every
function here except ``get_timestep_embedding`` (pinned by the reference's own copy,
ltx_video/models/transformers/embeddings.py:10-50): diffusers >= 0.31.0
(requirements.txt:4) is not installed and not vendored, so these follow the
package's published definitions; the call sites that constrain them are cited.

All functions are functional: ``sd`` is a flat state dict using the reference's
parameter names, ``p`` the key prefix (with trailing dot).
"""
import math

import torch
import torch.nn.functional as F


def linear(x, sd, p):
    """nn.Linear with reference key names ``{p}weight`` / ``{p}bias``."""
    return F.linear(x, sd[p + "weight"], sd.get(p + "bias"))


def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False,
                           downscale_freq_shift=1.0, scale=1.0, max_period=10000):
    """Sinusoid; follows ltx_video/models/transformers/embeddings.py:10-50."""
    assert timesteps.ndim == 1
    half = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32,
                                                    device=timesteps.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


def timestep_embedding_mlp(t_proj, sd, p):
    """diffusers TimestepEmbedding: linear_1 -> SiLU -> linear_2
    (keys pinned by checkpoints: adaln_single.emb.timestep_embedder.linear_{1,2})."""
    h = linear(t_proj, sd, p + "linear_1.")
    h = F.silu(h)
    return linear(h, sd, p + "linear_2.")


def combined_timestep_size_embeddings(timestep, sd, p, hidden_dtype):
    """diffusers PixArtAlphaCombinedTimestepSizeEmbeddings with
    use_additional_conditions=False: Timesteps(256, flip_sin_to_cos=True,
    downscale_freq_shift=0) -> TimestepEmbedding.
    Call sites: transformer3d.py:146-148 (via AdaLayerNormSingle),
    causal_video_autoencoder.py:728-730, 852-854."""
    proj = get_timestep_embedding(timestep, 256, flip_sin_to_cos=True, downscale_freq_shift=0.0)
    return timestep_embedding_mlp(proj.to(hidden_dtype), sd, p + "timestep_embedder.")


def adaln_single(timestep, sd, p, hidden_dtype):
    """diffusers AdaLayerNormSingle.forward -> (linear(silu(emb)), emb).
    Call site: transformer3d.py:428-433."""
    emb = combined_timestep_size_embeddings(timestep, sd, p + "emb.", hidden_dtype)
    return linear(F.silu(emb), sd, p + "linear."), emb


def text_projection(caption, sd, p):
    """diffusers PixArtAlphaTextProjection (act_fn="gelu_tanh").
    Call site: transformer3d.py:154-156, 448."""
    h = linear(caption, sd, p + "linear_1.")
    h = F.gelu(h, approximate="tanh")
    return linear(h, sd, p + "linear_2.")


def rms_norm(x, eps, weight=None):
    """diffusers RMSNorm.forward: fp32 variance, x * rsqrt(var + eps) promoted to
    fp32, then (a) with weight in half precision: cast to the weight dtype and
    multiply, (b) no weight: cast back to the input dtype.
    Call sites: attention.py:119-126,168 (norm1/norm2, no affine, eps=norm_eps)
    and attention.py:478-479 (q_norm/k_norm over heads*dim_head, eps=1e-5)."""
    in_dtype = x.dtype
    var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    h = x * torch.rsqrt(var + eps)
    if weight is not None:
        if weight.dtype in (torch.float16, torch.bfloat16):
            h = h.to(weight.dtype)
        h = h * weight
    else:
        h = h.to(in_dtype)
    return h


def gelu_proj(x, sd, p, approximate="tanh"):
    """diffusers activations.GELU: proj (Linear) then gelu.  attention.py:1296-1297."""
    return F.gelu(linear(x, sd, p + "proj."), approximate=approximate)
