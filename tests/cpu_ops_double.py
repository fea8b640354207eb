"""A CPU stand-in for ``ltxmi.ops`` -- TEST INFRASTRUCTURE ONLY.

The product's host logic (Transformer3DModel.forward, the attention processors, the Ulysses sharding of
ltxmi/distributed.py) calls its kernels through ``ltxmi.ops``.  ``install()`` swaps the kernel-calling functions of that
module for plain-torch CPU functions with the SAME argument meaning (fp32 arithmetic, one rounding to the output dtype,
in-place where the kernel is in place, strided views, K-blocked operands, segmented outputs ...), so that this host
logic can run in the CPU container: multi-process gloo tests of the sequence-parallel path (tests/test_distributed.py)
and CPU tests of the model plumbing against the oracle.  Nothing here is shipped or measured, and the product never
imports it."""
import math

import torch


def _f(t):
    return None if t is None else t.float()


def _rows(t):
    return t.reshape(-1, t.shape[-1]) if t.is_contiguous() else t


def gemm(a, w, bias=None, out=None, epilogue=0, residual=None, gate_table=None, gate_temb=None, rows_per_group=1,
         algo=0, rowsumsq=None, rowsumsq_cols=0, a_kblock=0, a_kblock_stride=0):
    from ltxmi import ops
    a2 = _rows(a)
    M = a2.shape[0]
    N, K = w.shape
    if a_kblock:
        nblk = K // a_kblock
        blocks = torch.as_strided(a2, (nblk, M, a_kblock), (a_kblock_stride, a2.stride(0), 1))
        a32 = blocks.permute(1, 0, 2).reshape(M, K).float()
    else:
        a32 = a2.float()
    acc = a32 @ w.float().t()
    if bias is not None:
        acc = acc + bias.float()
    if epilogue == ops.EPI_GELU_TANH:
        acc = torch.nn.functional.gelu(acc, approximate="tanh")
    elif epilogue == ops.EPI_SILU:
        acc = torch.nn.functional.silu(acc)
    elif epilogue == ops.EPI_GATE_RESIDUAL:
        if gate_table is not None:
            g = gate_table.float()[None, :] + gate_temb.float().repeat_interleave(rows_per_group, dim=0)[:M]
            acc = acc * g
        acc = acc + _rows(residual).float()
    dtype = a.dtype
    res = acc.to(dtype)
    if rowsumsq is not None:
        nb = rowsumsq_cols // 64
        rowsumsq[:, :nb] = res[:, :rowsumsq_cols].float().reshape(M, nb, 64).pow(2).sum(-1)
    if out is None:
        return res
    _rows(out).copy_(res)
    return out


def norm_modulate(x, out, eps, kind, scale_table, scale_temb, shift_table, shift_temb, rows_per_group):
    from ltxmi import ops
    x2 = _rows(x).float()
    rows, D = x2.shape
    if kind == ops.NORM_LAYER:
        mean = x2.mean(-1, keepdim=True)
        n = (x2 - mean) * torch.rsqrt((x2.pow(2).mean(-1, keepdim=True) - mean * mean).clamp_min(0) + eps)
    else:
        n = x2 * torch.rsqrt(x2.pow(2).mean(-1, keepdim=True) + eps)
    sc = scale_table.float()[None] + scale_temb.float().repeat_interleave(rows_per_group, dim=0)[:rows]
    sh = shift_table.float()[None] + shift_temb.float().repeat_interleave(rows_per_group, dim=0)[:rows]
    _rows(out).copy_((n * (1 + sc) + sh).to(out.dtype))
    return out


def _rope(o, cos, sin, period):
    rows = o.shape[0]
    idx = torch.arange(rows) % period
    c, s = cos.float()[idx], sin.float()[idx]
    r = torch.empty_like(o)
    r[:, 0::2] = o[:, 0::2] * c[:, 0::2] - o[:, 1::2] * s[:, 0::2]
    r[:, 1::2] = o[:, 1::2] * c[:, 1::2] + o[:, 0::2] * s[:, 1::2]
    return r


def rmsnorm_rope_(x, weight, eps, cos=None, sin=None, rope_period=0, rstd_of=None):
    if rstd_of is not None:
        ss, norm_dim, norm_eps, out = rstd_of
        out.copy_(torch.rsqrt(ss.float().sum(-1) / norm_dim + norm_eps))
    x2 = _rows(x)
    o = x2.float()
    o = o * torch.rsqrt(o.pow(2).mean(-1, keepdim=True) + eps) * weight.float()
    if cos is not None:
        o = _rope(o, cos, sin, rope_period or cos.shape[0])
    x2.copy_(o.to(x.dtype))
    return x


def rowsumsq_rstd(ss, norm_dim, eps, out=None):
    r = torch.rsqrt(ss.float().sum(-1) / norm_dim + eps)
    if out is None:
        return r
    out.copy_(r)
    return out


def attention_fuses_qnorm(B, H, Lq, Lk, dh, has_key_bias=False):
    return False            # the double keeps q's normalisation as a pass of its own (both forms are kernel-tested on the GPU)


def attention_kernel_id(B, H, Lq, Lk, dh, has_key_bias=False, k_stride_l=None, v_stride_l=None):
    return 0


def attention(q, k, v, out=None, key_bias=None, softmax_scale=None, q_norm=None, rope=None, out_segments=None):
    assert q_norm is None and rope is None
    B, Lq, H, dh = q.shape
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(dh)
    s = torch.einsum("blhd,bkhd->bhlk", q.float(), k.float()) * scale
    if key_bias is not None:
        s = s + key_bias.float()[:, None, None, :]
    o = torch.einsum("bhlk,bkhd->blhd", torch.softmax(s, dim=-1), v.float()).to(q.dtype).contiguous()
    if out is None:
        return o
    if out_segments is None:
        out.copy_(o)
        return out
    seg, sstride = out_segments
    nseg = Lq // seg
    full = torch.as_strided(out, (nseg, B, seg, H, dh), (sstride, out.stride(0), out.stride(1), out.stride(2), 1))
    full.copy_(o.view(B, nseg, seg, H, dh).permute(1, 0, 2, 3, 4))
    return out


def qkv_norm_rope_pack(qkv, B, Nl, D, P, q_weight, k_weight, eps, cos=None, sin=None, rope_period=0, out=None):
    x = _rows(qkv).float()
    parts = []
    for i, w in ((0, q_weight), (1, k_weight)):
        o = x[:, i * D:(i + 1) * D]
        o = o * torch.rsqrt(o.pow(2).mean(-1, keepdim=True) + eps) * w.float()
        if cos is not None:
            o = _rope(o, cos, sin, rope_period or cos.shape[0])
        parts.append(o)
    parts.append(x[:, 2 * D:])
    Dp = D // P
    t = torch.stack(parts, dim=1).to(qkv.dtype)                      # [B*Nl, 3, D]
    t = t.view(B, Nl, 3, P, Dp).permute(3, 1, 0, 2, 4).contiguous()  # [P, Nl, B, 3, Dp]
    if out is not None:
        out.copy_(t)
        return out
    return t


def silu(x, out=None):
    r = torch.nn.functional.silu(x.float()).to(x.dtype)
    if out is None:
        return r
    out.copy_(r)
    return out


def timestep_embedding(t_f32, dim=256):
    """diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0) -> bf16 [n, dim] (cos first)."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = t_f32[:, None].float() * freqs[None]
    return torch.cat([ang.cos(), ang.sin()], dim=-1).to(torch.bfloat16)


def stg_blend_(a, v, m_f32):
    B = a.shape[0]
    m = m_f32.float().view(B, 1, 1)
    a.copy_((a.float() * m + v.float().reshape(a.shape) * (1 - m)).to(a.dtype))
    return a


def stg_blend_grouped_(a, v, m_f32):
    G, B, L, Dg = a.shape
    m = m_f32.float().view(1, B, 1, 1)
    vv = v[:, :, :G * Dg].float().reshape(B, L, G, Dg).permute(2, 0, 1, 3)
    a.copy_((a.float() * m + vv * (1 - m)).to(a.dtype))
    return a


def guidance_step_(noise_pred, latents, dt, guidance_scale, stg_scale, rescaling_scale, do_cfg, do_stg, do_rescale,
                   workspace, cond_mask=None, t=0.0):
    """CFG-star / STG / std-rescale + Euler (+ conditioning mask), in place on ``latents`` -- the arithmetic of the
    oracle's guidance (pipeline_ltx_video.py:1183-1241, 1309-1342)."""
    from oracle import sched
    v = sched.guidance(noise_pred.float(), noise_pred.shape[0], guidance_scale, stg_scale, rescaling_scale,
                       bool(do_cfg), bool(do_stg), bool(do_rescale))
    new = latents.float() - dt * v
    if cond_mask is not None:
        new = torch.where((t - 1e-6 < 1.0 - cond_mask).unsqueeze(-1), new, latents.float())
    latents.copy_(new.to(latents.dtype))
    return latents


NAMES = ["guidance_step_", "gemm", "norm_modulate", "rmsnorm_rope_", "attention_fuses_qnorm", "attention_kernel_id", "attention", "qkv_norm_rope_pack", "silu",
         "timestep_embedding", "stg_blend_", "stg_blend_grouped_", "rowsumsq_rstd"]


def install():
    """Swap the kernel-calling functions of ltxmi.ops for the CPU functions above (this process only)."""
    from ltxmi import ops
    for n in NAMES:
        setattr(ops, n, globals()[n])
    return ops
