"""Sequence parallelism (Ulysses) for ``Transformer3DModel`` over RCCL/xGMI.

The reference carries an unbound USP implementation for its Wan model built on the third-party
``xfuser`` package (wan/distributed/xdit_context_parallel.py:66-192: ``usp_dit_forward`` chunks the
sequence :131-133, ``usp_attn_forward`` swaps sequence<->head sharding around attention :179-184 via
xFuserLongContextAttention, final ``all_gather`` :142).  LTX has no counterpart there; this module
provides the same two entry points for the LTX DiT, written directly on ``torch.distributed``
(backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests):

  * every op of a block except self-attention is token-local (per-token RMSNorm/AdaLN, RMSNorm
    across heads, RoPE with sliced tables, cross-attention against the replicated 256 text tokens,
    FF), so rank r simply owns N/P contiguous tokens;
  * inside self-attention each rank owns H/P heads over ALL tokens: ONE packed all-to-all carries
    q,k,v ([B, N/P, 3, H, dh] -> [B, N, 3, H/P, dh]) and one carries o back -- 2 collectives per
    layer instead of xfuser's 4.  xGMI is point-to-point, so an all-to-all uses all 7 links of a GPU
    at once (each peer pair its own link) and is not ring/per-link bound;
  * the model output [B, N/P, C] is all-gathered once per forward.

Requires H % P == 0 and N % P == 0 (and, for per-frame timesteps, whole frames per rank).
"""
from typing import Callable, Optional

import torch
import torch.distributed as dist

from .attention import BasicTransformerBlock, SkipLayerStrategy


# ------------------------------------------------------------------ layout + collectives
def shard_tokens(x, rank, world, dim=1):
    """Contiguous N/P slice of the token axis (xdit_context_parallel.py:131-133)."""
    n = x.shape[dim]
    if n % world != 0:
        raise ValueError(f"token count {n} is not divisible by the sequence-parallel degree {world}")
    return x.narrow(dim, rank * (n // world), n // world)


def gather_tokens(x, group=None, dim=1):
    """all_gather along the token axis (xdit_context_parallel.py:142)."""
    world = dist.get_world_size(group)
    parts = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(parts, x.contiguous(), group=group)
    return torch.cat(parts, dim=dim)


def seq_to_head_shard(qkv, group=None):
    """[B, N/P, 3, H, dh] (all heads, local tokens) -> [B, N, 3, H/P, dh] (local heads, all tokens)."""
    world = dist.get_world_size(group)
    B, Nl, three, H, dh = qkv.shape
    if H % world != 0:
        raise ValueError(f"heads {H} not divisible by the sequence-parallel degree {world}")
    Hl = H // world
    # destination-major send buffer: chunk j = heads [j*Hl, (j+1)*Hl)
    send = qkv.view(B, Nl, three, world, Hl, dh).permute(3, 0, 1, 2, 4, 5).contiguous()
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    # source-major receive buffer: chunk i = tokens [i*Nl, (i+1)*Nl)
    return recv.permute(1, 0, 2, 3, 4, 5).reshape(B, world * Nl, three, Hl, dh)


def head_to_seq_shard(o, group=None):
    """[B, N, H/P, dh] -> [B, N/P, H, dh] (inverse of seq_to_head_shard for the attention output)."""
    world = dist.get_world_size(group)
    B, N, Hl, dh = o.shape
    Nl = N // world
    send = o.view(B, world, Nl, Hl, dh).permute(1, 0, 2, 3, 4).contiguous()      # chunk j = tokens of rank j
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    return recv.permute(1, 2, 0, 3, 4).reshape(B, Nl, world * Hl, dh)            # chunk i = heads of rank i


def _default_attention(q, k, v, softmax_scale):
    from . import ops
    return ops.attention(q, k, v, softmax_scale=softmax_scale)


def usp_attn_forward(qkv, softmax_scale, group=None, attn_fn: Optional[Callable] = None):
    """Ulysses self-attention on a packed, already normed/roped projection buffer.
    qkv: [B, N/P, 3, H, dh] -> returns [B, N/P, H, dh]."""
    attn_fn = attn_fn or _default_attention
    full = seq_to_head_shard(qkv, group)
    o = attn_fn(full[:, :, 0], full[:, :, 1], full[:, :, 2], softmax_scale)
    return head_to_seq_shard(o.contiguous(), group)


# ------------------------------------------------------------------ processor + model forward
class UlyssesAttnProcessor:
    """Replaces AttnProcessor2_0 on ``attn1`` of every block (installed with
    ``Attention.set_processor``): identical math, plus the two all-to-alls."""

    def __init__(self, group=None):
        self.group = group

    def __call__(self, attn, hidden_states_wrapper, freqs_cis=None, encoder_hidden_states=None,
                 attention_mask=None, temb=None, skip_layer_mask=None, skip_layer_strategy=None,
                 fused_residual=None, *args, **kwargs):
        from . import ops
        from .attention import _host_mask
        hidden_states = hidden_states_wrapper[0]
        hidden_states_wrapper.clear()
        assert encoder_hidden_states is None, "UlyssesAttnProcessor is for self-attention"
        B, Nl, _ = hidden_states.shape
        D, H = attn.inner_dim, attn.heads
        dh = D // H
        wqkv, bqkv, _, _ = attn.packed()
        qkv = ops.gemm(hidden_states.reshape(B * Nl, -1), wqkv, bqkv)
        cos, sin = freqs_cis                                   # already sliced to the local tokens
        cos2, sin2 = cos.reshape(-1, D), sin.reshape(-1, D)
        ops.rmsnorm_rope_(qkv[:, :D], attn.q_norm.weight, attn.q_norm.eps, cos2, sin2, cos2.shape[0])
        ops.rmsnorm_rope_(qkv[:, D:2 * D], attn.k_norm.weight, attn.k_norm.eps, cos2, sin2, cos2.shape[0])
        a4 = usp_attn_forward(qkv.view(B, Nl, 3, H, dh), attn.scale, self.group)
        a3 = a4.reshape(B, Nl, D)
        host_mask = _host_mask(skip_layer_mask) if skip_layer_mask is not None else None
        if host_mask is not None and any(m != 1.0 for m in host_mask):
            m_dev = skip_layer_mask.reshape(B).to(torch.float32)
            if skip_layer_strategy == SkipLayerStrategy.AttentionValues:
                ops.stg_blend_(a3, qkv.view(B, Nl, 3 * D)[:, :, 2 * D:], m_dev)
            elif skip_layer_strategy == SkipLayerStrategy.AttentionSkip:
                ops.stg_blend_(a3, hidden_states, m_dev)
        w_o, b_o = attn.to_out[0].weight, attn.to_out[0].bias
        if fused_residual is not None:
            residual, gate_table, gate_temb, rpg = fused_residual
            ops.gemm(a3.reshape(B * Nl, D), w_o, b_o, out=residual.reshape(B * Nl, -1),
                     epilogue=ops.EPI_GATE_RESIDUAL, residual=residual.reshape(B * Nl, -1),
                     gate_table=gate_table, gate_temb=gate_temb, rows_per_group=rpg)
            return residual
        return ops.gemm(a3.reshape(B * Nl, D), w_o, b_o).view(B, Nl, -1)


def enable_sequence_parallel(model, group=None):
    """Install the Ulysses processor on every block's self-attention."""
    for blk in model.transformer_blocks:
        assert isinstance(blk, BasicTransformerBlock)
        blk.attn1.set_processor(UlyssesAttnProcessor(group))
    model._sp_group = group
    return model


def usp_dit_forward(model, hidden_states, freqs_cis, encoder_hidden_states=None, timestep=None,
                    encoder_attention_mask=None, skip_layer_mask=None, skip_layer_strategy=None,
                    latent_shape=None, group=None, **kw):
    """Sequence-parallel ``Transformer3DModel.forward``: shard tokens (and the per-token inputs that
    follow them), run the model on the local shard, all-gather the output.  Every rank passes the
    FULL inputs and receives the FULL output, so it is a drop-in for ``model(...)``."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    N = hidden_states.shape[1]
    hs = shard_tokens(hidden_states, rank, world)
    fc = tuple(shard_tokens(t, rank, world).contiguous() for t in freqs_cis)
    if timestep is not None and timestep.shape[-1] > 1:
        hw = latent_shape[-2] * latent_shape[-1]
        if (N // world) % hw != 0:
            raise ValueError("per-token timesteps need whole latent frames per rank")
        timestep = shard_tokens(timestep, rank, world)
        latent_shape = (latent_shape[0] // world,) + tuple(latent_shape[1:])
    out = model(hs, freqs_cis=fc, encoder_hidden_states=encoder_hidden_states, timestep=timestep,
                encoder_attention_mask=encoder_attention_mask, skip_layer_mask=skip_layer_mask,
                skip_layer_strategy=skip_layer_strategy, latent_shape=latent_shape, return_dict=False, **kw)
    if out[0] is None:
        return [None]
    return (gather_tokens(out[0], group),)
