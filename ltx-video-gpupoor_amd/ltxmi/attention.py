"""Transformer block of the LTX-Video DiT on libltxmi kernels.

Mirrors ltx_video/models/transformers/attention.py of the reference:
``BasicTransformerBlock`` (:45-364), ``Attention`` (:368-975), ``AttnProcessor2_0`` (:978-1173),
``FeedForward`` (:1263-1323) -- same constructor arguments, same parameter names (so the
reference's checkpoints load with ``load_state_dict``), same processor protocol
(``processor(attn, hidden_states_wrapper, freqs_cis=, encoder_hidden_states=, attention_mask=,
skip_layer_mask=, skip_layer_strategy=)`` with the 1-element list that the callee clears).

What differs is where the arithmetic runs:
  * norm1/norm2 + ``*= 1+scale; += shift``      -> one ltxmi_norm_modulate_bf16 pass
  * to_q/to_k/to_v                              -> one GEMM against the packed [3D, D] weight
  * q_norm/k_norm (RMSNorm over all heads) + RoPE -> one in-place ltxmi_rmsnorm_rope_bf16 pass each
  * SDPA                                        -> ltxmi_attention_fwd_bf16 reading q/k/v in place
  * to_out + ``*= gate`` + ``hidden += ..``     -> GEMM with the GATE_RESIDUAL epilogue (in place)
  * ff.net[0] (+GELU-tanh), ff.net[2] + gate + residual -> two GEMMs with fused epilogues
The token-chunking of the feed-forward (:333-343) existed only to save VRAM and is dropped.
"""
from enum import Enum, auto
from typing import Optional

import torch
from torch import nn

from . import ops
from .attention_seam import pay_attention

BF16 = torch.bfloat16


FUSE_CROSS_ATTENTION_Q = True      # see AttnProcessor2_0 (cross-attention branch)


class SkipLayerStrategy(Enum):          # ltx_video/utils/skip_layer_strategy.py:4-8
    AttentionSkip = auto()
    AttentionValues = auto()
    Residual = auto()
    TransformerBlock = auto()


def reshape_hidden_states(hidden_states, latent_frames):        # attention.py:36-37
    return hidden_states.reshape(hidden_states.shape[0], latent_frames, -1, hidden_states.shape[-1])


def restore_hidden_states_shape(hidden_states):                 # attention.py:40-41
    return hidden_states.reshape(hidden_states.shape[0], -1, hidden_states.shape[-1])


class RMSNorm(nn.Module):
    """Parameter holder with the diffusers RMSNorm interface used by the reference
    (attention.py:119-126, 478-479); the arithmetic is fused into the kernels above."""

    def __init__(self, dim, eps: float, elementwise_affine: bool = True):
        super().__init__()
        self.eps = eps
        self.dim = dim
        if elementwise_affine:
            self.weight = nn.Parameter(torch.ones(dim))
        else:
            self.register_parameter("weight", None)


class _NoAffineLayerNorm(nn.Module):
    def __init__(self, dim, eps: float, elementwise_affine: bool = False):
        super().__init__()
        if elementwise_affine:
            raise NotImplementedError("ltxmi: standardisation norms with affine parameters are not on this path")
        self.eps = eps
        self.dim = dim


class GELUProj(nn.Module):
    """diffusers activations.GELU(dim_in, dim_out, approximate="tanh"): key ``proj``."""

    def __init__(self, dim_in, dim_out, approximate="tanh", bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)
        self.approximate = approximate


class FeedForward(nn.Module):
    """attention.py:1263-1323; keys ``net.0.proj`` and ``net.2``."""

    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False,
                 inner_dim=None, bias=True):
        super().__init__()
        if activation_fn != "gelu-approximate":
            raise NotImplementedError(f"ltxmi: activation_fn '{activation_fn}' is not on this path "
                                      "(LTX-Video uses 'gelu-approximate')")
        inner_dim = int(dim * mult) if inner_dim is None else inner_dim
        dim_out = dim if dim_out is None else dim_out
        self.net = nn.ModuleList([GELUProj(dim, inner_dim, "tanh", bias), nn.Dropout(dropout),
                                  nn.Linear(inner_dim, dim_out, bias=bias)])

    def forward(self, hidden_states, scale: float = 1.0):
        h = ops.gemm(hidden_states, self.net[0].proj.weight, self.net[0].proj.bias, epilogue=ops.EPI_GELU_TANH)
        out = ops.gemm(h, self.net[2].weight, self.net[2].bias)
        return out.view(*hidden_states.shape[:-1], out.shape[-1])


class Attention(nn.Module):
    """attention.py:368-975 (the subset LTX-Video instantiates: no group/spatial norm, no added
    KV, no LoRA).  Parameters: to_q, to_k, to_v, to_out.0, q_norm, k_norm."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, out_bias=True, scale_qk=True, qk_norm=None,
                 eps=1e-5, rescale_output_factor=1.0, residual_connection=False, processor=None, out_dim=None,
                 use_tpu_flash_attention=False, use_rope=False, **unsupported):
        super().__init__()
        for k, v in unsupported.items():
            if v not in (None, False, 32):
                raise NotImplementedError(f"ltxmi.Attention: argument {k}={v} is not on this path")
        if use_tpu_flash_attention:
            raise NotImplementedError("ltxmi.Attention: TPU flash attention does not exist on MI355X")
        self.inner_dim = out_dim if out_dim is not None else dim_head * heads
        self.query_dim = query_dim
        self.use_bias = bias
        self.is_cross_attention = cross_attention_dim is not None
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.rescale_output_factor = rescale_output_factor
        self.residual_connection = residual_connection
        self.out_dim = out_dim if out_dim is not None else query_dim
        self.use_tpu_flash_attention = False
        self.use_rope = use_rope
        self.scale = dim_head ** -0.5 if scale_qk else 1.0
        self.heads = out_dim // dim_head if out_dim is not None else heads
        if qk_norm is None:
            self.q_norm, self.k_norm = nn.Identity(), nn.Identity()
        elif qk_norm == "rms_norm":
            self.q_norm = RMSNorm(dim_head * heads, eps=1e-5)
            self.k_norm = RMSNorm(dim_head * heads, eps=1e-5)
        else:
            raise NotImplementedError(f"ltxmi.Attention: qk_norm '{qk_norm}' is not on this path")
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, self.out_dim, bias=out_bias), nn.Dropout(dropout)])
        self._packs = {}
        self.set_processor(processor if processor is not None else AttnProcessor2_0())

    def set_processor(self, processor) -> None:                 # attention.py:575-595
        self.processor = processor

    def get_processor(self, return_deprecated_lora: bool = False):
        return self.processor

    # ---- packed projection weights (product-side layout; built lazily, rebuilt when a source changes)
    def _pack(self, slot, mods):
        """Contiguous [sum(out), in] weight (and bias) of ``mods``, cached per slot.  The key holds storage AND version
        counter of every source tensor, so an in-place edit that autograd sees (``weight.copy_``, ``+=``, ``add_`` --
        what a LoRA merge does) rebuilds the pack just like a reload does; the caches that derive from it (the
        per-layer text K/V) key on the pack and follow.  NOT seen: edits through ``weight.data`` (no counter is
        bumped) and any edit of an inference tensor (it has no counter, ``ops.tensor_version``): call
        ``invalidate_packed()`` after those."""
        srcs = [t for m in mods for t in (m.weight, m.bias) if t is not None]
        key = tuple((t.data_ptr(), ops.tensor_version(t), t.dtype, t.device) for t in srcs)
        hit = self._packs.get(slot)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                w = torch.cat([m.weight for m in mods], dim=0).contiguous()
                b = torch.cat([m.bias for m in mods], dim=0).contiguous() if mods[0].bias is not None else None
            hit = (key, w, b)
            self._packs[slot] = hit
        return hit[1], hit[2]

    def packed_qkv(self):
        """[to_q; to_k; to_v] -> ([3D, Din], [3D]) for self-attention (one projection GEMM)."""
        if self.cross_attention_dim != self.query_dim:
            raise ValueError("ltxmi.Attention.packed_qkv: query and key/value inputs differ")
        return self._pack("qkv", [self.to_q, self.to_k, self.to_v])

    def packed_kv(self):
        """[to_k; to_v] -> ([2D, Dkv], [2D]) for cross-attention (the text keys / values in one GEMM)."""
        return self._pack("kv", [self.to_k, self.to_v])

    def invalidate_packed(self):
        """Drop the packed copies (they are rebuilt on the next forward)."""
        self._packs = {}
        self.__dict__.pop("_text_kv_cache", None)

    def _load_from_state_dict(self, *a, **k):
        self.invalidate_packed()
        return super()._load_from_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self.invalidate_packed()
        return super()._apply(fn, *a, **k)

    def forward(self, hidden_states, freqs_cis=None, encoder_hidden_states=None, attention_mask=None,
                skip_layer_mask=None, skip_layer_strategy=None, fused_residual=None, **cross_attention_kwargs):
        """``hidden_states`` is the reference's 1-element list wrapper (attention.py:269-272);
        a bare tensor is accepted too (UNetMidBlock3D calls it that way, :951-956)."""
        wrapper = hidden_states if isinstance(hidden_states, list) else [hidden_states]
        return self.processor(self, wrapper, freqs_cis=freqs_cis, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, skip_layer_mask=skip_layer_mask,
                              skip_layer_strategy=skip_layer_strategy, fused_residual=fused_residual)


def _host_mask(mask):
    """Per-batch skip mask as a python list (the reference reads it with .item()/.min() on the
    hot path, attention.py:228,1061; we read it once per forward, see Transformer3DModel)."""
    if mask is None:
        return None
    host = getattr(mask, "_ltxmi_host", None)
    return host if host is not None else [float(x) for x in mask.flatten().tolist()]


class AttnProcessor2_0:
    """attention.py:978-1173 on libltxmi kernels.

    ``fused_residual = (residual, gate_table, gate_temb, rows_per_group)`` (an extension used
    by BasicTransformerBlock) folds ``to_out`` + gate + ``residual += ..`` into one GEMM
    epilogue that updates ``residual`` in place and returns it; without it the plain
    ``to_out[0]`` result is returned exactly like the reference."""

    def __call__(self, attn: Attention, hidden_states_wrapper, freqs_cis, encoder_hidden_states=None,
                 attention_mask=None, temb=None, skip_layer_mask=None, skip_layer_strategy=None,
                 fused_residual=None, *args, **kwargs):
        hidden_states = hidden_states_wrapper[0]
        hidden_states_wrapper.clear()
        if hidden_states.dim() != 3:
            raise NotImplementedError("ltxmi.AttnProcessor2_0: only [B, N, D] inputs are on this path")
        B, N, _ = hidden_states.shape
        D = attn.inner_dim
        H = attn.heads
        dh = D // H
        x2 = hidden_states.reshape(B * N, -1)
        is_cross = encoder_hidden_states is not None
        # only the pack this call uses is built: self-attention never materialises [to_k; to_v], cross-attention never
        # [to_q; to_k; to_v] (1.2 GB of HBM on the 2B model otherwise)
        if is_cross:
            wkv, bkv = attn.packed_kv()
        else:
            wqkv, bqkv = attn.packed_qkv()
        if not isinstance(attn.q_norm, RMSNorm):
            raise NotImplementedError("ltxmi.AttnProcessor2_0: qk_norm=None is not on this path")

        q_fused = None                      # (row sums of squares, weight, eps), (cos, sin, period): q finished on load
        if not is_cross:
            # K1, the fused QKV projection + q/k RMSNorm + RoPE (attention.py:1040-1059).  Where the attention kernel can
            # finish q while it loads it, the projection GEMM also emits q's per-row sums of squares and q gets NO pass of
            # its own: the only launch between the projection and attention is k's norm + RoPE.
            fuse_q = ops.attention_fuses_qnorm(B, H, N, N, dh) and D % 64 == 0 and attention_mask is None
            ss = torch.empty((B * N, D // 64), dtype=torch.float32, device=x2.device) if fuse_q else None
            qkv = ops.gemm(x2, wqkv, bqkv, rowsumsq=ss, rowsumsq_cols=D if fuse_q else 0)   # [B*N, 3D]
            q2, k2, v2 = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
            cos = sin = None
            period = 0
            if attn.use_rope:
                cos, sin = freqs_cis
                if cos.shape[0] not in (1, B):
                    raise ValueError("freqs_cis batch must be 1 or the batch size")
                cos2, sin2 = cos.reshape(-1, D), sin.reshape(-1, D)
                period = cos2.shape[0]                  # N (shared) or B*N (per sample)
                cos, sin = cos2, sin2
            if fuse_q:
                # k's pass also finalises q's row factor (one float per row) from the GEMM's partial sums: the attention
                # kernel's (head, query tile) workgroups then read 4 bytes per row instead of re-summing 32 partials each
                rstd = torch.empty((B * N,), dtype=torch.float32, device=x2.device)
                ops.rmsnorm_rope_(k2, attn.k_norm.weight, attn.k_norm.eps, cos, sin, period,
                                  rstd_of=(ss, D, attn.q_norm.eps, rstd))
                q_fused = ((rstd, attn.q_norm.weight, attn.q_norm.eps), (cos, sin, period) if cos is not None else None)
            else:
                ops.rmsnorm_rope_(q2, attn.q_norm.weight, attn.q_norm.eps, cos, sin, period)
                ops.rmsnorm_rope_(k2, attn.k_norm.weight, attn.k_norm.eps, cos, sin, period)
            Lk = N
            q4 = qkv.view(B, N, 3, H, dh)[:, :, 0]
            k4 = qkv.view(B, N, 3, H, dh)[:, :, 1]
            v4 = qkv.view(B, N, 3, H, dh)[:, :, 2]
            v3 = qkv.view(B, N, 3 * D)[:, :, 2 * D:]
        else:
            Bk, Lk, _ = encoder_hidden_states.shape
            # q's RMSNorm is applied by the attention kernel while it loads q, as in self-attention.  With 256 text keys the
            # kernel's short life is mostly prologue, and with every head's workgroups re-deriving the row factor from 32
            # partials the fusion did not pay (round 2: 83.2 us fused against 20.9 + 60.9 us).  With the factor finalised per
            # row by a launch of its own it does: 8.4 + 67.3 us against 21.2 + 63.2 us (tools/xattn_time.py; -0.28 ms per
            # step, tools/bench_xattn_fuse.py).
            fuse_q = FUSE_CROSS_ATTENTION_Q and D % 64 == 0 and bool(ops.attention_fuses_qnorm(B, H, N, Lk, dh, attention_mask is not None))
            ss = torch.empty((B * N, D // 64), dtype=torch.float32, device=x2.device) if fuse_q else None
            q2 = ops.gemm(x2, attn.to_q.weight, attn.to_q.bias, rowsumsq=ss, rowsumsq_cols=D if fuse_q else 0)   # [B*N, D]
            # The text keys/values depend on the prompt and this layer's weights only: within a generation
            # they are the same at every denoise step, so they are projected + normalised once and reused
            # (same kernels on the same inputs: identical values).  Keyed on storage + version of the inputs;
            # the source tensors are kept alive so an address cannot be reused while the entry exists.
            ehs = encoder_hidden_states
            key = (ehs.data_ptr(), tuple(ehs.shape), ops.tensor_version(ehs), wkv.data_ptr(), ops.tensor_version(wkv),
                   attn.k_norm.weight.data_ptr(), ops.tensor_version(attn.k_norm.weight))
            cache = attn.__dict__.setdefault("_text_kv_cache", {})
            hit = cache.get(key) if ops.STEP_INVARIANT_CACHING else None
            if hit is None:
                # this forward's stacked projection (Transformer3DModel._stacked_text_kv): rows of the FULL batch's text states;
                # a block run on the leading rows of the batch takes the leading rows
                # or on a micro-batch of rows (sequence-parallel overlap mode) takes ITS rows: any whole-row offset into the
                # full tensor is served; the entry lives until the end of the forward (Transformer3DModel.forward drops it)
                ready = attn.__dict__.get("_text_kv_ready")
                if ready is not None:
                    full, full_version, kv_full = ready
                    row_bytes = full.stride(0) * full.element_size()
                    off = ehs.data_ptr() - full.data_ptr()
                    if (row_bytes > 0 and off >= 0 and off % row_bytes == 0 and ehs.shape[1:] == full.shape[1:]
                            and off // row_bytes + Bk <= full.shape[0] and ehs.dtype == full.dtype and ehs.is_contiguous()
                            and ops.tensor_version(full) == full_version):
                        r0 = off // row_bytes
                        hit = (kv_full[r0 * Lk:(r0 + Bk) * Lk],)
            if hit is None:
                kv = ops.gemm(ehs.reshape(Bk * Lk, -1), wkv, bkv)                  # [B*Lk, 2D]
                ops.rmsnorm_rope_(kv[:, :D], attn.k_norm.weight, attn.k_norm.eps)
                if len(cache) >= 4:
                    cache.clear()
                if ops.STEP_INVARIANT_CACHING:
                    cache[key] = (kv, ehs, wkv)
            else:
                kv = hit[0]
            if fuse_q:
                # one factor per row (a 3 us launch over 2 MB) instead of a 122 MB pass over q
                q_fused = ((ops.rowsumsq_rstd(ss, D, attn.q_norm.eps), attn.q_norm.weight, attn.q_norm.eps), None)
            else:
                ops.rmsnorm_rope_(q2, attn.q_norm.weight, attn.q_norm.eps)
            q4 = q2.view(B, N, H, dh)
            k4 = kv.view(Bk, Lk, 2, H, dh)[:, :, 0]
            v4 = kv.view(Bk, Lk, 2, H, dh)[:, :, 1]
            v3 = kv.view(Bk, Lk, 2 * D)[:, :, D:]

        host_mask = _host_mask(skip_layer_mask) if skip_layer_mask is not None else None
        skip_attention = (host_mask is not None and skip_layer_strategy == SkipLayerStrategy.AttentionValues
                          and len(host_mask) == 1 and host_mask[0] == 0)           # attention.py:1060-1062
        if skip_attention:
            a3 = v3.contiguous()
        else:
            mask4 = None
            if attention_mask is not None:
                # [B,1,T] bias (transformer3d.py:411-415); the reference repeats it over heads
                # (attention.py:1026-1033) -- here it stays a per-key bias, broadcast in-kernel.
                if attention_mask.dim() != 3 or attention_mask.shape[1] != 1:
                    raise NotImplementedError("ltxmi.AttnProcessor2_0: attention_mask must be a [B,1,Lk] bias")
                mask4 = attention_mask.reshape(B, 1, 1, Lk)
            a4 = pay_attention([q4, k4, v4], attention_mask=mask4, softmax_scale=attn.scale,
                               q_norm=q_fused[0] if q_fused else None, rope=q_fused[1] if q_fused else None)   # [B,N,H,dh]
            a3 = a4.view(B, N, D)
            if host_mask is not None and any(m != 1.0 for m in host_mask):
                m_dev = skip_layer_mask.reshape(B).to(torch.float32)
                if skip_layer_strategy == SkipLayerStrategy.AttentionValues:
                    ops.stg_blend_(a3, v3, m_dev)                                   # :1134-1141
                elif skip_layer_strategy == SkipLayerStrategy.AttentionSkip:
                    ops.stg_blend_(a3, hidden_states, m_dev)                        # :1127-1133

        w_o, b_o = attn.to_out[0].weight, attn.to_out[0].bias
        if fused_residual is not None:
            residual, gate_table, gate_temb, rpg = fused_residual
            ops.gemm(a3.reshape(B * N, D), w_o, b_o, out=residual.reshape(B * N, -1),
                     epilogue=ops.EPI_GATE_RESIDUAL, residual=residual.reshape(B * N, -1),
                     gate_table=gate_table, gate_temb=gate_temb, rows_per_group=rpg)
            return residual
        out = ops.gemm(a3.reshape(B * N, D), w_o, b_o).view(B, N, -1)
        if attn.residual_connection or attn.rescale_output_factor != 1.0:
            raise NotImplementedError("ltxmi.AttnProcessor2_0: residual_connection / rescale_output_factor")
        return out


class BasicTransformerBlock(nn.Module):
    """attention.py:45-364 for adaptive_norm="single_scale_shift" (AdaLN-Zero), the only
    mode LTX-Video's checkpoints use.  Updates ``hidden_states`` IN PLACE like the reference
    (:288, :310) and returns it."""

    def __init__(self, dim, num_attention_heads, attention_head_dim, dropout=0.0, cross_attention_dim=None,
                 activation_fn="geglu", num_embeds_ada_norm=None, attention_bias=False, only_cross_attention=False,
                 double_self_attention=False, upcast_attention=False, norm_elementwise_affine=True,
                 adaptive_norm="single_scale_shift", standardization_norm="layer_norm", norm_eps=1e-5,
                 qk_norm=None, final_dropout=False, attention_type="default", ff_inner_dim=None, ff_bias=True,
                 attention_out_bias=True, use_tpu_flash_attention=False, use_rope=False):
        super().__init__()
        if adaptive_norm != "single_scale_shift" or only_cross_attention or double_self_attention:
            raise NotImplementedError("ltxmi.BasicTransformerBlock: only the AdaLN-Zero "
                                      "(single_scale_shift) self+cross block is on this path")
        assert standardization_norm in ("layer_norm", "rms_norm")
        self.adaptive_norm = adaptive_norm
        self.norm_kind = ops.NORM_RMS if standardization_norm == "rms_norm" else ops.NORM_LAYER
        make_norm = RMSNorm if standardization_norm == "rms_norm" else _NoAffineLayerNorm
        if norm_elementwise_affine:
            raise NotImplementedError("ltxmi.BasicTransformerBlock: norm_elementwise_affine=True is not on this path")
        self.norm1 = make_norm(dim, eps=norm_eps, elementwise_affine=False)
        self.attn1 = Attention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim,
                               dropout=dropout, bias=attention_bias, cross_attention_dim=None,
                               out_bias=attention_out_bias, qk_norm=qk_norm, use_rope=use_rope)
        if cross_attention_dim is not None:
            self.attn2 = Attention(query_dim=dim, cross_attention_dim=cross_attention_dim,
                                   heads=num_attention_heads, dim_head=attention_head_dim, dropout=dropout,
                                   bias=attention_bias, out_bias=attention_out_bias, qk_norm=qk_norm,
                                   use_rope=use_rope)
        else:
            self.attn2 = None
        self.norm2 = make_norm(dim, norm_eps, False)
        self.ff = FeedForward(dim, dropout=dropout, activation_fn=activation_fn, final_dropout=final_dropout,
                              inner_dim=ff_inner_dim, bias=ff_bias)
        self.scale_shift_table = nn.Parameter(torch.randn(6, dim) / dim ** 0.5)

    def prepare_shared_state(self):
        """Build, on the CURRENT stream, what the processors create lazily and cache for every later caller (the packed
        projection weights).  Called by the micro-batched block loop before its side streams fork."""
        self.attn1.packed_qkv()
        if self.attn2 is not None:
            self.attn2.packed_kv()

    def forward(self, hidden_states, freqs_cis=None, attention_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, timestep=None, cross_attention_kwargs=None, class_labels=None,
                added_cond_kwargs=None, skip_layer_mask=None, skip_layer_strategy: Optional[SkipLayerStrategy] = None):
        if attention_mask is not None:
            raise NotImplementedError("ltxmi.BasicTransformerBlock: a self-attention mask is not on this path")
        B, N, D = hidden_states.shape
        assert timestep.ndim == 3                                                   # attention.py:237
        T1 = timestep.shape[1]
        if N % T1 != 0:
            raise ValueError(f"per-frame modulation needs tokens ({N}) divisible by frames ({T1})")
        rpg = N // T1
        host_mask = _host_mask(skip_layer_mask)
        if host_mask is not None and min(host_mask) == 1.0:                          # attention.py:228-229
            skip_layer_mask, host_mask = None, None
        table = self.scale_shift_table                                               # [6, D]
        temb = timestep.reshape(B * T1, 6 * D)                                       # row g = b*T1 + f
        h2 = hidden_states.view(B * N, D)

        def chunk(i):                                     # (table row, temb column block) of ada value i
            return table[i], temb[:, i * D:(i + 1) * D]

        # 0/1. norm1 -> modulate(shift_msa=0, scale_msa=1) -> self-attention -> gate_msa(2) + residual
        norm_h = torch.empty_like(hidden_states)
        sc_t, sc_e = chunk(1)
        sh_t, sh_e = chunk(0)
        ops.norm_modulate(h2, norm_h.view(B * N, D), self.norm1.eps, self.norm_kind, sc_t, sc_e, sh_t, sh_e, rpg)
        g_t, g_e = chunk(2)
        self.attn1([norm_h], freqs_cis=freqs_cis, skip_layer_mask=skip_layer_mask,
                   skip_layer_strategy=skip_layer_strategy, fused_residual=(hidden_states, g_t, g_e, rpg))
        del norm_h

        # 3. cross-attention on the un-normalised stream (attention.py:294-311), plain residual
        if self.attn2 is not None:
            self.attn2([hidden_states], freqs_cis=freqs_cis, encoder_hidden_states=encoder_hidden_states,
                       attention_mask=encoder_attention_mask, fused_residual=(hidden_states, None, None, 1))

        block_skip = (host_mask is not None and skip_layer_strategy == SkipLayerStrategy.TransformerBlock)
        if block_skip:
            # the reference's `original_hidden_states` aliases the tensor updated in place above
            # (attention.py:231 vs :288,:310): what gets blended back is THIS state.
            original = hidden_states.clone()

        # 4. norm2 -> modulate(shift_mlp=3, scale_mlp=4) -> FF -> gate_mlp(5) + residual
        norm_h = torch.empty_like(hidden_states)
        sc_t, sc_e = chunk(4)
        sh_t, sh_e = chunk(3)
        ops.norm_modulate(h2, norm_h.view(B * N, D), self.norm2.eps, self.norm_kind, sc_t, sc_e, sh_t, sh_e, rpg)
        ff1 = ops.gemm(norm_h.view(B * N, D), self.ff.net[0].proj.weight, self.ff.net[0].proj.bias,
                       epilogue=ops.EPI_GELU_TANH)
        del norm_h
        g_t, g_e = chunk(5)
        ops.gemm(ff1, self.ff.net[2].weight, self.ff.net[2].bias, out=h2, epilogue=ops.EPI_GATE_RESIDUAL,
                 residual=h2, gate_table=g_t, gate_temb=g_e, rows_per_group=rpg)
        del ff1

        if block_skip:
            ops.stg_blend_(hidden_states, original, skip_layer_mask.reshape(B).to(torch.float32))
        return hidden_states
