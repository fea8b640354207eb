"""``pay_attention`` -- the attention seam shared by the LTX and Wan paths.

Drop-in for wan/modules/attention.py:162-399 (byte-identical copy utils/attention.py),
restricted to the reference's eager branch (``sdpa``): non-causal softmax(q k^T * scale
[+ additive mask]) v with fp32 accumulation.  The reference's backend selector
(``offload.shared_state["_attention"]``, sage/flash/xformers wheels) is gone: there is one
backend, the gfx950 flash kernel in libltxmi.so, and no fallback.

Same calling convention: ``qkv_list = [q, k, v]`` with q,k,v ``[B, L, H, dh]`` (NHD); the
list is CLEARED so the caller's references die early (:185-186); returns ``[B, Lq, H, dh]``
in q's dtype.
"""
import torch

from . import ops


def _key_bias_from_mask(attention_mask, B, Lk):
    """The seam receives the mask as [B, Lq|1, H|1, Lk] (it is transposed(1,2) before SDPA,
    :110-111).  The kernel supports a per-key bias shared by all heads and queries, which is
    what the LTX cross-attention produces (transformer3d.py:411-415 -> attention.py:1026-1033)."""
    m = attention_mask
    if m.dim() != 4 or m.shape[0] != B or m.shape[-1] != Lk:
        raise ValueError(f"pay_attention: attention_mask must be [B, 1|Lq, 1|H, Lk], got {tuple(m.shape)}")
    for d in (1, 2):
        if m.shape[d] != 1 and m.stride(d) != 0:
            raise NotImplementedError(
                "pay_attention: only key biases broadcast over queries and heads are supported "
                f"(mask dim {d} has size {m.shape[d]} with stride {m.stride(d)})")
    # every block of a forward passes the same mask: keep its fp32 form (one conversion kernel per
    # forward instead of one per layer); keyed on storage + version so an in-place edit invalidates it
    # (a few entries: the micro-batches of the sequence-parallel overlap mode pass row slices of one mask in turn)
    key = (m.data_ptr(), tuple(m.shape), tuple(m.stride()), m.dtype, ops.tensor_version(m))
    hit = _BIAS_CACHE.get(key)
    if hit is None:
        if len(_BIAS_CACHE) >= 4:
            _BIAS_CACHE.clear()
        hit = _BIAS_CACHE[key] = (m[:, 0, 0, :].to(torch.float32).contiguous(), m)     # m kept alive: no pointer reuse
    return hit[0]


_BIAS_CACHE = {}


@torch.compiler.disable()
def pay_attention(qkv_list, dropout_p=0., softmax_scale=None, causal=False, window_size=(-1, -1),
                  deterministic=False, version=None, force_attention=None, attention_mask=None,
                  cross_attn=False, q_lens=None, k_lens=None, q_norm=None, rope=None):
    """``q_norm`` / ``rope`` (extension, used by AttnProcessor2_0 only): q is the raw projection output and is
    RMS-normalised (+ rotated) by the kernel while it loads it -- see ops.attention.

    ``softmax_scale``: HONOURED here.  The reference's eager branch drops it on the floor (``sdpa_wrapper`` calls
    ``F.scaled_dot_product_attention`` without ``scale=``, wan/modules/attention.py:99-116, so that branch always uses
    1/sqrt(head_dim)) while its flash / sage branches pass it on (:344-347, :394-399); every caller in the reference
    leaves it ``None`` (= 1/sqrt(head_dim) in all branches), where the two readings coincide.  A caller that does pass a
    scale gets what the argument says -- the behaviour of the reference's non-eager back-ends, not of its sdpa branch."""
    q, k, v = qkv_list
    qkv_list.clear()
    if (q_norm is not None or rope is not None) and (q_lens is not None or k_lens is not None):
        # checked before ANY branch below: the var-k-len path re-enters this function per chunk without q_norm / rope
        # and would attend the raw projection un-normalised and un-rotated
        raise NotImplementedError("pay_attention: q_norm / rope on load with q_lens / k_lens")
    if causal or tuple(window_size) != (-1, -1) or dropout_p != 0.:
        raise NotImplementedError("pay_attention: causal / windowed / dropout attention is not on this path")
    if force_attention not in (None, "sdpa"):
        raise NotImplementedError(f"pay_attention: backend '{force_attention}' does not exist here (single HIP backend)")
    out_dtype = q.dtype
    if v.dtype != torch.bfloat16:
        raise TypeError(f"pay_attention: bf16 only on this path, got {v.dtype}")
    q = q.to(v.dtype)
    k = k.to(v.dtype)
    b, lq, lk = q.size(0), q.size(1), k.size(1)
    final_padding = 0

    if b > 1 and k_lens is not None:
        # "poor man's var-k-len": :197-229 -- runs of equal k_len are batched together
        assert attention_mask is None and q_lens is None
        k_lens = [int(x) for x in k_lens]
        chunk_sizes, k_sizes = [], []
        cur, cnt = k_lens[0], 1
        for kl in k_lens[1:]:
            if kl == cur:
                cnt += 1
            else:
                chunk_sizes.append(cnt)
                k_sizes.append(cur)
                cur, cnt = kl, 1
        chunk_sizes.append(cnt)
        k_sizes.append(cur)
        if len(chunk_sizes) > 1 or k_lens[0] != k.shape[1]:
            outs = []
            for sq, sk, sv, sz in zip(torch.split(q, chunk_sizes), torch.split(k, chunk_sizes),
                                      torch.split(v, chunk_sizes), k_sizes):
                outs.append(pay_attention([sq, sk[:, :sz], sv[:, :sz]], softmax_scale=softmax_scale))
            return torch.cat(outs, dim=0)
    elif q_lens is not None or k_lens is not None:
        assert b == 1
        szq = int(q_lens[0]) if q_lens is not None else lq
        szk = int(k_lens[0]) if k_lens is not None else lk
        final_padding = lq - szq
        q, k, v = q[:, :szq], k[:, :szk], v[:, :szk]

    bias = None if attention_mask is None else _key_bias_from_mask(attention_mask, b, k.size(1))
    x = ops.attention(q, k, v, key_bias=bias, softmax_scale=softmax_scale, q_norm=q_norm, rope=rope)
    x = x.type(out_dtype)
    if final_padding > 0:
        x = torch.cat([x, torch.empty((x.shape[0], final_padding, *x.shape[-2:]), dtype=x.dtype, device=x.device)], 1)
    return x
