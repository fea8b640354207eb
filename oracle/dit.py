"""Oracle restatement of the reference's DiT forward.  TEST INFRASTRUCTURE ONLY.

Follows, op for op (so that running it in bf16 reproduces the reference's eager
rounding points, and running it in fp32 gives the "truth" used by parity tests):

  Transformer3DModel.forward              ltx_video/models/transformers/transformer3d.py:328-507
  Transformer3DModel.precompute_freqs_cis transformer3d.py:202-255
  Transformer3DModel.create_skip_layer_mask  transformer3d.py:171-186
  BasicTransformerBlock.forward           ltx_video/models/transformers/attention.py:205-364
  AttnProcessor2_0.__call__               attention.py:986-1173
  Attention.apply_rotary_emb              attention.py:960-975
  Attention.prepare_attention_mask        attention.py:868-925
  pay_attention -> sdpa_wrapper           wan/modules/attention.py:162-199, 344-347, 99-116

Functional style: ``sd`` is a flat state dict with the reference's key names
(``transformer_blocks.{i}.attn1.to_q.weight`` ...), ``cfg`` a dict with the
reference's constructor argument names (transformer3d.py:50-81).
"""
import math

import torch
import torch.nn.functional as F

from . import leaves

# ltx_video/utils/skip_layer_strategy.py:4-8
ATTENTION_SKIP, ATTENTION_VALUES, RESIDUAL, TRANSFORMER_BLOCK = 1, 2, 3, 4


def default_2b_config():
    """OURS_TRANSFORMER_CONFIG, ltx_video/utils/diffusers_config_mapping.py:74-105."""
    return dict(
        num_attention_heads=32, attention_head_dim=64, in_channels=128, out_channels=128,
        num_layers=28, cross_attention_dim=2048, caption_channels=4096,
        attention_bias=True, activation_fn="gelu-approximate",
        norm_elementwise_affine=False, norm_eps=1e-6,
        qk_norm="rms_norm", standardization_norm="rms_norm",
        adaptive_norm="single_scale_shift",
        positional_embedding_type="rope", positional_embedding_theta=10000.0,
        positional_embedding_max_pos=[20, 2048, 2048], timestep_scale_multiplier=1000,
    )


# --------------------------------------------------------------------------- RoPE
def precompute_freqs_cis(indices_grid, cfg, out_dtype):
    """transformer3d.py:192-255 (spacing="exp")."""
    dim = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    theta = cfg["positional_embedding_theta"]
    max_pos = cfg["positional_embedding_max_pos"]
    dtype = torch.float32
    frac = torch.stack([indices_grid[:, i] / max_pos[i] for i in range(3)], dim=-1)  # [B,N,3]
    indices = theta ** torch.linspace(math.log(1, theta), math.log(theta, theta), dim // 6,
                                      device=frac.device, dtype=dtype)
    indices = indices.to(dtype) * math.pi / 2
    freqs = (indices * (frac.unsqueeze(-1) * 2 - 1)).transpose(-1, -2).flatten(2)
    cos = freqs.cos().repeat_interleave(2, dim=-1)
    sin = freqs.sin().repeat_interleave(2, dim=-1)
    if dim % 6 != 0:
        cos = torch.cat([torch.ones_like(cos[:, :, : dim % 6]), cos], dim=-1)
        sin = torch.cat([torch.zeros_like(cos[:, :, : dim % 6]), sin], dim=-1)
    return cos.to(out_dtype), sin.to(out_dtype)


def apply_rotary_emb(x, freqs_cis):
    """attention.py:960-975: interleaved pairs on the flat channel axis."""
    cos, sin = freqs_cis
    t = x.reshape(*x.shape[:-1], -1, 2)
    t1, t2 = t.unbind(dim=-1)
    rot = torch.stack((-t2, t1), dim=-1).reshape(x.shape)
    return x * cos + rot * sin


def create_skip_layer_mask(num_layers, batch_size, num_conds, ptb_index, skip_block_list, dtype):
    """transformer3d.py:171-186."""
    if skip_block_list is None or len(skip_block_list) == 0:
        return None
    mask = torch.ones((num_layers, batch_size * num_conds), dtype=dtype)
    for b in skip_block_list:
        mask[b, ptb_index::num_conds] = 0
    return mask


# ---------------------------------------------------------------- attention seam
def sdpa_nhd(q, k, v, attention_mask=None, head_chunk=8):
    """pay_attention's eager branch (wan/modules/attention.py:344-347 -> :99-116):
    q,k,v are [B, L, H, dh]; attention_mask, if given, is [B, Lq|1, H|1, Lk] additive
    (it is transposed(1,2) before SDPA, :110-111).  Non-causal, scale 1/sqrt(dh),
    fp32 accumulate, result in q's dtype.  Written out explicitly (no call to
    F.scaled_dot_product_attention) and chunked over heads to bound memory."""
    B, Lq, H, dh = q.shape
    scale = 1.0 / math.sqrt(dh)
    out = torch.empty(B, Lq, H, v.shape[-1], dtype=q.dtype, device=q.device)
    # bound the fp32 score block (and its softmax copy) to ~3 GB: at N = 13 376 x B 3 eight heads would be 17 GB
    head_chunk = max(1, min(head_chunk, int(3e9 // max(1, B * Lq * k.shape[1] * 4))))
    for h0 in range(0, H, head_chunk):
        h1 = min(H, h0 + head_chunk)
        qq = q[:, :, h0:h1].permute(0, 2, 1, 3).float()
        kk = k[:, :, h0:h1].permute(0, 2, 1, 3).float()
        vv = v[:, :, h0:h1].permute(0, 2, 1, 3).float()
        s = torch.matmul(qq, kk.transpose(-1, -2)) * scale
        if attention_mask is not None:
            m = attention_mask.transpose(1, 2)  # [B, H|1, Lq|1, Lk]
            if m.shape[1] != 1:
                m = m[:, h0:h1]
            s = s + m.float()
        p = torch.softmax(s, dim=-1)
        out[:, :, h0:h1] = torch.matmul(p, vv).permute(0, 2, 1, 3).to(q.dtype)
    return out


def _prepare_attention_mask(attention_mask, target_length, batch_size, heads):
    """attention.py:868-925 with out_dim=3, followed by the view at :1031-1033."""
    if attention_mask is None:
        return None
    if attention_mask.shape[-1] != target_length:
        attention_mask = F.pad(attention_mask, (0, target_length), value=0.0)
    if attention_mask.shape[0] < batch_size * heads:
        attention_mask = attention_mask.repeat_interleave(heads, dim=0)
    return attention_mask.view(batch_size, heads, -1, attention_mask.shape[-1])


def attention_processor(sd, p, cfg, hidden_states, freqs_cis=None, encoder_hidden_states=None,
                        attention_mask=None, skip_layer_mask=None, skip_layer_strategy=None,
                        use_rope=True):
    """AttnProcessor2_0.__call__ (attention.py:986-1173) for 3-D inputs, no
    spatial/group norm, no residual connection, rescale_output_factor 1."""
    heads = cfg["num_attention_heads"]
    batch_size, sequence_length, _ = (hidden_states.shape if encoder_hidden_states is None
                                      else encoder_hidden_states.shape)
    if skip_layer_mask is not None:
        skip_layer_mask = skip_layer_mask.reshape(batch_size, 1, 1)
    attention_mask = _prepare_attention_mask(attention_mask, sequence_length, batch_size, heads)

    qk_norm = cfg.get("qk_norm")

    def _qk_norm(x, name):
        if qk_norm is None:
            return x
        if qk_norm == "rms_norm":
            return leaves.rms_norm(x, 1e-5, sd[p + name + ".weight"])
        if qk_norm == "layer_norm":
            return F.layer_norm(x, (x.shape[-1],), sd[p + name + ".weight"], sd[p + name + ".bias"], 1e-5)
        raise ValueError(qk_norm)

    query = _qk_norm(leaves.linear(hidden_states, sd, p + "to_q."), "q_norm")
    if encoder_hidden_states is not None:
        key = _qk_norm(leaves.linear(encoder_hidden_states, sd, p + "to_k."), "k_norm")
    else:
        encoder_hidden_states = hidden_states
        key = _qk_norm(leaves.linear(hidden_states, sd, p + "to_k."), "k_norm")
        if use_rope:
            key = apply_rotary_emb(key, freqs_cis)
            query = apply_rotary_emb(query, freqs_cis)
    value = leaves.linear(encoder_hidden_states, sd, p + "to_v.")
    skip_attention = False
    value_for_stg = None
    if skip_layer_mask is not None and skip_layer_strategy == ATTENTION_VALUES:
        skip_attention = skip_layer_mask.shape[0] == 1 and skip_layer_mask[0].item() == 0
        value_for_stg = value

    inner_dim = key.shape[-1]
    head_dim = inner_dim // heads
    dtype = query.dtype
    if skip_attention:
        out = value_for_stg
    else:
        q = query.view(batch_size, -1, heads, head_dim)
        k = key.view(batch_size, -1, heads, head_dim)
        v = value.view(batch_size, -1, heads, head_dim)
        mask_nhd = None if attention_mask is None else attention_mask.transpose(1, 2)  # :1115-1116
        a = sdpa_nhd(q, k, v, mask_nhd)                     # pay_attention, NHD
        a = a.reshape(batch_size, -1, heads * head_dim).to(dtype)
        if skip_layer_mask is not None and skip_layer_strategy == ATTENTION_SKIP:
            a = a * skip_layer_mask + hidden_states * (1.0 - skip_layer_mask)
        elif skip_layer_mask is not None and skip_layer_strategy == ATTENTION_VALUES:
            a = a * skip_layer_mask                          # :1138-1140 (in place there)
            a = a + value_for_stg * (1.0 - skip_layer_mask)
        out = a
    return leaves.linear(out, sd, p + "to_out.0.")


# --------------------------------------------------------------------- the block
def _norm(x, sd, p, cfg):
    eps = cfg["norm_eps"]
    if cfg.get("standardization_norm", "layer_norm") == "rms_norm":
        return leaves.rms_norm(x, eps, sd.get(p + "weight"))
    return F.layer_norm(x, (x.shape[-1],), sd.get(p + "weight"), sd.get(p + "bias"), eps)


def _frames(x, t1):                       # attention.py:36-41
    return x.reshape(x.shape[0], t1, -1, x.shape[-1])


def _flat(x):
    return x.reshape(x.shape[0], -1, x.shape[-1])


def transformer_block(sd, p, cfg, hidden_states, freqs_cis, encoder_hidden_states,
                      encoder_attention_mask, timestep, attention_mask=None,
                      skip_layer_mask=None, skip_layer_strategy=None):
    """BasicTransformerBlock.forward, adaptive_norm == "single_scale_shift"
    (attention.py:205-364)."""
    assert cfg.get("adaptive_norm", "single_scale_shift") == "single_scale_shift"
    batch_size = hidden_states.shape[0]
    if skip_layer_mask is not None and skip_layer_mask.flatten().min() == 1.0:
        skip_layer_mask = None

    norm_h = _norm(hidden_states, sd, p + "norm1.", cfg)
    assert timestep.ndim == 3
    table = sd[p + "scale_shift_table"]
    ada = table[None, None] + timestep.reshape(batch_size, timestep.shape[1], table.shape[0], -1)
    ada = ada.unsqueeze(-2)
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = ada.unbind(dim=2)
    t1 = scale_msa.shape[1]
    norm_h = _frames(norm_h, t1)
    norm_h = norm_h * (1 + scale_msa)          # in place in the reference (:248)
    norm_h = norm_h + shift_msa                # (:249)
    norm_h = _flat(norm_h)

    attn_out = attention_processor(sd, p + "attn1.", cfg, norm_h, freqs_cis=freqs_cis,
                                   attention_mask=attention_mask,
                                   skip_layer_mask=skip_layer_mask,
                                   skip_layer_strategy=skip_layer_strategy)
    attn_out = _flat(_frames(attn_out, t1) * gate_msa)       # :283-286
    hidden_states = hidden_states + attn_out                 # :288 (in place there)

    if (p + "attn2.to_q.weight") in sd:
        attn_out = attention_processor(sd, p + "attn2.", cfg, hidden_states, freqs_cis=freqs_cis,
                                       encoder_hidden_states=encoder_hidden_states,
                                       attention_mask=encoder_attention_mask)
        hidden_states = hidden_states + attn_out             # :310
    # `original_hidden_states = hidden_states` (:231) aliases the tensor that :288 and
    # :310 then update IN PLACE, so what the TransformerBlock strategy blends back at
    # :355-362 is the state after both attention residuals, not the block input.
    original_hidden_states = hidden_states

    norm_h = _norm(hidden_states, sd, p + "norm2.", cfg)
    norm_h = _frames(norm_h, t1)
    norm_h = norm_h * (1 + scale_mlp)                        # :318
    norm_h = norm_h + shift_mlp                              # :319
    norm_h = _flat(norm_h)

    # :333-343 -- token-chunked in the reference only to save memory; rows are independent
    act = cfg.get("activation_fn", "geglu")
    assert act in ("gelu-approximate", "gelu")
    ff = leaves.gelu_proj(norm_h, sd, p + "ff.net.0.", "tanh" if act == "gelu-approximate" else "none")
    ff = leaves.linear(ff, sd, p + "ff.net.2.")
    ff = _flat(_frames(ff, t1) * gate_mlp)                   # :346-349
    hidden_states = ff + hidden_states                       # :351

    if skip_layer_mask is not None and skip_layer_strategy == TRANSFORMER_BLOCK:
        m = skip_layer_mask.view(-1, 1, 1)
        hidden_states = hidden_states * m + original_hidden_states * (1.0 - m)
    return hidden_states


# ------------------------------------------------------------------- the model
def transformer3d_forward(sd, cfg, hidden_states, freqs_cis, encoder_hidden_states, timestep,
                          encoder_attention_mask=None, attention_mask=None, skip_layer_mask=None,
                          skip_layer_strategy=None, latent_shape=None, num_layers=None):
    """Transformer3DModel.forward with joint_pass=True, mixed=False
    (transformer3d.py:328-507).  Returns the sample tensor [B, N, out_channels]."""
    dtype = hidden_states.dtype
    if attention_mask is not None and attention_mask.ndim == 2:
        attention_mask = ((1 - attention_mask.to(dtype)) * -10000.0).unsqueeze(1)
    if encoder_attention_mask is not None and encoder_attention_mask.ndim == 2:
        encoder_attention_mask = ((1 - encoder_attention_mask.to(dtype)) * -10000.0).unsqueeze(1)

    hidden_states = leaves.linear(hidden_states, sd, "patchify_proj.")
    if cfg.get("timestep_scale_multiplier"):
        timestep = cfg["timestep_scale_multiplier"] * timestep
    if timestep.shape[-1] > 1:
        timestep = timestep.reshape(timestep.shape[0], -1, latent_shape[-2] * latent_shape[-1])
        timestep = timestep[:, :, 0]
    batch_size = hidden_states.shape[0]
    timestep, embedded_timestep = leaves.adaln_single(timestep.flatten(), sd, "adaln_single.", dtype)
    timestep = timestep.view(batch_size, -1, timestep.shape[-1])
    embedded_timestep = embedded_timestep.view(batch_size, -1, embedded_timestep.shape[-1])

    if "caption_projection.linear_1.weight" in sd:
        encoder_hidden_states = leaves.text_projection(encoder_hidden_states, sd, "caption_projection.")
        encoder_hidden_states = encoder_hidden_states.view(batch_size, -1, hidden_states.shape[-1])

    L = cfg["num_layers"] if num_layers is None else num_layers
    for i in range(L):
        hidden_states = transformer_block(
            sd, f"transformer_blocks.{i}.", cfg, hidden_states, freqs_cis, encoder_hidden_states,
            encoder_attention_mask, timestep, attention_mask=attention_mask,
            skip_layer_mask=None if skip_layer_mask is None else skip_layer_mask[i],
            skip_layer_strategy=skip_layer_strategy)

    ssv = sd["scale_shift_table"][None, None] + embedded_timestep[:, :, None]
    shift, scale = ssv[:, :, 0].unsqueeze(-2), ssv[:, :, 1].unsqueeze(-2)
    hidden_states = F.layer_norm(hidden_states, (hidden_states.shape[-1],), None, None, 1e-6)
    hidden_states = _frames(hidden_states, scale.shape[1])
    hidden_states = hidden_states * (1 + scale)
    hidden_states = hidden_states + shift
    hidden_states = _flat(hidden_states)
    return leaves.linear(hidden_states, sd, "proj_out.")


def init_state_dict(cfg, seed=0, dtype=torch.float32, num_layers=None):
    """Random weights with the reference's key names and init distributions
    (nn.Linear default init; scale_shift_table ~ N(0,1)/sqrt(D): transformer3d.py:141-143,
    attention.py:183-185; RMSNorm weight = 1).  Used by tests / bench to create
    synthetic checkpoints (no real weights are available offline)."""
    g = torch.Generator().manual_seed(seed)
    D = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    sd = {}

    def lin(name, fin, fout, bias=True):
        bound = 1.0 / math.sqrt(fin)
        sd[name + ".weight"] = (torch.rand(fout, fin, generator=g) * 2 - 1) * bound
        if bias:
            sd[name + ".bias"] = (torch.rand(fout, generator=g) * 2 - 1) * bound

    lin("patchify_proj", cfg["in_channels"], D)
    lin("adaln_single.emb.timestep_embedder.linear_1", 256, D)
    lin("adaln_single.emb.timestep_embedder.linear_2", D, D)
    lin("adaln_single.linear", D, 6 * D)
    if cfg.get("caption_channels"):
        lin("caption_projection.linear_1", cfg["caption_channels"], D)
        lin("caption_projection.linear_2", D, D)
    sd["scale_shift_table"] = torch.randn(2, D, generator=g) / D ** 0.5
    lin("proj_out", D, cfg["out_channels"])
    L = cfg["num_layers"] if num_layers is None else num_layers
    ab = cfg.get("attention_bias", False)
    for i in range(L):
        p = f"transformer_blocks.{i}."
        sd[p + "scale_shift_table"] = torch.randn(6, D, generator=g) / D ** 0.5
        for a, kv_in in (("attn1", D), ("attn2", cfg["cross_attention_dim"])):
            lin(p + a + ".to_q", D, D, ab)
            lin(p + a + ".to_k", kv_in, D, ab)
            lin(p + a + ".to_v", kv_in, D, ab)
            lin(p + a + ".to_out.0", D, D, True)
            if cfg.get("qk_norm") == "rms_norm":
                # perturbed from the all-ones init so that tests exercise the weight
                sd[p + a + ".q_norm.weight"] = 1.0 + 0.1 * torch.randn(D, generator=g)
                sd[p + a + ".k_norm.weight"] = 1.0 + 0.1 * torch.randn(D, generator=g)
        lin(p + "ff.net.0.proj", D, 4 * D)
        lin(p + "ff.net.2", 4 * D, D)
    return {k: v.to(dtype) for k, v in sd.items()}
