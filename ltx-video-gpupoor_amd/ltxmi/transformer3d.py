"""``Transformer3DModel`` -- the LTX-Video DiT wrapper on libltxmi kernels.

Drop-in for ltx_video/models/transformers/transformer3d.py:46-507: same constructor arguments
(:50-81), same parameter names (so ``load_state_dict`` of a reference checkpoint works, incl.
the ``model.diffusion_model.`` prefix strip :257-269), same ``forward`` signature and return
convention (:328-345, :504-507), ``precompute_freqs_cis`` (:202-255) and
``create_skip_layer_mask`` (:171-186).

Weights live fully in HBM (3.85 GB bf16 for the 2B config): the reference's mmgp block
offloading has no counterpart here.
"""
import contextlib
import math
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import torch
from torch import nn

from . import ops
from .attention import BasicTransformerBlock, SkipLayerStrategy

BF16 = torch.bfloat16


@dataclass
class Transformer3DModelOutput:
    sample: torch.Tensor


class _TimestepEmbedder(nn.Module):            # diffusers TimestepEmbedding: keys linear_1 / linear_2
    def __init__(self, in_channels, dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, dim)
        self.linear_2 = nn.Linear(dim, dim)


class _CombinedTimestepEmbeddings(nn.Module):  # diffusers PixArtAlphaCombinedTimestepSizeEmbeddings
    def __init__(self, dim):
        super().__init__()
        self.timestep_embedder = _TimestepEmbedder(256, dim)

    def forward(self, timestep_f32):
        """timestep_f32: fp32 [n], already scaled.  sinusoid(256, flip, shift 0) -> linear_1 ->
        SiLU -> linear_2 (transformer3d.py:428-433 via AdaLayerNormSingle)."""
        proj = ops.timestep_embedding(timestep_f32.contiguous(), 256)
        te = self.timestep_embedder
        h = ops.gemm(proj, te.linear_1.weight, te.linear_1.bias, epilogue=ops.EPI_SILU)
        return ops.gemm(h, te.linear_2.weight, te.linear_2.bias)


class AdaLayerNormSingle(nn.Module):           # keys emb.timestep_embedder.linear_{1,2}, linear
    def __init__(self, embedding_dim, use_additional_conditions=False):
        super().__init__()
        assert not use_additional_conditions
        self.emb = _CombinedTimestepEmbeddings(embedding_dim)
        self.linear = nn.Linear(embedding_dim, 6 * embedding_dim, bias=True)

    def forward(self, timestep_f32):
        emb = self.emb(timestep_f32)
        return ops.gemm(ops.silu(emb), self.linear.weight, self.linear.bias), emb


class PixArtAlphaTextProjection(nn.Module):    # keys linear_1 / linear_2, GELU-tanh between
    def __init__(self, in_features, hidden_size):
        super().__init__()
        self.linear_1 = nn.Linear(in_features, hidden_size)
        self.linear_2 = nn.Linear(hidden_size, hidden_size)

    def forward(self, caption):
        h = ops.gemm(caption.reshape(-1, caption.shape[-1]), self.linear_1.weight, self.linear_1.bias,
                     epilogue=ops.EPI_GELU_TANH)
        return ops.gemm(h, self.linear_2.weight, self.linear_2.bias)


class _Config(dict):
    __getattr__ = dict.__getitem__


class Transformer3DModel(nn.Module):
    def __init__(self, num_attention_heads: int = 16, attention_head_dim: int = 88,
                 in_channels: Optional[int] = None, out_channels: Optional[int] = None, num_layers: int = 1,
                 dropout: float = 0.0, norm_num_groups: int = 32, cross_attention_dim: Optional[int] = None,
                 attention_bias: bool = False, num_vector_embeds: Optional[int] = None,
                 activation_fn: str = "geglu", num_embeds_ada_norm: Optional[int] = None,
                 use_linear_projection: bool = False, only_cross_attention: bool = False,
                 double_self_attention: bool = False, upcast_attention: bool = False,
                 adaptive_norm: str = "single_scale_shift", standardization_norm: str = "layer_norm",
                 norm_elementwise_affine: bool = True, norm_eps: float = 1e-5, attention_type: str = "default",
                 caption_channels: int = None, use_tpu_flash_attention: bool = False,
                 qk_norm: Optional[str] = None, positional_embedding_type: str = "rope",
                 positional_embedding_theta: Optional[float] = None,
                 positional_embedding_max_pos: Optional[List[int]] = None,
                 timestep_scale_multiplier: Optional[float] = None, causal_temporal_positioning: bool = False,
                 **ignored):
        super().__init__()
        self.config = _Config({k: v for k, v in locals().items() if k not in ("self", "ignored", "__class__")})
        if use_tpu_flash_attention:
            raise NotImplementedError("TPU flash attention does not exist on MI355X")
        if positional_embedding_type != "rope":
            raise ValueError("Absolute positional embedding is no longer supported")     # transformer3d.py:98-99
        if positional_embedding_theta is None or positional_embedding_max_pos is None:
            raise ValueError("rope needs positional_embedding_theta and positional_embedding_max_pos")
        self.num_attention_heads = num_attention_heads
        self.attention_head_dim = attention_head_dim
        inner_dim = num_attention_heads * attention_head_dim
        self.inner_dim = inner_dim
        self.in_channels = in_channels
        self.patchify_proj = nn.Linear(in_channels, inner_dim, bias=True)
        self.positional_embedding_theta = positional_embedding_theta
        self.positional_embedding_max_pos = positional_embedding_max_pos
        self.use_rope = True
        self.timestep_scale_multiplier = timestep_scale_multiplier
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner_dim, num_attention_heads, attention_head_dim, dropout=dropout,
                                  cross_attention_dim=cross_attention_dim, activation_fn=activation_fn,
                                  num_embeds_ada_norm=num_embeds_ada_norm, attention_bias=attention_bias,
                                  only_cross_attention=only_cross_attention,
                                  double_self_attention=double_self_attention, upcast_attention=upcast_attention,
                                  adaptive_norm=adaptive_norm, standardization_norm=standardization_norm,
                                  norm_elementwise_affine=norm_elementwise_affine, norm_eps=norm_eps,
                                  attention_type=attention_type, qk_norm=qk_norm, use_rope=True)
            for _ in range(num_layers)])
        self.out_channels = in_channels if out_channels is None else out_channels
        self.scale_shift_table = nn.Parameter(torch.randn(2, inner_dim) / inner_dim ** 0.5)
        self.proj_out = nn.Linear(inner_dim, self.out_channels)
        self.adaln_single = AdaLayerNormSingle(inner_dim, use_additional_conditions=False)
        self.caption_projection = None
        if caption_channels is not None:
            self.caption_projection = PixArtAlphaTextProjection(caption_channels, inner_dim)

    # ------------------------------------------------------------------ properties
    @property
    def dtype(self):
        return self.patchify_proj.weight.dtype

    @property
    def device(self):
        return self.patchify_proj.weight.device

    @classmethod
    def from_config(cls, config):
        return cls(**{k: v for k, v in dict(config).items() if not k.startswith("_")})

    @classmethod
    def from_pretrained(cls, pretrained_model_path, *args, device="cuda", dtype=torch.bfloat16, **kwargs):
        """transformer3d.py:271-326: diffusers directory or single-file safetensors with a config blob."""
        from .loading import load_transformer
        return load_transformer(pretrained_model_path, device=device, dtype=dtype)

    def load_state_dict(self, state_dict: Dict, *args, **kwargs):              # transformer3d.py:257-269
        if any(k.startswith("model.diffusion_model.") for k in state_dict.keys()):
            state_dict = {k.replace("model.diffusion_model.", ""): v for k, v in state_dict.items()
                          if k.startswith("model.diffusion_model.")}
        return super().load_state_dict(state_dict, *args, **kwargs)

    # ------------------------------------------------------------------ helpers kept verbatim
    def create_skip_layer_mask(self, batch_size: int, num_conds: int, ptb_index: int,
                               skip_block_list: Optional[List[int]] = None):     # transformer3d.py:171-186
        if skip_block_list is None or len(skip_block_list) == 0:
            return None
        num_layers = len(self.transformer_blocks)
        host = torch.ones((num_layers, batch_size * num_conds), dtype=torch.float32)
        for block_idx in skip_block_list:
            host[block_idx, ptb_index::num_conds] = 0
        mask = host.to(device=self.device, dtype=self.dtype)
        mask._ltxmi_host_rows = [[float(x) for x in row] for row in host.tolist()]
        return mask

    def get_fractional_positions(self, indices_grid):                          # transformer3d.py:192-200
        return torch.stack([indices_grid[:, i] / self.positional_embedding_max_pos[i] for i in range(3)], dim=-1)

    def precompute_freqs_cis(self, indices_grid, spacing="exp"):
        """transformer3d.py:202-255 -- once per generation, fp32 math, tables cast to the model
        dtype.  Host-side torch (not on the per-step path)."""
        if spacing != "exp":
            raise NotImplementedError("only the default 'exp' spacing is on this path")
        dtype = torch.float32
        dim = self.inner_dim
        theta = self.positional_embedding_theta
        frac = self.get_fractional_positions(indices_grid)
        indices = theta ** torch.linspace(math.log(1, theta), math.log(theta, theta), dim // 6,
                                          device=frac.device, dtype=dtype)
        indices = indices.to(dtype=dtype) * math.pi / 2
        freqs = (indices * (frac.unsqueeze(-1) * 2 - 1)).transpose(-1, -2).flatten(2)
        cos_freq = freqs.cos().repeat_interleave(2, dim=-1)
        sin_freq = freqs.sin().repeat_interleave(2, dim=-1)
        if dim % 6 != 0:
            cos_freq = torch.cat([torch.ones_like(cos_freq[:, :, : dim % 6]), cos_freq], dim=-1)
            sin_freq = torch.cat([torch.zeros_like(cos_freq[:, :, : dim % 6]), sin_freq], dim=-1)
        return cos_freq.to(self.dtype).contiguous(), sin_freq.to(self.dtype).contiguous()

    def _stacked_text_kv(self, ehs):
        """Every layer's text keys / values of this forward in ONE GEMM (attention.py:1042-1048 runs the projections once per
        block): ``ehs`` [B, T, D] against the stacked [to_k; to_v] of all cross-attention layers -> [B T, L 2D], each layer's
        key half RMS-normalised in place (k_norm, :1040-1041), each block's attn2 handed its column slice (a strided view, read
        by the attention kernel through its strides).  The values are those of the per-layer projections bit for bit: all GEMM
        kernels accumulate over K in the same order.  Stacked weights: built once, rebuilt when any source tensor changes."""
        blocks = [b for b in self.transformer_blocks if getattr(b, "attn2", None) is not None]
        if not blocks or ehs.dim() != 3 or not ehs.is_contiguous():
            return
        packs = [b.attn2.packed_kv() for b in blocks]
        if any(bk is None for _, bk in packs) or len({tuple(w.shape) for w, _ in packs}) != 1:
            return
        key = tuple((w.data_ptr(), ops.tensor_version(w), bk.data_ptr(), ops.tensor_version(bk)) for w, bk in packs)
        st = self.__dict__.get("_stacked_kv_weights")
        if st is None or st[0] != key:
            with torch.no_grad():
                st = (key, torch.cat([w for w, _ in packs], 0).contiguous(), torch.cat([bk for _, bk in packs], 0).contiguous(),
                      packs)                                                        # (packs kept alive: no address reuse)
            self.__dict__["_stacked_kv_weights"] = st
        _, w_all, b_all, _ = st
        Bk, Lk, _ = ehs.shape
        two_d = packs[0][0].shape[0]
        kv_all = ops.gemm(ehs.reshape(Bk * Lk, -1), w_all, b_all)                       # [B T, L 2D]
        for i, b in enumerate(blocks):
            kv = kv_all[:, i * two_d:(i + 1) * two_d]
            ops.rmsnorm_rope_(kv[:, :two_d // 2], b.attn2.k_norm.weight, b.attn2.k_norm.eps)
            b.attn2.__dict__["_text_kv_ready"] = (ehs, ops.tensor_version(ehs), kv)

    def _run_blocks_microbatched(self, hidden_states, slices, freqs_cis, attention_mask, encoder_hidden_states,
                                 encoder_attention_mask, temb, cross_attention_kwargs, class_labels, layer_mask,
                                 skip_layer_strategy, ltxv_model):
        """The block loop over micro-batches of batch rows, micro-batch i on side stream i (plain sequential on the CPU).
        The host issues block b of every micro-batch before block b + 1 of any, so the collectives of the slices
        interleave in one fixed order on every rank.  The blocks update ``hidden_states`` in place through the row views.
        Returns True when interrupted."""
        B = hidden_states.shape[0]
        on_gpu = hidden_states.is_cuda
        streams = None
        # Everything that is created lazily and SHARED between the slices is created here, on the main stream, before the
        # side streams fork from it: the packed projection weights (``Attention._pack`` caches whatever the first caller
        # built -- on a fresh model, or after a reload / LoRA merge invalidated the packs, slice 1 on stream 1 would
        # otherwise take slice 0's cache entry while stream 0's torch.cat may still be writing it).
        for block in self.transformer_blocks:
            block.prepare_shared_state()
        if on_gpu:
            main = torch.cuda.current_stream()
            key = (hidden_states.device, len(slices))
            pool = self.__dict__.setdefault("_mb_streams", {})
            streams = pool.get(key)
            if streams is None:
                streams = pool[key] = [torch.cuda.Stream(device=hidden_states.device) for _ in slices]
            for st in streams:
                st.wait_stream(main)

        def rows(t, sl):
            return t if (t is None or t.shape[0] != B) else t[sl]

        interrupted = False
        for block_idx, block in enumerate(self.transformer_blocks):
            for i, sl in enumerate(slices):
                ctx = torch.cuda.stream(streams[i]) if on_gpu else contextlib.nullcontext()
                with ctx:
                    block(hidden_states[sl], freqs_cis=tuple(rows(t, sl) for t in freqs_cis), attention_mask=attention_mask,
                          encoder_hidden_states=rows(encoder_hidden_states, sl),
                          encoder_attention_mask=rows(encoder_attention_mask, sl), timestep=temb[sl],
                          cross_attention_kwargs=cross_attention_kwargs, class_labels=class_labels,
                          skip_layer_mask=layer_mask(block_idx, sl), skip_layer_strategy=skip_layer_strategy)
            if ltxv_model is not None and ltxv_model._interrupt:
                interrupted = True
                break
        if on_gpu:
            for st in streams:
                main.wait_stream(st)
        return interrupted

    # ------------------------------------------------------------------ forward
    def forward(self, hidden_states: torch.Tensor, freqs_cis: list,
                encoder_hidden_states: Optional[torch.Tensor] = None, timestep: Optional[torch.Tensor] = None,
                class_labels: Optional[torch.LongTensor] = None, cross_attention_kwargs: Dict[str, Any] = None,
                attention_mask: Optional[torch.Tensor] = None, encoder_attention_mask: Optional[torch.Tensor] = None,
                skip_layer_mask: Optional[torch.Tensor] = None,
                skip_layer_strategy: Optional[SkipLayerStrategy] = None, latent_shape=None, joint_pass=True,
                ltxv_model=None, mixed=False, return_dict: bool = True, stg_alias_blocks: int = 0,
                _microbatches=None):
        """``stg_alias_blocks`` (extension, default off): the caller guarantees that the LAST batch row has
        exactly the inputs of the row before it (the STG "perturbed" row is the text row until its first
        skipped block, pipeline_ltx_video.py:1035-1051) -- the first ``stg_alias_blocks`` blocks then run on
        B - 1 rows and the last row is filled in by a copy.  Bit-identical to running all rows: every kernel
        computes a row independently of the others, the GEMM kernels all accumulate over K in the same order and share
        their epilogue arithmetic (so the tile choice made from M does not matter), and the request is ignored when
        B - 1 and B rows would be served by different self-attention kernels (``ops.attention_kernel_id``).
        ``_microbatches`` (extension, set by ltxmi.distributed): a list of batch-row slices; the block loop then runs
        every block once per slice, each slice on a stream of its own, so that one slice's collectives (sequence
        parallelism) are hidden behind the other slices' kernels.  Rows are independent: same result."""
        if self.dtype != BF16:
            raise TypeError("ltxmi.Transformer3DModel runs in bfloat16 only: call .to(torch.bfloat16)")
        if mixed:
            raise NotImplementedError("mixed (fp32 residual stream) precision is not on this path")
        if attention_mask is not None:
            raise NotImplementedError("a self-attention mask is not on this path")
        dtype = self.dtype
        hidden_states = hidden_states.to(dtype)
        B, N, _ = hidden_states.shape
        D = self.inner_dim

        # mask -> additive bias (keep +0 / discard -10000), transformer3d.py:411-415
        if encoder_attention_mask is not None and encoder_attention_mask.ndim == 2:
            encoder_attention_mask = ((1 - encoder_attention_mask.to(dtype)) * -10000.0).unsqueeze(1)

        # 1. input projection
        hidden_states = ops.gemm(hidden_states.reshape(B * N, -1), self.patchify_proj.weight,
                                 self.patchify_proj.bias).view(B, N, D)

        # timestep: [B,1] (t2v) or per-token [B,N] -> per-frame [B,F] (:420-425)
        timestep = timestep.to(torch.float32)
        if self.timestep_scale_multiplier:
            timestep = self.timestep_scale_multiplier * timestep
        if timestep.shape[-1] > 1:
            timestep = timestep.reshape(timestep.shape[0], -1, latent_shape[-2] * latent_shape[-1])[:, :, 0]
        temb, embedded_timestep = self.adaln_single(timestep.flatten())
        temb = temb.view(B, -1, 6 * D)
        embedded_timestep = embedded_timestep.view(B, -1, D)

        # 2. text projection
        if self.caption_projection is not None:
            # the prompt does not change between the denoise steps of a generation: project it once per
            # (prompt tensor, weights) and hand the SAME tensor to the blocks, whose text K/V caches key on it
            ehs = encoder_hidden_states
            cp = self.caption_projection
            key = (ehs.data_ptr(), tuple(ehs.shape), ehs.dtype, ops.tensor_version(ehs), cp.linear_1.weight.data_ptr(),
                   ops.tensor_version(cp.linear_1.weight), cp.linear_2.weight.data_ptr(), ops.tensor_version(cp.linear_2.weight))
            cache = self.__dict__.setdefault("_caption_cache", {})
            hit = cache.get(key) if ops.STEP_INVARIANT_CACHING else None
            if hit is None:
                if len(cache) >= 4:
                    cache.clear()
                hit = (cp(ehs.to(dtype)).view(B, -1, D), ehs)                    # ehs kept alive: no address reuse
                if ops.STEP_INVARIANT_CACHING:
                    cache[key] = hit
            encoder_hidden_states = hit[0]

        if joint_pass and ops.STACKED_TEXT_KV and not ops.STEP_INVARIANT_CACHING and encoder_hidden_states is not None:
            self._stacked_text_kv(encoder_hidden_states)

        host_rows = None
        if skip_layer_mask is not None:
            host_rows = getattr(skip_layer_mask, "_ltxmi_host_rows", None)
            if host_rows is None:                      # one D2H copy per forward instead of one per block
                host_rows = [[float(x) for x in row] for row in skip_layer_mask.float().cpu().tolist()]

        def layer_mask(block_idx, sl=None):
            if skip_layer_mask is None:
                return None
            m = skip_layer_mask[block_idx] if sl is None else skip_layer_mask[block_idx, sl]
            rows = host_rows[block_idx]
            m._ltxmi_host = rows if sl is None else rows[sl]
            return m

        first_block = 0
        if joint_pass and stg_alias_blocks and B >= 2:
            heads, dh = self.num_attention_heads, self.attention_head_dim
            if ops.attention_kernel_id(B, heads, N, N, dh) != ops.attention_kernel_id(B - 1, heads, N, N, dh):
                stg_alias_blocks = 0                                        # a sub-batch would not reproduce the full batch's bits
        if joint_pass and stg_alias_blocks and B >= 2:
            first_block = min(int(stg_alias_blocks), len(self.transformer_blocks))
            keep = slice(0, B - 1)
            hs = hidden_states[keep]                                        # a view: the blocks update it in place
            for block_idx in range(first_block):
                self.transformer_blocks[block_idx](
                    hs, freqs_cis=freqs_cis, attention_mask=attention_mask,
                    encoder_hidden_states=encoder_hidden_states[keep],
                    encoder_attention_mask=None if encoder_attention_mask is None else encoder_attention_mask[keep],
                    timestep=temb[keep], cross_attention_kwargs=cross_attention_kwargs, class_labels=class_labels,
                    skip_layer_mask=layer_mask(block_idx, keep), skip_layer_strategy=skip_layer_strategy)
                if ltxv_model is not None and ltxv_model._interrupt:
                    return [None]
            hidden_states[B - 1].copy_(hidden_states[B - 2])
        if joint_pass and _microbatches and len(_microbatches) > 1 and first_block == 0:
            if self._run_blocks_microbatched(hidden_states, _microbatches, freqs_cis, attention_mask, encoder_hidden_states,
                                             encoder_attention_mask, temb, cross_attention_kwargs, class_labels,
                                             layer_mask, skip_layer_strategy, ltxv_model):
                return [None]
        elif joint_pass:
            for block_idx, block in enumerate(self.transformer_blocks):
                if block_idx < first_block:
                    continue
                hidden_states = block(hidden_states, freqs_cis=freqs_cis, attention_mask=attention_mask,
                                      encoder_hidden_states=encoder_hidden_states,
                                      encoder_attention_mask=encoder_attention_mask, timestep=temb,
                                      cross_attention_kwargs=cross_attention_kwargs, class_labels=class_labels,
                                      skip_layer_mask=layer_mask(block_idx),
                                      skip_layer_strategy=skip_layer_strategy)
                if ltxv_model is not None and ltxv_model._interrupt:        # cooperative cancel, :468-469
                    return [None]
        else:
            # one cond at a time so a perturbed (STG) row can skip attention entirely (:471-487)
            for block_idx, block in enumerate(self.transformer_blocks):
                for i in range(B):
                    block(hidden_states[i:i + 1], freqs_cis=freqs_cis, attention_mask=attention_mask,
                          encoder_hidden_states=encoder_hidden_states[i:i + 1],
                          encoder_attention_mask=encoder_attention_mask[i:i + 1], timestep=temb[i:i + 1],
                          cross_attention_kwargs=cross_attention_kwargs, class_labels=class_labels,
                          skip_layer_mask=layer_mask(block_idx, slice(i, i + 1)),
                          skip_layer_strategy=skip_layer_strategy)
                    if ltxv_model is not None and ltxv_model._interrupt:
                        return [None]

        for block in self.transformer_blocks:                                    # this forward's hand-over ends here
            if getattr(block, "attn2", None) is not None:
                block.attn2.__dict__.pop("_text_kv_ready", None)

        # 3. output: LayerNorm (no affine, 1e-6) -> (1 + scale) x + shift -> proj_out (:489-503)
        T1 = embedded_timestep.shape[1]
        emb2 = embedded_timestep.reshape(B * T1, D)
        normed = torch.empty_like(hidden_states)
        ops.norm_modulate(hidden_states.view(B * N, D), normed.view(B * N, D), 1e-6, ops.NORM_LAYER,
                          self.scale_shift_table[1], emb2, self.scale_shift_table[0], emb2, N // T1)
        out = ops.gemm(normed.view(B * N, D), self.proj_out.weight, self.proj_out.bias).view(B, N, -1)
        if not return_dict:
            return (out,)
        return Transformer3DModelOutput(sample=out)
