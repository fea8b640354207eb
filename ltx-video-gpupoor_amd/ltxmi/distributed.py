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
    q,k,v and one carries o back -- 2 collectives per layer instead of xfuser's 4 -- and both run on
    the buffers the kernels write / read (no pack or unpack copies, see UlyssesAttnProcessor).  xGMI is point-to-point, so an all-to-all uses all 7 links of a GPU
    at once (each peer pair its own link) and is not ring/per-link bound;
  * the model output [B, N/P, C] is all-gathered once per forward.

Requires H % P == 0 and N % P == 0 (and, for per-frame timesteps, whole frames per rank).
"""
import math
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
    """all_gather along the token axis (xdit_context_parallel.py:142): ONE ``all_gather_into_tensor`` into a
    rank-major buffer; for a single batch row that buffer IS the result (a view), otherwise one transposing copy puts
    the batch axis back in front (the model output is [B, N/P, 128]: a few MB)."""
    world = dist.get_world_size(group)
    if dim != 1:
        x = x.transpose(1, dim)
    x = x.contiguous()
    buf = torch.empty((world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(buf.view(-1), x.view(-1), group=group)          # (flat: the form every backend takes)
    if x.shape[0] == 1:
        out = buf.view((1, world * x.shape[1]) + tuple(x.shape[2:]))
    else:
        out = buf.transpose(0, 1).reshape((x.shape[0], world * x.shape[1]) + tuple(x.shape[2:]))
    return out if dim == 1 else out.transpose(1, dim)


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


def usp_attn_forward(qkv, softmax_scale, group=None, attn_fn: Optional[Callable] = None):
    """Ulysses self-attention on a packed, already normed/roped projection buffer (generic form, any attention
    function; the product path below is the zero-copy form of the same exchange).
    qkv: [B, N/P, 3, H, dh] -> returns [B, N/P, H, dh].
    Name from wan/distributed/xdit_context_parallel.py:149; the signature there (``self, x, seq_lens, grid_sizes, freqs,
    dtype``) belongs to ``WanSelfAttention.forward`` and is not reproduced."""
    if attn_fn is None:
        from . import ops
        attn_fn = lambda q, k, v, scale: ops.attention(q, k, v, softmax_scale=scale)   # noqa: E731
    full = seq_to_head_shard(qkv, group)
    o = attn_fn(full[:, :, 0], full[:, :, 1], full[:, :, 2], softmax_scale)
    return head_to_seq_shard(o.contiguous(), group)


# ------------------------------------------------------------------ processor + model forward
class UlyssesAttnProcessor:
    """Replaces AttnProcessor2_0 on ``attn1`` of every block (installed with ``Attention.set_processor``): identical
    math, plus the two all-to-alls -- which run directly on the buffers the kernels write and read:

        QKV GEMM            [B*Nl, 3D]
        qkv_norm_rope_pack  q/k RMSNorm + RoPE, v; every channel written to its destination rank's chunk
                            send  [P dst][Nl][B][3][D/P]
        all_to_all_single   recv  [P src][Nl][B][3][D/P] = [N][B][3][H/P][dh]: q, k, v of the local heads over ALL
                            tokens with uniform strides -- the attention kernel reads them in place
        attention           writes o into the return exchange's send buffer  [P dst][B][Nl][D/P]  (segmented token axis)
        all_to_all_single   recv  [P src][B*Nl][D/P]: the K-blocked A operand of to_out, consumed in place
        to_out GEMM (+ gate + residual epilogue)

    No pack / unpack copy exists anywhere (the first version of this path spent 10 % of a step in them).

    Differences from AttnProcessor2_0, both deliberate: (1) the reference's shortcut for an all-perturbed batch under
    the AttentionValues strategy (skip attention and return v, attention.py:1060-1062) is not taken -- attention always
    runs and the STG blend then selects v for the perturbed rows: the same values, and every rank issues the same
    collectives whatever its rows are; (2) an ``attention_mask`` for self-attention is not supported (the reference's
    LTX path never passes one, transformer3d.py:411-415 builds the mask for the text keys only)."""

    def __init__(self, group=None, exchange_at_world_1=None, simulate_world=None):
        """``simulate_world`` (bench.py's compute-only projection, never a product setting): run ONE rank's share of a
        P-rank step on this GPU -- N/P local tokens, the pack kernel for P destinations, attention of H/P heads over all N
        keys, the K-blocked to_out -- with both exchanges replaced by the identity (the buffers have the exchanged
        buffers' shapes; their CONTENT is not what a real exchange would deliver, so only the time means anything).
        ``exchange_at_world_1``: at world size 1 both all-to-alls are identities and are skipped (RCCL would copy the
        buffers); True -- or LTXMI_SP_FORCE_EXCHANGE=1 in the environment -- issues them anyway, which puts
        ``dist.all_to_all_single`` (RCCL on a GPU) on the real zero-copy send / receive buffers and the K-blocked
        ``to_out`` operand on a one-GPU box (tests/test_gpu_model.py)."""
        import os
        self.group = group
        if exchange_at_world_1 is None:
            exchange_at_world_1 = os.environ.get("LTXMI_SP_FORCE_EXCHANGE", "0") == "1"
        self.exchange_at_world_1 = bool(exchange_at_world_1)
        self.simulate_world = int(simulate_world) if simulate_world else None

    def __call__(self, attn, hidden_states_wrapper, freqs_cis, encoder_hidden_states=None,
                 attention_mask=None, temb=None, skip_layer_mask=None, skip_layer_strategy=None,
                 fused_residual=None, *args, **kwargs):
        from . import ops
        from .attention import _host_mask
        hidden_states = hidden_states_wrapper[0]
        hidden_states_wrapper.clear()
        assert encoder_hidden_states is None, "UlyssesAttnProcessor is for self-attention"
        if attention_mask is not None:
            raise NotImplementedError("UlyssesAttnProcessor: a self-attention mask is not on this path")
        P = self.simulate_world or dist.get_world_size(self.group)
        B, Nl, _ = hidden_states.shape
        D, H = attn.inner_dim, attn.heads
        dh = D // H
        if H % P != 0:
            raise ValueError(f"heads {H} not divisible by the sequence-parallel degree {P}")
        Hl, Dp = H // P, D // P
        wqkv, bqkv = attn.packed_qkv()
        qkv = ops.gemm(hidden_states.reshape(B * Nl, -1), wqkv, bqkv)                       # [B*Nl, 3D]
        cos, sin = freqs_cis                                   # already sliced to the local tokens
        cos2, sin2 = cos.reshape(-1, D), sin.reshape(-1, D)
        assert attn.q_norm.eps == attn.k_norm.eps
        send = ops.qkv_norm_rope_pack(qkv, B, Nl, D, P, attn.q_norm.weight, attn.k_norm.weight, attn.q_norm.eps,
                                      cos2, sin2, cos2.shape[0])                            # [P, Nl, B, 3, Dp]
        exchange = (P > 1 or self.exchange_at_world_1) and not self.simulate_world
        if exchange:
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send, group=self.group)
        else:
            recv = send                    # one rank: the exchange is the identity (RCCL would copy the buffer)
        full = recv.view(P * Nl, B, 3, Hl, dh)                                              # [N, B, 3, Hl, dh]
        q, k, v = (full[:, :, i].permute(1, 0, 2, 3) for i in range(3))                     # [B, N, Hl, dh] views
        osend = torch.empty((P, B, Nl, Hl, dh), dtype=qkv.dtype, device=qkv.device)
        ops.attention(q, k, v, out=osend[0], softmax_scale=attn.scale, out_segments=(Nl, B * Nl * Dp))
        if exchange:
            orecv = torch.empty_like(osend)
            dist.all_to_all_single(orecv, osend, group=self.group)                          # [P src][B*Nl][Dp]
        else:
            orecv = osend
        host_mask = _host_mask(skip_layer_mask) if skip_layer_mask is not None else None
        if host_mask is not None and any(m != 1.0 for m in host_mask):
            # STG blends (attention.py:1127-1141) on the K-blocked layout [P][B, Nl][Dp] in ONE launch: channel block p of
            # token (b, n) of the blended-in tensor lies p * Dp further along its row
            m_dev = skip_layer_mask.reshape(B).to(torch.float32)
            if skip_layer_strategy == SkipLayerStrategy.AttentionValues:
                ops.stg_blend_grouped_(orecv.view(P, B, Nl, Dp), qkv.view(B, Nl, 3 * D)[:, :, 2 * D:], m_dev)
            elif skip_layer_strategy == SkipLayerStrategy.AttentionSkip:
                ops.stg_blend_grouped_(orecv.view(P, B, Nl, Dp), hidden_states, m_dev)
        w_o, b_o = attn.to_out[0].weight, attn.to_out[0].bias
        a_blk0 = orecv[0].view(B * Nl, Dp)
        # the K-blocked reading of the receive buffer ([P src][B Nl][Dp]); at P = 1 there is one block = the plain matrix,
        # described as K-blocked only when the exchange is forced (so that code path runs on one GPU too)
        kb = dict(a_kblock=Dp, a_kblock_stride=B * Nl * Dp) if (P > 1 or (exchange and Dp % 64 == 0)) else {}
        if fused_residual is not None:
            residual, gate_table, gate_temb, rpg = fused_residual
            ops.gemm(a_blk0, w_o, b_o, out=residual.reshape(B * Nl, -1),
                     epilogue=ops.EPI_GATE_RESIDUAL, residual=residual.reshape(B * Nl, -1),
                     gate_table=gate_table, gate_temb=gate_temb, rows_per_group=rpg, **kb)
            return residual
        return ops.gemm(a_blk0, w_o, b_o, **kb).view(B, Nl, -1)


def enable_sequence_parallel(model, group=None, overlap=True, bind_forward=False, exchange_at_world_1=None):
    """Install the Ulysses processor on every block's self-attention.

    ``bind_forward`` (the reference's pattern for its Wan model, ``wan/text2video.py``: ``model.forward =
    types.MethodType(usp_dit_forward, model)``): rebind ``model.forward`` to ``usp_dit_forward`` so that an unchanged caller
    -- ``LTXVideoPipeline.__call__`` -- runs sequence-parallel; every rank passes the full inputs and gets the full
    output.  ``disable_sequence_parallel`` undoes both.

    ``overlap`` (default on; takes effect at world size > 1 with >= 2 batch rows): the block loop runs the batch as TWO
    micro-batches, each on a stream of its own (``Transformer3DModel.forward(_microbatches=...)``).  Inside a block
    everything depends on the step before it, so an exchange can only be hidden behind work of ANOTHER batch row: while
    one micro-batch waits for its all-to-all (RCCL runs it on the process group's communication stream), the other
    one's projection / attention / to_out / FF kernels run.  Rows are computed independently of each other by every
    kernel, so the result is the same as without the split (tests: world size 2, torch.equal)."""
    for blk in model.transformer_blocks:
        assert isinstance(blk, BasicTransformerBlock)
        blk.attn1.set_processor(UlyssesAttnProcessor(group, exchange_at_world_1))
    model._sp_group = group
    model._sp_overlap = bool(overlap)
    model._sp_interrupt = _InterruptAgreement()
    model.__dict__.pop("forward", None)
    if bind_forward:
        import types

        def forward(self, hidden_states, freqs_cis, **kw):
            kw.setdefault("group", self._sp_group)
            return_dict = kw.pop("return_dict", True)
            out = usp_dit_forward(self, hidden_states, freqs_cis, _unbound=True, **kw)
            if out[0] is None or not return_dict:
                return out
            from .transformer3d import Transformer3DModelOutput
            return Transformer3DModelOutput(sample=out[0])

        model.forward = types.MethodType(forward, model)
    return model


def disable_sequence_parallel(model):
    """Back to the default processor and the class's own ``forward``."""
    from .attention import AttnProcessor2_0
    for blk in model.transformer_blocks:
        blk.attn1.set_processor(AttnProcessor2_0())
    for k in ("forward", "_sp_group", "_sp_overlap", "_sp_interrupt"):
        model.__dict__.pop(k, None)
    return model


def begin_generation(model):
    """Call at the start of every generation (``LTXVideoPipeline.__call__`` does): drops an interrupt agreement that was
    posted by the LAST forward of the previous generation and never consumed -- the reference resets ``_interrupt`` per
    generation, and a stale agreed flag would make the new generation's first forward return ``[None]`` on every rank.
    Must be called by every rank (it is local: no collective)."""
    agree = model.__dict__.get("_sp_interrupt")
    if agree is not None:
        agree.reset()


def microbatch_slices(B):
    """The two micro-batches of the overlap mode: rows [0, B/2) and [B/2, B) (B_eff = 3: the uncond row | text + STG rows)."""
    h = max(B // 2, 1)
    return [slice(0, h), slice(h, B)]


class _InterruptAgreement:
    """The reference's cooperative cancel (``ltxv_model._interrupt`` polled before every block, transformer3d.py:468-469)
    is a per-process flag: under sequence parallelism one rank leaving the block loop while the others wait in an
    all-to-all would hang the group, so the ranks must AGREE on it (MAX all-reduce).  Reading the agreed value on the
    host in the same forward would drain the launch queue once per denoise step (the host runs about a step ahead of the
    device); instead forward k posts its flag -- all-reduce, then an asynchronous copy into pinned host memory behind an
    event -- and forward k + 1 reads it: no host synchronisation on a step's critical path, every rank sees the same
    value in the same forward, and a cancel takes effect one forward (one denoise step) after it was raised.

    Contract for callers: leave the loop ONLY on the ``[None]`` a forward returns, never on the rank-local flag (a rank
    that stopped issuing forwards by itself would leave the others in forward k + 1's collectives); and call
    ``begin_generation(model)`` before the first forward of a generation, so that a flag posted by the previous
    generation's last forward (never read: there was no forward k + 1) is not taken for a cancel of the new one."""

    def __init__(self):
        self.pending = None
        self.consts = {}

    def reset(self):
        """Drop a posted-but-unread flag (start of a new generation, ``begin_generation``).  The copy into the pinned
        word is drained first, so a late write cannot land in the next generation's first read."""
        if self.pending is not None:
            _, ev = self.pending
            if ev is not None:
                ev.synchronize()
        self.pending = None

    def poll_and_post(self, local_flag, device, group):
        agreed = False
        if self.pending is not None:
            host, ev = self.pending
            if ev is not None:
                ev.synchronize()                 # recorded a whole forward ago: long complete
            agreed = bool(int(host[0]))
        if agreed:
            self.pending = None                  # consumed: the next forward starts afresh
            return True
        if device.type == "cuda":
            c = self.consts.get(device)
            if c is None:
                c = self.consts[device] = (torch.zeros(1, dtype=torch.int32, device=device),
                                           torch.ones(1, dtype=torch.int32, device=device),
                                           torch.zeros(1, dtype=torch.int32).pin_memory())
            flag = c[1 if local_flag else 0].clone()
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
            c[2].copy_(flag, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.pending = (c[2], ev)
        else:
            flag = torch.tensor([1 if local_flag else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
            self.pending = (flag, None)
        return False


def usp_dit_forward(model, hidden_states, freqs_cis, encoder_hidden_states=None, timestep=None,
                    encoder_attention_mask=None, skip_layer_mask=None, skip_layer_strategy=None,
                    latent_shape=None, group=None, _unbound=False, **kw):
    """Sequence-parallel ``Transformer3DModel.forward``: shard tokens (and the per-token inputs that
    follow them), run the model on the local shard, all-gather the output.  Every rank passes the
    FULL inputs and receives the FULL output, so it is a drop-in for ``model(...)``.

    The NAME is the reference's (wan/distributed/xdit_context_parallel.py:66), the signature is not: there it is a
    replacement for ``WanModel.forward(self, x, t, context, seq_len, clip_fea=None, y=None)`` -- a model outside this
    path -- and here it wraps ``Transformer3DModel.forward`` and so takes THAT method's arguments."""
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
    # cooperative cancel: agreed between the ranks without a host synchronisation (see _InterruptAgreement); the blocks
    # run with a frozen copy of the agreed value
    holder = kw.pop("ltxv_model", None)
    if holder is not None:
        agree = model.__dict__.setdefault("_sp_interrupt", _InterruptAgreement())
        if agree.poll_and_post(bool(getattr(holder, "_interrupt", False)), hidden_states.device, group):
            return [None]
        kw["ltxv_model"] = _Frozen()
    if getattr(model, "_sp_overlap", False) and world > 1 and hidden_states.shape[0] >= 2:
        kw["_microbatches"] = microbatch_slices(hidden_states.shape[0])
    kw.pop("return_dict", None)
    # (with ``model.forward`` rebound to this function, the shard goes to the class's own forward)
    run = (lambda *a, **k: type(model).forward(model, *a, **k)) if _unbound else model
    out = run(hs, freqs_cis=fc, encoder_hidden_states=encoder_hidden_states, timestep=timestep,
              encoder_attention_mask=encoder_attention_mask, skip_layer_mask=skip_layer_mask,
              skip_layer_strategy=skip_layer_strategy, latent_shape=latent_shape, return_dict=False, **kw)
    if out[0] is None:
        return [None]
    return (gather_tokens(out[0], group),)


class _Frozen:
    """Stand-in ``ltxv_model`` for the block loop under sequence parallelism: never interrupts mid-forward."""
    _interrupt = False


# ------------------------------------------------------------------ VAE decode: z-tiles over the ranks
def exchange_tiles(decoders, shapes, group=None, dtype=torch.float16, device=None, trace=None):
    """Two phases.  (1) This rank decodes ALL the tiles it owns (tile n belongs to rank n mod P), each straight into its
    slot of one flat send buffer -- no collective is issued before the last local decode has been enqueued, so the
    ranks' decodes run concurrently.  (2) ONE ``all_gather_into_tensor`` of the (padded) send buffers; every tile is then
    a contiguous view of the receive buffer.  The tile shapes are computed by the caller from the latent slices, so
    nothing about them has to be communicated and no host synchronisation happens anywhere.

    ``decoders[n](out)`` decodes tile n into ``out`` (a contiguous tensor of shape ``shapes[n]``); ``trace`` (tests)
    receives ("decode", n) / ("collective", name) in issue order."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    T = len(decoders)
    numel = [int(math.prod(s)) for s in shapes]
    per_rank = [sum(numel[n] for n in range(r, T, world)) for r in range(world)]
    cap = max(max(per_rank), 1)
    cap = -(-cap // 8) * 8                                    # 16-byte granules
    recv = torch.empty((world, cap), dtype=dtype, device=device)
    send = recv[rank] if world == 1 else torch.empty(cap, dtype=dtype, device=device)
    off = 0
    for n in range(rank, T, world):                            # phase 1: local decodes only
        if trace is not None:
            trace.append(("decode", n))
        decoders[n](send[off:off + numel[n]].view(shapes[n]))
        off += numel[n]
    if world > 1:                                              # phase 2: the one exchange
        if trace is not None:
            trace.append(("collective", "all_gather_into_tensor"))
        dist.all_gather_into_tensor(recv.view(-1), send, group=group)
    tiles, offs = [], [0] * world
    for n in range(T):
        r = n % world
        tiles.append(recv[r, offs[r]:offs[r] + numel[n]].view(shapes[n]))
        offs[r] += numel[n]
    return tiles


def tile_parallel_vae_decode(latents, vae, is_video=True, vae_per_channel_normalize=False, timestep=None, group=None,
                             _trace=None):
    """``vae_decode`` with the z-tiles of the tiled decode (vae.py:365-402) spread over the ranks (tiles are
    independent until the overlap blends): rank r decodes tiles r, r + P, ... back to back, ONE all-gather exchanges the
    fp16 pixels (``exchange_tiles``), then the cross-fades and the concatenation run on every rank, so every rank returns
    the full video -- the same bits as the single-rank tiled decode.  No collective inside a convolution, none before the
    last local decode.  Falls back to the plain decode when z-tiling is off or the clip is a single tile."""
    from .autoencoder import vae_decode
    world = dist.get_world_size(group)

    def exchange(decoders, shapes):
        return exchange_tiles(decoders, shapes, group=group, dtype=torch.float16, device=latents.device, trace=_trace)

    return vae_decode(latents, vae, is_video, vae_per_channel_normalize=vae_per_channel_normalize, timestep=timestep,
                      _tile_exchange=exchange if world > 1 else None)
