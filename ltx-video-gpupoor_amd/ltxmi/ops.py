"""Tensor-level wrappers over the C ABI (one function per entry point of include/ltxmi.h).

PyTorch is plumbing only here: it owns the device allocations and the stream; every
FLOP of the hot path runs in libltxmi.so.  All wrappers launch on
``torch.cuda.current_stream()`` and never synchronise.
"""
import ctypes
import math

import torch

from . import _lib
from ._lib import lib, check

BF16 = torch.bfloat16

EPI_NONE, EPI_GELU_TANH, EPI_SILU, EPI_GATE_RESIDUAL = 0, 1, 2, 3
NORM_RMS, NORM_LAYER = 0, 1


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- optional per-launch timing (bench.py): HIP events recorded on the launch stream around
# the launches whose key is being watched; nothing is recorded (or allocated) otherwise.
_watch = None
_events = {}


def watch_launches(keys):
    """keys: iterable of ("gemm", M, N, K, epilogue) / ("attention", B, H, Lq, Lk, dh) /
    ("conv3d", output positions, Cin, Cout, d2s[, "plain" | "add" | "post_norm" | "second_output"]) tuples, or None."""
    global _watch
    _watch = set(keys) if keys else None
    _events.clear()


_redo_counter = None


def count_attention_redos(counter):
    """Diagnostic: an int32 CUDA tensor of one element that EVERY attention launch from now on bumps once per (batch, head,
    query tile) item it had to redo in the exact form (``ops.attention(redo_counter=)``); None switches it off.  bench.py
    keeps it on through the timed steps: the count is part of the bench line."""
    global _redo_counter
    _redo_counter = counter


def launch_times_ms():
    """{key: [ms, ...]} for the watched launches (call after a stream/device synchronize)."""
    return {k: [a.elapsed_time(b) for a, b in v] for k, v in _events.items()}


def _prof_begin(key, key2=None):
    if _watch is None:
        return None
    if key not in _watch:
        if key2 is None or key2 not in _watch:
            return None
        key = key2
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    return key, e0, e1


def _prof_end(tok):
    if tok is not None:
        tok[2].record()
        _events.setdefault(tok[0], []).append((tok[1], tok[2]))


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _chk_bf16(*ts):
    for t in ts:
        if t is None:
            continue
        if t.dtype != BF16 or not t.is_cuda:
            raise TypeError(f"ltxmi: expected a CUDA bfloat16 tensor, got {t.dtype} on {t.device}")


def _rows(t):
    """View [..., K] with contiguous last dim as (rows, ld)."""
    if t.stride(-1) != 1:
        raise ValueError("ltxmi: innermost dimension must be contiguous")
    t2 = t.reshape(-1, t.shape[-1]) if t.is_contiguous() else t
    if t2.dim() != 2:
        raise ValueError("ltxmi: expected a 2-D tensor or a contiguous N-D tensor")
    return t2, t2.shape[0], t2.stride(0)


def tensor_version(t):
    """``t._version`` for the cache keys of packed weights and derived caches.  Inference tensors (weights created,
    loaded or cast under ``torch.inference_mode()``) track no version counter and raise on the read: the key then rests
    on storage, dtype and device alone (-1 here).  Edits made through ``.data`` bump no counter either: after those
    call ``invalidate_packed()`` on the module (``Attention`` / ``CausalConv3d``)."""
    try:
        return t._version
    except RuntimeError:
        return -1


def gemm(a, w, bias=None, out=None, epilogue=EPI_NONE, residual=None, gate_table=None, gate_temb=None,
         rows_per_group=1, algo=0, rowsumsq=None, rowsumsq_cols=0, a_kblock=0, a_kblock_stride=0):
    """out[M,N] = epi(a[M,K] @ w[N,K]^T + bias).  ``a``/``out``/``residual`` may be row-strided 2-D views.
    gate_temb: 2-D view [groups, N] (row stride = gate_ld)."""
    _chk_bf16(a, w, bias, out, residual, gate_table, gate_temb)
    a2, M, lda = _rows(a)
    N, K = w.shape
    if a_kblock:
        # ``a`` is block 0 of a K-blocked operand: [M, a_kblock] rows; block j lies j * a_kblock_stride elements further
        if a2.shape[1] != a_kblock or K % a_kblock:
            raise ValueError(f"ltxmi.gemm: K-blocked a must be [M, {a_kblock}] with K={K} a multiple of the block")
    elif a2.shape[1] != K:
        raise ValueError(f"ltxmi.gemm: a is [{M},{a2.shape[1]}] but w is [{N},{K}]")
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=a.device)
    o2, Mo, ldc = _rows(out)
    if Mo != M or o2.shape[1] != N:
        raise ValueError("ltxmi.gemm: bad out shape")
    args = _lib.GemmArgs()
    args.A, args.lda = a2.data_ptr(), lda
    args.W, args.ldw = w.data_ptr(), w.stride(0)
    args.bias = bias.data_ptr() if bias is not None else None
    args.C, args.ldc = o2.data_ptr(), ldc
    args.M, args.N, args.K = M, N, K
    args.epilogue = epilogue
    if residual is not None:
        r2, Mr, ldr = _rows(residual)
        args.residual, args.ldr = r2.data_ptr(), ldr
    if gate_table is not None:
        args.gate_table = gate_table.data_ptr()
        args.gate_temb = gate_temb.data_ptr()
        args.gate_ld = gate_temb.stride(0) if gate_temb.dim() == 2 else 0
    args.rows_per_group = rows_per_group
    args.algo = algo               # 0 = kernel chosen by shape; 128 / 256: diagnostics (include/ltxmi.h)
    args.a_kblock, args.a_kblock_stride = a_kblock, a_kblock_stride
    if rowsumsq is not None:
        # fp32 [M, rowsumsq_cols/64]: per-64-column sums of squares of the bf16 outputs, columns < rowsumsq_cols
        if rowsumsq.dtype != torch.float32 or rowsumsq.dim() != 2 or rowsumsq.shape[0] != M or rowsumsq.stride(1) != 1 \
                or rowsumsq.shape[1] * 64 < rowsumsq_cols:
            raise ValueError("ltxmi.gemm: rowsumsq must be fp32 [M, >= rowsumsq_cols/64]")
        args.rowsumsq, args.rowsumsq_cols, args.rowsumsq_ld = rowsumsq.data_ptr(), rowsumsq_cols, rowsumsq.stride(0)
    tok = _prof_begin(("gemm", M, N, K, epilogue))
    check(lib.ltxmi_gemm_bf16(ctypes.byref(args), _stream()), "ltxmi_gemm_bf16")
    _prof_end(tok)
    return out


def norm_modulate(x, out, eps, kind, scale_table, scale_temb, shift_table, shift_temb, rows_per_group):
    """out = norm(x) * (1 + scale_table + scale_temb[g]) + shift_table + shift_temb[g].
    scale_temb/shift_temb: 2-D views [groups, D] sharing one row stride."""
    _chk_bf16(x, out, scale_table, scale_temb, shift_table, shift_temb)
    x2, rows, ldx = _rows(x)
    o2, _, ldy = _rows(out)
    D = x2.shape[1]
    if scale_temb.stride(0) != shift_temb.stride(0):
        raise ValueError("ltxmi.norm_modulate: scale/shift tables must share a row stride")
    check(lib.ltxmi_norm_modulate_bf16(_ptr(x2), ldx, _ptr(o2), ldy, rows, D, eps, kind,
                                       _ptr(scale_table), _ptr(scale_temb), _ptr(shift_table), _ptr(shift_temb),
                                       scale_temb.stride(0), rows_per_group, _stream()),
          "ltxmi_norm_modulate_bf16")
    return out


def rmsnorm_rope_(x, weight, eps, cos=None, sin=None, rope_period=0, rstd_of=None):
    """In place: x = rope(rmsnorm(x) * weight).  x: 2-D row-strided view [rows, D];
    cos/sin: [period, D] tables (row r uses row r % period) or None.
    rstd_of = (rowsumsq fp32 [rows, blocks], norm_dim, eps, out fp32 [rows]): riding on the launch, out[r] =
    rsqrt(sum(rowsumsq[r]) / norm_dim + eps) -- q's RMSNorm factor from the projection GEMM's partial sums."""
    _chk_bf16(x, weight, cos, sin)
    x2, rows, ldx = _rows(x)
    D = x2.shape[1]
    ld_tab = 0
    if cos is not None:
        if cos.dim() != 2 or cos.shape != sin.shape or cos.stride(0) != sin.stride(0):
            raise ValueError("ltxmi.rmsnorm_rope_: cos/sin must be matching 2-D tables")
        ld_tab = cos.stride(0)
        rope_period = rope_period or cos.shape[0]
    if rstd_of is not None:
        ss, norm_dim, norm_eps, out = rstd_of
        if ss.dtype != torch.float32 or out.dtype != torch.float32 or ss.dim() != 2 or ss.shape[0] != rows or \
                ss.stride(1) != 1 or out.shape != (rows,) or not out.is_contiguous():
            raise ValueError("ltxmi.rmsnorm_rope_: rstd_of needs fp32 sums [rows, blocks] and a contiguous fp32 out [rows]")
        check(lib.ltxmi_rmsnorm_rope_rstd_bf16(_ptr(x2), ldx, rows, D, _ptr(weight), eps, _ptr(cos), _ptr(sin), ld_tab,
                                               rope_period, _ptr(ss), ss.stride(0), ss.shape[1], norm_dim, norm_eps,
                                               _ptr(out), _stream()), "ltxmi_rmsnorm_rope_rstd_bf16")
        return x
    check(lib.ltxmi_rmsnorm_rope_bf16(_ptr(x2), ldx, rows, D, _ptr(weight), eps, _ptr(cos), _ptr(sin), ld_tab,
                                      rope_period, _stream()), "ltxmi_rmsnorm_rope_bf16")
    return x


def rowsumsq_rstd(ss, norm_dim, eps, out=None):
    """fp32 [rows, blocks] partial sums of squares -> fp32 [rows] RMSNorm factors rsqrt(sum / norm_dim + eps)."""
    if ss.dtype != torch.float32 or ss.dim() != 2 or ss.stride(1) != 1 or not ss.is_cuda:
        raise ValueError("ltxmi.rowsumsq_rstd: expected CUDA fp32 sums [rows, blocks]")
    rows = ss.shape[0]
    out = torch.empty((rows,), dtype=torch.float32, device=ss.device) if out is None else out
    check(lib.ltxmi_rowsumsq_rstd_f32(_ptr(ss), ss.stride(0), ss.shape[1], rows, norm_dim, eps, _ptr(out), _stream()),
          "ltxmi_rowsumsq_rstd_f32")
    return out


def attention_fuses_qnorm(B, H, Lq, Lk, dh, has_key_bias=False):
    """Whether ``attention(..., q_norm=...)`` is available for this shape (every shape ``attention`` accepts)."""
    return bool(lib.ltxmi_attention_fuses_qnorm(B, H, Lq, Lk, dh, int(has_key_bias)))


def attention_kernel_id(B, H, Lq, Lk, dh, has_key_bias=False, k_stride_l=None, v_stride_l=None):
    """Which kernel instance ``attention`` runs for a shape: equal ids = the same arithmetic per (batch, head, row).
    k_stride_l / v_stride_l: the token strides of k and v in elements (default: slices of the packed [.., 3 H dh]
    projection, what AttnProcessor2_0 passes)."""
    ks = 3 * H * dh if k_stride_l is None else k_stride_l
    vs = ks if v_stride_l is None else v_stride_l
    return int(lib.ltxmi_attention_kernel_id(B, H, Lq, Lk, dh, int(has_key_bias), ks, vs))


def attention(q, k, v, out=None, key_bias=None, softmax_scale=None, q_norm=None, rope=None, out_segments=None,
              redo_counter=None, force_exact=False):
    """q [B,Lq,H,dh], k/v [B,Lk,H,dh] (NHD; batch and token strides free, (H,dh) contiguous).
    key_bias: fp32 [B,Lk] additive (broadcast over heads and queries).
    q_norm = (rowsumsq fp32 [B*Lq, H*dh/64] from ``gemm(..., rowsumsq=)`` OR the finalised factor fp32 [B*Lq] from
    ``rmsnorm_rope_(..., rstd_of=)``, weight bf16 [H*dh], eps): q is the raw projection output and is RMS-normalised over
    all heads (+ rotated with rope = (cos [period, H*dh], sin, period)) while the kernel loads it.
    out_segments = (tokens per segment, elements between segments): ``out`` is segment 0's [B, segment, H, dh] view of a
    buffer whose token axis is cut into such segments (the Ulysses return all-to-all's send buffer).
    redo_counter (diagnostic): int32 device tensor of one element, incremented once per (batch, head, query tile) item the
    pipelined kernels had to run again in the exact online-softmax form; force_exact: every item in that form."""
    _chk_bf16(q, k, v, out)
    B, Lq, H, dh = q.shape
    Lk = k.shape[1]
    for t in (q, k, v):
        if t.stride(3) != 1 or t.stride(2) != dh:
            raise ValueError("ltxmi.attention: (heads, head_dim) must be contiguous")
    if out is None:
        if out_segments is not None:
            raise ValueError("ltxmi.attention: out_segments needs an explicit out")
        out = torch.empty((B, Lq, H, dh), dtype=BF16, device=q.device)
    if out.dim() != 4 or out.stride(3) != 1 or out.stride(2) != dh:
        raise ValueError("ltxmi.attention: out must be 4-D with (heads, head_dim) contiguous")
    a = _lib.AttnArgs()
    if out_segments is not None:
        seg_len, seg_stride = out_segments
        if seg_len <= 0 or Lq % seg_len != 0 or tuple(out.shape) != (B, seg_len, H, dh):
            raise ValueError(f"ltxmi.attention: out_segments=({seg_len}, {seg_stride}) needs Lq={Lq} to be a multiple of the "
                             f"segment and out to be segment 0's [B, segment, H, dh] view, got {tuple(out.shape)}")
        a.o_segment_len, a.o_stride_segment = seg_len, seg_stride
    elif tuple(out.shape) != (B, Lq, H, dh):
        raise ValueError(f"ltxmi.attention: out must be [B, Lq, H, dh] = {(B, Lq, H, dh)}, got {tuple(out.shape)}")
    a.q, a.q_stride_b, a.q_stride_l = q.data_ptr(), q.stride(0), q.stride(1)
    a.k, a.k_stride_b, a.k_stride_l = k.data_ptr(), k.stride(0), k.stride(1)
    a.v, a.v_stride_b, a.v_stride_l = v.data_ptr(), v.stride(0), v.stride(1)
    a.o, a.o_stride_b, a.o_stride_l = out.data_ptr(), out.stride(0), out.stride(1)
    if key_bias is not None:
        if key_bias.dtype != torch.float32 or key_bias.shape != (B, Lk) or key_bias.stride(1) != 1:
            raise ValueError("ltxmi.attention: key_bias must be fp32 [B, Lk]")
        a.key_bias, a.bias_stride_b = key_bias.data_ptr(), key_bias.stride(0)
    a.B, a.H, a.Lq, a.Lk, a.head_dim = B, H, Lq, Lk, dh
    a.softmax_scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(dh)
    if q_norm is not None:
        ss, w, eps = q_norm
        nb = H * dh // 64
        _chk_bf16(w)
        if ss.dim() == 1:                                    # one finalised factor per row
            if ss.dtype != torch.float32 or ss.shape != (B * Lq,) or ss.stride(0) != 1 or not ss.is_cuda:
                raise ValueError("ltxmi.attention: q_norm row factors must be a contiguous fp32 [B*Lq]")
            a.q_rstd, a.q_rstd_stride_b, a.q_rstd_stride_l = ss.data_ptr(), Lq, 1
        else:
            if ss.dtype != torch.float32 or ss.dim() != 2 or ss.shape != (B * Lq, nb) or ss.stride(1) != 1:
                raise ValueError("ltxmi.attention: q_norm row sums must be fp32 [B*Lq, H*dh/64]")
            a.q_rowsumsq, a.q_rowsumsq_stride_b, a.q_rowsumsq_stride_l = ss.data_ptr(), Lq * ss.stride(0), ss.stride(0)
            a.q_rowsumsq_blocks = nb
        a.q_norm_weight, a.q_norm_eps = w.data_ptr(), eps
        if rope is not None:
            cos, sin, period = rope
            _chk_bf16(cos, sin)
            if period not in (Lq, B * Lq) or cos.stride(0) != sin.stride(0) or cos.stride(1) != 1:
                raise ValueError("ltxmi.attention: rope tables must be [Lq or B*Lq, H*dh] with one row stride")
            a.rope_cos, a.rope_sin = cos.data_ptr(), sin.data_ptr()
            a.rope_stride_l = cos.stride(0)
            a.rope_stride_b = 0 if period == Lq else Lq * cos.stride(0)
    elif rope is not None:
        raise ValueError("ltxmi.attention: rope needs q_norm")
    if redo_counter is None:
        redo_counter = _redo_counter
    if redo_counter is not None:
        if redo_counter.dtype != torch.int32 or redo_counter.numel() != 1 or not redo_counter.is_cuda:
            raise ValueError("ltxmi.attention: redo_counter must be a CUDA int32 tensor of one element")
        a.redo_counter = redo_counter.data_ptr()
    a.force_exact = int(bool(force_exact))
    tok = _prof_begin(("attention", B, H, Lq, Lk, dh))
    check(lib.ltxmi_attention_fwd_bf16(ctypes.byref(a), _stream()), "ltxmi_attention_fwd_bf16")
    _prof_end(tok)
    return out


def qkv_norm_rope_pack(qkv, B, Nl, D, P, q_weight, k_weight, eps, cos=None, sin=None, rope_period=0, out=None):
    """qkv [B*Nl, 3D] (row-strided) -> the Ulysses send buffer [P, Nl, B, 3, D/P]: q/k RMS-normalised over all D
    channels (+ weight) and rotated, v copied, each channel in its destination rank's chunk (include/ltxmi.h)."""
    _chk_bf16(qkv, q_weight, k_weight, cos, sin, out)
    x2, rows, ld = _rows(qkv)
    if rows != B * Nl or x2.shape[1] != 3 * D:
        raise ValueError("ltxmi.qkv_norm_rope_pack: qkv must be [B*Nl, 3D]")
    if out is None:
        out = torch.empty((P, Nl, B, 3, D // P), dtype=BF16, device=qkv.device)
    if not out.is_contiguous() or out.numel() != rows * 3 * D:
        raise ValueError("ltxmi.qkv_norm_rope_pack: out must be a contiguous [P, Nl, B, 3, D/P] buffer")
    ld_tab = cos.stride(0) if cos is not None else 0
    check(lib.ltxmi_qkv_norm_rope_pack_bf16(_ptr(x2), ld, B, Nl, D, P, _ptr(q_weight), _ptr(k_weight), eps, _ptr(cos),
                                            _ptr(sin), ld_tab, rope_period, _ptr(out), _stream()),
          "ltxmi_qkv_norm_rope_pack_bf16")
    return out


def stg_blend_grouped_(a, v, m_f32):
    """a [G, B, L, Dg] contiguous (the K-blocked layout: channel block g of every row), v [B, L, >= G*Dg] with channels
    contiguous: a[g] = a[g] * m + v[..., g*Dg:(g+1)*Dg] * (1 - m), one launch."""
    _chk_bf16(a, v)
    G, B, L, Dg = a.shape
    if not a.is_contiguous() or v.dim() != 3 or v.shape[0] != B or v.shape[1] != L or v.shape[2] < G * Dg or v.stride(2) != 1:
        raise ValueError("ltxmi.stg_blend_grouped_: a must be contiguous [G,B,L,Dg], v [B,L,>=G*Dg] with unit channel stride")
    check(lib.ltxmi_stg_blend_grouped_bf16(_ptr(a), _ptr(v), Dg, v.stride(0), v.stride(1), _ptr(m_f32), G, B, L, Dg, _stream()),
          "ltxmi_stg_blend_grouped_bf16")
    return a


def silu(x, out=None):
    _chk_bf16(x, out)
    out = torch.empty_like(x) if out is None else out
    check(lib.ltxmi_silu_bf16(_ptr(x), _ptr(out), x.numel(), _stream()), "ltxmi_silu_bf16")
    return out


def add(a, b, out=None):
    _chk_bf16(a, b, out)
    out = torch.empty_like(a) if out is None else out
    check(lib.ltxmi_add_bf16(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream()), "ltxmi_add_bf16")
    return out


def timestep_embedding(t_f32, dim=256):
    """t_f32: fp32 [n] (already multiplied by timestep_scale_multiplier) -> bf16 [n, dim]."""
    if t_f32.dtype != torch.float32 or not t_f32.is_cuda or not t_f32.is_contiguous():
        raise TypeError("ltxmi.timestep_embedding: expected a contiguous CUDA fp32 vector")
    out = torch.empty((t_f32.numel(), dim), dtype=BF16, device=t_f32.device)
    check(lib.ltxmi_timestep_embedding_bf16(_ptr(t_f32), _ptr(out), t_f32.numel(), dim, _stream()),
          "ltxmi_timestep_embedding_bf16")
    return out


def stg_blend_(a, v, m_f32):
    """a [B,L,D] (rows contiguous over (B,L)), v row-strided view with the same rows; a = a*m + v*(1-m)."""
    _chk_bf16(a, v)
    B, L, D = a.shape
    a2, _, lda = _rows(a)
    v2 = v.reshape(B * L, D) if v.is_contiguous() else v
    if v2.dim() == 3:
        if v2.stride(0) != L * v2.stride(1):
            raise ValueError("ltxmi.stg_blend_: v batch stride must equal L * row stride")
        ldv = v2.stride(1)
    else:
        ldv = v2.stride(0)
    check(lib.ltxmi_stg_blend_bf16(_ptr(a2), lda, _ptr(v2), ldv, _ptr(m_f32), B, L, D, _stream()),
          "ltxmi_stg_blend_bf16")
    return a


# Convolution implementation: 0 = chosen by shape (the product setting), 1 = implicit GEMM, 2 = direct convolution.
# The parity tests run a decode both ways to check the two implementations against each other.
CONV_ALGO = 0


# conv3d(post_norm=...): let the convolution apply the PixelNorm -> AdaLN -> SiLU in its epilogue where it can (False: always
# the second launch; the A/B switch of tools/vae_time.py --ab-post-norm and of the parity tests)
CONV_POST_NORM_FUSE = True
# conv3d(post_norm=..., keep_raw=True): the consumer's norm as a second output of the producing convolution where it can
# (False: always the second launch; tools/vae_time.py --ab-second-output)
CONV_SECOND_OUTPUT_FUSE = True
# conv3d: give the library the workspace it asks for (ltxmi_conv3d_workspace_bytes), i.e. let the wide, short layers run split over
# their input channels (False: never; tools/vae_time.py --ab-split)
CONV_SPLIT = True
_conv_workspace = {}


def conv3d(x, w_packed, bias, causal, pad_replicate, d2s=False, residual=None, add=None, out=None,
           stride=(1, 1, 1), tpad=0, out_T=0, kernel_t=3, time_pad_zeros=False, algo=None, post_norm=None, keep_raw=False):
    """x [B,T,H,W,Cin] NDHWC; w_packed [Cout, 27*Cin] (tap-major; (p1p2p3, c')-major rows when d2s).
    stride = (st, s, s) with st, s in {1, 2}; tpad / out_T: front time padding and output frames
    when they differ from CausalConv3d's (0 = default); kernel_t = 1: per-frame 3x3 Conv2d
    (w_packed [Cout, 9*Cin]); time_pad_zeros: nn.Conv3d zero padding in time.  See include/ltxmi.h.
    post_norm = (scale fp32 [B,Cout] or None, shift, eps): the result goes through PixelNorm -> (1+scale) x + shift -> SiLU
    (``pixelnorm_ada_silu``) -- in the convolution's epilogue where the kernel can (ltxmi_conv3d_fuses_post_norm), as a
    second launch on the result otherwise.
    keep_raw (with post_norm): return (raw, activated) -- the raw result too, for the skip path of the block that consumes the
    activated one (the NEXT block's norm1 -> AdaLN -> SiLU riding on this convolution: ltxmi_conv3d_args.y_norm); allowed
    with `add` and with d2s."""
    _chk_bf16(x, w_packed, bias, residual, add, out)
    B, T, H, W, Cin = x.shape
    Cout = w_packed.shape[0]
    if not x.is_contiguous() or not w_packed.is_contiguous():
        raise ValueError("ltxmi.conv3d: x and w must be contiguous")
    st, sh, sw = stride
    if sh != sw:
        raise ValueError("ltxmi.conv3d: height and width strides must be equal")
    front = tpad if tpad > 0 else (2 if causal else 1)
    back = 0 if (tpad > 0 or causal) else 1
    oT = out_T if out_T > 0 else (T + front + back - 3) // st + 1
    if kernel_t == 1:
        oT = T
    if w_packed.shape[1] != 9 * kernel_t * Cin:
        raise ValueError(f"ltxmi.conv3d: packed weight has K={w_packed.shape[1]}, expected {9 * kernel_t * Cin}")
    oH, oW = (H - 1) // sh + 1, (W - 1) // sh + 1
    if out is None:
        if d2s:
            out = torch.empty((B, 2 * T - 1, 2 * H, 2 * W, Cout // 8), dtype=BF16, device=x.device)
        else:
            out = torch.empty((B, oT, oH, oW, Cout), dtype=BF16, device=x.device)
    a = _lib.Conv3dArgs()
    a.x, a.w, a.bias, a.y = x.data_ptr(), w_packed.data_ptr(), (bias.data_ptr() if bias is not None else None), out.data_ptr()
    a.B, a.T, a.H, a.W, a.Cin, a.Cout = B, T, H, W, Cin, Cout
    a.causal, a.pad_replicate, a.d2s = int(causal), int(pad_replicate), int(d2s)
    a.stride_t, a.stride_hw, a.tpad, a.out_T = st, sh, tpad, out_T
    a.kernel_t, a.time_pad_zeros = kernel_t, int(time_pad_zeros)
    a.algo = CONV_ALGO if algo is None else algo
    if residual is not None:
        a.residual, a.res_channels = residual.data_ptr(), residual.shape[-1]
    if add is not None:
        if not add.is_contiguous() or add.shape != (B, oT, oH, oW, Cout):
            raise ValueError("ltxmi.conv3d: `add` must be a contiguous [B,T,H,W,Cout] tensor")
        a.add = add.data_ptr()
    second_launch = None
    out_norm = None
    if keep_raw and post_norm is None:
        raise ValueError("ltxmi.conv3d: keep_raw is for post_norm")
    if post_norm is not None:
        scale, shift, eps = post_norm
        if (scale is None) != (shift is None) or ((d2s or add is not None) and not keep_raw):
            raise ValueError("ltxmi.conv3d: post_norm needs scale and shift together, and (without keep_raw) the plain store "
                             "without `add`")
        c_norm = Cout // 8 if d2s else Cout
        for t in (scale, shift):
            if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.shape != (B, c_norm) or not t.is_cuda):
                raise ValueError("ltxmi.conv3d: post_norm scale / shift must be contiguous fp32 [B, C] device tensors")
        a.post_norm, a.post_scale, a.post_shift, a.post_eps = 1, _ptr(scale), _ptr(shift), eps
        if keep_raw:
            out_norm = torch.empty_like(out)
            a.y_norm = out_norm.data_ptr()
    # (after the norm request is in the arguments: at 512 input channels the split is only worth it when the norm rides along)
    # Scratch for the channel-split form of the wide, short layers (ltxmi_conv3d_args.workspace): one buffer per (device, stream),
    # grown on demand, shared by every call on that stream (stream-ordered: the next call's writes follow this call's reads)
    if CONV_SPLIT:
        want = int(lib.ltxmi_conv3d_workspace_bytes(ctypes.byref(a)))
        if want > 0:
            key = (x.device, torch.cuda.current_stream().cuda_stream)
            ws = _conv_workspace.get(key)
            if ws is None or ws.numel() < want:
                if ws is None and len(_conv_workspace) >= 8:          # streams come and go: do not keep their buffers for ever
                    _conv_workspace.clear()
                ws = _conv_workspace[key] = torch.empty(want, dtype=torch.uint8, device=x.device)
            a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    if post_norm is not None:
        if not (CONV_SECOND_OUTPUT_FUSE if keep_raw else CONV_POST_NORM_FUSE) or not lib.ltxmi_conv3d_fuses_post_norm(ctypes.byref(a)):
            a.post_norm, a.post_scale, a.post_shift, a.post_eps, a.y_norm = 0, None, None, 0.0, None
            second_launch = (scale, shift, eps)
    # (watched either by shape alone or by shape + epilogue: "plain" / "add" / "post_norm" / "second_output")
    kind = ("second_output" if a.y_norm else "post_norm") if a.post_norm else ("add" if add is not None else "plain")
    tok = _prof_begin(("conv3d", B * oT * oH * oW, Cin, Cout, int(d2s)), ("conv3d", B * oT * oH * oW, Cin, Cout, int(d2s), kind))
    check(lib.ltxmi_conv3d_ndhwc_bf16(ctypes.byref(a), _stream()), "ltxmi_conv3d_ndhwc_bf16")
    _prof_end(tok)
    if second_launch is not None:
        pixelnorm_ada_silu(out, second_launch[0], second_launch[1], True, second_launch[2], out=out_norm if keep_raw else out)
    return (out, out_norm) if keep_raw else out


def pixelnorm_ada_silu(x, scale=None, shift=None, apply_silu=True, eps=1e-8, out=None):
    """x NDHWC [B,T,H,W,C]; scale/shift fp32 [B,C] or None."""
    _chk_bf16(x, out)
    B, C = x.shape[0], x.shape[-1]
    rows = x.numel() // C
    out = torch.empty_like(x) if out is None else out
    check(lib.ltxmi_pixelnorm_ada_silu_bf16(_ptr(x), _ptr(out), rows, C, rows // B, _ptr(scale), _ptr(shift),
                                            int(apply_silu), eps, _stream()), "ltxmi_pixelnorm_ada_silu_bf16")
    return out


def layernorm_affine(x, gamma, beta, eps, out=None):
    _chk_bf16(x, gamma, beta, out)
    C = x.shape[-1]
    out = torch.empty_like(x) if out is None else out
    check(lib.ltxmi_layernorm_affine_bf16(_ptr(x), _ptr(out), x.numel() // C, C, _ptr(gamma), _ptr(beta), eps,
                                          _stream()), "ltxmi_layernorm_affine_bf16")
    return out


def ncdhw_to_ndhwc(z, std=None, mean=None):
    _chk_bf16(z)
    B, C, T, H, W = z.shape
    out = torch.empty((B, T, H, W, C), dtype=BF16, device=z.device)
    check(lib.ltxmi_ncdhw_to_ndhwc_bf16(_ptr(z.contiguous()), _ptr(out), B, C, T, H, W, _ptr(std), _ptr(mean),
                                        _stream()), "ltxmi_ncdhw_to_ndhwc_bf16")
    return out


def unpatchify_to_ncdhw(x, c_out, patch):
    _chk_bf16(x)
    B, T, H, W, _ = x.shape
    out = torch.empty((B, c_out, T, H * patch, W * patch), dtype=BF16, device=x.device)
    check(lib.ltxmi_unpatchify_to_ncdhw_bf16(_ptr(x), _ptr(out), B, T, H, W, c_out, patch, _stream()),
          "ltxmi_unpatchify_to_ncdhw_bf16")
    return out


def patchify_to_ndhwc(x, patch, c_pad):
    """pixels [B,C,T,H,W] -> NDHWC [B,T,H/p,W/p,c_pad] (zero channels beyond C*p*p)."""
    _chk_bf16(x)
    B, C, T, H, W = x.shape
    out = torch.empty((B, T, H // patch, W // patch, c_pad), dtype=BF16, device=x.device)
    check(lib.ltxmi_patchify_to_ndhwc_bf16(_ptr(x.contiguous()), _ptr(out), B, C, T, H, W, patch, c_pad, _stream()),
          "ltxmi_patchify_to_ndhwc_bf16")
    return out


def space_to_depth_skip(conv, x, stride):
    """conv [B,T',H,W,Cc] (T' = T+1 when stride[0] == 2), x [B,T,H,W,Cin] -> [B,T'/st,H/s,W/s,Cc*st*s*s]."""
    _chk_bf16(conv, x)
    B, T, H, W, Cin = x.shape
    Cc = conv.shape[-1]
    st, s, _ = stride
    Tc = T + 1 if st == 2 else T
    if tuple(conv.shape) != (B, Tc, H, W, Cc) or not conv.is_contiguous() or not x.is_contiguous() or Cin % Cc:
        raise ValueError(f"ltxmi.space_to_depth_skip: conv {tuple(conv.shape)} does not match x {tuple(x.shape)}")
    out = torch.empty((B, Tc // st, H // s, W // s, Cc * st * s * s), dtype=BF16, device=x.device)
    check(lib.ltxmi_space_to_depth_skip_bf16(_ptr(conv), _ptr(x), _ptr(out), B, T, H, W, Cin, Cc, st, s, Cin // Cc,
                                             _stream()), "ltxmi_space_to_depth_skip_bf16")
    return out


def ndhwc_to_ncdhw(x, c0, C, std=None, mean=None):
    """channels c0..c0+C of NDHWC x -> NCDHW [B,C,T,H,W], optionally (v - mean) / std per channel."""
    _chk_bf16(x)
    B, T, H, W, ld = x.shape
    if not x.is_contiguous():
        raise ValueError("ltxmi.ndhwc_to_ncdhw: x must be contiguous")
    out = torch.empty((B, C, T, H, W), dtype=BF16, device=x.device)
    check(lib.ltxmi_ndhwc_to_ncdhw_bf16(_ptr(x), ld, c0, _ptr(out), B, C, T, H, W, _ptr(std), _ptr(mean), _stream()),
          "ltxmi_ndhwc_to_ncdhw_bf16")
    return out


# Loop-invariant reuse across denoise steps (caption projection, text keys/values): exact, on by default;
# bench.py switches it off for its headline number so that every step of the timed region does all the work
# the reference's step does.
STEP_INVARIANT_CACHING = True
# Without that cache every step projects the prompt through every layer's to_k / to_v (the reference's own schedule:
# attention.py:1042-1048 once per block and step).  768 text rows make a skinny GEMM per layer (192 workgroups of 128 x 128 on 256
# CUs, 11 % matrix-pipe busy); Transformer3DModel.forward therefore runs ALL layers' projections as one GEMM against the
# stacked weights before the block loop (same kernels' arithmetic, bit-identical values) and hands each block its slice.
STACKED_TEXT_KV = True


def set_step_invariant_caching(enabled: bool):
    global STEP_INVARIANT_CACHING
    STEP_INVARIANT_CACHING = bool(enabled)


GUIDANCE_WORKSPACE_FLOATS = 2048      # include/ltxmi.h: LTXMI_GUIDANCE_WORKSPACE_FLOATS


def guidance_step_(noise_pred, latents, dt, guidance_scale, stg_scale, rescaling_scale, do_cfg, do_stg, do_rescale,
                   workspace, cond_mask=None, t=0.0):
    """noise_pred bf16 [num_conds, N, C] (one sample); latents fp32/bf16 [1, N, C], updated in place.
    cond_mask fp32 [1, N] (or None): only tokens with t - 1e-6 < 1 - cond_mask are advanced."""
    _chk_bf16(noise_pred)
    num_conds = noise_pred.shape[0]
    n = noise_pred[0].numel()
    is_bf16 = latents.dtype == BF16
    if not is_bf16 and latents.dtype != torch.float32:
        raise TypeError("ltxmi.guidance_step_: latents must be fp32 or bf16")
    if not latents.is_contiguous() or latents.numel() != n:
        raise ValueError("ltxmi.guidance_step_: latents must be contiguous and match one chunk of noise_pred")
    if workspace.dtype != torch.float32 or workspace.numel() < GUIDANCE_WORKSPACE_FLOATS:
        raise ValueError(f"ltxmi.guidance_step_: workspace must hold {GUIDANCE_WORKSPACE_FLOATS} fp32 values")
    if cond_mask is None:
        check(lib.ltxmi_guidance_step_bf16(_ptr(noise_pred), n, num_conds, guidance_scale, stg_scale, rescaling_scale,
                                           int(do_cfg), int(do_stg), int(do_rescale), _ptr(latents), int(is_bf16),
                                           float(dt), _ptr(workspace), _stream()), "ltxmi_guidance_step_bf16")
        return latents
    C = noise_pred.shape[-1]
    if cond_mask.dtype != torch.float32 or not cond_mask.is_contiguous() or cond_mask.numel() * C != n:
        raise ValueError("ltxmi.guidance_step_: cond_mask must be contiguous fp32 with one value per token")
    check(lib.ltxmi_guidance_step_masked_bf16(_ptr(noise_pred), n, num_conds, guidance_scale, stg_scale,
                                              rescaling_scale, int(do_cfg), int(do_stg), int(do_rescale),
                                              _ptr(latents), int(is_bf16), float(dt), _ptr(cond_mask), C, float(t),
                                              _ptr(workspace), _stream()), "ltxmi_guidance_step_masked_bf16")
    return latents


def image_cond_noise_(latents, init_latents, noise, cond_mask, noise_scale, t):
    """latents/init_latents/noise [1, N, C] (all fp32 or all bf16), cond_mask fp32 [1, N]; in place."""
    is_bf16 = latents.dtype == BF16
    for x in (latents, init_latents, noise):
        if x.dtype != latents.dtype or x.shape != latents.shape or not x.is_contiguous():
            raise ValueError("ltxmi.image_cond_noise_: latents, init_latents and noise must match and be contiguous")
    if latents.dtype not in (BF16, torch.float32) or cond_mask.dtype != torch.float32 or not cond_mask.is_contiguous():
        raise TypeError("ltxmi.image_cond_noise_: fp32/bf16 latents and a contiguous fp32 mask expected")
    C = latents.shape[-1]
    tokens = latents.numel() // C
    if cond_mask.numel() != tokens:
        raise ValueError("ltxmi.image_cond_noise_: one mask value per token expected")
    check(lib.ltxmi_image_cond_noise(_ptr(latents), _ptr(init_latents), _ptr(noise), int(is_bf16), _ptr(cond_mask),
                                     tokens, C, float(noise_scale), float(t), _stream()), "ltxmi_image_cond_noise")
    return latents


def groupnorm_silu(x, gamma, beta, groups, eps, residual=None, samples=None, out=None):
    """x channels-last [..., C]; statistics per sample over everything but the leading ``samples``
    rows-groups (samples = B for 5-D NDHWC input by default; pass B*T for per-frame GroupNorm)."""
    _chk_bf16(x, gamma, beta, residual, out)
    if not x.is_contiguous() or (residual is not None and (not residual.is_contiguous() or residual.shape != x.shape)):
        raise ValueError("ltxmi.groupnorm_silu: x / residual must be contiguous and equally shaped")
    C = x.shape[-1]
    samples = x.shape[0] if samples is None else samples
    S = x.numel() // C // samples
    out = torch.empty_like(x) if out is None else out
    ws = torch.empty(samples * (2 * C + 2 * groups), dtype=torch.float32, device=x.device)
    check(lib.ltxmi_groupnorm_silu_bf16(_ptr(x), _ptr(out), _ptr(residual), samples, S, C, groups, _ptr(gamma),
                                        _ptr(beta), eps, _ptr(ws), _stream()), "ltxmi_groupnorm_silu_bf16")
    return out


def pixel_shuffle2d(x):
    """x NDHWC [B,T,H,W,4C] (channel (p1 p2 c)) -> [B,T,2H,2W,C]."""
    _chk_bf16(x)
    B, T, H, W, C4 = x.shape
    if not x.is_contiguous() or C4 % 32:
        raise ValueError("ltxmi.pixel_shuffle2d: contiguous input with 4*C channels (C % 8 == 0) expected")
    out = torch.empty((B, T, 2 * H, 2 * W, C4 // 4), dtype=BF16, device=x.device)
    check(lib.ltxmi_pixel_shuffle2d_ndhwc_bf16(_ptr(x), _ptr(out), B * T, H, W, C4 // 4, _stream()),
          "ltxmi_pixel_shuffle2d_ndhwc_bf16")
    return out


def adain_filter(latents, reference, factor=1.0):
    """latents [B,C,...], reference [B,C,...] (NCDHW, fp32 or bf16): per-(b,c) statistics transfer."""
    if latents.dtype != reference.dtype or latents.dtype not in (BF16, torch.float32):
        raise TypeError("ltxmi.adain_filter: fp32 or bf16 latents and reference of the same dtype expected")
    if latents.shape[:2] != reference.shape[:2]:
        raise ValueError("ltxmi.adain_filter: batch/channel sizes differ")
    latents, reference = latents.contiguous(), reference.contiguous()
    planes = latents.shape[0] * latents.shape[1]
    out = torch.empty_like(latents)
    check(lib.ltxmi_adain_filter(_ptr(latents), _ptr(reference), _ptr(out), int(latents.dtype == BF16), planes,
                                 latents.numel() // planes, reference.numel() // planes, float(factor), _stream()),
          "ltxmi_adain_filter")
    return out


def tile_blend_(a, b, extent, dim):
    """In place in ``b``: the first ``extent`` slices of ``b`` along ``dim`` cross-faded with the last ``extent`` of ``a``
    (blend_z / blend_v / blend_h, vae.py:193-221).  a, b: contiguous, same dtype (fp32 or bf16), equal sizes on every
    other axis."""
    codes = {torch.float32: 0, BF16: 1, torch.float16: 2}
    if a.dtype != b.dtype or b.dtype not in codes:
        raise TypeError("ltxmi.tile_blend_: fp32, bf16 or fp16 tensors of the same dtype expected")
    if not (a.is_contiguous() and b.is_contiguous()):
        raise ValueError("ltxmi.tile_blend_: contiguous tensors expected")
    sa, sb = list(a.shape), list(b.shape)
    if len(sa) != len(sb) or sa[:dim] + sa[dim + 1:] != sb[:dim] + sb[dim + 1:]:
        raise ValueError("ltxmi.tile_blend_: shapes differ off the blended axis")
    extent = min(sa[dim], sb[dim], int(extent))
    if extent <= 0:
        return b
    outer = 1
    for n in sb[:dim]:
        outer *= n
    inner = 1
    for n in sb[dim + 1:]:
        inner *= n
    check(lib.ltxmi_tile_blend(_ptr(a), _ptr(b), codes[b.dtype], outer, sa[dim], sb[dim], inner, extent, _stream()),
          "ltxmi_tile_blend")
    return b
