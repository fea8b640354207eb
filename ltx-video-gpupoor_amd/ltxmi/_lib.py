"""ctypes binding of libltxmi.so (include/ltxmi.h).

The library is the ONLY compute path of this package: if it cannot be loaded the
import fails loudly -- there is no PyTorch/CPU fallback.
"""
import ctypes
import os

# torch FIRST: its wheel ships a HIP runtime of its own (torch/lib/libamdhip64.so).  Loaded before torch, libltxmi.so pulls in the
# system's /opt/rocm runtime instead, the process then holds two HIP runtimes, and the library's calls (hipGetDevice, launches on
# torch's streams) go to the one torch never initialised: "cannot query the current device" on the first kernel.  With torch's
# runtime already in the process the library's DT_NEEDED entry resolves to it.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# LTXMI_LIB: tuning knob only (A/B runs of two builds of the SAME library); never a fallback path
LIB_PATH = os.environ.get("LTXMI_LIB") or os.path.join(_HERE, "libltxmi.so")

c_void_p, c_int, c_int64, c_float = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float


class GemmArgs(ctypes.Structure):
    _fields_ = [("A", c_void_p), ("lda", c_int64), ("W", c_void_p), ("ldw", c_int64), ("bias", c_void_p),
                ("C", c_void_p), ("ldc", c_int64), ("M", c_int), ("N", c_int), ("K", c_int),
                ("epilogue", c_int), ("residual", c_void_p), ("ldr", c_int64), ("gate_table", c_void_p),
                ("gate_temb", c_void_p), ("gate_ld", c_int64), ("rows_per_group", c_int), ("algo", c_int),
                ("rowsumsq", c_void_p), ("rowsumsq_cols", c_int), ("rowsumsq_ld", c_int64),
                ("a_kblock", c_int), ("a_kblock_stride", c_int64)]


class AttnArgs(ctypes.Structure):
    _fields_ = [("q", c_void_p), ("q_stride_b", c_int64), ("q_stride_l", c_int64),
                ("k", c_void_p), ("k_stride_b", c_int64), ("k_stride_l", c_int64),
                ("v", c_void_p), ("v_stride_b", c_int64), ("v_stride_l", c_int64),
                ("o", c_void_p), ("o_stride_b", c_int64), ("o_stride_l", c_int64),
                ("key_bias", c_void_p), ("bias_stride_b", c_int64),
                ("B", c_int), ("H", c_int), ("Lq", c_int), ("Lk", c_int), ("head_dim", c_int),
                ("softmax_scale", c_float),
                ("q_rowsumsq", c_void_p), ("q_rowsumsq_stride_b", c_int64), ("q_rowsumsq_stride_l", c_int64),
                ("q_rowsumsq_blocks", c_int), ("q_norm_weight", c_void_p), ("q_norm_eps", c_float),
                ("rope_cos", c_void_p), ("rope_sin", c_void_p), ("rope_stride_b", c_int64), ("rope_stride_l", c_int64),
                ("o_segment_len", c_int), ("o_stride_segment", c_int64),
                ("q_rstd", c_void_p), ("q_rstd_stride_b", c_int64), ("q_rstd_stride_l", c_int64),
                ("redo_counter", c_void_p), ("force_exact", c_int)]


class Conv3dArgs(ctypes.Structure):
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("y", c_void_p),
                ("B", c_int), ("T", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int), ("Cout", c_int),
                ("causal", c_int), ("pad_replicate", c_int), ("d2s", c_int), ("residual", c_void_p),
                ("res_channels", c_int), ("add", c_void_p),
                ("stride_t", c_int), ("stride_hw", c_int), ("tpad", c_int), ("out_T", c_int),
                ("kernel_t", c_int), ("time_pad_zeros", c_int), ("algo", c_int),
                ("post_norm", c_int), ("post_scale", c_void_p), ("post_shift", c_void_p), ("post_eps", c_float),
                ("y_norm", c_void_p), ("workspace", c_void_p), ("workspace_bytes", c_int64)]


# name -> (restype, argtypes); mirrors include/ltxmi.h one to one
SIGNATURES = {
    "ltxmi_version": (ctypes.c_char_p, []),
    "ltxmi_last_error": (ctypes.c_char_p, []),
    "ltxmi_arch": (ctypes.c_char_p, []),
    "ltxmi_gemm_bf16": (c_int, [ctypes.POINTER(GemmArgs), c_void_p]),
    "ltxmi_norm_modulate_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_float, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "ltxmi_rmsnorm_rope_bf16": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p,
                                        c_int64, c_int, c_void_p]),
    "ltxmi_rmsnorm_rope_rstd_bf16": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p,
                                             c_int64, c_int, c_void_p, c_int64, c_int, c_int, c_float, c_void_p, c_void_p]),
    "ltxmi_rowsumsq_rstd_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "ltxmi_qkv_norm_rope_pack_bf16": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float,
                                              c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "ltxmi_attention_fwd_bf16": (c_int, [ctypes.POINTER(AttnArgs), c_void_p]),
    "ltxmi_attention_fuses_qnorm": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "ltxmi_attention_kernel_id": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64]),
    "ltxmi_silu_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "ltxmi_timestep_embedding_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ltxmi_stg_blend_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ltxmi_stg_blend_grouped_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_int,
                                             c_void_p]),
    "ltxmi_conv3d_ndhwc_bf16": (c_int, [ctypes.POINTER(Conv3dArgs), c_void_p]),
    "ltxmi_conv3d_fuses_post_norm": (c_int, [ctypes.POINTER(Conv3dArgs)]),
    "ltxmi_conv3d_workspace_bytes": (c_int64, [ctypes.POINTER(Conv3dArgs)]),
    "ltxmi_pixelnorm_ada_silu_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p,
                                              c_int, c_float, c_void_p]),
    "ltxmi_add_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "ltxmi_layernorm_affine_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_float,
                                            c_void_p]),
    "ltxmi_ncdhw_to_ndhwc_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                          c_void_p, c_void_p]),
    "ltxmi_unpatchify_to_ncdhw_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                               c_void_p]),
    "ltxmi_guidance_step_masked_bf16": (c_int, [c_void_p, c_int64, c_int, c_float, c_float, c_float, c_int, c_int,
                                                c_int, c_void_p, c_int, c_float, c_void_p, c_int, c_float, c_void_p,
                                                c_void_p]),
    "ltxmi_image_cond_noise": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_float,
                                       c_float, c_void_p]),
    "ltxmi_groupnorm_silu_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p,
                                          c_void_p, c_float, c_void_p, c_void_p]),
    "ltxmi_pixel_shuffle2d_ndhwc_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "ltxmi_adain_filter": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_float, c_void_p]),
    "ltxmi_tile_blend": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "ltxmi_patchify_to_ndhwc_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                             c_void_p]),
    "ltxmi_space_to_depth_skip_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                               c_int, c_int, c_int, c_int, c_void_p]),
    "ltxmi_ndhwc_to_ncdhw_bf16": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                          c_void_p, c_void_p, c_void_p]),
    "ltxmi_guidance_step_bf16": (c_int, [c_void_p, c_int64, c_int, c_float, c_float, c_float, c_int, c_int, c_int,
                                         c_void_p, c_int, c_float, c_void_p, c_void_p]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"ltxmi: {LIB_PATH} is missing -- build it with `make -C ltx-video-gpupoor_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class LtxmiError(RuntimeError):
    pass


def check(status, what):
    if status != 0:
        raise LtxmiError(f"{what} failed ({status}): {lib.ltxmi_last_error().decode()}")
