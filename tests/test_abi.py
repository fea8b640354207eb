"""CPU-side checks of the drop-in boundary: libltxmi.so loads, exports every symbol that
include/ltxmi.h declares, and rejects bad arguments before touching a GPU (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ltxmi.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ltxmi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from ltxmi import _lib
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} is declared in include/ltxmi.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in ltxmi/_lib.py"
    assert set(_lib.SIGNATURES) == set(names)
    assert _lib.lib.ltxmi_arch() == b"gfx950"
    assert b"ltxmi" in _lib.lib.ltxmi_version()


def test_header_cites_the_reference_for_every_entry_point():
    text = open(HEADER).read()
    for needle in ("attention.py:", "transformer3d.py:", "wan/modules/attention.py:", "causal_conv3d.py:",
                   "causal_video_autoencoder.py:", "pipeline_ltx_video.py:", "rf.py:"):
        assert needle in text, needle


def test_argument_validation_without_gpu():
    from ltxmi import _lib
    lib = _lib.lib
    # NULL struct
    assert lib.ltxmi_gemm_bf16(None, None) == -1
    assert b"NULL" in lib.ltxmi_last_error()
    a = _lib.GemmArgs()
    buf = ctypes.create_string_buffer(4096 + 64)
    base = (ctypes.addressof(buf) + 63) & ~63
    a.A = a.W = a.C = base
    a.M, a.N, a.K = 16, 16, 100            # K not a multiple of 64
    a.lda = a.ldw = 128
    a.ldc = 16
    assert lib.ltxmi_gemm_bf16(ctypes.byref(a), None) == -2
    assert b"multiple of 64" in lib.ltxmi_last_error()
    at = _lib.AttnArgs()
    at.q = at.k = at.v = at.o = base
    at.B, at.H, at.Lq, at.Lk, at.head_dim = 1, 1, 8, 8, 48
    assert lib.ltxmi_attention_fwd_bf16(ctypes.byref(at), None) == -2
    assert b"head_dim" in lib.ltxmi_last_error()
    at.head_dim = 64
    at.Lk = 0
    assert lib.ltxmi_attention_fwd_bf16(ctypes.byref(at), None) == -1
    c = _lib.Conv3dArgs()
    c.x = c.w = c.y = base
    c.B, c.T, c.H, c.W, c.Cin, c.Cout = 1, 1, 2, 2, 32, 64     # Cin % 64 != 0
    assert lib.ltxmi_conv3d_ndhwc_bf16(ctypes.byref(c), None) == -2
    assert lib.ltxmi_silu_bf16(None, None, 8, None) == -1
    assert lib.ltxmi_rmsnorm_rope_bf16(base, 64, 4, 60, base, 1e-5, None, None, 0, 0, None) == -2


def test_host_ops_refuse_cpu_tensors():
    """The product path has no CPU fallback: CPU tensors are an error, not a slow path."""
    import torch
    from ltxmi import ops
    x = torch.zeros(4, 64, dtype=torch.bfloat16)
    w = torch.zeros(8, 64, dtype=torch.bfloat16)
    with pytest.raises(TypeError):
        ops.gemm(x, w)
    with pytest.raises(TypeError):
        ops.gemm(x.float(), w.float())


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ltx-video-gpupoor_amd", "ltxmi")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("the oracle", ""), f"{fn} mentions the oracle package"
            assert "/root/reference" not in src
