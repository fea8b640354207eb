#!/usr/bin/env python3
"""A/B builds of libltxmi.so on the self-attention shapes inside ONE process (alternating launches on the
same tensors: clocks / box / data are common to all arms), plus a cross-check of every arm against arm 0.
    python tools/ab_attn.py libA.so libB.so [libC.so ...]
AB_Q_ON_LOAD=1: launch as the DiT does (q's row factor and weight applied on load: the head_dim-64 kernel's QSCALED instance).
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from ltxmi import _lib  # noqa: E402


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    lib.ltxmi_attention_fwd_bf16.restype = ctypes.c_int32
    lib.ltxmi_attention_fwd_bf16.argtypes = [ctypes.POINTER(_lib.AttnArgs), ctypes.c_void_p]
    lib.ltxmi_last_error.restype = ctypes.c_char_p
    return lib


def main():
    libs = [load(p) for p in sys.argv[1:]]
    shapes = [(3, 32, 4992, 64, 10), (1, 32, 4992, 64, 10), (1, 32, 13376, 64, 5), (1, 32, 32768, 64, 2), (1, 32, 98304, 64, 1)]
    if os.environ.get("AB_SHAPES") == "small":
        shapes = shapes[:3]
    if os.environ.get("AB_SHAPES") == "ulysses":       # heads per rank of the Ulysses mode at P = 2, 4, 8 (N 4992 and 13376)
        shapes = [(3, 16, 4992, 64, 10), (3, 8, 4992, 64, 10), (3, 4, 4992, 64, 10), (3, 16, 13376, 64, 5), (3, 8, 13376, 64, 5),
                  (3, 4, 13376, 64, 5)]
    if os.environ.get("AB_SHAPES") == "dh128":         # config 4 (Wan 1.3B self-attention) and 13B-like LTX shapes
        shapes = [(1, 12, 32760, 128, 3), (1, 32, 4992, 128, 10), (3, 32, 4992, 128, 5), (1, 32, 13376, 128, 3)]
    stream = torch.cuda.current_stream().cuda_stream
    for (B, H, N, dh, it) in shapes:
        qkv = torch.randn(B, N, 3, H, dh, device="cuda").to(torch.bfloat16)
        outs = [torch.empty(B, N, H, dh, device="cuda", dtype=torch.bfloat16) for _ in libs]
        args, keep = [], []
        for o in outs:
            a = _lib.AttnArgs()
            q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
            a.q, a.q_stride_b, a.q_stride_l = q.data_ptr(), q.stride(0), q.stride(1)
            a.k, a.k_stride_b, a.k_stride_l = k.data_ptr(), k.stride(0), k.stride(1)
            a.v, a.v_stride_b, a.v_stride_l = v.data_ptr(), v.stride(0), v.stride(1)
            a.o, a.o_stride_b, a.o_stride_l = o.data_ptr(), o.stride(0), o.stride(1)
            a.key_bias, a.bias_stride_b = None, 0
            a.B, a.H, a.Lq, a.Lk, a.head_dim, a.softmax_scale = B, H, N, N, dh, dh ** -0.5
            if os.environ.get("AB_Q_ON_LOAD") == "1":      # the DiT's launch form: q finished on load (row factor x weight)
                keep.append((torch.ones(B * N, device="cuda", dtype=torch.float32), torch.ones(H * dh, device="cuda", dtype=torch.bfloat16)))
                a.q_rstd, a.q_rstd_stride_b, a.q_rstd_stride_l = keep[-1][0].data_ptr(), N, 1
                a.q_norm_weight, a.q_norm_eps = keep[-1][1].data_ptr(), 1e-6
            args.append(a)
        times = [[] for _ in libs]
        for rep in range(6):
            for i, lib in enumerate(libs):
                rc = lib.ltxmi_attention_fwd_bf16(ctypes.byref(args[i]), stream)
                assert rc == 0, lib.ltxmi_last_error()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(it):
                    lib.ltxmi_attention_fwd_bf16(ctypes.byref(args[i]), stream)
                e1.record()
                torch.cuda.synchronize()
                if rep > 0:
                    times[i].append(e0.elapsed_time(e1) / it)
        med = [sorted(t)[len(t) // 2] for t in times]
        tf = [4.0 * B * H * N * N * dh / m / 1e9 for m in med]
        ref = outs[0].float()
        errs = [float((o.float() - ref).norm() / ref.norm()) for o in outs]
        mx = [float((o.float() - ref).abs().max()) for o in outs]
        line = f"B{B} H{H} N{N} dh{dh}: " + "  ".join(
            f"[{i}] {med[i]:8.3f} ms {tf[i]:7.1f} TF ({tf[i] / 25:4.1f}%) relL2-vs-0 {errs[i]:.2e} max {mx[i]:.2e}" for i in range(len(libs)))
        print(line, flush=True)
        del qkv, outs


if __name__ == "__main__":
    main()
