#!/usr/bin/env python3
"""A/B several builds of libltxmi.so on the hot GEMM shapes inside ONE process (alternating launches on the same tensors,
so clocks / box / data are common to all arms); the first library is the reference of the ratios and of a bit-equality check.
    python tools/ab_gemm.py path/to/libA.so path/to/libB.so [libC.so ...]"""
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
    lib.ltxmi_gemm_bf16.restype = ctypes.c_int32
    lib.ltxmi_gemm_bf16.argtypes = [ctypes.POINTER(_lib.GemmArgs), ctypes.c_void_p]
    return lib


def main():
    libs = [load(p) for p in sys.argv[1:]]
    names = [os.path.basename(p) for p in sys.argv[1:]]
    shapes = [(14976, 8192, 2048, 1, "ff1"), (14976, 2048, 8192, 3, "ff2"), (14976, 6144, 2048, 0, "qkv"),
              (14976, 6144, 2048, 6, "qkv+ss"), (14976, 2048, 2048, 6, "q2+ss"),
              (14976, 2048, 2048, 3, "to_out"), (14976, 2048, 2048, 7, "to_out gated"), (14976, 2048, 8192, 7, "ff2 gated"),
              (4992, 8192, 2048, 1, "ff1 B1"), (8192, 8192, 8192, 0, "8k^3")]
    stream = torch.cuda.current_stream().cuda_stream
    for (M, N, K, epi, name) in shapes:
        a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device="cuda").to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        res = torch.randn(M, N, device="cuda").to(torch.bfloat16)
        g = _lib.GemmArgs()
        g.A, g.lda, g.W, g.ldw, g.bias, g.C, g.ldc = a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), out.data_ptr(), N
        g.M, g.N, g.K, g.epilogue = M, N, K, (0 if epi == 6 else 3 if epi == 7 else epi)
        if epi == 6:                                 # plain epilogue + row sums of squares over the first 2048 columns
            ss = torch.empty(M, 32, device="cuda", dtype=torch.float32)
            g.rowsumsq, g.rowsumsq_cols, g.rowsumsq_ld = ss.data_ptr(), 2048, 32
        if epi in (3, 7):
            g.residual, g.ldr = res.data_ptr(), N
        if epi == 7:                                 # x + (table + temb[batch element]) * (A W^T + b): attn1.to_out / ff.net.2 of a block
            g.epilogue = 3
            gtab = torch.randn(N, device="cuda").to(torch.bfloat16)
            gemb = torch.randn(3, 6 * N, device="cuda").to(torch.bfloat16)
            g.gate_table, g.gate_temb, g.gate_ld, g.rows_per_group = gtab.data_ptr(), gemb.data_ptr() + 2 * 2 * N, 6 * N, M // 3
        times = [[] for _ in libs]
        ref = None
        for rep in range(7):
            for i, lib in enumerate(libs):
                for _ in range(3):
                    assert lib.ltxmi_gemm_bf16(ctypes.byref(g), stream) == 0
                if rep == 0:
                    if ref is None:
                        ref = out.clone()
                    elif not torch.equal(ref, out):
                        print(f"  !! {names[i]} differs from {names[0]}: max {float((ref.float() - out.float()).abs().max()):.4g}")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    lib.ltxmi_gemm_bf16(ctypes.byref(g), stream)
                e1.record()
                torch.cuda.synchronize()
                if rep > 0:
                    times[i].append(e0.elapsed_time(e1) / 10)
        med = [sorted(t)[len(t) // 2] for t in times]
        print(f"{name:7s} {M}x{N}x{K} epi{epi}: " + " | ".join(f"{n} {m:.4f} ms {2.0 * M * N * K / m / 1e9:7.1f} TF x{med[0] / m:.3f}" for n, m in zip(names, med)),
              flush=True)


if __name__ == "__main__":
    main()
