#!/usr/bin/env python3
"""A/B two builds of libltxmi.so on the hot GEMM shapes inside ONE process (alternating launches on the
same tensors, so clocks / box / data are common to both arms).
    python tools/ab_gemm.py path/to/libA.so path/to/libB.so
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
    lib.ltxmi_gemm_bf16.restype = ctypes.c_int32
    lib.ltxmi_gemm_bf16.argtypes = [ctypes.POINTER(_lib.GemmArgs), ctypes.c_void_p]
    return lib


def main():
    libs = [load(p) for p in sys.argv[1:3]]
    shapes = [(14976, 8192, 2048, 1), (14976, 2048, 8192, 0), (14976, 6144, 2048, 0), (14976, 2048, 2048, 0),
              (8192, 8192, 8192, 0)]
    stream = torch.cuda.current_stream().cuda_stream
    for (M, N, K, epi) in shapes:
        a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device="cuda").to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        g = _lib.GemmArgs()
        g.A, g.lda, g.W, g.ldw, g.bias, g.C, g.ldc = a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), out.data_ptr(), N
        g.M, g.N, g.K, g.epilogue = M, N, K, epi
        times = [[], []]
        for rep in range(6):
            for i, lib in enumerate(libs):
                for _ in range(3):
                    lib.ltxmi_gemm_bf16(ctypes.byref(g), stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    lib.ltxmi_gemm_bf16(ctypes.byref(g), stream)
                e1.record()
                torch.cuda.synchronize()
                if rep > 0:
                    times[i].append(e0.elapsed_time(e1) / 10)
        med = [sorted(t)[len(t) // 2] for t in times]
        tf = [2.0 * M * N * K / m / 1e9 for m in med]
        print(f"{M}x{N}x{K} epi{epi}:  A {med[0]:.4f} ms {tf[0]:7.1f} TF   B {med[1]:.4f} ms {tf[1]:7.1f} TF   B/A {tf[1] / tf[0]:.3f}",
              flush=True)


if __name__ == "__main__":
    main()
