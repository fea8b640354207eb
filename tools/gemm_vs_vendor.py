#!/usr/bin/env python3
"""This library's GEMM against the vendor's (hipBLASLt behind torch.matmul) on the hot shapes of a denoise step, in ONE process
(same box, same clocks): per shape the product launch (with its fused epilogue), the same kernel with the plain epilogue, and the
vendor kernel (which has no epilogue at all here), plus the arithmetic of where the difference comes from:
  rounds    tiles of 256 x 256 on the device's CUs: a partial last round costs the persistent kernel a whole tile time per
            workgroup (the vendor's SK3 solutions are stream-K: profiles/r04_blas_solutions.md)
  epilogue  product launch minus plain launch (bias / GELU / gate + residual, the residual's HBM burst)
Diagnostic (calibration, not a product path); usage: python tools/gemm_vs_vendor.py"""
import ctypes
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from ltxmi import _lib  # noqa: E402

lib = _lib.lib
stream = torch.cuda.current_stream().cuda_stream
CUS = torch.cuda.get_device_properties(0).multi_processor_count


def timeit(fn, reps=7, inner=10):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    return sorted(ts)[len(ts) // 2]


def main():
    shapes = [(14976, 8192, 2048, 1, "ff.net.0 (+bias+GELU)"), (14976, 2048, 8192, 7, "ff.net.2 (+gate+residual)"),
              (14976, 6144, 2048, 6, "qkv (+bias, q row sums)"), (14976, 2048, 2048, 7, "attn1.to_out (+gate+residual)"),
              (14976, 2048, 2048, 3, "attn2.to_out (+residual)"), (8192, 8192, 8192, 0, "8192^3 (plain)")]
    print(f"# {torch.cuda.get_device_name(0)}, {CUS} CUs; times in ms (median of 7 x 10 launches), TFLOP/s in brackets")
    print("| shape | product launch | same kernel, plain epilogue | vendor (no epilogue) | rounds of 256x256 tiles | epilogue share | "
          "plain vs vendor | what a stream-K split of the last round would remove |")
    print("|---|---|---|---|---|---|---|---|")
    for (M, N, K, epi, name) in shapes:
        a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device="cuda").to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        res = torch.randn(M, N, device="cuda").to(torch.bfloat16)

        def args(e):
            g = _lib.GemmArgs()
            g.A, g.lda, g.W, g.ldw, g.bias, g.C, g.ldc = a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), out.data_ptr(), N
            g.M, g.N, g.K, g.epilogue = M, N, K, (0 if e in (0, 6) else 3 if e == 7 else e)
            keep = []
            if e == 6:
                ss = torch.empty(M, 32, device="cuda", dtype=torch.float32)
                g.rowsumsq, g.rowsumsq_cols, g.rowsumsq_ld = ss.data_ptr(), 2048, 32
                keep.append(ss)
            if e in (3, 7):
                g.residual, g.ldr = res.data_ptr(), N
            if e == 7:
                gtab = torch.randn(N, device="cuda").to(torch.bfloat16)
                gemb = torch.randn(3, 6 * N, device="cuda").to(torch.bfloat16)
                g.gate_table, g.gate_temb, g.gate_ld, g.rows_per_group = gtab.data_ptr(), gemb.data_ptr() + 4 * N, 6 * N, M // 3
                keep += [gtab, gemb]
            return g, keep

        g_prod, k1 = args(epi)
        g_plain, k2 = args(0)
        t_prod = timeit(lambda: lib.ltxmi_gemm_bf16(ctypes.byref(g_prod), stream))
        t_plain = timeit(lambda: lib.ltxmi_gemm_bf16(ctypes.byref(g_plain), stream))
        t_vendor = timeit(lambda: torch.matmul(a, w.t()))
        tiles = math.ceil(M / 256) * math.ceil(N / 256)
        rounds = tiles / CUS
        quant = 1.0 - rounds / math.ceil(rounds)
        tf = lambda t: 2.0 * M * N * K / t / 1e9      # noqa: E731
        print(f"| {name} {M}x{N}x{K} | {t_prod:.4f} ({tf(t_prod):.0f}) | {t_plain:.4f} ({tf(t_plain):.0f}) | {t_vendor:.4f} ({tf(t_vendor):.0f}) | "
              f"{rounds:.2f} | {100 * (t_prod - t_plain) / t_prod:.1f} % | x{t_vendor / t_plain:.3f} | {100 * quant:.1f} % |", flush=True)


if __name__ == "__main__":
    main()
