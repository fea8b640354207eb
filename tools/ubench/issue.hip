// issue.hip -- micro-benchmark: what one wave per SIMD can issue beside v_mfma_f32_32x32x16_bf16 (gfx950).
// Diagnostic only (not part of libltxmi).  Build twice: with and without -mllvm -amdgpu-mfma-vgpr-form.
//   hipcc -O3 --offload-arch=gfx950 [-mllvm -amdgpu-mfma-vgpr-form] tools/ubench/issue.hip -o issue_[v|a]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

// V: 0 MFMA only (one dependent chain)   1 MFMA + 2 fma + 2 exp + cvt   2 VALU only (2 fma + 2 exp + cvt)
//    3 MFMA + 2 fma + cvt (no exp)       4 MFMA + 2 exp                  5 MFMA + 4 fma
//    6 MFMA (two alternating chains) + 2 fma + 2 exp + cvt               7 MFMA + 1 exp + 3 fma
template <int V>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters, float c) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (lane + e)); b[e] = (__bf16)(0.02f * (lane - e)); }
    f32x16 acc0, acc1;
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    float x[16];
    for (int e = 0; e < 16; ++e) x[e] = 0.001f * (lane + e);
    uint32_t sink = 0;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (V != 2) {
                if (V == 6 && (j & 1)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            }
            asm volatile("" : "+v"(x[2 * j]), "+v"(x[2 * j + 1]));     // opaque per iteration: no hoisting
            if (V == 1 || V == 2 || V == 6) {
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(x[2 * j], c, -1.0f));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(x[2 * j + 1], c, -1.0f));
                uint32_t w = pack_bf16(p0, p1);
                asm volatile("" : "+v"(w));
                sink ^= w;
            } else if (V == 3) {
                const float p0 = __builtin_fmaf(x[2 * j], c, -1.0f);
                const float p1 = __builtin_fmaf(x[2 * j + 1], c, -1.0f);
                uint32_t w = pack_bf16(p0, p1);
                asm volatile("" : "+v"(w));
                sink ^= w;
            } else if (V == 4) {
                float p0 = __builtin_amdgcn_exp2f(x[2 * j]);
                float p1 = __builtin_amdgcn_exp2f(x[2 * j + 1]);
                asm volatile("" : "+v"(p0), "+v"(p1));
                sink ^= __float_as_uint(p0) ^ __float_as_uint(p1);
            } else if (V == 5) {
                float p0 = __builtin_fmaf(x[2 * j], c, -1.0f), p1 = __builtin_fmaf(x[2 * j + 1], c, -1.0f);
                float p2 = __builtin_fmaf(x[2 * j], c, -2.0f), p3 = __builtin_fmaf(x[2 * j + 1], c, -3.0f);
                asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
                sink ^= __float_as_uint(p0) ^ __float_as_uint(p1) ^ __float_as_uint(p2) ^ __float_as_uint(p3);
            } else if (V == 7) {
                float p0 = __builtin_amdgcn_exp2f(x[2 * j]);
                float p1 = __builtin_fmaf(x[2 * j + 1], c, -1.0f), p2 = __builtin_fmaf(x[2 * j], c, -2.0f), p3 = __builtin_fmaf(x[2 * j + 1], c, -3.0f);
                asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
                sink ^= __float_as_uint(p0) ^ __float_as_uint(p1) ^ __float_as_uint(p2) ^ __float_as_uint(p3);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e];
    out[blockIdx.x * 256 + threadIdx.x] = s + __uint_as_float(sink & 1);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V> void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, out, cyc, iters, 0.5f);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mn = ~0ull, sum = 0;
    for (int i = 0; i < 256; ++i) { mn = h[i] < mn ? h[i] : mn; sum += h[i]; }
    printf("%-48s  %7.1f cycles per MFMA gap (min %7.1f)\n", name, (double)sum / 256 / (iters * 8), (double)mn / (iters * 8));
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<0>("MFMA only (one chain)", out, cyc);
    run<2>("VALU only: 2 fma + 2 exp + cvt", out, cyc);
    run<1>("MFMA + 2 fma + 2 exp + cvt", out, cyc);
    run<6>("MFMA (2 chains) + 2 fma + 2 exp + cvt", out, cyc);
    run<3>("MFMA + 2 fma + cvt", out, cyc);
    run<4>("MFMA + 2 exp", out, cyc);
    run<5>("MFMA + 4 fma", out, cyc);
    run<7>("MFMA + 1 exp + 3 fma", out, cyc);
    return 0;
}
