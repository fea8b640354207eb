// mix.hip -- micro-benchmark: the softmax instruction mix of head_dim-64 attention beside v_mfma_f32_32x32x16_bf16,
// at 1 / 2 / 4 waves per SIMD (gfx950).  Diagnostic only.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/ubench/mix.hip -o mix
// Group = what one MFMA of the kernel travels with: 2 v_fma + 2 v_exp + 1 v_cvt_pk (exp two instructions behind its
// fma, cvt of the previous group's pair).  Reported: SIMD cycles per group = elapsed / (iters * 8 * waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// V: 0 MFMA only   1 VALU group only   2 MFMA + group   3 MFMA + group with c in a VGPR   4 MFMA + 2 exp   5 MFMA + 4 plain VALU
//    6 MFMA + group, 2 LDS reads (ds_read_b128) per group   7 two 16x16x32 MFMAs + group
template <int V>
__global__ void k(float* out, unsigned long long* cyc, int iters, float c, float d) {
    __shared__ __attribute__((aligned(16))) char lds[16384];
    const int lane = threadIdx.x & 63;
    float x[16];
    for (int e = 0; e < 16; ++e) x[e] = 0.001f * (lane + e);
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (lane + e)); b[e] = (__bf16)(0.02f * (lane - e)); }
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) ((float*)lds)[i] = 0.5f;
    f32x16 acc;
    f32x4 acc4a = {0, 0, 0, 0}, acc4b = {0, 0, 0, 0};
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    uint32_t w[8];
    for (int e = 0; e < 8; ++e) w[e] = 0;
    float cv = c;
    asm volatile("" : "+v"(cv));
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    float p0 = x[14], p1 = x[15];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (V == 7) {
                acc4a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4a, 0, 0, 0);
                acc4b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4b, 0, 0, 0);
            } else if (V != 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            if (V == 6) {
                bf16x8 r0 = *(const bf16x8*)(lds + lane * 16 + g * 1024), r1 = *(const bf16x8*)(lds + lane * 16 + g * 1024 + 8192);
                asm volatile("" ::"v"(r0), "v"(r1));
            }
            float f0, f1, e0, e1;
            if (V == 1 || V == 2 || V == 6 || V == 7) {
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f0) : "v"(x[2 * g]), "s"(c), "v"(d));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f1) : "v"(x[2 * g + 1]), "s"(c), "v"(d));
            } else if (V == 3) {
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f0) : "v"(x[2 * g]), "v"(cv), "v"(d));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f1) : "v"(x[2 * g + 1]), "v"(cv), "v"(d));
            } else if (V == 5) {
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f0) : "v"(x[2 * g]), "v"(cv), "v"(d));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f1) : "v"(x[2 * g + 1]), "v"(cv), "v"(d));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(x[2 * g]), "v"(cv), "v"(cv));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(x[2 * g + 1]), "v"(cv), "v"(cv));
                asm volatile("" ::"v"(f0), "v"(f1));
            } else { f0 = x[2 * g]; f1 = x[2 * g + 1]; }
            if (V != 0 && V != 5) {
                asm volatile("v_exp_f32 %0, %1" : "=v"(e0) : "v"(f0));
                asm volatile("v_exp_f32 %0, %1" : "=v"(e1) : "v"(f1));
                if (V != 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[g]) : "v"(p0), "v"(p1));
            }
            if (V != 0) { p0 = e0; p1 = e1; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = p0 + p1;
    for (int e = 0; e < 16; ++e) s += acc[e];
    for (int e = 0; e < 4; ++e) s += acc4a[e] + acc4b[e];
    for (int e = 0; e < 8; ++e) s += __uint_as_float(w[e] & 1);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int V> void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 1000;
    printf("%-44s", name);
    for (int wps : {1, 2, 4}) {
        const int threads = 256 * wps;
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<V>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 0.999f, -0.001f);
        hipDeviceSynchronize();
        static unsigned long long h[256 * 16];
        hipMemcpy(h, cyc, sizeof(unsigned long long) * 256 * 4 * wps, hipMemcpyDeviceToHost);
        // per CU the LAST wave to finish defines the throughput (the SIMD serves its waves oldest-first, so
        // equal streams finish one after another); average that over the CUs
        double sum = 0;
        for (int cu = 0; cu < 256; ++cu) {
            unsigned long long mx = 0;
            for (int i = 0; i < 4 * wps; ++i) mx = h[cu * 4 * wps + i] > mx ? h[cu * 4 * wps + i] : mx;
            sum += (double)mx;
        }
        printf("  %dw/SIMD %6.2f", wps, sum / 256 / (iters * 8.0 * wps));
    }
    printf("   (SIMD cycles per group)\n");
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    run<0>("MFMA 32x32x16 only", out, cyc);
    run<1>("group {2 fma, 2 exp, cvt} only", out, cyc);
    run<2>("MFMA + group", out, cyc);
    run<3>("MFMA + group (scale in a VGPR)", out, cyc);
    run<4>("MFMA + 2 exp", out, cyc);
    run<5>("MFMA + 4 fma", out, cyc);
    run<6>("MFMA + group + 2 ds_read_b128", out, cyc);
    run<7>("2 x MFMA 16x16x32 + group", out, cyc);
    return 0;
}
