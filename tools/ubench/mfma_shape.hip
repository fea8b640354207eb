// mfma_shape.hip -- micro-benchmark behind "16x16x32 or 32x32x16 for the attention chunk?" (cdna_hip_programming.md rule 28:
// the chip may hold a higher clock on one bf16 MFMA shape, so build both at the same output tile per wave and keep the faster
// BY WALL on RANDOM data).  Diagnostic only.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/ubench/mfma_shape.hip -o tools/ubench/mfma_shape
// One "group" = what one 32x32x16 MFMA of the head_dim-64 attention chunk travels with: 2 v_fma + 2 v_exp + 1 v_cvt_pk
// (+ optionally the chunk's two transposed LDS reads).  Variant S32: one v_mfma_f32_32x32x16_bf16 per group; variant S16:
// two v_mfma_f32_16x16x32_bf16 per group (the same 32 x 32 x 16 of work, the same accumulator count).  Operands are random
// bf16 in [-1, 1) and rotate over four register sets so the matrix pipe sees changing data as in the kernel.
// Reported per variant and waves/SIMD: SIMD cycles per group (s_memtime), the in-kernel clock (s_memtime / s_memrealtime)
// and the wall time of the launch (hipEvents) -> delivered TFLOP/s.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

template <int SHAPE, bool LDS>
__global__ void k(const uint32_t* __restrict__ rnd, float* out, unsigned long long* cyc, int iters, float c, float d) {
    __shared__ __attribute__((aligned(16))) char lds[16384];
    const int lane = threadIdx.x & 63;
    bf16x8 a[4], b[4];
    for (int s = 0; s < 4; ++s)
        for (int e = 0; e < 8; ++e) {
            const uint32_t r = rnd[(s * 64 + lane) * 16 + e], r2 = rnd[(s * 64 + lane) * 16 + 8 + e];
            a[s][e] = (__bf16)(((int)(r & 0xffff) - 32768) / 32768.f);
            b[s][e] = (__bf16)(((int)(r2 & 0xffff) - 32768) / 32768.f);
        }
    float x[16];
    for (int e = 0; e < 16; ++e) x[e] = ((int)(rnd[4096 + lane * 16 + e] & 0xffff) - 32768) / 8192.f;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) ((uint32_t*)lds)[i] = rnd[i];
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 q0 = {0, 0, 0, 0}, q1 = {0, 0, 0, 0}, q2 = {0, 0, 0, 0}, q3 = {0, 0, 0, 0};
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t0, t1, r0, r1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    float p0 = x[14], p1 = x[15];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (SHAPE == 32) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g & 3], b[(g + 1) & 3], acc, 0, 0, 0);
            } else {
                // the same 32 x 32 x 16 MACs: four 16 x 16 output tiles x K 16 = two 16x16x32 instructions' worth per
                // group, alternating over the four accumulator tiles
                if (g & 1) {
                    q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[g & 3], b[(g + 1) & 3], q0, 0, 0, 0);
                    q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(g + 2) & 3], b[(g + 3) & 3], q1, 0, 0, 0);
                } else {
                    q2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[g & 3], b[(g + 1) & 3], q2, 0, 0, 0);
                    q3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(g + 2) & 3], b[(g + 3) & 3], q3, 0, 0, 0);
                }
            }
            if (LDS) {
                const char* base = lds + lane * 8 + g * 1024;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 512));
                asm volatile("" ::"v"(lo), "v"(hi));
            }
            float f0, f1, e0, e1;
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f0) : "v"(x[2 * g]), "s"(c), "v"(d));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f1) : "v"(x[2 * g + 1]), "s"(c), "v"(d));
            asm volatile("v_exp_f32 %0, %1" : "=v"(e0) : "v"(f0));
            asm volatile("v_exp_f32 %0, %1" : "=v"(e1) : "v"(f1));
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[g]) : "v"(p0), "v"(p1));
            p0 = e0;
            p1 = e1;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float s = p0 + p1;
    for (int e = 0; e < 16; ++e) s += acc[e];
    for (int e = 0; e < 4; ++e) s += q0[e] + q1[e] + q2[e] + q3[e];
    for (int e = 0; e < 8; ++e) s += __uint_as_float(w[e] & 1);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) {
        cyc[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2] = t1 - t0;
        cyc[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 1] = r1 - r0;
    }
}

template <int SHAPE, bool LDS>
static void run(const char* name, const uint32_t* rnd, float* out, unsigned long long* cyc) {
    const int iters = 20000;                       // ~ 5-10 ms per launch: long enough for the clock to settle
    printf("%-52s\n", name);
    for (int wps : {1, 2}) {
        const int threads = 256 * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {        // back-to-back launches; the last ones run at the settled clock
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL((k<SHAPE, LDS>), dim3(256), dim3(threads), 0, 0, rnd, out, cyc, iters, 0.999f, -0.001f);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 3 && ms < best) best = ms;
        }
        static unsigned long long h[256 * 8 * 2];
        hipMemcpy(h, cyc, sizeof(unsigned long long) * 256 * 4 * wps * 2, hipMemcpyDeviceToHost);
        double cyc_sum = 0, clk_sum = 0;
        for (int cu = 0; cu < 256; ++cu) {
            unsigned long long mx = 0, rt = 1;
            for (int i = 0; i < 4 * wps; ++i)
                if (h[(cu * 4 * wps + i) * 2] > mx) { mx = h[(cu * 4 * wps + i) * 2]; rt = h[(cu * 4 * wps + i) * 2 + 1]; }
            cyc_sum += (double)mx;
            clk_sum += (double)mx / (double)rt * 0.1;      // s_memrealtime ticks at 100 MHz -> GHz
        }
        const double flop = 2.0 * 32 * 32 * 16 * 8.0 * iters * 256 * 4 * wps;
        printf("   %dw/SIMD: %6.2f SIMD cycles per group, in-kernel clock %.3f GHz, launch %.3f ms = %7.1f TFLOP/s\n", wps,
               cyc_sum / 256 / (iters * 8.0 * wps), clk_sum / 256, best, flop / best / 1e9);
    }
}

int main() {
    uint32_t* hr = (uint32_t*)malloc(8192 * 4);
    srand(12345);
    for (int i = 0; i < 8192; ++i) hr[i] = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    uint32_t* rnd;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&rnd, 8192 * 4);
    hipMemcpy(rnd, hr, 8192 * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 8 * 2 * 8);
    run<32, false>("S32: 1 x 32x32x16 + {2 fma, 2 exp, cvt}", rnd, out, cyc);
    run<16, false>("S16: 2 x 16x16x32 + {2 fma, 2 exp, cvt}", rnd, out, cyc);
    run<32, true>("S32 + 2 ds_read_b64_tr_b16", rnd, out, cyc);
    run<16, true>("S16 + 2 ds_read_b64_tr_b16", rnd, out, cyc);
    return 0;
}
