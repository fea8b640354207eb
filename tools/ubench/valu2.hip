// valu2.hip -- micro-benchmark: SIMD-level throughput of VALU instruction forms on gfx950 at 1 / 2 / 4 waves per SIMD.
// Diagnostic only.  hipcc -O3 --offload-arch=gfx950 tools/ubench/valu2.hip -o valu2
// Each wave runs 16 independent copies of one instruction per loop iteration (each depends only on its own
// result of the previous iteration); reported: SIMD cycles per wave-instruction = elapsed / (iters * 16 * waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define BODY16(ASM, ...)                                              \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) asm volatile(ASM : "+v"(x[e]) : __VA_ARGS__);
template <int V>
__global__ void k(float* out, unsigned long long* cyc, int iters, float c, float d) {
    const int lane = threadIdx.x & 63;
    float x[16], y[16];
    for (int e = 0; e < 16; ++e) { x[e] = 0.001f * (lane + e); y[e] = 1.0f + 0.001f * e; }
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 xp[16];
    for (int e = 0; e < 16; ++e) xp[e] = f2{x[e], y[e]};
    const f2 cp = {c, c}, dp = {d, d};
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if (V == 0) { BODY16("v_fma_f32 %0, %0, %1, %2", "s"(c), "v"(d)) }
        if (V == 1) { BODY16("v_fma_f32 %0, %0, %1, %2", "v"(y[e]), "v"(d)) }
        if (V == 2) { BODY16("v_fmac_f32_e32 %0, %1, %2", "v"(y[e]), "v"(d)) }
        if (V == 3) { BODY16("v_sub_f32_e32 %0, %0, %1", "v"(d)) }
        if (V == 4) { BODY16("v_mul_f32_e32 %0, %1, %0", "s"(c)) }
        if (V == 5) { BODY16("v_exp_f32_e32 %0, %0", "v"(d)) }
        if (V == 6) { BODY16("v_max_f32_e32 %0, %0, %1", "v"(y[e])) }
        if (V == 7) { BODY16("v_max3_f32 %0, %0, %1, %2", "v"(y[e]), "v"(d)) }
        if (V == 8) { BODY16("v_cvt_pk_bf16_f32 %0, %0, %1", "v"(y[e])) }
        if (V == 9) { BODY16("v_perm_b32 %0, %0, %1, %2", "v"(y[e]), "s"(0x07060302)) }
        if (V == 10) {
#pragma unroll
            for (int e = 0; e < 16; ++e) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(xp[e]) : "v"(cp), "v"(dp));
        }
        if (V == 11) {
#pragma unroll
            for (int e = 0; e < 16; ++e) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xp[e]) : "v"(dp));
        }
        if (V == 12) {
#pragma unroll
            for (int e = 0; e < 16; ++e) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(xp[e]) : "v"(cp));
        }
        if (V == 13) { BODY16("v_exp_f16_e32 %0, %0", "v"(d)) }
        if (V == 14) { BODY16("v_add_u32_e32 %0, %0, %1", "v"(d)) }
        if (V == 15) { BODY16("v_mov_b32_e32 %0, %1", "v"(d)) }
        if (V == 16) { BODY16("v_fma_f32 %0, %0, %1, %1", "v"(d)) }
        if (V == 17) { BODY16("v_add_f32_e32 %0, %0, %0", "v"(d)) }
        if (V == 18) { BODY16("v_ldexp_f32 %0, %0, %1", "v"(d)) }
        if (V == 19) { BODY16("v_fract_f32_e32 %0, %0", "v"(d)) }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += x[e] + xp[e][0] + xp[e][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int V> void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 1000;
    printf("%-40s", name);
    for (int wps : {1, 2, 4}) {            // waves per SIMD = block of 256 * wps threads, one block per CU
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
        printf("  %dw/SIMD %6.2f", wps, sum / 256 / (iters * 16.0 * wps));
    }
    printf("   (SIMD cycles per wave-instruction)\n");
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    run<0>("v_fma_f32 v,v,s,v (VOP3)", out, cyc);
    run<1>("v_fma_f32 v,v,v,v (VOP3)", out, cyc);
    run<16>("v_fma_f32 v,v,v1,v1 (VOP3, 2 distinct)", out, cyc);
    run<2>("v_fmac_f32_e32 (VOP2)", out, cyc);
    run<3>("v_sub_f32_e32 (VOP2)", out, cyc);
    run<17>("v_add_f32_e32 v,v,v same reg", out, cyc);
    run<4>("v_mul_f32_e32 s (VOP2)", out, cyc);
    run<5>("v_exp_f32_e32", out, cyc);
    run<13>("v_exp_f16_e32", out, cyc);
    run<6>("v_max_f32_e32 (VOP2)", out, cyc);
    run<7>("v_max3_f32 (VOP3)", out, cyc);
    run<8>("v_cvt_pk_bf16_f32 (VOP3)", out, cyc);
    run<9>("v_perm_b32 (VOP3)", out, cyc);
    run<10>("v_pk_fma_f32", out, cyc);
    run<11>("v_pk_add_f32", out, cyc);
    run<12>("v_pk_mul_f32", out, cyc);
    run<14>("v_add_u32_e32", out, cyc);
    run<15>("v_mov_b32_e32", out, cyc);
    run<18>("v_ldexp_f32 (VOP3)", out, cyc);
    run<19>("v_fract_f32_e32", out, cyc);
    return 0;
}
