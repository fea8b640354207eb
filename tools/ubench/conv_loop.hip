// conv_loop.hip -- micro-benchmark: what makes a tap of the direct convolution (conv_direct.hip) take ~1830 cycles for the 1024
// cycles of MFMAs of a SIMD pair?  The tap's ingredients are added one at a time to a bare MFMA loop of the same shape (8 waves per
// workgroup = 2 per SIMD, one workgroup per CU, 4 position blocks x 2 k-steps x 4 channel blocks of v_mfma_f32_16x16x32_bf16 per wave
// and iteration, 64 accumulator registers, two fragment register sets).  Diagnostic only.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/conv_loop.hip -o tools/ubench/conv_loop && tools/ubench/conv_loop
// Variants (cumulative bits): 1 = sched_barrier(0) behind every block; 2 = the next iteration's 16 fragment reads (ds_read_b128)
// behind the blocks; 4 = their address arithmetic (6 VALU per block) from a per-iteration scalar; 8 = one s_barrier per iteration;
// 16 = ~85 dependent scalar instructions behind block 2 (the per-tap control); 32 = two 1-KB LDS-DMA pieces at the top + counted wait.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((address_space(1))) void* glb_ptr;

// SH = 1: the same 64 x 64 x 64 per wave and iteration as 2 x 2 blocks x 4 k-steps of v_mfma_f32_32x32x16_bf16 (16 MFMAs of 32
// cycles instead of 32 of 16), the same 16 fragment reads
template <int V, int THREADS, int SH = 0>
__global__ __launch_bounds__(THREADS) void k(const uint32_t* __restrict__ rnd, float* out, unsigned long long* cyc, int iters, int salt) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    for (int i = tid; i < (THREADS == 512 ? 36864 : 19200); i += THREADS) ((uint32_t*)lds)[i] = rnd[i & 8191] & 0x3f803f80u;   // small bf16 values
    __syncthreads();
    const int frow = lane & 15, fchunk = lane >> 4;
    bf16x8 af[2][4][2], bfr[2][4][2];
    for (int s = 0; s < 2; ++s)
        for (int i = 0; i < 4; ++i)
            for (int ks = 0; ks < 2; ++ks) {
                af[s][i][ks] = *(const bf16x8*)(lds + ((s * 8 + i * 2 + ks) * 64 + lane) * 16);
                bfr[s][i][ks] = *(const bf16x8*)(lds + 32768 + ((s * 8 + i * 2 + ks) * 64 + lane) * 16);
            }
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x16 acc32[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
    int b_off[4];
    constexpr int WOFF = THREADS == 512 ? 92160 : 32768;       // (the four-wave form's LDS is half the size)
    for (int j = 0; j < 4; ++j) b_off[j] = WOFF + (wn * 64 + j * 16 + frow) * 128 + ((fchunk ^ (frow & 7)) << 4);
    int n_off = salt, ctl = salt;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    auto body = [&](auto s_tag, auto n_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(s_tag)::value, N = decltype(n_tag)::value;
        if (V & 32) {
            __builtin_amdgcn_global_load_lds((glb_ptr)(rnd + (wave * 2) * 256 + lane * 4), (lds_ptr)(lds + WOFF + 32768 + wave * 2048), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr)(rnd + (wave * 2 + 1) * 256 + lane * 4), (lds_ptr)(lds + WOFF + 32768 + wave * 2048 + 1024), 16, 0, 0);
        }
        const char* ws = lds + ((n_off & 1) << 13);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (SH == 0) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[S][j][ks], af[S][i][ks], acc[i][j], 0, 0, 0);
            } else {
                // k-step i: fragments [i][block]
#pragma unroll
                for (int pi = 0; pi < 2; ++pi)
#pragma unroll
                    for (int cj = 0; cj < 2; ++cj)
                        acc32[pi][cj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[S][i][cj], af[S][i][pi], acc32[pi][cj], 0, 0, 0);
            }
            if (V & 2) {
                int row;
                if (V & 4) {
                    const int blk = wm * 4 + i;
                    row = (((blk >> 3) * 10 + (blk & 7)) * 18 + (n_off & 255) + frow) & 255;
                } else {
                    row = ((wm * 4 + i) * 18 + frow) & 255;
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    af[N][i][ks] = *(const bf16x8*)(lds + row * 128 + (((fchunk + 4 * ks) ^ (row & 7)) << 4));
                    bfr[N][i][ks] = *(const bf16x8*)(ws + (b_off[i] ^ (ks << 6)));
                }
            }
            if ((V & 16) && i == 2) {
                int x = ctl;
                asm volatile("" : "+s"(x));
#pragma unroll
                for (int q = 0; q < 28; ++q) {           // ~3 scalar instructions each, dependent
                    x = x * 3 + q;
                    x = x > 1000 ? x - 997 : x;
                }
                ctl = x;
            }
            if (V & 1) __builtin_amdgcn_sched_barrier(0);
        }
        if (V & 4) n_off = (n_off + 19) & 0x1ff;
        if (V & 32) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (V & 8) asm volatile("s_barrier" ::: "memory");
    };
    using s0 = std::integral_constant<int, 0>;
    using s1 = std::integral_constant<int, (V & 2) ? 1 : 0>;      // (without reads both iterations use set 0)
    for (int it = 0; it < iters; it += 2) {
        body(s0{}, s1{});
        body(s1{}, s0{});
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float sum = (float)ctl;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e) sum += acc32[i][j][e];
    out[blockIdx.x * THREADS + tid] = sum;
    if (lane == 0) cyc[blockIdx.x * (THREADS / 64) + wave] = t1 - t0;
}

template <int V, int THREADS = 512, int SH = 0>
static void run(const uint32_t* rnd, float* out, unsigned long long* cyc, const char* what) {
    const int iters = 2048, grid = THREADS == 512 ? 256 : 512;      // one 8-wave workgroup or two 4-wave workgroups per CU
    const int smem = THREADS == 512 ? 147456 : 76800;
    hipFuncSetAttribute((const void*)k<V, THREADS, SH>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<V, THREADS, SH>), dim3(grid), dim3(THREADS), smem, 0, rnd, out, cyc, iters, 5);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<V, THREADS, SH>), dim3(grid), dim3(THREADS), smem, 0, rnd, out, cyc, iters, 5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * (THREADS / 64));
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cpi = (double)h[h.size() / 2] / iters;
    const double flop = 2.0 * 64 * 64 * 64 * (THREADS / 64) * grid * (double)iters;      // per wave and iteration: 64 x 64 x 64
    printf("%s V=%2d x%d %-58s: %7.1f cycles / iteration (matrix pipe: 1024 per SIMD pair)  %6.3f ms  %6.0f TFLOP/s\n", SH ? "32x32x16" : "16x16x32", V, THREADS / 64, what, cpi, ms, flop / ms / 1e9);
}

int main() {
    uint32_t* rnd; float* out; unsigned long long* cyc;
    std::vector<uint32_t> h(8192 * 4);
    srand(1);
    for (auto& x : h) x = (uint32_t)rand() * 2654435761u;
    hipMalloc(&rnd, h.size() * 4); hipMalloc(&out, 512 * 512 * 4); hipMalloc(&cyc, 512 * 8 * 8);
    hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0>(rnd, out, cyc, "bare MFMAs");
    run<1>(rnd, out, cyc, "+ sched_barrier per block");
    run<3>(rnd, out, cyc, "+ 16 fragment reads per iteration");
    run<7>(rnd, out, cyc, "+ their address arithmetic");
    run<15>(rnd, out, cyc, "+ s_barrier per iteration");
    run<31>(rnd, out, cyc, "+ ~85 scalar instructions behind block 2");
    run<63>(rnd, out, cyc, "+ two LDS-DMA pieces + vmcnt(0)");
    run<9>(rnd, out, cyc, "bare MFMAs + s_barrier");
    run<17>(rnd, out, cyc, "bare MFMAs + scalar blob");
    run<33>(rnd, out, cyc, "bare MFMAs + LDS-DMA");
    run<11>(rnd, out, cyc, "MFMAs + reads + s_barrier");
    // the same with FOUR waves per workgroup and two workgroups per CU (the two waves of a SIMD belong to different workgroups)
    run<0, 256>(rnd, out, cyc, "bare MFMAs");
    run<15, 256>(rnd, out, cyc, "+ reads, addresses, s_barrier");
    run<31, 256>(rnd, out, cyc, "+ ~140 scalar instructions behind block 2");
    run<63, 256>(rnd, out, cyc, "+ two LDS-DMA pieces + vmcnt(0)");
    run<17, 256>(rnd, out, cyc, "bare MFMAs + scalar blob");
    // the 32x32x16 shape: half the MFMA instructions for the same work
    run<0, 512, 1>(rnd, out, cyc, "bare MFMAs");
    run<3, 512, 1>(rnd, out, cyc, "+ 16 fragment reads per iteration");
    run<15, 512, 1>(rnd, out, cyc, "+ addresses, s_barrier");
    run<31, 512, 1>(rnd, out, cyc, "+ scalar chain");
    run<63, 512, 1>(rnd, out, cyc, "+ two LDS-DMA pieces + vmcnt(0)");
    run<0, 256, 1>(rnd, out, cyc, "bare MFMAs");
    run<15, 256, 1>(rnd, out, cyc, "+ reads, addresses, s_barrier");
    run<31, 256, 1>(rnd, out, cyc, "+ scalar chain");
    run<63, 256, 1>(rnd, out, cyc, "+ two LDS-DMA pieces + vmcnt(0)");
    return 0;
}
