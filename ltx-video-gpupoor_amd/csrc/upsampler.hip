// upsampler.hip -- the HBM-bound pieces of the multi-scale bridge between pass 1 and pass 2
// (LatentUpsampler, latent_upsampler.py:15-149; adain_filter_latent, pipeline_ltx_video.py:1709-1737):
// GroupNorm(32) (+ residual) + SiLU over channels-last activations, the 2-D pixel shuffle after the
// upsampling convolution, and the per-channel AdaIN statistics transfer.  The convolutions themselves
// are the implicit GEMM of gemm.hip (kernel_t / time_pad_zeros options).
#include <math.h>

#include "common.h"

namespace ltxmi {

constexpr int UP_THREADS = 256;

// ---------------------------------------------------------------- GroupNorm, channels-last
// x [Bn, S, C]: per (sample, channel) sum / sum of squares -> ws[Bn][C][2] (fp32, zeroed by the caller).
// Thread -> (row offset, 8-channel slot); a block reduces its rows through LDS and issues ONE atomic
// pair per channel.  C/8 is a power of two <= 256.
__global__ void gn_stats_kernel(const uint16_t* __restrict__ x, float* __restrict__ ws, int64_t S, int C,
                                int rows_per_block) {
    __shared__ float red[UP_THREADS][17];                 // [thread][8 sums + 8 squares], padded
    const int slots = C >> 3;
    const int rpi = UP_THREADS / slots;                    // rows per iteration
    const int slot = threadIdx.x % slots, roff = threadIdx.x / slots;
    const int64_t b = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > S) r1 = S;
    float s[8], q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = q[k] = 0.f;
    for (int64_t r = r0 + roff; r < r1; r += rpi) {
        const u32x4 w = *(const u32x4*)(x + (b * S + r) * C + slot * 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float lo = bf_lo(w[k]), hi = bf_hi(w[k]);
            s[2 * k] += lo; q[2 * k] += lo * lo;
            s[2 * k + 1] += hi; q[2 * k + 1] += hi * hi;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[threadIdx.x][k] = s[k]; red[threadIdx.x][8 + k] = q[k]; }
    __syncthreads();
    // thread t < slots*16 finalises value (slot, k): sum over the rpi row-offset copies
    for (int v = threadIdx.x; v < slots * 16; v += UP_THREADS) {
        const int sl = v >> 4, k = v & 15;
        float acc = 0.f;
        for (int j = 0; j < rpi; ++j) acc += red[j * slots + sl][k];
        const int ch = sl * 8 + (k & 7);
        atomicAdd(ws + ((b * C + ch) << 1) + (k >> 3), acc);
    }
}

// ws[Bn][C][2] -> stat[Bn][G][2] = (mean, rstd) over S * C/G values
__global__ void gn_finalize_kernel(const float* __restrict__ ws, float* __restrict__ stat, int C, int G, int64_t S,
                                   float eps, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int b = i / G, g = i % G, cg = C / G;
    float s = 0.f, q = 0.f;
    for (int c = g * cg; c < (g + 1) * cg; ++c) { s += ws[((int64_t)b * C + c) * 2]; q += ws[((int64_t)b * C + c) * 2 + 1]; }
    const float n = (float)S * (float)cg;
    const float mean = s / n;
    const float var = fmaxf(q / n - mean * mean, 0.f);      // biased, like nn.GroupNorm
    stat[i * 2] = mean;
    stat[i * 2 + 1] = rsqrtf(var + eps);
}

// y = silu((x - mean) * rstd * gamma + beta (+ residual))
__global__ void gn_apply_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                const uint16_t* __restrict__ res, const float* __restrict__ stat,
                                const uint16_t* __restrict__ gamma, const uint16_t* __restrict__ beta, int64_t S, int C,
                                int G, int64_t total_chunks) {
    const int slots = C >> 3, cg = C / G;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total_chunks;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int slot = (int)(i % slots);
        const int64_t row = i / slots;
        const int64_t b = row / S;
        const u32x4 w = *(const u32x4*)(x + i * 8);
        const u32x4 gm = *(const u32x4*)(gamma + slot * 8);
        const u32x4 bt = *(const u32x4*)(beta + slot * 8);
        u32x4 rr = {0u, 0u, 0u, 0u};
        if (res) rr = *(const u32x4*)(res + i * 8);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c0 = slot * 8 + 2 * k;
            const float* st0 = stat + (b * G + c0 / cg) * 2;
            const float* st1 = stat + (b * G + (c0 + 1) / cg) * 2;
            float lo = (bf_lo(w[k]) - st0[0]) * st0[1] * bf_lo(gm[k]) + bf_lo(bt[k]);
            float hi = (bf_hi(w[k]) - st1[0]) * st1[1] * bf_hi(gm[k]) + bf_hi(bt[k]);
            if (res) { lo += bf_lo(rr[k]); hi += bf_hi(rr[k]); }
            o[k] = pack_bf16(silu_f(lo), silu_f(hi));
        }
        *(u32x4*)(y + i * 8) = o;
    }
}

// ---------------------------------------------------------------- 2-D pixel shuffle, channels-last
// x [R, H, W, 4C] with channel (p1*2 + p2)*C + c (conv rows packed that way) -> y [R, 2H, 2W, C]
// = PixelShuffleND(2) "b (c p1 p2) h w -> b c (h p1) (w p2)" (pixel_shuffle.py:12-21); 16-byte copies.
__global__ void pixel_shuffle2d_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int H, int W, int C,
                                       int64_t total_chunks) {
    const int slots = C >> 3;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total_chunks;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int slot = (int)(i % slots);
        int64_t r = i / slots;
        const int xo = (int)(r % (2 * W)); r /= 2 * W;
        const int yo = (int)(r % (2 * H));
        const int64_t f = r / (2 * H);
        const int p = (yo & 1) * 2 + (xo & 1);
        const int64_t src = (((f * H + (yo >> 1)) * W + (xo >> 1)) * 4 + p) * C + slot * 8;
        *(u32x4*)(y + i * 8) = *(const u32x4*)(x + src);
    }
}

// ---------------------------------------------------------------- AdaIN (adain_filter_latent)
template <typename T> __device__ __forceinline__ float ld(const T* p, int64_t i);
template <> __device__ __forceinline__ float ld<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld<uint16_t>(const uint16_t* p, int64_t i) { return bf2f(p[i]); }
__device__ __forceinline__ void st(float* p, int64_t i, float v) { p[i] = v; }
__device__ __forceinline__ void st(uint16_t* p, int64_t i, float v) { p[i] = f2bf(v); }
template <> __device__ __forceinline__ float ld<_Float16>(const _Float16* p, int64_t i) { return (float)p[i]; }
__device__ __forceinline__ void st(_Float16* p, int64_t i, float v) { p[i] = (_Float16)v; }

__device__ __forceinline__ void block_sum2(float& a, float& b) {
    __shared__ float red[2][UP_THREADS / 64];
    a = wave_sum(a);
    b = wave_sum(b);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    a = b = 0.f;
    for (int k = 0; k < UP_THREADS / 64; ++k) { a += red[0][k]; b += red[1][k]; }
}

// One block per (b, c): latents [B*C, n], reference [B*C, m] (NCDHW planes are contiguous).
// out = lerp(x, (x - mean_x) / sd_x * sd_r + mean_r, factor), sd = torch.std (unbiased).
template <typename T>
__global__ void adain_kernel(const T* __restrict__ x, const T* __restrict__ ref, T* __restrict__ out, int64_t n,
                             int64_t m, float factor) {
    const T* xp = x + (int64_t)blockIdx.x * n;
    const T* rp = ref + (int64_t)blockIdx.x * m;
    T* op = out + (int64_t)blockIdx.x * n;
    // two-pass statistics (mean, then centred squares): planes are small and L2-resident
    float sx = 0.f, sr = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += UP_THREADS) sx += ld(xp, i);
    for (int64_t i = threadIdx.x; i < m; i += UP_THREADS) sr += ld(rp, i);
    block_sum2(sx, sr);
    const float mx = sx / (float)n, mr = sr / (float)m;
    float qx = 0.f, qr = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += UP_THREADS) { const float d = ld(xp, i) - mx; qx += d * d; }
    for (int64_t i = threadIdx.x; i < m; i += UP_THREADS) { const float d = ld(rp, i) - mr; qr += d * d; }
    block_sum2(qx, qr);
    const float sdx = sqrtf(qx / (float)(n - 1)), sdr = sqrtf(qr / (float)(m - 1));
    const float scale = sdr / sdx;
    for (int64_t i = threadIdx.x; i < n; i += UP_THREADS) {
        const float v = ld(xp, i);
        const float tform = (v - mx) * scale + mr;
        st(op, i, v + factor * (tform - v));
    }
}

// Tile cross-fade of the tiled VAE paths: b[o, z, i] = a[o, La - extent + z, i] * (1 - z / extent) + b[o, z, i] * (z / extent)
// for z < extent, with both tensors viewed as [outer][L][inner] around the blended axis (vae.py:193-221 blend_z / _v / _h).
template <typename T>
__global__ void tile_blend_kernel(const T* __restrict__ a, T* __restrict__ b, int64_t outer, int64_t La, int64_t Lb,
                                  int64_t inner, int extent) {
    const int64_t total = outer * extent * inner;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = idx % inner, r = idx / inner;
        const int z = (int)(r % extent);
        const int64_t o = r / extent;
        const float w = (float)z / (float)extent;
        const int64_t ia = (o * La + (La - extent + z)) * inner + i, ib = (o * Lb + z) * inner + i;
        const float va = ld(a, ia), vb = ld(b, ib);
        st(b, ib, va * (1.0f - w) + vb * w);
    }
}

}  // namespace ltxmi

using namespace ltxmi;

extern "C" int ltxmi_groupnorm_silu_bf16(const void* x, void* y, const void* residual, int32_t samples, int64_t S,
                                         int32_t C, int32_t groups, const void* gamma, const void* beta, float eps,
                                         float* workspace, void* stream) {
    LTXMI_REQUIRE(x && y && gamma && beta && workspace, LTXMI_ERR_INVALID_ARG, "ltxmi_groupnorm_silu_bf16: NULL argument");
    LTXMI_REQUIRE(samples > 0 && S > 0 && C > 0 && groups > 0, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_groupnorm_silu_bf16: non-positive size");
    const int slots = C / 8;
    LTXMI_REQUIRE(C % 8 == 0 && slots <= UP_THREADS && (slots & (slots - 1)) == 0 && C % groups == 0,
                  LTXMI_ERR_UNSUPPORTED,
                  "ltxmi_groupnorm_silu_bf16: C=%d must be 8 * 2^k <= 2048 and divisible by groups=%d", C, groups);
    hipStream_t s = (hipStream_t)stream;
    float* sums = workspace;                                   // [samples][C][2]
    float* stat = workspace + (int64_t)samples * C * 2;        // [samples][groups][2]
    if (hipMemsetAsync(sums, 0, sizeof(float) * (size_t)samples * C * 2, s) != hipSuccess) {
        set_error("ltxmi_groupnorm_silu_bf16: hipMemsetAsync failed");
        return LTXMI_ERR_LAUNCH;
    }
    // ~2048 blocks over the (rows, samples) grid
    int64_t want = 2048 / samples;
    if (want < 1) want = 1;
    int64_t rpb = (S + want - 1) / want;
    const int rpi = UP_THREADS / slots;
    rpb = (rpb + rpi - 1) / rpi * rpi;
    const unsigned gx = (unsigned)((S + rpb - 1) / rpb);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(gx, samples), dim3(UP_THREADS), 0, s, (const uint16_t*)x, sums, S, C,
                       (int)rpb);
    const int total = samples * groups;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((total + 255) / 256), dim3(256), 0, s, sums, stat, C, groups, S, eps,
                       total);
    const int64_t chunks = (int64_t)samples * S * slots;
    int64_t g = (chunks + UP_THREADS - 1) / UP_THREADS;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)g), dim3(UP_THREADS), 0, s, (const uint16_t*)x, (uint16_t*)y,
                       (const uint16_t*)residual, stat, (const uint16_t*)gamma, (const uint16_t*)beta, S, C, groups,
                       chunks);
    return check_launch("ltxmi_groupnorm_silu_bf16");
}

extern "C" int ltxmi_pixel_shuffle2d_ndhwc_bf16(const void* x, void* y, int64_t frames, int32_t H, int32_t W,
                                                int32_t C, void* stream) {
    LTXMI_REQUIRE(x && y, LTXMI_ERR_INVALID_ARG, "ltxmi_pixel_shuffle2d_ndhwc_bf16: NULL argument");
    LTXMI_REQUIRE(frames > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_pixel_shuffle2d_ndhwc_bf16: bad sizes (C=%d must be a multiple of 8)", C);
    const int64_t chunks = frames * 4 * H * W * (C / 8);
    int64_t g = (chunks + UP_THREADS - 1) / UP_THREADS;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(pixel_shuffle2d_kernel, dim3((unsigned)g), dim3(UP_THREADS), 0, (hipStream_t)stream,
                       (const uint16_t*)x, (uint16_t*)y, H, W, C, chunks);
    return check_launch("ltxmi_pixel_shuffle2d_ndhwc_bf16");
}

extern "C" int ltxmi_tile_blend(const void* a, void* b, int32_t dtype, int64_t outer, int64_t len_a, int64_t len_b,
                                int64_t inner, int32_t extent, void* stream) {
    LTXMI_REQUIRE(dtype >= 0 && dtype <= 2, LTXMI_ERR_INVALID_ARG, "ltxmi_tile_blend: dtype %d not in {0 fp32, 1 bf16, 2 fp16}", dtype);
    LTXMI_REQUIRE(a && b, LTXMI_ERR_INVALID_ARG, "ltxmi_tile_blend: NULL argument");
    LTXMI_REQUIRE(outer > 0 && inner > 0 && extent > 0 && extent <= len_a && extent <= len_b, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_tile_blend: need outer, inner > 0 and 0 < extent <= both lengths (got %d, %lld, %lld)", extent,
                  (long long)len_a, (long long)len_b);
    const int64_t total = outer * extent * inner;
    const int grid = (int)((total + UP_THREADS - 1) / UP_THREADS < 65535 ? (total + UP_THREADS - 1) / UP_THREADS : 65535);
    if (dtype == 1)
        hipLaunchKernelGGL(tile_blend_kernel<uint16_t>, dim3(grid), dim3(UP_THREADS), 0, (hipStream_t)stream,
                           (const uint16_t*)a, (uint16_t*)b, outer, len_a, len_b, inner, extent);
    else if (dtype == 2)
        hipLaunchKernelGGL(tile_blend_kernel<_Float16>, dim3(grid), dim3(UP_THREADS), 0, (hipStream_t)stream,
                           (const _Float16*)a, (_Float16*)b, outer, len_a, len_b, inner, extent);
    else
        hipLaunchKernelGGL(tile_blend_kernel<float>, dim3(grid), dim3(UP_THREADS), 0, (hipStream_t)stream,
                           (const float*)a, (float*)b, outer, len_a, len_b, inner, extent);
    return check_launch("ltxmi_tile_blend");
}

extern "C" int ltxmi_adain_filter(const void* latents, const void* reference, void* out, int32_t is_bf16,
                                  int32_t planes, int64_t n, int64_t n_ref, float factor, void* stream) {
    LTXMI_REQUIRE(latents && reference && out, LTXMI_ERR_INVALID_ARG, "ltxmi_adain_filter: NULL argument");
    LTXMI_REQUIRE(planes > 0 && n > 1 && n_ref > 1, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_adain_filter: need planes > 0 and more than one value per plane");
    if (is_bf16)
        hipLaunchKernelGGL(adain_kernel<uint16_t>, dim3(planes), dim3(UP_THREADS), 0, (hipStream_t)stream,
                           (const uint16_t*)latents, (const uint16_t*)reference, (uint16_t*)out, n, n_ref, factor);
    else
        hipLaunchKernelGGL(adain_kernel<float>, dim3(planes), dim3(UP_THREADS), 0, (hipStream_t)stream,
                           (const float*)latents, (const float*)reference, (float*)out, n, n_ref, factor);
    return check_launch("ltxmi_adain_filter");
}
