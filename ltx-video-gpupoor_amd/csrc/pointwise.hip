// pointwise.hip -- small elementwise / layout / step-math kernels (all HBM-bound or tiny).
#include <math.h>

#include "common.h"

namespace ltxmi {

constexpr int PW_THREADS = 256;
static inline unsigned pw_grid(int64_t work_items) {
    int64_t g = (work_items + PW_THREADS - 1) / PW_THREADS;
    const int64_t cap = 256 * 8;   // 8 blocks per CU, grid-stride beyond that
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// y = silu(x), 8 elements (16 B) per thread-iteration; n % 8 == 0
__global__ void silu_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int64_t nchunks) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nchunks; i += (int64_t)gridDim.x * blockDim.x) {
        const u32x4 w = *(const u32x4*)(x + i * 8);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack_bf16(silu_f(bf_lo(w[k])), silu_f(bf_hi(w[k])));
        *(u32x4*)(y + i * 8) = o;
    }
}

__global__ void add_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b, uint16_t* __restrict__ y,
                           int64_t nchunks) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nchunks; i += (int64_t)gridDim.x * blockDim.x) {
        const u32x4 u = *(const u32x4*)(a + i * 8);
        const u32x4 v = *(const u32x4*)(b + i * 8);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack_bf16(bf_lo(u[k]) + bf_lo(v[k]), bf_hi(u[k]) + bf_hi(v[k]));
        *(u32x4*)(y + i * 8) = o;
    }
}

// sinusoid: out[i, :half] = cos(t_i * f_k), out[i, half:] = sin(t_i * f_k), f_k = exp(-ln(1e4) k / half)
// (flip_sin_to_cos=True, downscale_freq_shift=0: embeddings.py:29-45)
__global__ void timestep_embedding_kernel(const float* __restrict__ t, uint16_t* __restrict__ out, int n, int dim) {
    const int half = dim >> 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * half) return;
    const int i = idx / half, k = idx % half;
    const float f = expf(-9.210340371976184f * (float)k / (float)half);
    const float a = t[i] * f;
    out[(int64_t)i * dim + k] = f2bf(cosf(a));
    out[(int64_t)i * dim + half + k] = f2bf(sinf(a));
}

// a = a*m[b] + v*(1-m[b])
__global__ void stg_blend_kernel(uint16_t* __restrict__ a, int64_t lda, const uint16_t* __restrict__ v, int64_t ldv,
                                 const float* __restrict__ m, int L, int D, int64_t total_chunks) {
    const int cpr = D >> 3;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total_chunks;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cpr;
        const int ch = (int)(i % cpr);
        const float mm = m[row / L];
        if (mm == 1.0f) continue;
        uint16_t* ap = a + row * lda + ch * 8;
        const u32x4 u = *(const u32x4*)ap;
        const u32x4 w = *(const u32x4*)(v + row * ldv + ch * 8);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[k] = pack_bf16(bf_lo(u[k]) * mm + bf_lo(w[k]) * (1.f - mm), bf_hi(u[k]) * mm + bf_hi(w[k]) * (1.f - mm));
        *(u32x4*)ap = o;
    }
}

// The same blend over a contiguous a [G][B][L][D] against a v addressed by (group, batch, token) strides: the K-blocked
// receive buffer of the Ulysses return exchange ([P source ranks][B Nl][D / P]) blended in ONE launch with the values /
// the block input, whose channel block p of token (b, n) lies at p * v_sg + b * v_sb + n * v_sl.
__global__ void stg_blend_grouped_kernel(uint16_t* __restrict__ a, const uint16_t* __restrict__ v, int64_t v_sg, int64_t v_sb,
                                         int64_t v_sl, const float* __restrict__ m, int B, int L, int D, int64_t total_chunks) {
    const int cpr = D >> 3;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total_chunks;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cpr;                       // (g, b, l)
        const int ch = (int)(i % cpr);
        const int l = (int)(row % L);
        const int64_t gb = row / L;
        const int b = (int)(gb % B);
        const int64_t g = gb / B;
        const float mm = m[b];
        if (mm == 1.0f) continue;
        uint16_t* ap = a + row * D + ch * 8;
        const u32x4 u = *(const u32x4*)ap;
        const u32x4 w = *(const u32x4*)(v + g * v_sg + b * v_sb + l * v_sl + ch * 8);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[k] = pack_bf16(bf_lo(u[k]) * mm + bf_lo(w[k]) * (1.f - mm), bf_hi(u[k]) * mm + bf_hi(w[k]) * (1.f - mm));
        *(u32x4*)ap = o;
    }
}

// z [B,C,T,H,W] bf16 -> y [B,T,H,W,C] bf16, optionally * std[c] + mean[c]
__global__ void ncdhw_to_ndhwc_kernel(const uint16_t* __restrict__ z, uint16_t* __restrict__ y, int C, int64_t thw,
                                      int64_t total, const float* __restrict__ std, const float* __restrict__ mean) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t pos = (i / C) % thw;
        const int64_t b = i / ((int64_t)C * thw);
        float v = bf2f(z[(b * C + c) * thw + pos]);
        if (std) v = v * std[c] + mean[c];
        y[i] = f2bf(v);
    }
}

// x [B,T,H,W, C_out*p*p] (channel n = (c*p + r)*p + q) -> y [B,C_out,T,H*p,W*p], pixel (h*p+q, w*p+r)
__global__ void unpatchify_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int T, int H, int W,
                                  int Co, int P, int64_t total) {
    const int Wp = W * P, Hp = H * P;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xo = (int)(i % Wp);
        const int yo = (int)((i / Wp) % Hp);
        const int t = (int)((i / ((int64_t)Wp * Hp)) % T);
        const int c = (int)((i / ((int64_t)Wp * Hp * T)) % Co);
        const int64_t b = i / ((int64_t)Wp * Hp * T * Co);
        const int w = xo / P, r = xo % P, h = yo / P, q = yo % P;
        const int n = (c * P + r) * P + q;
        y[i] = x[(((b * T + t) * H + h) * W + w) * ((int64_t)Co * P * P) + n];
    }
}

// x [B,C,T,H,W] pixels -> y [B,T,H/P,W/P,Cpad], channel n = (c*P + r)*P + q holds pixel (h*P+q, w*P+r);
// channels >= C*P*P are zero (pads K of the first convolution to a multiple of 64)
__global__ void patchify_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int C, int T, int H, int W,
                                int P, int Cpad, int64_t total) {
    const int Hp = H / P, Wp = W / P;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % Cpad);
        int64_t r0 = i / Cpad;
        const int w = (int)(r0 % Wp); r0 /= Wp;
        const int h = (int)(r0 % Hp); r0 /= Hp;
        const int t = (int)(r0 % T);
        const int64_t b = r0 / T;
        uint16_t v = 0;
        if (n < C * P * P) {
            const int c = n / (P * P), r = (n / P) % P, q = n % P;
            v = x[(((b * C + c) * T + t) * H + h * P + q) * (int64_t)W + w * P + r];
        }
        y[i] = v;
    }
}

// SpaceToDepthDownsample tail: out[b,d,h,w, c*P + p] = conv[b, d*st+p1, h*s+p2, w*s+p3, c]
//   + mean_j xdup[.. channel k = co*g + j -> (ci = k / P, p' = k % P)], xdup = x with its first frame
// duplicated when st == 2 (conv then has T+1 frames).
__global__ void space_to_depth_skip_kernel(const uint16_t* __restrict__ conv, const uint16_t* __restrict__ x,
                                           uint16_t* __restrict__ out, int T, int H, int W, int Cin, int Cc,
                                           int st, int s, int group, int64_t total) {
    const int P = st * s * s, Co = Cc * P;
    const int Tc = st == 2 ? T + 1 : T;
    const int Do = Tc / st, Ho = H / s, Wo = W / s;
    const float inv_g = 1.0f / (float)group;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Co);
        int64_t r0 = i / Co;
        const int w = (int)(r0 % Wo); r0 /= Wo;
        const int h = (int)(r0 % Ho); r0 /= Ho;
        const int d = (int)(r0 % Do);
        const int64_t b = r0 / Do;
        const int c = co / P, p = co % P;
        const int p1 = p / (s * s), p2 = (p / s) % s, p3 = p % s;
        float v = bf2f(conv[(((b * Tc + d * st + p1) * H + h * s + p2) * (int64_t)W + w * s + p3) * Cc + c]);
        float acc = 0.f;
        for (int j = 0; j < group; ++j) {
            const int k = co * group + j;
            const int ci = k / P, pp = k % P;
            int tt = d * st + pp / (s * s);
            if (st == 2) tt = tt > 0 ? tt - 1 : 0;
            acc += bf2f(x[(((b * T + tt) * H + h * s + (pp / s) % s) * (int64_t)W + w * s + pp % s) * Cin + ci]);
        }
        out[i] = f2bf(v + acc * inv_g);
    }
}

// x rows [B*thw, ldx], channels c0..c0+C  ->  y [B,C,T,H,W], optionally (v - mean[c]) / std[c]
__global__ void ndhwc_to_ncdhw_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int64_t ldx, int c0,
                                      int C, int64_t thw, int64_t total, const float* __restrict__ std,
                                      const float* __restrict__ mean) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pos = i % thw;
        const int c = (int)((i / thw) % C);
        const int64_t b = i / (thw * C);
        float v = bf2f(x[(b * thw + pos) * ldx + c0 + c]);
        if (std) v = (v - mean[c]) / std[c];
        y[i] = f2bf(v);
    }
}

// patch 4, 3 output channels (the LTX-Video decoder): one thread per input position reads its 48 channels
// (6 x 16 B) and writes, for every (c, q), the 4 horizontally adjacent pixels r = 0..3 as one 8-byte
// store; neighbouring threads (w, w+1) write neighbouring 8 bytes.  n = (c*4 + r)*4 + q.
__global__ void unpatchify4_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int T, int H, int W,
                                   int64_t total_pos) {
    const int Wp = W * 4, Hp = H * 4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total_pos; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const int h = (int)((i / W) % H);
        const int t = (int)((i / ((int64_t)W * H)) % T);
        const int64_t b = i / ((int64_t)W * H * T);
        uint16_t v[48];
#pragma unroll
        for (int k = 0; k < 6; ++k) *(u32x4*)(v + 8 * k) = *(const u32x4*)(x + i * 48 + 8 * k);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u32x2 o;
                o[0] = (uint32_t)v[(c * 4 + 0) * 4 + q] | ((uint32_t)v[(c * 4 + 1) * 4 + q] << 16);
                o[1] = (uint32_t)v[(c * 4 + 2) * 4 + q] | ((uint32_t)v[(c * 4 + 3) * 4 + q] << 16);
                *(u32x2*)(y + (((b * 3 + c) * T + t) * Hp + h * 4 + q) * (int64_t)Wp + w * 4) = o;
            }
    }
}

// ---------------------------------------------------------------- guidance + Euler step
// pass 1 sums: text*uncond, uncond^2, text, text^2; pass 2 sums: out, out^2 (per-block partials in the workspace)
struct GuidanceP {
    const uint16_t* np; int64_t n; int num_conds;
    float gs, stg, rs; int do_cfg, do_stg, do_rescale;
    float* lat_f32; uint16_t* lat_bf16; float dt; float* ws;
    const float* cond_mask; int channels; float t;      // conditioning: token n steps iff t - 1e-6 < 1 - cond_mask[n]
};

// Deterministic two-level sums (no atomics: run-to-run identical latents): every block writes its partial
// to part[block][NV]; the consumer kernel adds the partials of all blocks in block order.
// Workspace layout (floats): [8 .. 8 + 4*256) partials of pass 1, [1032 .. 1032 + 2*256) partials of pass 2.
constexpr int GD_MAX_BLOCKS = 256;
constexpr int GD_PART1 = 8, GD_PART2 = 8 + 4 * GD_MAX_BLOCKS;
template <int NV>
__device__ __forceinline__ void block_partial_store(float (&v)[NV], float* part) {
    __shared__ float red[NV][PW_THREADS / 64];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float s = wave_sum(v[i]);
        if ((threadIdx.x & 63) == 0) red[i][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < PW_THREADS / 64; ++w) s += red[threadIdx.x][w];
        part[blockIdx.x * NV + threadIdx.x] = s;
    }
}
template <int NV>
__device__ __forceinline__ void sum_partials(const float* part, int nblocks, float (&out)[NV]) {
    __shared__ float tot[NV];
    if (threadIdx.x < NV) {
        float s = 0.f;
        for (int b = 0; b < nblocks; ++b) s += part[b * NV + threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) out[i] = tot[i];
}

__device__ __forceinline__ void guidance_fetch(const GuidanceP& p, int64_t i, float& unc, float& txt, float& ptb) {
    // chunk order is (uncond, text, perturbed): pipeline_ltx_video.py:1035-1051
    int c = 0;
    unc = txt = ptb = 0.f;
    if (p.do_cfg) unc = bf2f(p.np[(int64_t)(c++) * p.n + i]);
    txt = bf2f(p.np[(int64_t)(c++) * p.n + i]);
    if (p.do_stg) ptb = bf2f(p.np[(int64_t)(c++) * p.n + i]);
}

__device__ __forceinline__ float guidance_combine(const GuidanceP& p, float unc, float txt, float ptb, float alpha) {
    float out = txt;
    if (p.do_cfg && p.gs != 0.f && p.gs != 1.f) {
        const float u = alpha * unc;
        out = u + p.gs * (txt - u);
    }
    if (p.do_stg) out = out + p.stg * (txt - ptb);
    return out;
}

__global__ void guidance_reduce1(GuidanceP p) {
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * blockDim.x) {
        float u, t, q;
        guidance_fetch(p, i, u, t, q);
        a0 += t * u; a1 += u * u; a2 += t; a3 += t * t;
    }
    float acc[4] = {a0, a1, a2, a3};
    block_partial_store(acc, p.ws + GD_PART1);
}

__global__ void guidance_reduce2(GuidanceP p) {
    float s1[4];
    sum_partials(p.ws + GD_PART1, gridDim.x, s1);
    const float alpha = s1[0] / (s1[1] + 1e-8f);
    float a4 = 0, a5 = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * blockDim.x) {
        float u, t, q;
        guidance_fetch(p, i, u, t, q);
        const float o = guidance_combine(p, u, t, q, alpha);
        a4 += o; a5 += o * o;
    }
    float acc[2] = {a4, a5};
    block_partial_store(acc, p.ws + GD_PART2);
}

__global__ void guidance_apply(GuidanceP p) {
    float s1[4], s2[2];
    sum_partials(p.ws + GD_PART1, gridDim.x, s1);
    sum_partials(p.ws + GD_PART2, gridDim.x, s2);
    const float alpha = s1[0] / (s1[1] + 1e-8f);
    float factor = 1.f;
    if (p.do_stg && p.do_rescale && p.stg > 0.f) {
        const float n = (float)p.n;
        const float var_t = fmaxf((s1[3] - s1[2] * s1[2] / n) / (n - 1.f), 0.f);   // torch .std(): unbiased
        const float var_o = fmaxf((s2[1] - s2[0] * s2[0] / n) / (n - 1.f), 0.f);
        factor = p.rs * (sqrtf(var_t) / sqrtf(var_o)) + (1.f - p.rs);
    }
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * blockDim.x) {
        float u, t, q;
        guidance_fetch(p, i, u, t, q);
        // the reference holds noise_pred in bf16 after the guidance arithmetic; the Euler
        // update itself runs in the latents' dtype (rf.py:375)
        const float o = guidance_combine(p, u, t, q, alpha) * factor;
        if (p.cond_mask && !(p.t - 1e-6f < 1.0f - p.cond_mask[i / p.channels])) continue;   // :1341-1342
        if (p.lat_f32) p.lat_f32[i] = p.lat_f32[i] - p.dt * o;
        else p.lat_bf16[i] = f2bf(bf2f(p.lat_bf16[i]) - p.dt * o);
    }
}

// add_noise_to_image_conditioning_latents (pipeline_ltx_video.py:606-629): hard-conditioned tokens
// (mask > 1 - eps) are reset to init + noise_scale * noise * t^2, the others keep their value
template <typename T>
__global__ void cond_noise_kernel(T* __restrict__ lat, const T* __restrict__ init, const T* __restrict__ noise,
                                  const float* __restrict__ mask, int channels, float scale_t2, int64_t total) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (!(mask[i / channels] > 1.0f - 1e-6f)) continue;
        if constexpr (sizeof(T) == 4) lat[i] = init[i] + scale_t2 * noise[i];
        else lat[i] = f2bf(bf2f(init[i]) + scale_t2 * bf2f(noise[i]));
    }
}

}  // namespace ltxmi

using namespace ltxmi;

extern "C" int ltxmi_silu_bf16(const void* x, void* y, int64_t n, void* stream) {
    LTXMI_REQUIRE(x && y, LTXMI_ERR_INVALID_ARG, "ltxmi_silu_bf16: NULL argument");
    LTXMI_REQUIRE(n > 0 && n % 8 == 0, LTXMI_ERR_UNSUPPORTED, "ltxmi_silu_bf16: n=%lld must be a positive multiple of 8", (long long)n);
    hipLaunchKernelGGL(silu_kernel, dim3(pw_grid(n / 8)), dim3(PW_THREADS), 0, (hipStream_t)stream,
                       (const uint16_t*)x, (uint16_t*)y, n / 8);
    return check_launch("ltxmi_silu_bf16");
}

extern "C" int ltxmi_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream) {
    LTXMI_REQUIRE(a && b && y, LTXMI_ERR_INVALID_ARG, "ltxmi_add_bf16: NULL argument");
    LTXMI_REQUIRE(n > 0 && n % 8 == 0, LTXMI_ERR_UNSUPPORTED, "ltxmi_add_bf16: n=%lld must be a positive multiple of 8", (long long)n);
    hipLaunchKernelGGL(add_kernel, dim3(pw_grid(n / 8)), dim3(PW_THREADS), 0, (hipStream_t)stream,
                       (const uint16_t*)a, (const uint16_t*)b, (uint16_t*)y, n / 8);
    return check_launch("ltxmi_add_bf16");
}

extern "C" int ltxmi_timestep_embedding_bf16(const float* t, void* out, int32_t n, int32_t dim, void* stream) {
    LTXMI_REQUIRE(t && out, LTXMI_ERR_INVALID_ARG, "ltxmi_timestep_embedding_bf16: NULL argument");
    LTXMI_REQUIRE(n > 0 && dim > 0 && dim % 2 == 0, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_timestep_embedding_bf16: n=%d dim=%d (dim must be even)", n, dim);
    const int total = n * (dim / 2);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, t,
                       (uint16_t*)out, n, dim);
    return check_launch("ltxmi_timestep_embedding_bf16");
}

extern "C" int ltxmi_stg_blend_bf16(void* a, int64_t lda, const void* v, int64_t ldv, const float* m, int32_t B,
                                    int32_t L, int32_t D, void* stream) {
    LTXMI_REQUIRE(a && v && m, LTXMI_ERR_INVALID_ARG, "ltxmi_stg_blend_bf16: NULL argument");
    LTXMI_REQUIRE(B > 0 && L > 0 && D > 0, LTXMI_ERR_INVALID_ARG, "ltxmi_stg_blend_bf16: non-positive size");
    LTXMI_REQUIRE(D % 8 == 0 && lda % 8 == 0 && ldv % 8 == 0, LTXMI_ERR_UNSUPPORTED,
                  "ltxmi_stg_blend_bf16: D and strides must be multiples of 8");
    const int64_t total = (int64_t)B * L * (D / 8);
    hipLaunchKernelGGL(stg_blend_kernel, dim3(pw_grid(total)), dim3(PW_THREADS), 0, (hipStream_t)stream, (uint16_t*)a,
                       lda, (const uint16_t*)v, ldv, m, L, D, total);
    return check_launch("ltxmi_stg_blend_bf16");
}

extern "C" int ltxmi_stg_blend_grouped_bf16(void* a, const void* v, int64_t v_stride_g, int64_t v_stride_b, int64_t v_stride_l,
                                            const float* m, int32_t G, int32_t B, int32_t L, int32_t D, void* stream) {
    LTXMI_REQUIRE(a && v && m, LTXMI_ERR_INVALID_ARG, "ltxmi_stg_blend_grouped_bf16: NULL argument");
    LTXMI_REQUIRE(G > 0 && B > 0 && L > 0 && D > 0, LTXMI_ERR_INVALID_ARG, "ltxmi_stg_blend_grouped_bf16: non-positive size");
    LTXMI_REQUIRE(D % 8 == 0 && v_stride_g % 8 == 0 && v_stride_b % 8 == 0 && v_stride_l % 8 == 0 &&
                      (((uintptr_t)a | (uintptr_t)v) & 15) == 0,
                  LTXMI_ERR_UNSUPPORTED, "ltxmi_stg_blend_grouped_bf16: D and strides must be multiples of 8, pointers 16-byte aligned");
    const int64_t total = (int64_t)G * B * L * (D / 8);
    hipLaunchKernelGGL(stg_blend_grouped_kernel, dim3(pw_grid(total)), dim3(PW_THREADS), 0, (hipStream_t)stream, (uint16_t*)a,
                       (const uint16_t*)v, v_stride_g, v_stride_b, v_stride_l, m, B, L, D, total);
    return check_launch("ltxmi_stg_blend_grouped_bf16");
}

extern "C" int ltxmi_ncdhw_to_ndhwc_bf16(const void* z, void* y, int32_t B, int32_t C, int32_t T, int32_t H,
                                         int32_t W, const float* std, const float* mean, void* stream) {
    LTXMI_REQUIRE(z && y, LTXMI_ERR_INVALID_ARG, "ltxmi_ncdhw_to_ndhwc_bf16: NULL argument");
    LTXMI_REQUIRE((std == nullptr) == (mean == nullptr), LTXMI_ERR_INVALID_ARG,
                  "ltxmi_ncdhw_to_ndhwc_bf16: std and mean must both be given or both be NULL");
    LTXMI_REQUIRE(B > 0 && C > 0 && T > 0 && H > 0 && W > 0, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_ncdhw_to_ndhwc_bf16: non-positive size");
    const int64_t thw = (int64_t)T * H * W, total = (int64_t)B * C * thw;
    hipLaunchKernelGGL(ncdhw_to_ndhwc_kernel, dim3(pw_grid(total)), dim3(PW_THREADS), 0, (hipStream_t)stream,
                       (const uint16_t*)z, (uint16_t*)y, C, thw, total, std, mean);
    return check_launch("ltxmi_ncdhw_to_ndhwc_bf16");
}

extern "C" int ltxmi_unpatchify_to_ncdhw_bf16(const void* x, void* y, int32_t B, int32_t T, int32_t H, int32_t W,
                                              int32_t C_out, int32_t patch, void* stream) {
    LTXMI_REQUIRE(x && y, LTXMI_ERR_INVALID_ARG, "ltxmi_unpatchify_to_ncdhw_bf16: NULL argument");
    LTXMI_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0 && C_out > 0 && patch > 0, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_unpatchify_to_ncdhw_bf16: non-positive size");
    if (patch == 4 && C_out == 3 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
        const int64_t pos = (int64_t)B * T * H * W;
        hipLaunchKernelGGL(unpatchify4_kernel, dim3(pw_grid(pos)), dim3(PW_THREADS), 0, (hipStream_t)stream,
                           (const uint16_t*)x, (uint16_t*)y, T, H, W, pos);
        return check_launch("ltxmi_unpatchify_to_ncdhw_bf16");
    }
    const int64_t total = (int64_t)B * C_out * T * H * patch * W * patch;
    hipLaunchKernelGGL(unpatchify_kernel, dim3(pw_grid(total)), dim3(PW_THREADS), 0, (hipStream_t)stream,
                       (const uint16_t*)x, (uint16_t*)y, T, H, W, C_out, patch, total);
    return check_launch("ltxmi_unpatchify_to_ncdhw_bf16");
}

extern "C" int ltxmi_guidance_step_masked_bf16(const void* noise_pred, int64_t n, int32_t num_conds,
                                               float guidance_scale, float stg_scale, float rescaling_scale,
                                               int32_t do_cfg, int32_t do_stg, int32_t do_rescale, void* latents,
                                               int32_t latents_bf16, float dt, const float* cond_mask,
                                               int32_t channels, float t, float* workspace, void* stream) {
    LTXMI_REQUIRE(noise_pred && latents && workspace, LTXMI_ERR_INVALID_ARG, "ltxmi_guidance_step_bf16: NULL argument");
    LTXMI_REQUIRE(n > 1, LTXMI_ERR_INVALID_ARG, "ltxmi_guidance_step_bf16: n must be > 1");
    LTXMI_REQUIRE(num_conds == 1 + (do_cfg ? 1 : 0) + (do_stg ? 1 : 0), LTXMI_ERR_INVALID_ARG,
                  "ltxmi_guidance_step_bf16: num_conds=%d inconsistent with do_cfg=%d do_stg=%d", num_conds, do_cfg, do_stg);
    hipStream_t s = (hipStream_t)stream;
    GuidanceP p;
    p.np = (const uint16_t*)noise_pred; p.n = n; p.num_conds = num_conds;
    p.gs = guidance_scale; p.stg = stg_scale; p.rs = rescaling_scale;
    p.do_cfg = do_cfg; p.do_stg = do_stg; p.do_rescale = do_rescale;
    p.lat_f32 = latents_bf16 ? nullptr : (float*)latents;
    p.lat_bf16 = latents_bf16 ? (uint16_t*)latents : nullptr;
    p.dt = dt; p.ws = workspace;
    p.cond_mask = cond_mask; p.channels = channels > 0 ? channels : 1; p.t = t;
    LTXMI_REQUIRE(!cond_mask || (channels > 0 && n % channels == 0), LTXMI_ERR_INVALID_ARG,
                  "ltxmi_guidance_step_bf16: n=%lld is not a multiple of channels=%d", (long long)n, channels);
    unsigned g = pw_grid(n);
    if (g > 256) g = 256;
    hipLaunchKernelGGL(guidance_reduce1, dim3(g), dim3(PW_THREADS), 0, s, p);
    hipLaunchKernelGGL(guidance_reduce2, dim3(g), dim3(PW_THREADS), 0, s, p);
    hipLaunchKernelGGL(guidance_apply, dim3(g), dim3(PW_THREADS), 0, s, p);
    return check_launch("ltxmi_guidance_step_bf16");
}

extern "C" int ltxmi_guidance_step_bf16(const void* noise_pred, int64_t n, int32_t num_conds, float guidance_scale,
                                        float stg_scale, float rescaling_scale, int32_t do_cfg, int32_t do_stg,
                                        int32_t do_rescale, void* latents, int32_t latents_bf16, float dt,
                                        float* workspace, void* stream) {
    return ltxmi_guidance_step_masked_bf16(noise_pred, n, num_conds, guidance_scale, stg_scale, rescaling_scale, do_cfg,
                                           do_stg, do_rescale, latents, latents_bf16, dt, nullptr, 0, 0.f, workspace,
                                           stream);
}

extern "C" int ltxmi_image_cond_noise(void* latents, const void* init_latents, const void* noise, int32_t is_bf16,
                                      const float* cond_mask, int64_t tokens, int32_t channels, float noise_scale,
                                      float t, void* stream) {
    LTXMI_REQUIRE(latents && init_latents && noise && cond_mask, LTXMI_ERR_INVALID_ARG, "ltxmi_image_cond_noise: NULL argument");
    LTXMI_REQUIRE(tokens > 0 && channels > 0, LTXMI_ERR_INVALID_ARG, "ltxmi_image_cond_noise: non-positive size");
    const int64_t total = tokens * channels;
    const float s = noise_scale * t * t;
    if (is_bf16)
        hipLaunchKernelGGL(cond_noise_kernel<uint16_t>, dim3(pw_grid(total)), dim3(PW_THREADS), 0, (hipStream_t)stream,
                           (uint16_t*)latents, (const uint16_t*)init_latents, (const uint16_t*)noise, cond_mask, channels, s, total);
    else
        hipLaunchKernelGGL(cond_noise_kernel<float>, dim3(pw_grid(total)), dim3(PW_THREADS), 0, (hipStream_t)stream,
                           (float*)latents, (const float*)init_latents, (const float*)noise, cond_mask, channels, s, total);
    return check_launch("ltxmi_image_cond_noise");
}

extern "C" int ltxmi_patchify_to_ndhwc_bf16(const void* x, void* y, int32_t B, int32_t C, int32_t T, int32_t H,
                                            int32_t W, int32_t patch, int32_t C_pad, void* stream) {
    LTXMI_REQUIRE(x && y, LTXMI_ERR_INVALID_ARG, "ltxmi_patchify_to_ndhwc_bf16: NULL argument");
    LTXMI_REQUIRE(B > 0 && C > 0 && T > 0 && H > 0 && W > 0 && patch > 0, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_patchify_to_ndhwc_bf16: non-positive size");
    LTXMI_REQUIRE(H % patch == 0 && W % patch == 0 && C_pad >= C * patch * patch, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_patchify_to_ndhwc_bf16: H=%d W=%d must be multiples of patch=%d and C_pad=%d >= C*patch^2",
                  H, W, patch, C_pad);
    const int64_t total = (int64_t)B * T * (H / patch) * (W / patch) * C_pad;
    hipLaunchKernelGGL(patchify_kernel, dim3(pw_grid(total)), dim3(PW_THREADS), 0, (hipStream_t)stream,
                       (const uint16_t*)x, (uint16_t*)y, C, T, H, W, patch, C_pad, total);
    return check_launch("ltxmi_patchify_to_ndhwc_bf16");
}

extern "C" int ltxmi_space_to_depth_skip_bf16(const void* conv, const void* x, void* out, int32_t B, int32_t T,
                                              int32_t H, int32_t W, int32_t Cin, int32_t Cconv, int32_t stride_t,
                                              int32_t stride_hw, int32_t group, void* stream) {
    LTXMI_REQUIRE(conv && x && out, LTXMI_ERR_INVALID_ARG, "ltxmi_space_to_depth_skip_bf16: NULL argument");
    LTXMI_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0 && Cin > 0 && Cconv > 0 && group > 0, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_space_to_depth_skip_bf16: non-positive size");
    LTXMI_REQUIRE((stride_t == 1 || stride_t == 2) && (stride_hw == 1 || stride_hw == 2), LTXMI_ERR_UNSUPPORTED,
                  "ltxmi_space_to_depth_skip_bf16: strides must be 1 or 2");
    const int P = stride_t * stride_hw * stride_hw;
    const int Tc = stride_t == 2 ? T + 1 : T;
    LTXMI_REQUIRE(Tc % stride_t == 0 && H % stride_hw == 0 && W % stride_hw == 0, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_space_to_depth_skip_bf16: frames(+1)=%d H=%d W=%d not divisible by the strides", Tc, H, W);
    LTXMI_REQUIRE((int64_t)Cconv * P * group == (int64_t)Cin * P, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_space_to_depth_skip_bf16: Cconv*group=%d must equal Cin=%d", Cconv * group, Cin);
    const int64_t total = (int64_t)B * (Tc / stride_t) * (H / stride_hw) * (W / stride_hw) * Cconv * P;
    hipLaunchKernelGGL(space_to_depth_skip_kernel, dim3(pw_grid(total)), dim3(PW_THREADS), 0, (hipStream_t)stream,
                       (const uint16_t*)conv, (const uint16_t*)x, (uint16_t*)out, T, H, W, Cin, Cconv, stride_t,
                       stride_hw, group, total);
    return check_launch("ltxmi_space_to_depth_skip_bf16");
}

extern "C" int ltxmi_ndhwc_to_ncdhw_bf16(const void* x, int64_t ldx, int32_t c0, void* y, int32_t B, int32_t C,
                                         int32_t T, int32_t H, int32_t W, const float* std, const float* mean,
                                         void* stream) {
    LTXMI_REQUIRE(x && y, LTXMI_ERR_INVALID_ARG, "ltxmi_ndhwc_to_ncdhw_bf16: NULL argument");
    LTXMI_REQUIRE(B > 0 && C > 0 && T > 0 && H > 0 && W > 0 && c0 >= 0 && ldx >= c0 + C, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_ndhwc_to_ncdhw_bf16: bad sizes");
    LTXMI_REQUIRE((std == nullptr) == (mean == nullptr), LTXMI_ERR_INVALID_ARG,
                  "ltxmi_ndhwc_to_ncdhw_bf16: std and mean come together");
    const int64_t thw = (int64_t)T * H * W, total = (int64_t)B * C * thw;
    hipLaunchKernelGGL(ndhwc_to_ncdhw_kernel, dim3(pw_grid(total)), dim3(PW_THREADS), 0, (hipStream_t)stream,
                       (const uint16_t*)x, (uint16_t*)y, ldx, c0, C, thw, total, std, mean);
    return check_launch("ltxmi_ndhwc_to_ncdhw_bf16");
}
