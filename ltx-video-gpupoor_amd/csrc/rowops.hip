// rowops.hip -- HBM-bound row kernels of the DiT and VAE paths.
//
// All of them are "one wave per row": a row of D channels is read once as 16-byte chunks
// (lane l takes chunks l, l+64, ...: 1 KiB coalesced per wave-instruction), reduced with
// a 6-step wave butterfly (no LDS, no barrier), transformed in registers and written once.
// Algorithmic traffic = one read + one write of the activation (+ the small tables).
#include "common.h"

namespace ltxmi {

constexpr int ROWS_PER_WG = 4;  // 4 waves, one row each

struct Chunk {
    float v[8];
};
__device__ __forceinline__ Chunk load_chunk(const uint16_t* p) {
    const u32x4 w = *(const u32x4*)p;
    Chunk c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c.v[2 * i] = bf_lo(w[i]);
        c.v[2 * i + 1] = bf_hi(w[i]);
    }
    return c;
}
__device__ __forceinline__ void store_chunk(uint16_t* p, const Chunk& c) {
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf16(c.v[2 * i], c.v[2 * i + 1]);
    *(u32x4*)p = w;
}

// ------------------------------------------------------------------ norm + AdaLN modulate
template <int NCH, bool LAYER>
__global__ __launch_bounds__(256) void norm_modulate_kernel(
    const uint16_t* __restrict__ x, int64_t ldx, uint16_t* __restrict__ y, int64_t ldy, int rows, int D,
    float eps, const uint16_t* __restrict__ sc_tab, const uint16_t* __restrict__ sc_temb,
    const uint16_t* __restrict__ sh_tab, const uint16_t* __restrict__ sh_temb, int64_t temb_ld,
    int rows_per_group) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROWS_PER_WG + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = D >> 3;
    const uint16_t* xr = x + (int64_t)row * ldx;
    Chunk c[NCH];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int ch = lane + 64 * j;
        if (ch < nchunk) {
            c[j] = load_chunk(xr + ch * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1 += c[j].v[e];
                s2 += c[j].v[e] * c[j].v[e];
            }
        }
    }
    s2 = wave_sum(s2);
    float mean = 0.f, rstd;
    if (LAYER) {
        s1 = wave_sum(s1);
        mean = s1 / D;
        rstd = rsqrtf(fmaxf(s2 / D - mean * mean, 0.f) + eps);
    } else {
        rstd = rsqrtf(s2 / D + eps);
    }
    const int64_t g = (int64_t)(row / rows_per_group) * temb_ld;
    uint16_t* yr = y + (int64_t)row * ldy;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int ch = lane + 64 * j;
        if (ch < nchunk) {
            const Chunk a = load_chunk(sc_tab + ch * 8), a2 = load_chunk(sc_temb + g + ch * 8);
            const Chunk b = load_chunk(sh_tab + ch * 8), b2 = load_chunk(sh_temb + g + ch * 8);
            Chunk o;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                o.v[e] = (c[j].v[e] - mean) * rstd * (1.0f + a.v[e] + a2.v[e]) + (b.v[e] + b2.v[e]);
            store_chunk(yr + ch * 8, o);
        }
    }
}

// ------------------------------------------------- q/k RMSNorm(weight) + interleaved RoPE
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_rope_kernel(uint16_t* __restrict__ x, int64_t ldx, int rows, int D,
                                                           const uint16_t* __restrict__ w, float eps,
                                                           const uint16_t* __restrict__ cs,
                                                           const uint16_t* __restrict__ sn, int64_t ld_tab,
                                                           int rope_period, const float* __restrict__ side_ss,
                                                           int64_t side_ld, int side_n, float side_inv_d, float side_eps,
                                                           float* __restrict__ side_rstd) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROWS_PER_WG + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (side_ss) {
        // riding on this launch (k's pass of self-attention): the RMSNorm factor of the SAME row of q from the projection
        // GEMM's per-64-column sums of squares -- one float per row, so that the attention kernel's workgroups (one per
        // head and query tile) read 4 bytes per row instead of re-summing the row's partials each
        float t = 0.f;
        for (int j = lane; j < side_n; j += 64) t += side_ss[(int64_t)row * side_ld + j];
        t = wave_sum(t);
        if (lane == 0) side_rstd[row] = rsqrtf(t * side_inv_d + side_eps);
    }
    const int nchunk = D >> 3;
    uint16_t* xr = x + (int64_t)row * ldx;
    Chunk c[NCH];
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int ch = lane + 64 * j;
        if (ch < nchunk) {
            c[j] = load_chunk(xr + ch * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s2 += c[j].v[e] * c[j].v[e];
        }
    }
    s2 = wave_sum(s2);
    const float rstd = rsqrtf(s2 / D + eps);
    const int64_t trow = cs ? (int64_t)(row % rope_period) * ld_tab : 0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int ch = lane + 64 * j;
        if (ch < nchunk) {
            const Chunk wt = load_chunk(w + ch * 8);
            Chunk o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.v[e] = c[j].v[e] * rstd * wt.v[e];
            if (cs) {
                const Chunk co = load_chunk(cs + trow + ch * 8), si = load_chunk(sn + trow + ch * 8);
                Chunk r2;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    r2.v[e] = o.v[e] * co.v[e] - o.v[e + 1] * si.v[e];
                    r2.v[e + 1] = o.v[e + 1] * co.v[e + 1] + o.v[e] * si.v[e + 1];
                }
                o = r2;
            }
            store_chunk(xr + ch * 8, o);
        }
    }
}

// --------------------------------------- PixelNorm -> (1+scale) x + shift -> SiLU (NDHWC rows)
template <int NCH>
__global__ __launch_bounds__(256) void pixelnorm_ada_silu_kernel(const uint16_t* __restrict__ x,
                                                                 uint16_t* __restrict__ y, int64_t rows, int C,
                                                                 int64_t rows_per_batch,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, int apply_silu,
                                                                 float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_WG + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = C >> 3;
    const uint16_t* xr = x + row * C;
    Chunk c[NCH];
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int ch = lane + 64 * j;
        if (ch < nchunk) {
            c[j] = load_chunk(xr + ch * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s2 += c[j].v[e] * c[j].v[e];
        }
    }
    s2 = wave_sum(s2);
    const float rstd = rsqrtf(s2 / C + eps);
    const int64_t bofs = (row / rows_per_batch) * C;
    uint16_t* yr = y + row * C;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int ch = lane + 64 * j;
        if (ch < nchunk) {
            Chunk o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = c[j].v[e] * rstd;
                if (scale) v = v * (1.0f + scale[bofs + ch * 8 + e]) + shift[bofs + ch * 8 + e];
                o.v[e] = apply_silu ? silu_f(v) : v;
            }
            store_chunk(yr + ch * 8, o);
        }
    }
}

// The same for narrow rows (C = 8 * LPR <= 256, the high-resolution end of the VAE decoder): a wave holds
// 64 / LPR rows, LPR lanes x 16 bytes per row, so every lane carries data (the row-per-wave form above
// leaves 3/4 of the lanes idle at C = 128 and ran at ~1 TB/s); grid-stride over row groups.
template <int LPR>
__global__ __launch_bounds__(256) void pixelnorm_ada_silu_narrow_kernel(const uint16_t* __restrict__ x,
                                                                        uint16_t* __restrict__ y, int64_t rows,
                                                                        int64_t rows_per_batch,
                                                                        const float* __restrict__ scale,
                                                                        const float* __restrict__ shift,
                                                                        int apply_silu, float eps) {
    constexpr int C = LPR * 8, RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPR, rsub = lane / LPR;
    const int64_t stride = (int64_t)gridDim.x * (ROWS_PER_WG * RPW);
    for (int64_t row = ((int64_t)blockIdx.x * ROWS_PER_WG + wave) * RPW + rsub; row < rows; row += stride) {
        Chunk c = load_chunk(x + row * C + sub * 8);
        float s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s2 += c.v[e] * c.v[e];
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
        const float rstd = rsqrtf(s2 * (1.0f / C) + eps);
        float sc[8], sh[8];
        if (scale) {
            const int64_t bofs = (row / rows_per_batch) * C + sub * 8;
            *(f32x4*)sc = *(const f32x4*)(scale + bofs);
            *(f32x4*)(sc + 4) = *(const f32x4*)(scale + bofs + 4);
            *(f32x4*)sh = *(const f32x4*)(shift + bofs);
            *(f32x4*)(sh + 4) = *(const f32x4*)(shift + bofs + 4);
        }
        Chunk o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = c.v[e] * rstd;
            if (scale) v = v * (1.0f + sc[e]) + sh[e];
            o.v[e] = apply_silu ? silu_f(v) : v;
        }
        store_chunk(y + row * C + sub * 8, o);
    }
}

// ----------------------------------------------------- channel LayerNorm with affine (norm3)
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_affine_kernel(const uint16_t* __restrict__ x,
                                                               uint16_t* __restrict__ y, int64_t rows, int C,
                                                               const uint16_t* __restrict__ gamma,
                                                               const uint16_t* __restrict__ beta, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_WG + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = C >> 3;
    const uint16_t* xr = x + row * C;
    Chunk c[NCH];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int ch = lane + 64 * j;
        if (ch < nchunk) {
            c[j] = load_chunk(xr + ch * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1 += c[j].v[e];
                s2 += c[j].v[e] * c[j].v[e];
            }
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const float mean = s1 / C;
    const float rstd = rsqrtf(fmaxf(s2 / C - mean * mean, 0.f) + eps);
    uint16_t* yr = y + row * C;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int ch = lane + 64 * j;
        if (ch < nchunk) {
            const Chunk gm = load_chunk(gamma + ch * 8), bt = load_chunk(beta + ch * 8);
            Chunk o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.v[e] = (c[j].v[e] - mean) * rstd * gm.v[e] + bt.v[e];
            store_chunk(yr + ch * 8, o);
        }
    }
}

// ---------------- Ulysses send buffer: q/k RMSNorm(weight) + RoPE and v, written destination-major
// One pass over a row of the packed projection [q | k | v] (3 D channels): q and k are normalised + rotated exactly as
// rmsnorm_rope_kernel does, v is copied, and every 16-byte chunk goes straight to its place in the all-to-all send
// buffer  [P dst][Nl tokens][B][3][D / P]  (destination rank = head group of the channel).  Token-major inside a
// destination's chunk: after the exchange rank r holds [N = P Nl][B][3][D / P], i.e. q, k, v of ITS heads over ALL
// tokens with uniform batch / token strides -- the attention kernel reads that in place, no unpack copy.
template <int NCH>
__global__ __launch_bounds__(256) void qkv_norm_rope_pack_kernel(const uint16_t* __restrict__ x, int64_t ldx, int B, int Nl,
                                                                 int D, int P, const uint16_t* __restrict__ wq,
                                                                 const uint16_t* __restrict__ wk, float eps,
                                                                 const uint16_t* __restrict__ cs,
                                                                 const uint16_t* __restrict__ sn, int64_t ld_tab,
                                                                 int rope_period, uint16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROWS_PER_WG + (threadIdx.x >> 6);          // = b * Nl + n
    if (row >= B * Nl) return;
    const int b = row / Nl, n = row - b * Nl;
    const int nchunk = D >> 3, Dp = D / P;
    const uint16_t* xr = x + (int64_t)row * ldx;
    Chunk cq[NCH], ck[NCH];
    float sq = 0.f, sk = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int ch = lane + 64 * j;
        if (ch < nchunk) {
            cq[j] = load_chunk(xr + ch * 8);
            ck[j] = load_chunk(xr + D + ch * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sq += cq[j].v[e] * cq[j].v[e];
                sk += ck[j].v[e] * ck[j].v[e];
            }
        }
    }
    sq = wave_sum(sq);
    sk = wave_sum(sk);
    const float rq = rsqrtf(sq / D + eps), rk = rsqrtf(sk / D + eps);
    const int64_t trow = cs ? (int64_t)(row % rope_period) * ld_tab : 0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int ch = lane + 64 * j;
        if (ch < nchunk) {
            const int c0 = ch * 8;
            const int p = c0 / Dp, cl = c0 - p * Dp;                        // destination rank, channel within its D / P
            uint16_t* dst = out + ((((int64_t)p * Nl + n) * B + b) * 3) * Dp + cl;
            const Chunk wqv = load_chunk(wq + c0), wkv = load_chunk(wk + c0);
            Chunk oq, ok;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                oq.v[e] = cq[j].v[e] * rq * wqv.v[e];
                ok.v[e] = ck[j].v[e] * rk * wkv.v[e];
            }
            if (cs) {
                const Chunk co = load_chunk(cs + trow + c0), si = load_chunk(sn + trow + c0);
                Chunk r2, r3;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    r2.v[e] = oq.v[e] * co.v[e] - oq.v[e + 1] * si.v[e];
                    r2.v[e + 1] = oq.v[e + 1] * co.v[e + 1] + oq.v[e] * si.v[e + 1];
                    r3.v[e] = ok.v[e] * co.v[e] - ok.v[e + 1] * si.v[e];
                    r3.v[e + 1] = ok.v[e + 1] * co.v[e + 1] + ok.v[e] * si.v[e + 1];
                }
                oq = r2;
                ok = r3;
            }
            store_chunk(dst, oq);
            store_chunk(dst + Dp, ok);
            *(u32x4*)(dst + 2 * Dp) = *(const u32x4*)(xr + 2 * D + c0);     // v: plain copy
        }
    }
}

static inline int nch_for(int D) { return D <= 512 ? 1 : (D <= 2048 ? 4 : 16); }

}  // namespace ltxmi

using namespace ltxmi;

#define DISPATCH_NCH(D, CALL)                  \
    switch (nch_for(D)) {                      \
        case 1: { constexpr int NCH = 1; CALL; } break;   \
        case 4: { constexpr int NCH = 4; CALL; } break;   \
        default: { constexpr int NCH = 16; CALL; } break; \
    }

extern "C" int ltxmi_norm_modulate_bf16(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t D,
                                        float eps, int32_t kind, const void* scale_table, const void* scale_temb,
                                        const void* shift_table, const void* shift_temb, int64_t temb_ld,
                                        int32_t rows_per_group, void* stream) {
    LTXMI_REQUIRE(x && y && scale_table && scale_temb && shift_table && shift_temb, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_norm_modulate_bf16: NULL argument");
    LTXMI_REQUIRE(rows > 0 && D > 0 && rows_per_group > 0, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_norm_modulate_bf16: non-positive size rows=%d D=%d", rows, D);
    LTXMI_REQUIRE(D % 8 == 0 && D <= 8192 && ldx % 8 == 0 && ldy % 8 == 0 && temb_ld % 8 == 0, LTXMI_ERR_UNSUPPORTED,
                  "ltxmi_norm_modulate_bf16: D=%d must be a multiple of 8 and <= 8192 (strides multiples of 8)", D);
    LTXMI_REQUIRE(kind == LTXMI_NORM_RMS || kind == LTXMI_NORM_LAYER, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_norm_modulate_bf16: bad kind %d", kind);
    const int grid = (rows + ROWS_PER_WG - 1) / ROWS_PER_WG;
    hipStream_t s = (hipStream_t)stream;
#define CALLK(L)                                                                                            \
    hipLaunchKernelGGL((norm_modulate_kernel<NCH, L>), dim3(grid), dim3(256), 0, s, (const uint16_t*)x, ldx, \
                       (uint16_t*)y, ldy, rows, D, eps, (const uint16_t*)scale_table,                       \
                       (const uint16_t*)scale_temb, (const uint16_t*)shift_table, (const uint16_t*)shift_temb, \
                       temb_ld, rows_per_group)
    if (kind == LTXMI_NORM_LAYER) {
        DISPATCH_NCH(D, CALLK(true))
    } else {
        DISPATCH_NCH(D, CALLK(false))
    }
#undef CALLK
    return check_launch("ltxmi_norm_modulate_bf16");
}

static int rmsnorm_rope_launch(void* x, int64_t ldx, int32_t rows, int32_t D, const void* weight, float eps,
                               const void* cos_tab, const void* sin_tab, int64_t ld_tab, int32_t rope_period,
                               const float* side_ss, int64_t side_ld, int32_t side_n, int32_t side_d, float side_eps,
                               float* side_rstd, void* stream);

// rstd[r] = rsqrt(sum_j ss[r * ld + j] / norm_dim + eps): the row factor of an RMSNorm from per-64-column partial sums.  Eight
// lanes per row, 16 bytes each (coalesced: a row of 32 partials is 128 B), reduced with three shuffles.
__global__ __launch_bounds__(256) void rowsumsq_rstd_kernel(const float* __restrict__ ss, int64_t ld, int n, int rows, float inv_d,
                                                            float eps, float* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int r = t >> 3, part = t & 7;
    float acc = 0.f;
    if (r < rows) {
        const float* p = ss + (int64_t)r * ld;
        if ((n & 3) == 0 && ((ld & 3) == 0) && (((uintptr_t)ss) & 15) == 0) {
            for (int j = 4 * part; j < n; j += 32) {
                const f32x4 v = *(const f32x4*)(p + j);
                acc += (v[0] + v[1]) + (v[2] + v[3]);
            }
        } else {
            for (int j = part; j < n; j += 8) acc += p[j];
        }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (r < rows && part == 0) out[r] = rsqrtf(acc * inv_d + eps);
}

extern "C" int ltxmi_rowsumsq_rstd_f32(const float* rowsumsq, int64_t rowsumsq_ld, int32_t rowsumsq_blocks, int32_t rows,
                                       int32_t norm_dim, float norm_eps, float* rstd_out, void* stream) {
    LTXMI_REQUIRE(rowsumsq && rstd_out, LTXMI_ERR_INVALID_ARG, "ltxmi_rowsumsq_rstd_f32: NULL argument");
    LTXMI_REQUIRE(rows > 0 && rowsumsq_blocks > 0 && norm_dim > 0 && rowsumsq_ld >= rowsumsq_blocks &&
                      ((((uintptr_t)rowsumsq) | ((uintptr_t)rstd_out)) & 3) == 0,
                  LTXMI_ERR_INVALID_ARG, "ltxmi_rowsumsq_rstd_f32: bad geometry (rows %d, blocks %d, ld %lld, dim %d)", rows,
                  rowsumsq_blocks, (long long)rowsumsq_ld, norm_dim);
    hipLaunchKernelGGL(rowsumsq_rstd_kernel, dim3((rows * 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, rowsumsq, rowsumsq_ld,
                       rowsumsq_blocks, rows, 1.0f / (float)norm_dim, norm_eps, rstd_out);
    return check_launch("ltxmi_rowsumsq_rstd_f32");
}

extern "C" int ltxmi_rmsnorm_rope_bf16(void* x, int64_t ldx, int32_t rows, int32_t D, const void* weight, float eps,
                                       const void* cos_tab, const void* sin_tab, int64_t ld_tab,
                                       int32_t rope_period, void* stream) {
    return rmsnorm_rope_launch(x, ldx, rows, D, weight, eps, cos_tab, sin_tab, ld_tab, rope_period, nullptr, 0, 0, 1, 0.f,
                               nullptr, stream);
}

extern "C" int ltxmi_rmsnorm_rope_rstd_bf16(void* x, int64_t ldx, int32_t rows, int32_t D, const void* weight, float eps,
                                            const void* cos_tab, const void* sin_tab, int64_t ld_tab, int32_t rope_period,
                                            const float* rowsumsq, int64_t rowsumsq_ld, int32_t rowsumsq_blocks,
                                            int32_t norm_dim, float norm_eps, float* rstd_out, void* stream) {
    LTXMI_REQUIRE(rowsumsq && rstd_out, LTXMI_ERR_INVALID_ARG, "ltxmi_rmsnorm_rope_rstd_bf16: NULL rowsumsq / rstd_out");
    LTXMI_REQUIRE(rowsumsq_blocks > 0 && norm_dim > 0 && rowsumsq_ld >= rowsumsq_blocks &&
                      ((((uintptr_t)rowsumsq) | ((uintptr_t)rstd_out)) & 3) == 0,
                  LTXMI_ERR_INVALID_ARG, "ltxmi_rmsnorm_rope_rstd_bf16: bad rowsumsq geometry (blocks %d, ld %lld, dim %d)",
                  rowsumsq_blocks, (long long)rowsumsq_ld, norm_dim);
    return rmsnorm_rope_launch(x, ldx, rows, D, weight, eps, cos_tab, sin_tab, ld_tab, rope_period, rowsumsq, rowsumsq_ld,
                               rowsumsq_blocks, norm_dim, norm_eps, rstd_out, stream);
}

static int rmsnorm_rope_launch(void* x, int64_t ldx, int32_t rows, int32_t D, const void* weight, float eps,
                               const void* cos_tab, const void* sin_tab, int64_t ld_tab, int32_t rope_period,
                               const float* side_ss, int64_t side_ld, int32_t side_n, int32_t side_d, float side_eps,
                               float* side_rstd, void* stream) {
    LTXMI_REQUIRE(x && weight, LTXMI_ERR_INVALID_ARG, "ltxmi_rmsnorm_rope_bf16: NULL argument");
    LTXMI_REQUIRE((cos_tab == nullptr) == (sin_tab == nullptr), LTXMI_ERR_INVALID_ARG,
                  "ltxmi_rmsnorm_rope_bf16: cos and sin must both be given or both be NULL");
    LTXMI_REQUIRE(rows > 0 && D > 0, LTXMI_ERR_INVALID_ARG, "ltxmi_rmsnorm_rope_bf16: non-positive size");
    LTXMI_REQUIRE(D % 8 == 0 && D <= 8192 && ldx % 8 == 0, LTXMI_ERR_UNSUPPORTED,
                  "ltxmi_rmsnorm_rope_bf16: D=%d must be a multiple of 8 and <= 8192", D);
    if (cos_tab) LTXMI_REQUIRE(rope_period > 0 && ld_tab % 8 == 0 && ld_tab >= D, LTXMI_ERR_INVALID_ARG,
                               "ltxmi_rmsnorm_rope_bf16: bad rope table geometry");
    const int grid = (rows + ROWS_PER_WG - 1) / ROWS_PER_WG;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_NCH(D, hipLaunchKernelGGL((rmsnorm_rope_kernel<NCH>), dim3(grid), dim3(256), 0, s, (uint16_t*)x, ldx,
                                       rows, D, (const uint16_t*)weight, eps, (const uint16_t*)cos_tab,
                                       (const uint16_t*)sin_tab, ld_tab, rope_period > 0 ? rope_period : 1, side_ss, side_ld,
                                       side_n, 1.0f / (float)side_d, side_eps, side_rstd))
    return check_launch("ltxmi_rmsnorm_rope_bf16");
}

extern "C" int ltxmi_qkv_norm_rope_pack_bf16(const void* qkv, int64_t ld, int32_t B, int32_t Nl, int32_t D, int32_t P,
                                             const void* q_weight, const void* k_weight, float eps, const void* cos_tab,
                                             const void* sin_tab, int64_t ld_tab, int32_t rope_period, void* out,
                                             void* stream) {
    LTXMI_REQUIRE(qkv && q_weight && k_weight && out, LTXMI_ERR_INVALID_ARG, "ltxmi_qkv_norm_rope_pack_bf16: NULL argument");
    LTXMI_REQUIRE((cos_tab == nullptr) == (sin_tab == nullptr), LTXMI_ERR_INVALID_ARG,
                  "ltxmi_qkv_norm_rope_pack_bf16: cos and sin must both be given or both be NULL");
    LTXMI_REQUIRE(B > 0 && Nl > 0 && D > 0 && P > 0, LTXMI_ERR_INVALID_ARG, "ltxmi_qkv_norm_rope_pack_bf16: non-positive size");
    LTXMI_REQUIRE(D % (8 * P) == 0 && D <= 8192 && ld % 8 == 0 && ld >= 3 * D, LTXMI_ERR_UNSUPPORTED,
                  "ltxmi_qkv_norm_rope_pack_bf16: D=%d must be a multiple of 8 * P (P=%d) and <= 8192, ld >= 3 D", D, P);
    LTXMI_REQUIRE((int64_t)B * Nl < (1ll << 31) - 4, LTXMI_ERR_UNSUPPORTED, "ltxmi_qkv_norm_rope_pack_bf16: too many rows");
    if (cos_tab) LTXMI_REQUIRE(rope_period > 0 && ld_tab % 8 == 0 && ld_tab >= D, LTXMI_ERR_INVALID_ARG,
                               "ltxmi_qkv_norm_rope_pack_bf16: bad rope table geometry");
    const int rows = B * Nl, grid = (rows + ROWS_PER_WG - 1) / ROWS_PER_WG;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_NCH(D, hipLaunchKernelGGL((qkv_norm_rope_pack_kernel<NCH>), dim3(grid), dim3(256), 0, s, (const uint16_t*)qkv,
                                       ld, B, Nl, D, P, (const uint16_t*)q_weight, (const uint16_t*)k_weight, eps,
                                       (const uint16_t*)cos_tab, (const uint16_t*)sin_tab, ld_tab,
                                       rope_period > 0 ? rope_period : 1, (uint16_t*)out))
    return check_launch("ltxmi_qkv_norm_rope_pack_bf16");
}

extern "C" int ltxmi_pixelnorm_ada_silu_bf16(const void* x, void* y, int64_t rows, int32_t C,
                                             int64_t rows_per_batch, const float* scale, const float* shift,
                                             int32_t apply_silu, float eps, void* stream) {
    LTXMI_REQUIRE(x && y, LTXMI_ERR_INVALID_ARG, "ltxmi_pixelnorm_ada_silu_bf16: NULL argument");
    LTXMI_REQUIRE((scale == nullptr) == (shift == nullptr), LTXMI_ERR_INVALID_ARG,
                  "ltxmi_pixelnorm_ada_silu_bf16: scale and shift must both be given or both be NULL");
    LTXMI_REQUIRE(rows > 0 && C > 0 && rows_per_batch > 0, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_pixelnorm_ada_silu_bf16: non-positive size");
    LTXMI_REQUIRE(C % 8 == 0 && C <= 8192, LTXMI_ERR_UNSUPPORTED,
                  "ltxmi_pixelnorm_ada_silu_bf16: C=%d must be a multiple of 8 and <= 8192", C);
    const int64_t grid = (rows + ROWS_PER_WG - 1) / ROWS_PER_WG;
    LTXMI_REQUIRE(grid < (1ll << 31), LTXMI_ERR_UNSUPPORTED, "ltxmi_pixelnorm_ada_silu_bf16: too many rows");
    hipStream_t s = (hipStream_t)stream;
    if ((C == 64 || C == 128 || C == 256) && ((((uintptr_t)scale | (uintptr_t)shift) & 15) == 0)) {
        const int rpw = 512 / C;                                  // rows per wave
        int64_t g = (rows + ROWS_PER_WG * rpw - 1) / (ROWS_PER_WG * rpw);
        if (g > 256 * 16) g = 256 * 16;                           // grid-stride beyond 16 blocks per CU
#define CALLN(LPR_)                                                                                             \
    hipLaunchKernelGGL((pixelnorm_ada_silu_narrow_kernel<LPR_>), dim3((unsigned)g), dim3(256), 0, s,             \
                       (const uint16_t*)x, (uint16_t*)y, rows, rows_per_batch, scale, shift, apply_silu, eps)
        if (C == 64) CALLN(8);
        else if (C == 128) CALLN(16);
        else CALLN(32);
#undef CALLN
        return check_launch("ltxmi_pixelnorm_ada_silu_bf16");
    }
    DISPATCH_NCH(C, hipLaunchKernelGGL((pixelnorm_ada_silu_kernel<NCH>), dim3((unsigned)grid), dim3(256), 0, s,
                                       (const uint16_t*)x, (uint16_t*)y, rows, C, rows_per_batch, scale, shift,
                                       apply_silu, eps))
    return check_launch("ltxmi_pixelnorm_ada_silu_bf16");
}

extern "C" int ltxmi_layernorm_affine_bf16(const void* x, void* y, int64_t rows, int32_t C, const void* gamma,
                                           const void* beta, float eps, void* stream) {
    LTXMI_REQUIRE(x && y && gamma && beta, LTXMI_ERR_INVALID_ARG, "ltxmi_layernorm_affine_bf16: NULL argument");
    LTXMI_REQUIRE(rows > 0 && C > 0, LTXMI_ERR_INVALID_ARG, "ltxmi_layernorm_affine_bf16: non-positive size");
    LTXMI_REQUIRE(C % 8 == 0 && C <= 8192, LTXMI_ERR_UNSUPPORTED,
                  "ltxmi_layernorm_affine_bf16: C=%d must be a multiple of 8 and <= 8192", C);
    const int64_t grid = (rows + ROWS_PER_WG - 1) / ROWS_PER_WG;
    LTXMI_REQUIRE(grid < (1ll << 31), LTXMI_ERR_UNSUPPORTED, "ltxmi_layernorm_affine_bf16: too many rows");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_NCH(C, hipLaunchKernelGGL((layernorm_affine_kernel<NCH>), dim3((unsigned)grid), dim3(256), 0, s,
                                       (const uint16_t*)x, (uint16_t*)y, rows, C, (const uint16_t*)gamma,
                                       (const uint16_t*)beta, eps))
    return check_launch("ltxmi_layernorm_affine_bf16");
}
