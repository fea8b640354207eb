// attention.hip -- flash-style attention forward for the pay_attention() seam.
//
// One workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 query
// rows and walks the keys in tiles of 64.  Per tile and wave:
//
//   S^T[key][query]  = K . Q^T          v_mfma_f32_32x32x16_bf16, A = K tile (LDS, ds_read_b128),
//                                        B = Q^T (registers, loaded once)
//   online softmax                        entirely in registers: the SWAPPED product leaves the
//                                        query on the lane (col = lane&31) and 16 of the tile's
//                                        keys in that lane's 16 accumulator registers, so the
//                                        row max / row sum are in-lane reductions plus ONE
//                                        v_permlane32_swap with the partner lane (lane^32)
//   O^T[d][query]   += V^T . P^T         A = V^T via ds_read_b64_tr_b16 (hardware transposed LDS
//                                        read), B = P^T = the S^T accumulator registers converted
//                                        pairwise to bf16 (no LDS round trip, no lane movement)
//
// Both products keep the query on the lane, so m, l and the O rescale factor are lane-local.
//
// LDS: K image = rows of DH bf16, 16-byte chunk c of row r at slot c ^ swz(r) (swz chosen so
// the 16-lane ds_read_b128 groups of the 32x32x16 A operand hit 16 distinct slots);
// V image = [8 key][32 col] sub-tiles of 512 B, which makes each 32-lane half of a transposed
// read cover 256 contiguous bytes (conflict-free).  K/V tiles are register-staged
// (global_load_dwordx4 issued before the tile's MFMAs, ds_write_b128 after them: the "async
// STAGE split"), two LDS buffers, one barrier per tile.
//
// Work mapping is XCD-aware: each XCD walks whole (batch, head) pairs, so the 32 CUs that share
// an L2 stream the same K/V at the same time.
#include <type_traits>

#include "attention.h"

namespace ltxmi {

constexpr int KV_TILE = 64;
constexpr int Q_PER_WAVE = 32;
constexpr int Q_PER_WG = 128;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int ATTN_QB_BIG = 2;        // 32-row query blocks per wave when the grid still fills the chip

template <int DH>
struct AttnCfg {
    static constexpr int ROW_BYTES = DH * 2;
    static constexpr int TILE_BYTES = KV_TILE * DH * 2;         // one K or V tile
    static constexpr int CHUNKS_PER_ROW = DH / 8;               // 16-byte chunks per key row
    static constexpr int LD_PER_THREAD = (KV_TILE * CHUNKS_PER_ROW) / 256;
    static constexpr int KSTEPS = DH / 16;                      // QK^T MFMA k-steps
    static constexpr int DBLK = DH / 32;                        // O^T row blocks
    static constexpr int BIAS_BYTES = KV_TILE * 4;
    static constexpr int STAGE_BYTES = 2 * TILE_BYTES + BIAS_BYTES;
    static constexpr int SMEM = 2 * STAGE_BYTES;
    __device__ static __forceinline__ int k_swz(int row) {
        return DH == 64 ? ((row >> 1) & 7) : (row & 15);
    }
};


// QB = 32-row query blocks per wave (1 or 2).  With QB = 2 a wave owns 64 query rows; the two
// blocks are independent softmax streams that share every K / V^T fragment read, and the source
// order  QK(0) QK(1) | softmax(0) | PV(0) | softmax(1) | PV(1)  lets the in-order wave run the
// VALU softmax of one block underneath the MFMAs of the other.
template <int DH, bool HAS_BIAS, int QB>
__global__ __launch_bounds__(256, (QB == 2 || DH == 128) ? 2 : 1) void attn_fwd_kernel(AttnParams p) {
    using C = AttnCfg<DH>;
    constexpr int QW = Q_PER_WAVE * QB;            // query rows per wave
    // FOLD (no key bias): the scale lives in Q and the running max enters the QK^T product itself
    // through one extra MFMA k-step ([1,1,0..] x [-m_hi,-m_lo,0..]), so the accumulator already
    // holds x - m and the per-element VALU work is max + exp2 + cvt: no multiply, no subtract.
    // The VALU issue port is the binding resource of this kernel (PMC: VALU busy 61 % of SIMD
    // time at 52 % MFMA utilisation), the matrix pipe has the headroom for the extra k-step.
    constexpr bool FOLD = !HAS_BIAS && DH == 128;   // measured: pays at head_dim 128, not at 64
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;      // query column of this lane
    const int hh = lane >> 5;     // which 4-row group of every 8 accumulator rows

    // ---- XCD-aware work id (bijective chunking, see gemm.hip)
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, qn = nwg >> 3, rn = nwg & 7;
    const int work = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (orig >> 3);
    const int bh = work / p.q_tiles, qt = work % p.q_tiles;
    const int b = bh / p.H, head = bh % p.H;

    const uint16_t* qb = p.q + (int64_t)b * p.q_sb + head * DH;
    const uint16_t* kb_ = p.k + (int64_t)b * p.k_sb + head * DH;
    const uint16_t* vb = p.v + (int64_t)b * p.v_sb + head * DH;
    uint16_t* ob = p.o + (int64_t)b * p.o_sb + head * DH;
    const float* biasb = HAS_BIAS ? p.bias + (int64_t)b * p.bias_sb : nullptr;

    // ---- Q^T fragments (B operand): lane (r, hh), k-step s holds Q[q][16 s + 8 hh .. +7]
    int q_row[QB];
    bf16x8 qf[QB][C::KSTEPS];
#pragma unroll
    for (int i = 0; i < QB; ++i) {
        q_row[i] = qt * (4 * QW) + wave * QW + 32 * i + r;
        const int q_ld = q_row[i] < p.Lq ? q_row[i] : p.Lq - 1;
        // optional: q is the raw projection output; q_norm (RMSNorm over all H * dh channels from the projection GEMM's
        // partial sums of squares, x weight) and the interleaved-pair RoPE are applied here, with the arithmetic of
        // rmsnorm_rope_kernel (rowops.hip): fp32, one rounding to bf16 at the end (see attention_pipe.hip)
        const float rstd = p.q_on_load() ? p.q_row_rstd(b, q_ld, p.H * DH) : 0.f;
#pragma unroll
        for (int s = 0; s < C::KSTEPS; ++s) {
            qf[i][s] = *(const bf16x8*)(qb + (int64_t)q_ld * p.q_sl + 16 * s + 8 * hh);
            if (p.q_on_load()) {
                const int col = head * DH + 16 * s + 8 * hh;
                const bf16x8 wv = *(const bf16x8*)(p.q_w + col);
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (float)qf[i][s][e] * rstd * (float)wv[e];
                if (p.rope_cos) {
                    const int64_t trow = (int64_t)b * p.rope_sb + (int64_t)q_ld * p.rope_sl;
                    const bf16x8 cv = *(const bf16x8*)(p.rope_cos + trow + col), sv = *(const bf16x8*)(p.rope_sin + trow + col);
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const float r0 = o[e] * (float)cv[e] - o[e + 1] * (float)sv[e];
                        const float r1 = o[e + 1] * (float)cv[e + 1] + o[e] * (float)sv[e + 1];
                        o[e] = r0;
                        o[e + 1] = r1;
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[i][s][e] = (__bf16)o[e];
            }
            if (FOLD) {
                // fold softmax_scale * log2(e) into Q once (bf16 re-rounding of Q: ~2^-9 relative per
                // element, averaged over head_dim in the dot product -- below the bf16 rounding of P)
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[i][s][e] = (__bf16)((float)qf[i][s][e] * p.scale_log2e);
            }
        }
    }

    // ---- staging geometry: thread handles chunks c = tid + 256 i of the [64][DH/8] tile
    int st_key[C::LD_PER_THREAD], st_koff[C::LD_PER_THREAD], st_voff[C::LD_PER_THREAD], st_col[C::LD_PER_THREAD];
#pragma unroll
    for (int i = 0; i < C::LD_PER_THREAD; ++i) {
        const int c = tid + 256 * i;
        const int key = c / C::CHUNKS_PER_ROW, dc = c % C::CHUNKS_PER_ROW;
        st_key[i] = key;
        st_col[i] = dc * 8;
        st_koff[i] = key * C::ROW_BYTES + ((dc ^ C::k_swz(key)) << 4);
        st_voff[i] = C::TILE_BYTES + ((key >> 3) * C::DBLK + (dc >> 2)) * 512 + (key & 7) * 64 + (dc & 3) * 16;
    }
    u32x4 kreg[C::LD_PER_THREAD], vreg[C::LD_PER_THREAD];
    float breg = 0.f;

    // per-thread global offsets of its chunks inside a tile (keys beyond Lk are clamped only
    // when loading the ragged last tile)
    // (32-bit byte offsets from a wave-uniform tile base -> saddr + voffset loads, no 64-bit VALU math)
    uint32_t k_goff[C::LD_PER_THREAD], v_goff[C::LD_PER_THREAD];
#pragma unroll
    for (int i = 0; i < C::LD_PER_THREAD; ++i) {
        k_goff[i] = (uint32_t)(((int64_t)st_key[i] * p.k_sl + st_col[i]) * 2);
        v_goff[i] = (uint32_t)(((int64_t)st_key[i] * p.v_sl + st_col[i]) * 2);
    }
    auto load_tile = [&](int k0, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        if (!TAIL) {
            const char* kt = (const char*)(kb_ + (int64_t)k0 * p.k_sl);
            const char* vt = (const char*)(vb + (int64_t)k0 * p.v_sl);
#pragma unroll
            for (int i = 0; i < C::LD_PER_THREAD; ++i) {
                kreg[i] = *(const u32x4*)(kt + k_goff[i]);
                vreg[i] = *(const u32x4*)(vt + v_goff[i]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < C::LD_PER_THREAD; ++i) {
                int key = k0 + st_key[i];
                key = key < p.Lk ? key : p.Lk - 1;
                kreg[i] = *(const u32x4*)(kb_ + (int64_t)key * p.k_sl + st_col[i]);
                vreg[i] = *(const u32x4*)(vb + (int64_t)key * p.v_sl + st_col[i]);
            }
        }
        if (HAS_BIAS && tid < KV_TILE) {
            const int key = k0 + tid;
            breg = key < p.Lk ? biasb[key] * LOG2E : 0.f;
        }
    };
    auto write_tile = [&](int buf) {
        char* s = smem + buf * C::STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < C::LD_PER_THREAD; ++i) {
            *(u32x4*)(s + st_koff[i]) = kreg[i];
            *(u32x4*)(s + st_voff[i]) = vreg[i];
        }
        if (HAS_BIAS && tid < KV_TILE) *(float*)(s + 2 * C::TILE_BYTES + tid * 4) = breg;
    };

    // ---- per-lane LDS read offsets
    // K (A operand of QK^T): row 32 kb + r, chunk 2 s + hh
    int k_rd[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) k_rd[kb] = (32 * kb + r) * C::ROW_BYTES;
    const int k_sw0 = C::k_swz(r);                 // swz(32 + r) == swz(r) for both head dims
    // V^T (A operand of PV) via transposed reads: block rows 16 s' + 4 hh + q (+8), cols 32 db + r
    const int g16 = lane >> 4, i16 = lane & 15;
    const int v_rd = C::TILE_BYTES + (4 * (g16 >> 1) + (i16 >> 2)) * 64 + (16 * (g16 & 1) + 4 * (i16 & 3)) * 2;

    f32x16 oT[QB][C::DBLK];
    float m_run[QB];           // running max: raw scores (no bias) or scaled+biased log2 domain (bias)
    // Row sums on the matrix pipe (the VALU is the co-limiting pipe at head_dim 64): the P^T fragment
    // of the 32x32x16 PV product, re-read as the B operand of a 16x16x32 MFMA, puts query (l & 15)
    // [+16 for odd 16-lane groups] on the column and this lane's 8 keys in k-group (l >> 4).  With
    // A = 1 on (row 0, even k-groups) and (row 1, odd k-groups), D[0][n] = sum over the tile's keys of
    // P[query n] and D[1][n] = the same for query n + 16: lanes 0..15 hold them in registers 0 and 1.
    // One 16-cycle MFMA per 16 keys replaces 8 v_add per lane; 4 accumulator registers per block.
    f32x4 lT[QB];
    bf16x8 ones;
    {
        const bool on = ((lane & 15) == 0 && ((lane >> 4) & 1) == 0) || ((lane & 15) == 1 && ((lane >> 4) & 1) == 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = on ? (__bf16)1.0f : (__bf16)0.0f;
    }
    bf16x8 kaug, maug[QB];      // FOLD: A operand (ones at k = 0, 1) and B operand (-m as hi + lo bf16)
#pragma unroll
    for (int e = 0; e < 8; ++e) kaug[e] = (hh == 0 && e < 2) ? (__bf16)1.0f : (__bf16)0.0f;
#pragma unroll
    for (int i = 0; i < QB; ++i) {
        m_run[i] = FOLD ? 0.f : -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) maug[i][e] = (__bf16)0.0f;
        lT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < C::DBLK; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) oT[i][d][e] = 0.f;
    }
    const float c = p.scale_log2e;

    const int nt = (p.Lk + KV_TILE - 1) / KV_TILE;
    const int n_full = p.Lk / KV_TILE;            // tiles with all 64 keys valid

    // One key tile: S^T, online softmax, O^T update.  TAIL (ragged last tile only) masks keys >= Lk.
    auto tile_body = [&](int t, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        const int cur = t & 1;
        const char* s = smem + cur * C::STAGE_BYTES;

        // ---------------- S^T = K Q^T, one query block after the other
        f32x16 sT[QB][2];
#pragma unroll
        for (int i = 0; i < QB; ++i)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int e = 0; e < 16; ++e) sT[i][kb][e] = 0.f;
                if (FOLD) sT[i][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kaug, maug[i], sT[i][kb], 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < C::KSTEPS; ++ks) {
                    const bf16x8 kf = *(const bf16x8*)(s + k_rd[kb] + (((2 * ks + hh) ^ k_sw0) << 4));
                    sT[i][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[i][ks], sT[i][kb], 0, 0, 0);
                }
            }

#pragma unroll
        for (int i = 0; i < QB; ++i) {
            // ---------------- scores -> log2 domain (bias variant), mask, running max
            if (HAS_BIAS) {
                const float* bl = (const float*)(s + 2 * C::TILE_BYTES);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 b4 = *(const f32x4*)(bl + 32 * kb + 8 * g + 4 * hh);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            sT[i][kb][4 * g + e] = __builtin_fmaf(sT[i][kb][4 * g + e], c, b4[e]);
                    }
            }
            if (TAIL) {
                const int k0 = t * KV_TILE;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = k0 + 32 * kb + (e & 3) + 8 * (e >> 2) + 4 * hh;
                        if (key >= p.Lk) sT[i][kb][e] = -INFINITY;
                    }
            }
            // row max as a shallow tree (depth 4 with v_max3_f32) instead of one 32-deep dependent
            // chain.  Built with -fno-honor-nans (attention.o only): without it hipcc puts a
            // canonicalising v_max in front of every MFMA output that feeds fmaxf.
            float mt;
            {
                auto max3 = [](float a, float b, float c3) { return fmaxf(fmaxf(a, b), c3); };
                float l1[11];
#pragma unroll
                for (int g = 0; g < 5; ++g) {
                    l1[g] = max3(sT[i][0][3 * g], sT[i][0][3 * g + 1], sT[i][0][3 * g + 2]);
                    l1[5 + g] = max3(sT[i][1][3 * g], sT[i][1][3 * g + 1], sT[i][1][3 * g + 2]);
                }
                l1[10] = max3(sT[i][0][15], sT[i][1][15], l1[0]);
                const float a = max3(l1[1], l1[2], l1[3]), b2 = max3(l1[4], l1[5], l1[6]), c2 = max3(l1[7], l1[8], l1[9]);
                mt = fmaxf(max3(a, b2, c2), l1[10]);
            }
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
                mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            float nmoff = 0.f;                                   // -(max in the exponent's domain)
            if (FOLD) {
                // sT already holds x - m_run.  The max moved iff some element is > 0 (always on the
                // first tile, where m_run = 0 is only a placeholder).  m is kept on a 1/4 grid
                // (rounded UP, so p <= 1) which makes -m exactly representable as hi + lo bf16.
                const bool first = (t == 0);
                if (first || __any(mt > 0.f)) {
                    asm volatile("; max-moved branch (kept a real branch: not if-converted)" ::: "memory");
                    const float dq = (first || mt > 0.f) ? ceilf(mt * 4.0f) * 0.25f : 0.f;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int e = 0; e < 16; ++e) sT[i][kb][e] -= dq;
                    if (!first) {
                        const float alpha = fast_exp2(-dq);
                        lT[i][0] *= alpha;
                        lT[i][1] *= __shfl(alpha, (lane + 16) & 63, 64);
#pragma unroll
                        for (int d = 0; d < C::DBLK; ++d)
#pragma unroll
                            for (int e = 0; e < 16; ++e) oT[i][d][e] *= alpha;
                    }
                    m_run[i] += dq;
                    const float nm = -m_run[i];
                    const __bf16 mh = (__bf16)nm;
                    const __bf16 ml = (__bf16)(nm - (float)mh);
                    maug[i][0] = hh == 0 ? mh : (__bf16)0.0f;
                    maug[i][1] = hh == 0 ? ml : (__bf16)0.0f;
                }
            } else {
                const float m_new = fmaxf(m_run[i], mt);
                // rescale only when some row's max moved (wave-uniform branch): after the first few
                // tiles the running max is stable for most tiles and the O-wide multiply is skipped
                if (__any(m_new != m_run[i])) {
                    asm volatile("; rescale branch (kept a real branch: not if-converted)" ::: "memory");
                    const float alpha = HAS_BIAS ? fast_exp2(m_run[i] - m_new) : fast_exp2((m_run[i] - m_new) * c);
                    // lane n (< 16) holds the sums of queries n (reg 0) and n + 16 (reg 1)
                    lT[i][0] *= alpha;
                    lT[i][1] *= __shfl(alpha, (lane + 16) & 63, 64);
#pragma unroll
                    for (int d = 0; d < C::DBLK; ++d)
#pragma unroll
                        for (int e = 0; e < 16; ++e) oT[i][d][e] *= alpha;
                    m_run[i] = m_new;
                }
                nmoff = HAS_BIAS ? -m_run[i] : -m_run[i] * c;
            }

            // ---------------- P = exp2(x - m), bf16 fragments
            bf16x8 pf[4];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    sT[i][kb][e] = fast_exp2(FOLD ? sT[i][kb][e] : (HAS_BIAS ? sT[i][kb][e] + nmoff : __builtin_fmaf(sT[i][kb][e], c, nmoff)));
                }
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[2 * kb + h2][e] = (__bf16)sT[i][kb][8 * h2 + e];
            }

            // ---------------- O^T += V^T P^T  (;  l += ones . P^T)
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
                lT[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[sp], lT[i], 0, 0, 0);
#pragma unroll
                for (int d = 0; d < C::DBLK; ++d) {
                    const char* base = s + v_rd + (2 * sp * C::DBLK + d) * 512;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(base));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(base + C::DBLK * 512));
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    const bf16x8 vf = __builtin_bit_cast(bf16x8, both);
                    oT[i][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[sp], oT[i][d], 0, 0, 0);
                }
            }
        }
    };

    using no_tail = std::integral_constant<bool, false>;
    using with_tail = std::integral_constant<bool, true>;
    if (n_full > 0) load_tile(0, no_tail{}); else load_tile(0, with_tail{});
    write_tile(0);
    __syncthreads();

    // full tiles: prefetch t+1 (itself full, or the ragged last one), compute t, stage t+1
    for (int t = 0; t < n_full; ++t) {
        const bool has_next = t + 1 < nt;
        if (has_next) {
            if (t + 1 < n_full) load_tile((t + 1) * KV_TILE, no_tail{});
            else load_tile((t + 1) * KV_TILE, with_tail{});
        }
        tile_body(t, no_tail{});
        if (has_next) write_tile((t & 1) ^ 1);
        __syncthreads();
    }
    if (n_full < nt) tile_body(n_full, with_tail{});      // ragged last tile (nothing left to prefetch)

    // ---------------- epilogue: O = O^T / l
#pragma unroll
    for (int i = 0; i < QB; ++i) {
        // query r's sum sits in lane (r & 15), register (r >> 4)
        const float l0 = __shfl(lT[i][0], r & 15, 64), l1 = __shfl(lT[i][1], r & 15, 64);
        const float l = (r & 16) ? l1 : l0;
        const float inv = 1.0f / l;
        // A lane holds 8-byte pieces of ONE output row; stored from registers, every store instruction
        // would touch 32 different 128-byte lines with 16 bytes each (store-issue bound, cdna guide T21).
        // The 32 x DH block goes through a per-wave LDS scratch (the K/V stages are free now) and leaves
        // as whole rows: 16 bytes per lane, 64 / (DH/8) complete rows per instruction.
        char* scr = smem + wave * (32 * C::ROW_BYTES);
        if (i == 0) __syncthreads();                      // every wave is done reading the last K/V tile
#pragma unroll
        for (int d = 0; d < C::DBLK; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 w;
                w[0] = pack_bf16(oT[i][d][4 * g + 0] * inv, oT[i][d][4 * g + 1] * inv);
                w[1] = pack_bf16(oT[i][d][4 * g + 2] * inv, oT[i][d][4 * g + 3] * inv);
                const int chunk = 4 * d + g;              // 16-byte chunk of the row; this lane's half = hh
                *(u32x2*)(scr + r * C::ROW_BYTES + ((chunk ^ (r & 7)) << 4) + hh * 8) = w;
            }
        constexpr int LPR = C::CHUNKS_PER_ROW;            // lanes per output row
        constexpr int RPI = 64 / LPR;                     // rows per store instruction
        const int q0 = qt * (4 * QW) + wave * QW + 32 * i;
#pragma unroll
        for (int t = 0; t < 32 / RPI; ++t) {
            const int row = t * RPI + lane / LPR, chunk = lane % LPR;
            const u32x4 w = *(const u32x4*)(scr + row * C::ROW_BYTES + ((chunk ^ (row & 7)) << 4));
            if (q0 + row < p.Lq) *(u32x4*)(ob + p.o_row(q0 + row) + chunk * 8) = w;
        }
    }
}

template <int DH, bool HAS_BIAS, int QB>
static int launch(AttnParams p, hipStream_t stream) {
    using C = AttnCfg<DH>;
    auto kern = attn_fwd_kernel<DH, HAS_BIAS, QB>;
    static unsigned long long lds_done = 0;
    if (const int rc = reserve_lds((const void*)kern, C::SMEM, &lds_done, "ltxmi_attention_fwd_bf16")) return rc;
    const int q_per_wg = Q_PER_WG * QB;
    p.q_tiles = (p.Lq + q_per_wg - 1) / q_per_wg;
    const int64_t grid = (int64_t)p.B * p.H * p.q_tiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), C::SMEM, stream, p);
    return check_launch("ltxmi_attention_fwd_bf16");
}

}  // namespace ltxmi

using namespace ltxmi;

extern "C" int ltxmi_attention_fwd_bf16(const ltxmi_attn_args* a, void* stream) {
    LTXMI_REQUIRE(a && a->q && a->k && a->v && a->o, LTXMI_ERR_INVALID_ARG, "ltxmi_attention_fwd_bf16: NULL argument");
    LTXMI_REQUIRE(a->B > 0 && a->H > 0 && a->Lq > 0 && a->Lk > 0, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_attention_fwd_bf16: non-positive shape B=%d H=%d Lq=%d Lk=%d", a->B, a->H, a->Lq, a->Lk);
    LTXMI_REQUIRE(a->head_dim == 64 || a->head_dim == 128, LTXMI_ERR_UNSUPPORTED,
                  "ltxmi_attention_fwd_bf16: head_dim %d not in {64, 128}", a->head_dim);
    const int64_t strides[] = {a->q_stride_b, a->q_stride_l, a->k_stride_b, a->k_stride_l,
                               a->v_stride_b, a->v_stride_l, a->o_stride_b, a->o_stride_l};
    for (int64_t s : strides)
        LTXMI_REQUIRE(s % 8 == 0, LTXMI_ERR_UNSUPPORTED, "ltxmi_attention_fwd_bf16: strides must be multiples of 8 elements");
    LTXMI_REQUIRE((((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->o) & 15) == 0,
                  LTXMI_ERR_UNSUPPORTED, "ltxmi_attention_fwd_bf16: q/k/v/o must be 16-byte aligned");
    LTXMI_REQUIRE((int64_t)a->B * a->H * ((a->Lq + Q_PER_WG - 1) / Q_PER_WG) < (1ll << 31), LTXMI_ERR_UNSUPPORTED,
                  "ltxmi_attention_fwd_bf16: grid too large");
    AttnParams p;
    p.q = (const uint16_t*)a->q; p.q_sb = a->q_stride_b; p.q_sl = a->q_stride_l;
    p.k = (const uint16_t*)a->k; p.k_sb = a->k_stride_b; p.k_sl = a->k_stride_l;
    p.v = (const uint16_t*)a->v; p.v_sb = a->v_stride_b; p.v_sl = a->v_stride_l;
    p.o = (uint16_t*)a->o; p.o_sb = a->o_stride_b; p.o_sl = a->o_stride_l;
    p.bias = a->key_bias; p.bias_sb = a->bias_stride_b;
    p.B = a->B; p.H = a->H; p.Lq = a->Lq; p.Lk = a->Lk;
    p.scale_log2e = a->softmax_scale * LOG2E;
    p.q_tiles = (a->Lq + Q_PER_WG - 1) / Q_PER_WG;
    p.q_ss = a->q_rowsumsq; p.q_ss_sb = a->q_rowsumsq_stride_b; p.q_ss_sl = a->q_rowsumsq_stride_l;
    p.q_ss_n = a->q_rowsumsq_blocks;
    p.q_rstd = a->q_rstd; p.q_rstd_sb = a->q_rstd_stride_b; p.q_rstd_sl = a->q_rstd_stride_l;
    p.q_w = (const uint16_t*)a->q_norm_weight; p.q_eps = a->q_norm_eps;
    p.rope_cos = (const uint16_t*)a->rope_cos; p.rope_sin = (const uint16_t*)a->rope_sin;
    p.rope_sb = a->rope_stride_b; p.rope_sl = a->rope_stride_l;
    p.o_seg = a->o_segment_len; p.o_sseg = a->o_stride_segment;
    p.redo_count = a->redo_counter; p.force_exact = a->force_exact != 0;
    LTXMI_REQUIRE(a->o_segment_len >= 0 && a->o_stride_segment % 8 == 0, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_attention_fwd_bf16: bad output segment geometry");
    const bool span_ok = attn_pipe_span_ok(a->Lk, a->k_stride_l, a->v_stride_l, a->head_dim);
    const bool pipe_ok = span_ok && attn_pipe_takes(a->B, a->H, a->Lq, a->Lk, a->head_dim, a->key_bias != nullptr);
    if (a->q_rowsumsq || a->q_rstd) {
        LTXMI_REQUIRE(a->q_norm_weight && (((uintptr_t)a->q_norm_weight) & 15) == 0 &&
                          (a->q_rstd ? (((uintptr_t)a->q_rstd) & 3) == 0
                                     : (a->q_rowsumsq_blocks == a->H * a->head_dim / 64 && (((uintptr_t)a->q_rowsumsq) & 3) == 0)),
                      LTXMI_ERR_INVALID_ARG, "ltxmi_attention_fwd_bf16: bad q_rowsumsq / q_rstd / q_norm_weight geometry");
        LTXMI_REQUIRE((a->rope_cos == nullptr) == (a->rope_sin == nullptr), LTXMI_ERR_INVALID_ARG,
                      "ltxmi_attention_fwd_bf16: rope_cos and rope_sin must both be given or both be NULL");
        if (a->rope_cos)
            LTXMI_REQUIRE((((uintptr_t)a->rope_cos | (uintptr_t)a->rope_sin) & 15) == 0 && a->rope_stride_l % 8 == 0 &&
                              a->rope_stride_b % 8 == 0,
                          LTXMI_ERR_INVALID_ARG, "ltxmi_attention_fwd_bf16: RoPE tables must be 16-byte aligned rows");
    }
    hipStream_t s = (hipStream_t)stream;
    // 64 query rows per wave (two blocks sharing each K/V fragment) once there is enough work to
    // fill the chip with 256-row workgroups; 32 rows per wave otherwise (and always at head_dim 128,
    // where two blocks of accumulators do not fit the register file at 2 waves per SIMD)
    const int64_t wg256 = (int64_t)a->B * a->H * ((a->Lq + 255) / 256);
    if (a->head_dim == 64) {
        // short key sequences (the T5 cross-attention): K / V resident in LDS, single-pass softmax (attention_cross.hip)
        if (attn_cross_takes(a->B, a->H, a->Lq, a->Lk, a->head_dim)) return launch_attn_cross(p, s);
        // large self-attention: the software-pipelined LDS-DMA kernel (attention_pipe.hip)
        if (pipe_ok) {
            const int rc = launch_attn_pipe(p, s);
            if (rc != -1) return rc;
        }
        if (wg256 >= 512 && !a->key_bias) return launch<64, false, ATTN_QB_BIG>(p, s);
        return a->key_bias ? launch<64, true, 1>(p, s) : launch<64, false, 1>(p, s);
    }
    // head_dim 128, large self-attention: the one-wave-per-SIMD pipelined kernel (attention_pipe128.hip)
    if (span_ok && attn_pipe128_takes(a->B, a->H, a->Lq, a->Lk, a->head_dim, a->key_bias != nullptr)) {
        const int rc = launch_attn_pipe128(p, s);
        if (rc != -1) return rc;
    }
    return a->key_bias ? launch<128, true, 1>(p, s) : launch<128, false, 1>(p, s);
}

// Which kernel instance ltxmi_attention_fwd_bf16 runs for a shape (mirrors the dispatch above, INCLUDING the 2 GiB span test
// of the pipelined kernels, which depends on the key / value token strides).  Two shapes with the same id are computed with
// the same arithmetic per (batch, head, query row): what ltxmi.Transformer3DModel checks before it runs a sub-batch of rows
// and claims bit-identity with the full batch.
extern "C" int ltxmi_attention_kernel_id(int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t head_dim, int32_t has_key_bias,
                                         int64_t k_stride_l, int64_t v_stride_l) {
    if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0 || k_stride_l <= 0 || v_stride_l <= 0) return -1;
    if (head_dim != 64 && head_dim != 128) return -1;
    const bool span_ok = attn_pipe_span_ok(Lk, k_stride_l, v_stride_l, head_dim);
    if (head_dim == 128) {
        if (span_ok && attn_pipe128_takes(B, H, Lq, Lk, head_dim, has_key_bias != 0)) return 6;
        return has_key_bias ? 5 : 4;
    }
    if (attn_cross_takes(B, H, Lq, Lk, head_dim)) return 7;
    if (span_ok && attn_pipe_takes(B, H, Lq, Lk, head_dim, has_key_bias != 0)) return 3;
    const int64_t wg256 = (int64_t)B * H * ((Lq + 255) / 256);
    if (wg256 >= 512 && !has_key_bias) return 2;
    return has_key_bias ? 1 : 0;
}

extern "C" int ltxmi_attention_fuses_qnorm(int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t head_dim, int32_t has_key_bias) {
    // every kernel behind ltxmi_attention_fwd_bf16 normalises q on load (round 2: also the key-bias / small-shape /
    // head_dim-128 kernel); kept as a query so that callers written against it keep working
    return (B > 0 && H > 0 && Lq > 0 && Lk > 0 && (head_dim == 64 || head_dim == 128) && (H * head_dim) % 64 == 0) ? 1 : 0;
}
