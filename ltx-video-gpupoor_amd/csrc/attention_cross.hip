// attention_cross.hip -- attention against a SHORT key sequence (<= 256 keys, head_dim 64): the DiT's T5 cross-attention
// (attention.py:294-311, 1042-1048 of the reference -> pay_attention's eager branch with the additive key mask built at
// transformer3d.py:411-415), 28 launches per denoise step with Lq = 4992 x B_eff queries against 256 text keys.
//
// attention.hip serves this shape with one short-lived workgroup per 128 query rows: each loads its Q rows from HBM, then
// walks the four key tiles through two LDS buffers (register-staged loads from L2, a barrier per tile, the online-softmax
// rescale machinery) and is gone after ~4.5 us, most of it load latency -- 3744 workgroups, 14.6 rounds, the matrix pipe
// 20 % busy (profiles/r03_pmc_mfma.md).  With 256 keys the whole K and V of a (batch, head) are 64 KB: here a workgroup
//   * loads them (and the key bias) into LDS ONCE -- the two LDS images of attention.hip, four tiles side by side --
//     one barrier, none afterwards;
//   * then walks its share of the (batch, head)'s query rows, 32 per wave and iteration, with the next iteration's Q rows
//     requested from HBM under the current one's arithmetic;
//   * takes all (up to) 256 scores of a query row in registers at once, so the softmax is the textbook single pass -- row
//     maximum, exp2, sum -- with no running maximum, no rescale of O and no branch;
//   * writes O through a 2-KB per-wave LDS scratch as whole 64-byte row halves.
// Same lane layouts and the same arithmetic per score as attention.hip (swapped product S^T = K Q^T with the query on the
// lane, P^T straight from the accumulator registers into the PV product, V^T by transposed LDS reads, row sums on the matrix
// pipe).  Two workgroups per CU (75 KB of LDS each); grid = B x H x G with G row groups per (batch, head) so that the chip's
// 512 slots are filled once.
#include "attention.h"

namespace ltxmi {

namespace cross {

constexpr int DH = 64;
constexpr int KV_TILE = 64;
constexpr int MAX_TILES = 4;                         // <= 256 keys
constexpr int ROW_BYTES = DH * 2;
constexpr int TILE_BYTES = KV_TILE * DH * 2;         // 8 KiB: one K or V tile
constexpr int K_OFF = 0, V_OFF = MAX_TILES * TILE_BYTES, BIAS_OFF = 2 * MAX_TILES * TILE_BYTES;
constexpr int SCR_OFF = BIAS_OFF + MAX_TILES * KV_TILE * 4;
constexpr int SCR_BYTES = 32 * 64;                   // per wave: 32 rows x 32 columns of bf16
constexpr int SMEM = SCR_OFF + 4 * SCR_BYTES;        // 64 K + 1 K + 8 K = 74 752 B
constexpr int ROWS_PER_IT = 128;                     // 4 waves x 32 query rows
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int k_swz(int row) { return (row >> 1) & 7; }

__global__ __launch_bounds__(256, 2) void attn_cross_kernel(AttnParams p, int groups, int rows_per_group) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;

    // ---- XCD-aware work id (bijective chunking): the row groups of a (batch, head) run on one XCD and share its K / V in L2
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, qn = nwg >> 3, rn = nwg & 7;
    const int work = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (orig >> 3);
    const int bh = work / groups, grp = work % groups;
    const int b = bh / p.H, head = bh % p.H;
    const int row_begin = grp * rows_per_group;
    const int row_end = min(p.Lq, row_begin + rows_per_group);
    if (row_begin >= row_end) return;                                      // (whole workgroup: before any barrier)
    const int n_it = (row_end - row_begin + ROWS_PER_IT - 1) / ROWS_PER_IT;

    const uint16_t* qb = p.q + (int64_t)b * p.q_sb + head * DH;
    const uint16_t* kb_ = p.k + (int64_t)b * p.k_sb + head * DH;
    const uint16_t* vb = p.v + (int64_t)b * p.v_sb + head * DH;
    uint16_t* ob = p.o + (int64_t)b * p.o_sb + head * DH;
    const int nt = (p.Lk + KV_TILE - 1) / KV_TILE;                          // 1 .. 4 (wave-uniform)

    // ---- K / V / bias -> LDS, all tiles at once.  Thread handles 16-byte chunks c = tid + 256 i of a [64][8] tile (attention.hip's
    // staging geometry); keys past Lk re-read the last key (finite data) and get a bias of -inf, i.e. P = 0.
    {
        u32x4 kreg[MAX_TILES][2], vreg[MAX_TILES][2];
#pragma unroll
        for (int t = 0; t < MAX_TILES; ++t)
            if (t < nt) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int c = tid + 256 * i;
                    int key = t * KV_TILE + (c >> 3);
                    key = key < p.Lk ? key : p.Lk - 1;
                    kreg[t][i] = *(const u32x4*)(kb_ + (int64_t)key * p.k_sl + (c & 7) * 8);
                    vreg[t][i] = *(const u32x4*)(vb + (int64_t)key * p.v_sl + (c & 7) * 8);
                }
            }
        if (tid < MAX_TILES * KV_TILE) {
            float bv = -INFINITY;
            if (tid < p.Lk) bv = p.bias ? p.bias[(int64_t)b * p.bias_sb + tid] * LOG2E : 0.f;
            *(float*)(smem + BIAS_OFF + tid * 4) = bv;
        }
#pragma unroll
        for (int t = 0; t < MAX_TILES; ++t)
            if (t < nt) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int c = tid + 256 * i;
                    const int key = c >> 3, dc = c & 7;
                    *(u32x4*)(smem + K_OFF + t * TILE_BYTES + key * ROW_BYTES + ((dc ^ k_swz(key)) << 4)) = kreg[t][i];
                    *(u32x4*)(smem + V_OFF + t * TILE_BYTES + ((key >> 3) * 2 + (dc >> 2)) * 512 + (key & 7) * 64 + (dc & 3) * 16) = vreg[t][i];
                }
            }
    }
    __syncthreads();                                   // the only barrier: from here on K / V / bias are read-only

    // ---- per-lane LDS read offsets (attention.hip)
    const int k_rd0 = r * ROW_BYTES, k_rd1 = (32 + r) * ROW_BYTES;
    const int k_sw0 = k_swz(r);                        // swz(32 + r) == swz(r)
    const int g16 = lane >> 4, i16 = lane & 15;
    const int v_rd = (4 * (g16 >> 1) + (i16 >> 2)) * 64 + (16 * (g16 & 1) + 4 * (i16 & 3)) * 2;
    bf16x8 ones;
    {
        const bool on = ((lane & 15) == 0 && ((lane >> 4) & 1) == 0) || ((lane & 15) == 1 && ((lane >> 4) & 1) == 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = on ? (__bf16)1.0f : (__bf16)0.0f;
    }
    const float c = p.scale_log2e;
    char* scr = smem + SCR_OFF + wave * SCR_BYTES;

    // Q^T fragments of one iteration's rows, as loaded (the q-on-load arithmetic runs when the iteration starts)
    auto load_q = [&](int it, bf16x8 (&raw)[4], int& q_ld) {
        const int row = row_begin + it * ROWS_PER_IT + wave * 32 + r;
        q_ld = row < p.Lq ? row : p.Lq - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) raw[s] = *(const bf16x8*)(qb + (int64_t)q_ld * p.q_sl + 16 * s + 8 * hh);
    };
    bf16x8 qraw[4];
    int q_ld;
    load_q(0, qraw, q_ld);

    for (int it = 0; it < n_it; ++it) {
        // ---- this iteration's Q^T fragments; optional q_norm (+ RoPE) on load with rmsnorm_rope_kernel's arithmetic (see
        // attention_pipe.hip): x * rstd * weight, interleaved-pair rotation, ONE rounding to bf16
        bf16x8 qf[4];
        if (p.q_on_load()) {
            const float rstd = p.q_row_rstd(b, q_ld, p.H * DH);
            const int64_t trow = (int64_t)b * p.rope_sb + (int64_t)q_ld * p.rope_sl;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int col = head * DH + 16 * s + 8 * hh;
                const bf16x8 wv = *(const bf16x8*)(p.q_w + col);
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (float)qraw[s][e] * rstd * (float)wv[e];
                if (p.rope_cos) {
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
                for (int e = 0; e < 8; ++e) qf[s][e] = (__bf16)o[e];
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) qf[s] = qraw[s];
        }
        const int q0 = row_begin + it * ROWS_PER_IT + wave * 32;           // this wave's first row (wave-uniform)

        // ---- S^T = K Q^T for every key tile: the whole score row of a query stays in registers
        f32x16 sT[MAX_TILES][2];
#pragma unroll
        for (int t = 0; t < MAX_TILES; ++t)
            if (t < nt) {
                const char* ks = smem + K_OFF + t * TILE_BYTES;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) sT[t][kb][e] = 0.f;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const bf16x8 kf = *(const bf16x8*)(ks + (kb ? k_rd1 : k_rd0) + (((2 * s4 + hh) ^ k_sw0) << 4));
                        sT[t][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s4], sT[t][kb], 0, 0, 0);
                    }
                }
            }
        // the next iteration's Q rows: requested now, used ~2000 cycles from here
        int q_ld_next = q_ld;
        if (it + 1 < n_it) load_q(it + 1, qraw, q_ld_next);

        // ---- scores -> log2 domain with the key bias; row maximum over all keys
        float mt = -INFINITY;
#pragma unroll
        for (int t = 0; t < MAX_TILES; ++t)
            if (t < nt) {
                const float* bl = (const float*)(smem + BIAS_OFF) + t * KV_TILE;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 b4 = *(const f32x4*)(bl + 32 * kb + 8 * g + 4 * hh);
#pragma unroll
                        for (int e = 0; e < 4; ++e) sT[t][kb][4 * g + e] = __builtin_fmaf(sT[t][kb][4 * g + e], c, b4[e]);
                    }
                auto max3 = [](float a, float b2, float c3) { return fmaxf(fmaxf(a, b2), c3); };
                float l1[11];
#pragma unroll
                for (int g = 0; g < 5; ++g) {
                    l1[g] = max3(sT[t][0][3 * g], sT[t][0][3 * g + 1], sT[t][0][3 * g + 2]);
                    l1[5 + g] = max3(sT[t][1][3 * g], sT[t][1][3 * g + 1], sT[t][1][3 * g + 2]);
                }
                l1[10] = max3(sT[t][0][15], sT[t][1][15], l1[0]);
                const float a = max3(l1[1], l1[2], l1[3]), b2 = max3(l1[4], l1[5], l1[6]), c2 = max3(l1[7], l1[8], l1[9]);
                mt = max3(mt, max3(a, b2, c2), l1[10]);
            }
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
            mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        // (every key masked: the reference's softmax over a row of -10000s is uniform; here the bias is finite for every
        //  real key, so mt is finite whenever Lk >= 1)

        // ---- P = 2^(x - m), O^T += V^T P^T, l += sum P, tile by tile (a tile's score registers are dead after its P)
        f32x16 oT[2];
        f32x4 lT = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) oT[d][e] = 0.f;
#pragma unroll
        for (int t = 0; t < MAX_TILES; ++t)
            if (t < nt) {
                bf16x8 pf[4];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) sT[t][kb][e] = fast_exp2(sT[t][kb][e] - mt);
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                        for (int e = 0; e < 8; ++e) pf[2 * kb + h2][e] = (__bf16)sT[t][kb][8 * h2 + e];
                }
                const char* vs = smem + V_OFF + t * TILE_BYTES + v_rd;
#pragma unroll
                for (int sp = 0; sp < 4; ++sp) {
                    lT = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[sp], lT, 0, 0, 0);
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const char* base = vs + (2 * sp * 2 + d) * 512;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 2 * 512));
                        typedef __attribute__((ext_vector_type(8))) short s16x8;
                        const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        oT[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, both), pf[sp], oT[d], 0, 0, 0);
                    }
                }
            }

        // ---- O = O^T / l: query r's sum sits in lane (r & 15), register (r >> 4).  The 32 x 64 block leaves in two halves of
        // 32 columns through the wave's 2-KB scratch: 16 bytes per lane, 16 whole 64-byte row halves per store instruction.
        const float l0 = __shfl(lT[0], r & 15, 64), l1v = __shfl(lT[1], r & 15, 64);
        const float inv = 1.0f / ((r & 16) ? l1v : l0);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 w;
                w[0] = pack_bf16(oT[d][4 * g + 0] * inv, oT[d][4 * g + 1] * inv);
                w[1] = pack_bf16(oT[d][4 * g + 2] * inv, oT[d][4 * g + 3] * inv);
                *(u32x2*)(scr + r * 64 + ((g ^ (r & 3)) << 4) + hh * 8) = w;       // 16-byte chunk g of the half row, this lane's 8 bytes
            }
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int row = t2 * 16 + (lane >> 2), chunk = lane & 3;
                const u32x4 w = *(const u32x4*)(scr + row * 64 + ((chunk ^ (row & 3)) << 4));
                if (q0 + row < row_end) *(u32x4*)(ob + p.o_row(q0 + row) + 32 * d + chunk * 8) = w;
            }
        }
        q_ld = q_ld_next;
    }
}

}  // namespace cross

// Shapes this kernel takes: head_dim 64, at most 256 keys, and enough query rows that the one-time K / V load pays (every
// workgroup gets at least two iterations of 128 rows).
bool attn_cross_takes(int B, int H, int Lq, int Lk, int head_dim) {
    return head_dim == 64 && Lk >= 1 && Lk <= cross::MAX_TILES * cross::KV_TILE && Lq >= 1024;
}

int launch_attn_cross(AttnParams p, hipStream_t stream) {
    const int n_cu = device_cu_count("ltxmi_attention_fwd_bf16");
    if (n_cu <= 0) return LTXMI_ERR_LAUNCH;
    static unsigned long long lds_done = 0;
    if (const int rc = reserve_lds((const void*)cross::attn_cross_kernel, cross::SMEM, &lds_done, "ltxmi_attention_fwd_bf16")) return rc;
    // row groups per (batch, head): fill the chip's 2 x CUs slots once, never below two iterations per workgroup
    const int slots = 2 * n_cu, bh = p.B * p.H;
    int groups = slots / bh;
    if (groups < 1) groups = 1;
    const int max_groups = (p.Lq + 2 * cross::ROWS_PER_IT - 1) / (2 * cross::ROWS_PER_IT);
    if (groups > max_groups) groups = max_groups;
    int rows = (p.Lq + groups - 1) / groups;
    rows = (rows + cross::ROWS_PER_IT - 1) / cross::ROWS_PER_IT * cross::ROWS_PER_IT;
    groups = (p.Lq + rows - 1) / rows;
    const int64_t grid = (int64_t)bh * groups;
    hipLaunchKernelGGL(cross::attn_cross_kernel, dim3((unsigned)grid), dim3(256), cross::SMEM, stream, p, groups, rows);
    return check_launch("ltxmi_attention_fwd_bf16");
}

}  // namespace ltxmi
