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
#include <type_traits>

#include "attention.h"

namespace ltxmi {

namespace cross {

#ifndef LTXMI_XATTN_STAGGER
#define LTXMI_XATTN_STAGGER 32        // x 64 cycles: how much later a CU's second resident workgroup starts its first iteration
#endif

constexpr int DH = 64;
constexpr int KV_TILE = 64;
constexpr int MAX_TILES = 4;                         // <= 256 keys
constexpr int ROW_BYTES = DH * 2;
constexpr int TILE_BYTES = KV_TILE * DH * 2;         // 8 KiB: one K or V tile
constexpr int K_OFF = 0, V_OFF = MAX_TILES * TILE_BYTES, BIAS_OFF = 2 * MAX_TILES * TILE_BYTES;
constexpr int SCR_OFF = BIAS_OFF + MAX_TILES * KV_TILE * 16;           // 16 bytes per key: [bias hi, bias lo, 0 x 6] (bf16)
constexpr int SCR_BYTES = 32 * 64;                   // per wave: 32 rows x 32 columns of bf16
constexpr int QW_OFF = SCR_OFF + 4 * SCR_BYTES;       // this head's 64 q_norm weights (q finished on load)
constexpr int SMEM = QW_OFF + DH * 2;                // 64 K + 4 K + 8 K + 128 B = 77 952 B
constexpr int ROWS_PER_IT = 128;                     // 4 waves x 32 query rows
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int k_swz(int row) { return (row >> 1) & 7; }

// NT = key tiles (ceil(Lk / 64), 1 .. 4): a compile-time constant, so that an iteration is one straight-line block hipcc can
// schedule across (a run-time tile count put a wave-uniform branch around every tile's instructions)
template <int NT>
__global__ __launch_bounds__(256, 2) void attn_cross_kernel(AttnParams p, int groups, int rows_per_group, int stagger_from) {
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
    constexpr int nt = NT;

    // ---- K / V / bias -> LDS, all tiles at once.  Thread handles 16-byte chunks c = tid + 256 i of a [64][8] tile (attention.hip's
    // staging geometry); keys past Lk re-read the last key (finite data) and get a bias of -inf, i.e. P = 0.
    {
        u32x4 kreg[NT][2], vreg[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
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
            // The key bias rides on the QK^T product as one more k-step: K gets the columns [b', b' - bf16(b')] (hi + lo: exact to
            // 2^-16 relative), Q the columns [1, 1], with b' = bias / softmax_scale so that c (K Q^T + b') = c K Q^T + bias log2(e).
            // Keys past Lk get -1e30 (finite: -inf would meet the zero columns of Q as NaN), i.e. P = 0.
            float bv = -1e30f;
            if (tid < p.Lk) bv = p.bias ? p.bias[(int64_t)b * p.bias_sb + tid] * (LOG2E / p.scale_log2e) : 0.f;
            const __bf16 hi = (__bf16)bv;
            const __bf16 lo = (__bf16)(bv - (float)hi);
            u32x4 w = {0u, 0u, 0u, 0u};
            w[0] = (uint32_t)__builtin_bit_cast(uint16_t, hi) | ((uint32_t)__builtin_bit_cast(uint16_t, lo) << 16);
            *(u32x4*)(smem + BIAS_OFF + tid * 16) = w;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int c = tid + 256 * i;
                    const int key = c >> 3, dc = c & 7;
                    *(u32x4*)(smem + K_OFF + t * TILE_BYTES + key * ROW_BYTES + ((dc ^ k_swz(key)) << 4)) = kreg[t][i];
                    *(u32x4*)(smem + V_OFF + t * TILE_BYTES + ((key >> 3) * 2 + (dc >> 2)) * 512 + (key & 7) * 64 + (dc & 3) * 16) = vreg[t][i];
                }
            }
    }
    if (p.q_on_load() && tid < 8) *(u32x4*)(smem + QW_OFF + tid * 16) = *(const u32x4*)(p.q_w + head * DH + tid * 8);
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

    // ---- the Q pipeline.  An iteration's rows are requested one iteration ahead (raw, into qraw, with the row's RMSNorm
    // factor when q is finished on load) and turned into the Q^T fragments qf in the MIDDLE of the iteration before theirs,
    // right after the QK^T phase has read qf for the last time; the request for the iteration after goes out behind that.
    // Where that wait sits matters: hipcc waits for vector memory with vmcnt(0) across the loop's back edge, i.e. for
    // EVERYTHING outstanding.  At the top of an iteration that included the output stores issued a moment earlier (~2 us of
    // HBM write latency exposed per iteration: 47 us per launch); in the middle, everything outstanding is half an iteration
    // old or more.
    bf16x8 qraw[4], qf[4];
    float q_rstd = 0.f;
    int q_ld = 0;
    auto load_q = [&](int it) {
        const int row = row_begin + it * ROWS_PER_IT + wave * 32 + r;
        q_ld = row < p.Lq ? row : p.Lq - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) qraw[s] = *(const bf16x8*)(qb + (int64_t)q_ld * p.q_sl + 16 * s + 8 * hh);
        if (p.q_on_load()) q_rstd = p.q_row_rstd(b, q_ld, p.H * DH);
    };
    // qraw -> qf; optional q_norm (+ RoPE) with rmsnorm_rope_kernel's arithmetic (see attention_pipe.hip): x * rstd * weight,
    // interleaved-pair rotation, ONE rounding to bf16
    auto finish_q = [&]() {
        if (p.q_on_load()) {
            const int64_t trow = (int64_t)b * p.rope_sb + (int64_t)q_ld * p.rope_sl;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 wv = *(const bf16x8*)(smem + QW_OFF + (16 * s + 8 * hh) * 2);
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (float)qraw[s][e] * q_rstd * (float)wv[e];
                if (p.rope_cos) {
                    const int col = head * DH + 16 * s + 8 * hh;
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
    };
    load_q(0);
    finish_q();
    if (n_it > 1) load_q(1);
    // The two workgroups of a CU run the same program from the same start: both in their matrix phase, then both in their
    // VALU phase.  The second resident (the dispatcher fills every CU once before it doubles up) starts half an iteration
    // late, so that one's exp2 / cvt stream runs under the other's MFMAs.
    if (blockIdx.x >= (unsigned)stagger_from) __builtin_amdgcn_s_sleep(LTXMI_XATTN_STAGGER);

    for (int it = 0; it < n_it; ++it) {
        const int q0 = row_begin + it * ROWS_PER_IT + wave * 32;           // this wave's first row (wave-uniform)

        // ---- S'^T = K Q^T + b' for every key tile: the whole score row of a query stays in registers (NG accumulator tuples,
        // never copied).  Five MFMAs per group of 32 keys -- the bias k-step, then the four of head_dim 64 -- with the next
        // group's fragments requested behind the MFMA that frees their registers (hipcc left to itself read each K fragment
        // right in front of its MFMA and waited for it).
        constexpr int NG = 2 * NT;
        f32x16 sT[NG];
        bf16x8 kfr[5];                                   // [0] = the bias fragment, [1 + s4] = K, k-step s4
        bf16x8 qaug;                                     // Q's side of the bias k-step: [1, 1, 0 ...] on the hh = 0 lanes
        {
            uint32_t w0 = hh == 0 ? 0x3f803f80u : 0u, w1 = 0u;
            asm volatile("" : "+v"(w0), "+v"(w1));       // (made here every iteration: two registers less to keep across it)
            const u32x4 w = {w0, w1, w1, w1};
            qaug = __builtin_bit_cast(bf16x8, w);
        }
        auto read_k1 = [&](int g, int j) {
            if (j == 0) {
                kfr[0] = *(const bf16x8*)(smem + BIAS_OFF + (32 * g + r) * 16);
            } else {
                const char* ks = smem + K_OFF + (g >> 1) * TILE_BYTES + ((g & 1) ? k_rd1 : k_rd0);
                kfr[j] = *(const bf16x8*)(ks + (((2 * (j - 1) + hh) ^ k_sw0) << 4));
            }
        };
        auto max3 = [](float a, float b2, float c3) { return fmaxf(fmaxf(a, b2), c3); };
        float mt = -INFINITY;
        auto group_max = [&](int g) {
            const f32x16& x = sT[g];
            const float a = max3(x[0], x[1], x[2]), b2 = max3(x[3], x[4], x[5]), c2 = max3(x[6], x[7], x[8]);
            const float d2 = max3(x[9], x[10], x[11]), e2 = max3(x[12], x[13], x[14]);
            mt = max3(max3(a, b2, c2), max3(d2, e2, x[15]), mt);
            asm volatile("" : "+v"(mt));                 // (pinned under the next group's MFMAs)
        };
#pragma unroll
        for (int j = 0; j < 5; ++j) read_k1(0, j);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                if (j == 0) {
                    const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    sT[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[0], qaug, z16, 0, 0, 0);
                } else {
                    sT[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[j], qf[j - 1], sT[g], 0, 0, 0);
                }
                if (g + 1 < NG) read_k1(g + 1, j);         // the same fragment of the next group, five MFMAs ahead of its use
                __builtin_amdgcn_sched_barrier(0);
            }
            if (g >= 1) group_max(g - 1);                   // (VALU under the next group's MFMAs)
        }
        // qf has been read for the last time: the next iteration's fragments, then the request for the one after
        if (it + 1 < n_it) finish_q();
        if (it + 2 < n_it) load_q(it + 2);
        group_max(NG - 1);
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
            mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const float nmc = -mt * c;                           // P = 2^(c s' - c m): one FMA per score
        // (every key masked: the reference's softmax over a row of -10000s is uniform; here the bias is finite for every
        //  real key, so mt is finite whenever Lk >= 1)

        // ---- P = 2^(x - m), O^T += V^T P^T, l += sum P, tile by tile (a tile's score registers are dead after its P)
        // (O^T and l start as the FIRST product's zero C operand: they become live when tile 0's 32 score registers have
        // already turned into P -- zeroed up front they sat on top of all 128 scores and the kernel spilled)
        f32x16 oT[2];
        f32x4 lT;
        const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        // Software-pipelined by hand like the first phase: step (t, sp) = 16 keys = one ones-MFMA + two PV MFMAs, with the V^T
        // fragments of the NEXT step requested before them and one quarter of the NEXT tile's exp2 / cvt (8 scores -> one P
        // fragment) behind them.  (Left to hipcc every PV MFMA waited out the LDS latency of the transposed reads issued right
        // in front of it.)
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        auto read_v = [&](int t, int sp, bf16x8 (&dst)[2]) {
            const char* vs = smem + V_OFF + t * TILE_BYTES + v_rd;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const char* base = vs + (2 * sp * 2 + d) * 512;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 2 * 512));
                const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                dst[d] = __builtin_bit_cast(bf16x8, both);
            }
        };
        // P fragment sp' of tile t: scores 8 h2 .. 8 h2 + 7 of key block kb, sp' = 2 kb + h2
        auto make_p = [&](int t, int sp2, bf16x8& dst) {
            const f32x16& x = sT[2 * t + (sp2 >> 1)];
            const f32x2 c2v = {c, c}, m2v = {nmc, nmc};
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                // (two scores per v_pk_fma_f32: the VALU port, not the matrix pipe, paces this phase)
                const f32x2 y = f32x2{x[8 * (sp2 & 1) + e], x[8 * (sp2 & 1) + e + 1]} * c2v + m2v;
                dst[e] = (__bf16)fast_exp2(y[0]);
                dst[e + 1] = (__bf16)fast_exp2(y[1]);
            }
            asm volatile("" : "+v"(dst));                // (pinned: hipcc otherwise gathers the converts where it likes)
        };
        bf16x8 pfr[2][4], vfr[2][2];
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) make_p(0, sp, pfr[0][sp]);
        read_v(0, 0, vfr[0]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
                const int st = 4 * t + sp;                             // step number: V buffers alternate by it
                if (st + 1 < 4 * NT) read_v((st + 1) >> 2, (st + 1) & 3, vfr[(st + 1) & 1]);
                const bf16x8& pf = pfr[t & 1][sp];
                lT = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf, st == 0 ? z4 : lT, 0, 0, 0);
                oT[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[st & 1][0], pf, st == 0 ? z16 : oT[0], 0, 0, 0);
                oT[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[st & 1][1], pf, st == 0 ? z16 : oT[1], 0, 0, 0);
                if (t + 1 < NT) make_p(t + 1, sp, pfr[(t + 1) & 1][sp]);
                __builtin_amdgcn_sched_barrier(0);
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
    const int nt = (p.Lk + cross::KV_TILE - 1) / cross::KV_TILE;
#define LTXMI_XATTN_LAUNCH(N)                                                                                              \
    {                                                                                                                      \
        static unsigned long long lds_done = 0;                                                                            \
        if (const int rc = reserve_lds((const void*)cross::attn_cross_kernel<N>, cross::SMEM, &lds_done,                   \
                                       "ltxmi_attention_fwd_bf16"))                                                        \
            return rc;                                                                                                     \
        hipLaunchKernelGGL(cross::attn_cross_kernel<N>, dim3((unsigned)grid), dim3(256), cross::SMEM, stream, p, groups,   \
                           rows, n_cu);                                                                                    \
    }
    switch (nt) {
        case 1: LTXMI_XATTN_LAUNCH(1) break;
        case 2: LTXMI_XATTN_LAUNCH(2) break;
        case 3: LTXMI_XATTN_LAUNCH(3) break;
        default: LTXMI_XATTN_LAUNCH(4) break;
    }
#undef LTXMI_XATTN_LAUNCH
    return check_launch("ltxmi_attention_fwd_bf16");
}

}  // namespace ltxmi
