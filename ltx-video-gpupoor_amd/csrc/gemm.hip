// gemm.hip -- bf16 "NT" GEMM with fused epilogues for the DiT linears.
//
//   C[M,N] = epi( A[M,K] . W[N,K]^T + bias[N] )
//
// Both operands are K-contiguous (activations [tokens, K]; nn.Linear weights [out, in]),
// which is exactly the per-lane fragment order of the gfx950 bf16 MFMA (8 consecutive k
// per lane), so tiles go HBM -> LDS by 16-byte LDS-DMA (global_load_lds_dwordx4) with no
// register staging and LDS -> VGPR by ds_read_b128.
//
// Tiling: BM x BN x 64 block tile, WAVES_M x WAVES_N waves of 64 lanes, each wave owning a
// (BM/WAVES_M) x (BN/WAVES_N) sub-tile built from v_mfma_f32_16x16x32_bf16 (the shape the
// chip clocks highest on, MI355X_MICROARCH "DVFS give-back" item 7).  The MFMA is issued
// with W as the A operand and the activations as the B operand, so a lane's 4 accumulator
// registers are 4 CONSECUTIVE output columns of one output row -> 8-byte row-major stores.
//
// LDS image: rows of 64 bf16 (128 B), 16-byte chunk c of row r stored at slot c ^ (r & 7).
// The LDS-DMA destination is lane-linear, so the XOR is applied to the per-lane SOURCE
// address and again on the read (cdna_hip_programming.md rule 21).  With that swizzle the
// 16-lane ds_read_b128 groups of the 16x16x32 operand map hit 16 distinct 16-byte slots.
//
// Pipeline: two LDS stages, one __syncthreads() per K-tile; the LDS-DMA runs a full K-tile ahead
// and the MFMA fragments are double-buffered in registers (see the main loop).
//
// Block -> tile mapping is XCD-aware: the 8 XCDs each get a contiguous band of M-tiles so
// the W panel and the A rows they share stay in that XCD's private L2.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace ltxmi {

struct GemmParams {
    const uint16_t* A; int64_t lda;
    const uint16_t* W; int64_t ldw;
    const uint16_t* bias; int bias_stride;   // never NULL: a zero page with stride 0 stands in for "no bias"
    uint16_t* C; int64_t ldc;
    int M, N, K;
    const uint16_t* R; int64_t ldr;
    const uint16_t* gate_table;
    const uint16_t* gate_temb;
    int64_t gate_ld;
    int rows_per_group;
    int tiles_m, tiles_n;
    // implicit-GEMM convolution (MODE == 1): A rows are gathered from x [B,T,H,W,Cin]
    int cB, cT, cH, cW, cCin;      // input grid
    int oT, oH, oW;                 // output grid (== input grid unless strided / extended in time)
    int sT, sHW;                    // strides (1 or 2)
    int tzero;                      // time padding with zeros (plain nn.Conv3d) instead of frame replication
    int tpad;            // frames replicated in front (2 causal, 1 otherwise)
    int pad_replicate;   // spatial padding mode
    // depth-to-space epilogue (EPI == EPI_D2S)
    const uint16_t* res; int res_ch;
    // EPI_SUMSQ: fp32 [M, sumsq_ld] partial row sums of squares of the bf16 outputs, one per 64-column block
    float* sumsq; int sumsq_cols; int64_t sumsq_ld;
    // A whose K axis is cut into blocks of a_kblk elements lying a_kblk_stride elements apart (0 = plain): the receive
    // buffer of the Ulysses return all-to-all, [P source ranks][rows][D / P], is consumed in place by to_out
    int a_kblk; int64_t a_kblk_stride;

    __device__ __forceinline__ int64_t a_koff_elems(int kk) const { // element offset of column kk inside a row of A
        if (a_kblk <= 0) return kk;
        const int blk = kk / a_kblk;
        return (int64_t)blk * a_kblk_stride + (kk - blk * a_kblk);
    }
    __device__ __forceinline__ int64_t a_koff(int kt) const { return a_koff_elems(kt * 64); }   // ... of k-tile kt
};

constexpr int EPI_D2S = 4;   // internal: conv + pixel-shuffle(2,2,2) scatter (+ residual)
constexpr int EPI_RESIDUAL = 5;   // internal: GATE_RESIDUAL without a gate (C = R + acc + bias)
constexpr int EPI_SUMSQ = 6;      // internal: plain store + per-(row, 64-column block) sum of squares of the stored bf16
                                  // values for the columns < sumsq_cols (the q part of a fused QKV projection: the
                                  // attention kernel turns them into q's RMSNorm factor, attention.py:1040-1041)

// 64 zero bytes: LDS-DMA source for zero-padded taps
__device__ __attribute__((aligned(16))) uint32_t g_zero_page[16];

constexpr int BK = 64;

// s + the squares of four stored bf16 values, as ONE fixed chain of fused multiply-adds: under -ffast-math hipcc is free to
// associate "s += a*a + b*b + c*c + d*d" differently in every kernel instance, and the row sums of squares (which become
// q's RMSNorm factor) must not depend on which GEMM kernel the dispatcher picked for a given M
__device__ __forceinline__ float sumsq4(float s, u32x2 o) {
    s = __builtin_fmaf(bf_lo(o[0]), bf_lo(o[0]), s);
    s = __builtin_fmaf(bf_hi(o[0]), bf_hi(o[0]), s);
    s = __builtin_fmaf(bf_lo(o[1]), bf_lo(o[1]), s);
    s = __builtin_fmaf(bf_hi(o[1]), bf_hi(o[1]), s);
    return s;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int MODE>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_bf16_nt_kernel(GemmParams p) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MI = WM / 16, NI = WN / 16;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int A_INSTR = (BM / 8) / NW;   // LDS-DMA wave-instructions per wave per stage
    constexpr int B_INSTR = (BN / 8) / NW;
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile rows must split over the waves");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // ---- XCD-aware tile id: blocks b and b+8 share an XCD (round-robin dispatch); give
    // each XCD a contiguous run of tile ids (bijective form for any grid size).
    const int nwg = gridDim.x;
    const int orig = blockIdx.x;
    const int xcd = orig & 7;
    const int q = nwg >> 3, r = nwg & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    // tiles are numbered in bands of GN N-tiles, M-major inside a band: the ~32 tiles an XCD
    // runs concurrently form a (few M) x (GN N) patch that shares A rows AND W rows in L2
    constexpr int GN = 8;
    const int band_sz = p.tiles_m * GN;
    const int band = tile / band_sz, rem = tile % band_sz;
    const int gn = min(GN, p.tiles_n - band * GN);
    const int tm = rem / gn, tn = band * GN + rem % gn;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-lane source rows for the LDS-DMA (clamped: out-of-range rows re-read the
    // last valid row, their results are never stored)
    const int srow = lane >> 3;          // row within an 8-row piece
    const int sslot = lane & 7;          // 16-byte slot within the 128-byte LDS row
    const uint16_t* a_src[A_INSTR];
    const uint16_t* b_src[B_INSTR];
    int cv_t[A_INSTR], cv_y[A_INSTR], cv_x[A_INSTR];   // conv: output position of the lane's rows
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
        const int row = (wave * A_INSTR + j) * 8 + srow;
        int g = m0 + row;
        g = g < p.M ? g : p.M - 1;
        if (MODE == 0) {
            a_src[j] = p.A + (int64_t)g * p.lda + ((sslot ^ (row & 7)) << 3);
        } else {
            cv_x[j] = (g % p.oW) * p.sHW;
            const int r1 = g / p.oW;
            cv_y[j] = (r1 % p.oH) * p.sHW;
            const int r2 = r1 / p.oH;
            cv_t[j] = (r2 % p.oT) * p.sT;
            const int bb = r2 / p.oT;
            a_src[j] = p.A + (int64_t)bb * p.cT * p.cH * p.cW * p.cCin + ((sslot ^ (row & 7)) << 3);
        }
    }
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
        const int row = (wave * B_INSTR + j) * 8 + srow;
        int g = n0 + row;
        g = g < p.N ? g : p.N - 1;
        b_src[j] = p.W + (int64_t)g * p.ldw + ((sslot ^ (row & 7)) << 3);
    }

    // one 1-KB LDS-DMA piece (8 rows x 128 B) of k-tile kt: pieces 0..A_INSTR-1 are A rows, the rest W rows
    auto piece = [&](int buf, int kt, int g) {
        char* sa = smem + buf * STAGE_BYTES;
        char* sb = sa + A_BYTES;
        if (g >= A_INSTR) {
            const int j = g - A_INSTR;
            glds16(b_src[j] + kt * BK, sb + (wave * B_INSTR + j) * 1024);
        } else if (MODE == 0) {
            glds16(a_src[g] + p.a_koff(kt), sa + (wave * A_INSTR + g) * 1024);
        } else {
            // K index = tap * Cin + cin; a 64-wide K-tile never straddles taps (Cin % 64 == 0)
            const int kk = kt * BK;
            const int tap = kk / p.cCin, c0 = kk - tap * p.cCin;
            const int dt = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            const int j = g;
            int tt = cv_t[j] + dt - p.tpad;
            const bool toob = (tt < 0) | (tt >= p.cT);
            tt = tt < 0 ? 0 : (tt >= p.cT ? p.cT - 1 : tt);          // time: replicate (CausalConv3d)
            int yy = cv_y[j] + dy - 1, xx = cv_x[j] + dx - 1;
            const bool oob = (yy < 0) | (yy >= p.cH) | (xx < 0) | (xx >= p.cW);
            yy = yy < 0 ? 0 : (yy >= p.cH ? p.cH - 1 : yy);
            xx = xx < 0 ? 0 : (xx >= p.cW ? p.cW - 1 : xx);
            const uint16_t* src = a_src[j] + ((int64_t)(tt * p.cH + yy) * p.cW + xx) * p.cCin + c0;
            if ((oob && !p.pad_replicate) | (toob && p.tzero)) src = (const uint16_t*)g_zero_page;
            glds16(src, sa + (wave * A_INSTR + j) * 1024);
        }
    };
    auto stage = [&](int buf, int kt) {
#pragma unroll
        for (int g = 0; g < A_INSTR + B_INSTR; ++g) piece(buf, kt, g);
    };

    // ---- fragment read offsets (bytes within a stage's A or B image), k-step 0
    // lane l reads row (l & 15) of a 16-row tile, 16-byte chunk (l >> 4) [+4 for k-step 1]
    const int frow = lane & 15;
    const int fchunk = lane >> 4;
    int a_off[MI], b_off[NI];      // activations (MFMA B operand) / weights (MFMA A operand)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = wm * WM + i * 16 + frow;
        a_off[i] = row * 128 + ((fchunk ^ (row & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = wn * WN + i * 16 + frow;
        b_off[i] = A_BYTES + row * 128 + ((fchunk ^ (row & 7)) << 4);
    }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- main loop.  Two LDS stages, ONE barrier per K-tile, and the fragments double-buffered in
    // registers so that no MFMA ever waits on an LDS read issued after a barrier:
    //   phase A: MFMAs on the k-step-0 fragments (already in registers)  ||  ds_read k-step 1
    //   __syncthreads(): k-step-1 fragments landed, LDS-DMA of tile t+1 landed and visible
    //   phase B: LDS-DMA of tile t+2 into the stage just drained; MFMAs on k-step 1  ||  ds_read
    //            k-step 0 of tile t+1
    const int nk = p.K / BK;
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    bf16x8 af0[MI], bf0[NI], af1[MI], bf1[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) af0[i] = *(const bf16x8*)(smem + a_off[i]);
#pragma unroll
    for (int j = 0; j < NI; ++j) bf0[j] = *(const bf16x8*)(smem + b_off[j]);

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const char* s = smem + cur * STAGE_BYTES;
        // ---- phase A: the first MFMA row group issues before the k-step-1 reads, so the only LDS
        // wait in front of an MFMA is for fragments fetched half a K-tile ago
        constexpr int MI_HEAD = MI > 2 ? 2 : 1;
#pragma unroll
        for (int i = 0; i < MI_HEAD; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf0[j], af0[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MI; ++i) af1[i] = *(const bf16x8*)(s + (a_off[i] ^ 64));
#pragma unroll
        for (int j = 0; j < NI; ++j) bf1[j] = *(const bf16x8*)(s + (b_off[j] ^ 64));
#pragma unroll
        for (int i = MI_HEAD; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf0[j], af0[i], acc[i][j], 0, 0, 0);
        // Keep phase A's MFMAs on this side of the barrier (hipcc otherwise sinks the register-only
        // MFMAs below it, which puts the ds_read wait back in front of them), and wait for the
        // LDS-DMA explicitly: across the loop back-edge hipcc's own __syncthreads() lowering was
        // observed to wait for lgkmcnt only, not for the pending global_load_lds.
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase B.  The LDS-DMA pieces of k-tile kt+2 go out spread over the MFMA rows instead of
        // as a burst (a wave back-pressured at its VMEM instruction issues no MFMA; with the conv gather
        // the address arithmetic of a piece also hides under the previous row's MFMAs)
        const bool refill = kt + 2 < nk;
        if (kt + 1 < nk) {
            const char* sn = smem + (cur ^ 1) * STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < MI; ++i) af0[i] = *(const bf16x8*)(sn + a_off[i]);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf0[j] = *(const bf16x8*)(sn + b_off[j]);
        }
        constexpr int PPR = (A_INSTR + B_INSTR + MI - 1) / MI;        // pieces per MFMA row
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (refill) {
#pragma unroll
                for (int q = 0; q < PPR; ++q)
                    if (i * PPR + q < A_INSTR + B_INSTR) piece(cur, kt + 2, i * PPR + q);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf1[j], af1[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue.  acc[i][j][e]: row m = m0 + wm*WM + i*16 + (lane&15),
    //                               col n = n0 + wn*WN + j*16 + (lane>>4)*4 + e
    // Every load below is unconditional (masked lanes read a clamped address): a runtime-conditional
    // load makes hipcc branch around it and wait vmcnt(0) per element -- 32 dependent L2 round trips.
    const int erow = lane & 15;
    const int ecol = (lane >> 4) * 4;
    constexpr bool GATED = (EPI == LTXMI_EPI_GATE_RESIDUAL);
    constexpr bool RESID = (EPI == LTXMI_EPI_GATE_RESIDUAL || EPI == EPI_RESIDUAL);
    u32x2 bias_v[NI], gt_v[NI];
    int ncl[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * WN + j * 16 + ecol;
        ncl[j] = n < p.N ? n : p.N - 4;
        bias_v[j] = *(const u32x2*)(p.bias + ncl[j] * p.bias_stride);
        if (GATED) gt_v[j] = *(const u32x2*)(p.gate_table + ncl[j]);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * WM + i * 16 + erow;
        const bool m_ok = m < p.M;
        const int mc = m_ok ? m : p.M - 1;
        const uint16_t* gate_row = GATED ? p.gate_temb + (int64_t)(mc / p.rows_per_group) * p.gate_ld : nullptr;
        float ss[NI / 4 > 0 ? NI / 4 : 1];                       // EPI_SUMSQ: one partial per 64 columns (4 fragments)
#pragma unroll
        for (int q = 0; q < (NI / 4 > 0 ? NI / 4 : 1); ++q) ss[q] = 0.f;
        u32x2 ge_v[NI], rr_v[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            if (GATED) ge_v[j] = *(const u32x2*)(gate_row + ncl[j]);
            if (RESID) rr_v[j] = *(const u32x2*)(p.R + (int64_t)mc * p.ldr + ncl[j]);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wn * WN + j * 16 + ecol;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            v[0] += bf_lo(bias_v[j][0]); v[1] += bf_hi(bias_v[j][0]);
            v[2] += bf_lo(bias_v[j][1]); v[3] += bf_hi(bias_v[j][1]);
            if (EPI == LTXMI_EPI_GELU_TANH) {
                // the packed form of the persistent kernel: every GEMM kernel produces the same bits for a given row
                // (the accumulation order over K is the same in all of them), whatever M made the dispatcher choose
                gelu_tanh_pk(v[0], v[1]);
                gelu_tanh_pk(v[2], v[3]);
            } else if (EPI == LTXMI_EPI_SILU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
            }
            if (GATED) {
                v[0] *= bf_lo(gt_v[j][0]) + bf_lo(ge_v[j][0]);
                v[1] *= bf_hi(gt_v[j][0]) + bf_hi(ge_v[j][0]);
                v[2] *= bf_lo(gt_v[j][1]) + bf_lo(ge_v[j][1]);
                v[3] *= bf_hi(gt_v[j][1]) + bf_hi(ge_v[j][1]);
            }
            if (RESID) {
                v[0] += bf_lo(rr_v[j][0]); v[1] += bf_hi(rr_v[j][0]);
                v[2] += bf_lo(rr_v[j][1]); v[3] += bf_hi(rr_v[j][1]);
            }
            if (!(m_ok && n < p.N)) continue;
            u32x2 o;
            if (EPI == EPI_D2S) {
                // weight rows are packed (p1 p2 p3)-major: n = pp * C' + c'
                const int Cp = p.N >> 3;
                const int pp = n / Cp, cp = n - pp * Cp;
                const int x_ = m % p.cW, r1 = m / p.cW, y_ = r1 % p.cH, r2 = r1 / p.cH;
                const int t_ = r2 % p.cT, b_ = r2 / p.cT;
                const int to = 2 * t_ + (pp >> 2) - 1;
                if (to < 0) continue;                       // first upsampled frame is dropped
                const int yo = 2 * y_ + ((pp >> 1) & 1), xo = 2 * x_ + (pp & 1);
                if (p.res) {
                    // x_in = repeat(pixel_shuffle(x)): channel c' <- x[(c' mod (Cres/8)) * 8 + pp]
                    const uint16_t* rrow = p.res + (int64_t)m * p.res_ch + pp;
                    const int cm = p.res_ch >> 3;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += bf2f(rrow[((cp + e) % cm) * 8]);
                }
                o[0] = pack_bf16(v[0], v[1]);
                o[1] = pack_bf16(v[2], v[3]);
                const int64_t opos = (((int64_t)b_ * (2 * p.cT - 1) + to) * (2 * p.cH) + yo) * (2 * p.cW) + xo;
                *(u32x2*)(p.C + opos * Cp + cp) = o;
                continue;
            }
            o[0] = pack_bf16(v[0], v[1]);
            o[1] = pack_bf16(v[2], v[3]);
            *(u32x2*)(p.C + (int64_t)m * p.ldc + n) = o;
            if (EPI == EPI_SUMSQ)
                ss[j / 4] = sumsq4(ss[j / 4], o);
        }
        if (EPI == EPI_SUMSQ) {
            static_assert(EPI != EPI_SUMSQ || NI % 4 == 0, "sum-of-squares partials are per 64 columns");
#pragma unroll
            for (int q = 0; q < NI / 4; ++q) {
                float s = ss[q];
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                const int nb = n0 + wn * WN + q * 64;            // first column of this 64-column block
                if (lane < 16 && m_ok && nb < p.sumsq_cols) p.sumsq[(int64_t)m * p.sumsq_ld + (nb >> 6)] = s;
            }
        }
    }
}

// =====================================================================================
// Persistent variant for the dense (MODE 0) 256x256 tile: one workgroup per CU walks its list of
// output tiles and treats (tile, k-tile) as ONE stream, so the LDS-DMA pipeline never drains:
// the loads of the next tile's first two K-tiles are in flight while the current tile's
// accumulators are converted and stored, the stores drain under the next tile's MFMAs, and there
// is no per-tile workgroup dispatch.  Per-tile fixed cost measured before this: ~13-15 us per
// round against ~48 us of main loop at K = 2048.
//   * epilogue stores are raw buffer stores, unconditionally issued (out-of-range rows/columns get
//     an out-of-range offset and are dropped by the hardware bounds check), so their COUNT per wave
//     is the compile-time constant MI*NI: the first barrier after an epilogue waits with
//     s_waitcnt vmcnt(MI*NI), i.e. for the older LDS-DMA only, not for the stores;
// =====================================================================================
// Diagnostic build only (-DLTXMI_GEMM_STAMPS, tools/gemm_stamps.py): s_memtime stamps around the sections of a K-tile,
// summed per wave into a debug buffer.  No stamp executes in the product build.
#ifdef LTXMI_GEMM_STAMPS
#define GSTAMP(i)                                                                              \
    do {                                                                                       \
        unsigned long long t_;                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        gst_acc[i] += t_ - gst_prev;                                                           \
        gst_prev = t_;                                                                         \
    } while (0)
__device__ unsigned long long* g_gemm_stamps = nullptr;
#else
#define GSTAMP(i) do { } while (0)
#endif

#ifndef LTXMI_GEMM_DMA_WAVES
#define LTXMI_GEMM_DMA_WAVES 4      // waves of the workgroup that issue the LDS-DMA: the older wave of each SIMD pair
                                    // wins MFMA-issue arbitration and otherwise idles ~900 cycles at the barrier (8 = all: -0.7 %)
#endif
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_bf16_nt_persistent_kernel(GemmParams p) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MI = WM / 16, NI = WN / 16;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int A_INSTR = (BM / 8) / NW;
    constexpr int B_INSTR = (BN / 8) / NW;
    constexpr int NH = WN / 64;             // 64-column halves of a wave's sub-tile (epilogue works on 32 x 64 chunks)
    constexpr int NE = 4;                   // fragments (16 columns each) per half
    constexpr int N_STORES = NH * ((MI / 2) * 4 + (EPI == EPI_SUMSQ ? MI : 0));   // buffer stores per wave and tile
    static_assert(MI % 2 == 0 && WN % 64 == 0, "epilogue scratch is a 32 x 64 bf16 chunk per wave");
    static_assert(N_STORES <= 63, "vmcnt immediate is 6 bits");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // ---- this workgroup's tiles: the tile ids are cut into 8 contiguous ranges (one per XCD,
    // blocks b and b+8 share an XCD) and the workgroups of an XCD stride through its range, so
    // the tiles running concurrently on an XCD are neighbours in the band order
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nwg = gridDim.x, wg = blockIdx.x;
    const int xcd = wg & 7, lw = wg >> 3;
    const int tq = ntiles >> 3, tr = ntiles & 7;
    const int x0 = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int xcnt = tq + (xcd < tr ? 1 : 0);
    const int nw_x = (nwg >> 3) + (xcd < (nwg & 7) ? 1 : 0);
    const int my_n = lw < xcnt ? (xcnt - lw + nw_x - 1) / nw_x : 0;
    if (my_n == 0) return;
    constexpr int GN = 8;
    auto tile_origin = [&](int i, int& m0, int& n0) __attribute__((always_inline)) {
        const int tile = x0 + lw + i * nw_x;
        const int band_sz = p.tiles_m * GN;
        const int band = tile / band_sz, rem = tile % band_sz;
        const int gn = min(GN, p.tiles_n - band * GN);
        m0 = (rem / gn) * BM;
        n0 = (band * GN + rem % gn) * BN;
    };

    // ---- LDS-DMA sources.  A tile's A / W panels are addressed through per-tile BUFFER DESCRIPTORS
    // (base = the tile's first row, num_records = its valid rows), so everything that changes from tile
    // to tile is scalar (SGPRs) and the per-lane byte offsets (row * ld + swizzled 16-byte slot) are the
    // same for every tile; rows past M / N are out of range for the descriptor and read as zeros.
    const int srow = lane >> 3, sslot = lane & 7;
    // Piece q of an operand is rows 8q .. 8q+7 of the tile: its per-lane offset is the offset of piece 0
    // (row srow, 16-byte slot sslot ^ srow) plus q * 8 rows -- one VGPR per operand instead of one per piece.
    // DMA_WAVES waves issue the pieces (PPW per operand each).
    constexpr int DMA_WAVES = LTXMI_GEMM_DMA_WAVES < NW ? LTXMI_GEMM_DMA_WAVES : NW;
    constexpr int PPW_A = (BM / 8) / DMA_WAVES, PPW_B = (BN / 8) / DMA_WAVES;     // pieces per DMA wave
    const uint32_t aoff0 = (uint32_t)(srow * (int)p.lda * 2 + ((sslot ^ srow) << 4));
    const uint32_t boff0 = (uint32_t)(srow * (int)p.ldw * 2 + ((sslot ^ srow) << 4));
    const int a_step = 16 * (int)p.lda, b_step = 16 * (int)p.ldw;                  // bytes per 8 rows
    const bool dma_wave = wave < DMA_WAVES;
    // (descriptors are kept in plain locals: the resource type cannot be a struct member in the host pass)
    using rsrc_t = __amdgpu_buffer_rsrc_t;
    // (the resource type can be neither a struct member, nor bound to a reference, nor captured by a
    // lambda in the host pass: descriptors are plain locals passed BY VALUE)
    auto rsrc_a = [&](int m0) __attribute__((always_inline)) {
        // (K-blocked A: the descriptor spans all blocks; rows past M then read other rows' data instead of zeros --
        // their results are never stored)
        const int64_t span = p.a_kblk > 0 ? (int64_t)(p.K / p.a_kblk - 1) * p.a_kblk_stride + p.a_kblk : p.K;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)m0 * p.lda), 0,
                                                 (int)(((int64_t)(min(BM, p.M - m0) - 1) * p.lda + span) * 2), 0x00020000);
    };
    auto rsrc_w = [&](int n0) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0 * p.ldw), 0,
                                                 (int)(((int64_t)(min(BN, p.N - n0) - 1) * p.ldw + p.K) * 2), 0x00020000);
    };
    // one 1-KB piece (8 rows x 128 B) of k-tile kt: this wave's pieces 0..PPW_A-1 are A, the rest W
    // (akb: byte offset of k-tile kt inside a row of A, computed once per k-tile -- a division when A is K-blocked)
    auto piece = [&](int buf, rsrc_t ta, rsrc_t tw, int kt, int akb, int g) __attribute__((always_inline)) {
        char* sa = smem + buf * STAGE_BYTES;
        if (g < PPW_A) {
            const int q = wave * PPW_A + g;
            blds16(ta, sa + q * 1024, aoff0 + (uint32_t)(q * a_step), akb);
        } else {
            const int q = wave * PPW_B + (g - PPW_A);
            blds16(tw, sa + A_BYTES + q * 1024, boff0 + (uint32_t)(q * b_step), kt * (BK * 2));
        }
    };
    auto stage = [&](int buf, rsrc_t ta, rsrc_t tw, int kt) __attribute__((always_inline)) {
        if (dma_wave) {
            const int akb = (int)(p.a_koff(kt) * 2);
#pragma unroll
            for (int g = 0; g < PPW_A + PPW_B; ++g) piece(buf, ta, tw, kt, akb, g);
        }
    };

    // fragment (i) of a wave sits 16 rows = 2048 bytes below fragment (i-1) and has the same
    // (row & 7), so one base offset per operand and k-step plus immediates addresses all of them
    const int frow = lane & 15, fchunk = lane >> 4;
    const int a_off0 = (wm * WM + frow) * 128 + ((fchunk ^ (frow & 7)) << 4);
    const int b_off0 = A_BYTES + (wn * WN + frow) * 128 + ((fchunk ^ (frow & 7)) << 4);
    const int a_off1 = a_off0 ^ 64, b_off1 = b_off0 ^ 64;

    // output / residual through buffer descriptors (hardware bounds check drops masked stores)
    const uint32_t c_bytes = (uint32_t)(((int64_t)(p.M - 1) * p.ldc + p.N) * 2);
    const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, c_bytes, 0x00020000);
    const int erow = lane & 15, ecol = (lane >> 4) * 4;
    // EPI_SUMSQ: the partial sums leave through a descriptor too (their count per tile must be a constant)
    const __amdgpu_buffer_rsrc_t ss_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.sumsq, 0, EPI == EPI_SUMSQ ? (uint32_t)(((int64_t)(p.M - 1) * p.sumsq_ld + (p.sumsq_cols >> 6)) * 4) : 0u, 0x00020000);

    f32x4 acc[MI][NI];
    bf16x8 af0[MI], bf0[NI], af1[MI], bf1[NI];

    constexpr bool GATED = (EPI == LTXMI_EPI_GATE_RESIDUAL);
    constexpr bool RESID = (EPI == LTXMI_EPI_GATE_RESIDUAL || EPI == EPI_RESIDUAL);
    // Epilogue.  All loads are unconditional and issued in batches (see the note in the kernel
    // above).  The stores go through a 4 KB per-wave LDS scratch (the 32 KB left beside the two
    // 64 KB stages): a lane's accumulator fragment is 4 columns of one row, so storing from
    // registers means 8-byte pieces of 16 different 128-byte lines per instruction; transposed
    // through LDS every store instruction writes 8 whole 128-byte lines, 16 bytes per lane, and the
    // instruction count halves (the store tail is issue-bound: cdna guide T21).
    char* scr = smem + 2 * STAGE_BYTES + wave * 4096;
    // The residual comes in ROW-MAJOR, 16 bytes per lane through a buffer descriptor (8 whole 128-byte lines per instruction,
    // 32-bit offsets), one 32 x 64 chunk ahead of its use, and is turned into the accumulator layout through the wave's LDS
    // scratch -- the store path backwards.  (Requested per 16-row block in the accumulator layout -- 8-byte pieces, every line
    // touched by four instructions, 64-bit address arithmetic per piece -- it was 2 % slower on the K = 2048 launches.  The
    // loads stay the expensive part of this epilogue: stamps, 20.8 k cycles per tile against 7.2 k without them; hoisting all
    // of them, a second chunk in flight, issuing them ahead of the next tile's LDS-DMA pieces: measured, none helped --
    // profiles/r03_gemm_residual_rowmajor.log.)
    static_assert(!RESID || NH == 1, "the residual prefetch numbers its chunks within one 64-column half");
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.R, 0, RESID ? (uint32_t)(((int64_t)(p.M - 1) * p.ldr + p.N) * 2) : 0u, 0x00020000);
    u32x4 rres[4];
    auto load_res = [&](int m0, int n0, int c) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // (rows >= M / columns >= N: outside the descriptor, read as zeros; their results are never stored)
            const int m = m0 + wm * WM + c * 32 + t * 8 + (lane >> 3);
            const int n = n0 + wn * WN + (lane & 7) * 8;
            const uint32_t off = (m < p.M && n < p.N) ? (uint32_t)(((int64_t)m * p.ldr + n) * 2) : 0xfffffff0u;
            rres[t] = __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, off, 0, 0);
        }
    };
    auto epilogue = [&](int m0, int n0) __attribute__((always_inline)) {
#pragma unroll
      for (int jh = 0; jh < NH; ++jh) {                                   // one 64-column half of the sub-tile at a time
        const int nh0 = n0 + wn * WN + jh * 64;
        u32x2 bias_v[NE], gt_v[NE];
        int ncl[NE];
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int n = nh0 + j * 16 + ecol;
            ncl[j] = n < p.N ? n : p.N - 4;
            bias_v[j] = *(const u32x2*)(p.bias + ncl[j] * p.bias_stride);
            if (GATED) gt_v[j] = *(const u32x2*)(p.gate_table + ncl[j]);
        }
        if (RESID) load_res(m0, n0, 0);
#pragma unroll
        for (int c = 0; c < MI / 2; ++c) {
            u32x2 rr_c[2][NE];
            if (RESID) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int row_l = t * 8 + (lane >> 3), chunk = lane & 7;
                    *(u32x4*)(scr + row_l * 128 + ((chunk ^ (row_l & 7)) << 4)) = rres[t];
                }
                if (c + 1 < MI / 2) load_res(m0, n0, c + 1);
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int j = 0; j < NE; ++j) {
                        const int row_l = ii * 16 + erow, chunk = j * 2 + (lane >> 5);
                        rr_c[ii][j] = *(const u32x2*)(scr + row_l * 128 + ((chunk ^ (row_l & 7)) << 4) + ((lane >> 4) & 1) * 8);
                    }
            }
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = 2 * c + ii;
                const int m = m0 + wm * WM + i * 16 + erow;
                const int mc = m < p.M ? m : p.M - 1;                 // clamped row for the reads
                const uint16_t* gate_row = GATED ? p.gate_temb + (int64_t)(mc / p.rows_per_group) * p.gate_ld : nullptr;
                u32x2 ge_v[NE], rr_v[NE];
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    if (GATED) ge_v[j] = *(const u32x2*)(gate_row + ncl[j]);
                    if (RESID) rr_v[j] = rr_c[ii][j];
                }
                const int row_l = ii * 16 + erow;
                float ss = 0.f;
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    const f32x4 a4 = acc[i][jh * NE + j];
                    float v[4] = {a4[0], a4[1], a4[2], a4[3]};
                    v[0] += bf_lo(bias_v[j][0]); v[1] += bf_hi(bias_v[j][0]);
                    v[2] += bf_lo(bias_v[j][1]); v[3] += bf_hi(bias_v[j][1]);
                    if (EPI == LTXMI_EPI_GELU_TANH) {
                        gelu_tanh_pk(v[0], v[1]);
                        gelu_tanh_pk(v[2], v[3]);
                    } else if (EPI == LTXMI_EPI_SILU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                    }
                    if (GATED) {
                        v[0] *= bf_lo(gt_v[j][0]) + bf_lo(ge_v[j][0]);
                        v[1] *= bf_hi(gt_v[j][0]) + bf_hi(ge_v[j][0]);
                        v[2] *= bf_lo(gt_v[j][1]) + bf_lo(ge_v[j][1]);
                        v[3] *= bf_hi(gt_v[j][1]) + bf_hi(ge_v[j][1]);
                    }
                    if (RESID) {
                        v[0] += bf_lo(rr_v[j][0]); v[1] += bf_hi(rr_v[j][0]);
                        v[2] += bf_lo(rr_v[j][1]); v[3] += bf_hi(rr_v[j][1]);
                    }
                    u32x2 o;
                    o[0] = pack_bf16(v[0], v[1]);
                    o[1] = pack_bf16(v[2], v[3]);
                    if (EPI == EPI_SUMSQ)
                        ss = sumsq4(ss, o);
                    // 8-byte piece (j*4 + lane>>4) of scratch row row_l; 16-byte chunks XOR-swizzled by row
                    const int chunk = j * 2 + (lane >> 5);
                    *(u32x2*)(scr + row_l * 128 + ((chunk ^ (row_l & 7)) << 4) + ((lane >> 4) & 1) * 8) = o;
                }
                if (EPI == EPI_SUMSQ) {
                    // these 64 columns of row m: lanes l, l+16, l+32, l+48 hold its four 16-column pieces
                    ss += __shfl_xor(ss, 16, 64);
                    ss += __shfl_xor(ss, 32, 64);
                    const uint32_t off = (lane < 16 && m < p.M && nh0 < p.sumsq_cols)
                                             ? (uint32_t)(((int64_t)m * p.sumsq_ld + (nh0 >> 6)) * 4) : 0xfffffff0u;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ss), ss_rsrc, off, 0, 0);
                }
            }
            // read the 32 x 64 chunk back row-major: lane -> (row l>>3 [+8t], 16-byte chunk l&7)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row_l = t * 8 + (lane >> 3), chunk = lane & 7;
                const u32x4 w = *(const u32x4*)(scr + row_l * 128 + ((chunk ^ (row_l & 7)) << 4));
                const int m = m0 + wm * WM + c * 32 + row_l;
                const int n = nh0 + chunk * 8;
                const uint32_t off = (m < p.M && n < p.N) ? (uint32_t)(((int64_t)m * p.ldc + n) * 2) : 0xfffffff0u;
                __builtin_amdgcn_raw_buffer_store_b128(w, c_rsrc, off, 0, 0);
            }
        }
      }
    };

#ifdef LTXMI_GEMM_STAMPS
    unsigned long long gst_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gst_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(gst_prev)::"memory");
#endif
    const int nk = p.K / BK;                 // >= 2 on this path
    int cs_m0, cs_n0, ns_m0 = 0, ns_n0 = 0;
    tile_origin(0, cs_m0, cs_n0);
    rsrc_t cs_a = rsrc_a(cs_m0), cs_w = rsrc_w(cs_n0);     // current / next tile descriptors (scalars)
    rsrc_t ns_a = cs_a, ns_w = cs_w;
    stage(0, cs_a, cs_w, 0);
    stage(1, cs_a, cs_w, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MI; ++i) af0[i] = *(const bf16x8*)(smem + a_off0 + i * 2048);
#pragma unroll
    for (int j = 0; j < NI; ++j) bf0[j] = *(const bf16x8*)(smem + b_off0 + j * 2048);

    constexpr int MI_HEAD = MI > 2 ? 2 : 1;
    // phase A of one K-tile: MFMAs on the k-step-0 fragments, k-step-1 fragment reads in between.
    // FIRST: first K-tile of an output tile, accumulate onto 0 (fresh accumulator values per tile).
    constexpr int PPW = PPW_A + PPW_B;                          // LDS-DMA pieces of a DMA wave per K-tile
    auto phase_a = [&](const char* s, auto first_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
#pragma unroll
        for (int i = 0; i < MI_HEAD; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    bf0[j], af0[i], FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MI; ++i) af1[i] = *(const bf16x8*)(s + a_off1 + i * 2048);
#pragma unroll
        for (int j = 0; j < NI; ++j) bf1[j] = *(const bf16x8*)(s + b_off1 + j * 2048);
#pragma unroll
        for (int i = MI_HEAD; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    bf0[j], af0[i], FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto read_f0 = [&](const char* sn) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MI; ++i) af0[i] = *(const bf16x8*)(sn + a_off0 + i * 2048);
#pragma unroll
        for (int j = 0; j < NI; ++j) bf0[j] = *(const bf16x8*)(sn + b_off0 + j * 2048);
    };
    auto mfma_f1 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf1[j], af1[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // barrier between the phases: the LDS-DMA of the NEXT k-tile must have landed
    auto sync_all = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    };
    // phase B of a K-tile that is not the tile's last: refill the drained stage with k-tile (+2) of the
    // stream (this tile's, or the next tile's first), fetch the next k-step-0 fragments, MFMAs on k-step 1
    // The 8 LDS-DMA pieces of a wave are NOT issued as a burst: 64 pieces arriving at the CU's address
    // path together back-pressure every wave at its VMEM instruction, and a wave stalled there issues
    // no MFMA.  One piece goes out per MFMA row, so a wave waiting on its piece is covered by its SIMD
    // partner's MFMAs.  READ_F0: also fetch the next k-step-0 fragments (not on a tile's last K-tile:
    // they would have to live through the epilogue).
    // (akb = byte offset of k-tile kts inside a row of A: the caller computes it, and chooses the descriptors, BEFORE
    // phase A -- with one wave per SIMD every scalar instruction between the barrier and the first MFMA of phase B is
    // matrix-pipe idle time)
    auto phase_b_rows = [&](int cur, rsrc_t ta, rsrc_t tw, int kts, int akb, auto dma_tag, auto f0_tag) __attribute__((always_inline)) {
        constexpr bool READ_F0 = decltype(f0_tag)::value;
        constexpr bool DMA = decltype(dma_tag)::value;
        constexpr int NP = PPW;
        const char* sn = smem + (cur ^ 1) * STAGE_BYTES;
#pragma unroll
        for (int g = 0; g < MI; ++g) {
            if (DMA) {
                // row g's share: pieces [g NP / MI, (g + 1) NP / MI).  Issued unconditionally: when there is nothing left
                // to stage (the last tile's last two K-tiles) the descriptors are empty and the pieces arrive as zeros in a
                // stage nobody reads -- no branch inside the K loop, and a constant count for the counted waits.
#pragma unroll
                for (int q = (g * NP) / MI; q < ((g + 1) * NP) / MI; ++q) piece(cur, ta, tw, kts, akb, q);
            }
            if (READ_F0) {
                // A rows first (two per MFMA row), then the W fragments
                if (2 * g < MI) {
                    af0[2 * g] = *(const bf16x8*)(sn + a_off0 + (2 * g) * 2048);
                    af0[2 * g + 1] = *(const bf16x8*)(sn + a_off0 + (2 * g + 1) * 2048);
                } else if (2 * (g - MI / 2) < NI) {
                    const int j = 2 * (g - MI / 2);
                    bf0[j] = *(const bf16x8*)(sn + b_off0 + j * 2048);
                    bf0[j + 1] = *(const bf16x8*)(sn + b_off0 + (j + 1) * 2048);
                }
            }
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf1[j], af1[g], acc[g][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using first_t = std::integral_constant<bool, true>;
    using next_t = std::integral_constant<bool, false>;

    // The tile walk exists twice, once per role: the waves that issue the LDS-DMA (0 .. DMA_WAVES-1) and the waves that only
    // compute.  With the role a run-time condition every MFMA row of phase B carried a wave-uniform branch (8 per K-tile).
    const rsrc_t null_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0, 0x00020000);
    auto walk = [&](auto dma_tag) __attribute__((always_inline)) {
    using dma_t = decltype(dma_tag);
    int flat = 0;                            // running k-tile count: stage parity
    for (int ti = 0; ti < my_n; ++ti) {
        const bool has_next = ti + 1 < my_n;
        if (has_next) {
            tile_origin(ti + 1, ns_m0, ns_n0);
            ns_a = rsrc_a(ns_m0);
            ns_w = rsrc_w(ns_n0);
        } else {
            ns_a = null_rsrc;                // nothing follows: what phase B still "stages" reads as zeros
            ns_w = null_rsrc;
        }
        // What phase B of K-tile kt stages: k-tile (+2) of the (tile, k-tile) stream -- this tile's, or one of the next
        // tile's first two (its descriptors are scalars).  Chosen before phase A (see phase_b_rows).
#define LTXMI_GEMM_PICK_STAGE(KT)                                                         \
        const bool same_tile = (KT) + 2 < nk;                                              \
        const int kts = same_tile ? (KT) + 2 : (KT) + 2 - nk;                              \
        const rsrc_t st_a = same_tile ? cs_a : ns_a, st_w = same_tile ? cs_w : ns_w;      \
        const int akb = (int)(p.a_koff(kts) * 2);                                          \
        __builtin_amdgcn_sched_barrier(0);
        // ---- K-tile 0 (never the last: nk >= 2)
        {
            const int cur = flat & 1;
            LTXMI_GEMM_PICK_STAGE(0)
            phase_a(smem + cur * STAGE_BYTES, first_t{});
            // right after an epilogue the N_STORES younger buffer stores may still be in flight:
            // wait for everything older than them (the LDS-DMA) only
            if (ti > 0) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_STORES) : "memory");
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            } else {
                sync_all();
            }
            phase_b_rows(cur, st_a, st_w, kts, akb, dma_t{}, std::integral_constant<bool, true>{});
            ++flat;
        }
        // ---- middle K-tiles
        for (int kt = 1; kt < nk - 1; ++kt) {
            const int cur = flat & 1;
            LTXMI_GEMM_PICK_STAGE(kt)
            GSTAMP(4);                                                   // (scalar set-up + loop overhead + first/last K-tiles)
            phase_a(smem + cur * STAGE_BYTES, next_t{});
            GSTAMP(0);
#ifdef LTXMI_GEMM_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GSTAMP(1);
#endif
            sync_all();
            GSTAMP(2);
            phase_b_rows(cur, st_a, st_w, kts, akb, dma_t{}, std::integral_constant<bool, true>{});
            GSTAMP(3);
#ifdef LTXMI_GEMM_STAMPS
            gst_acc[5] += 1;
#endif
            ++flat;
        }
#undef LTXMI_GEMM_PICK_STAGE
        // ---- last K-tile, then the epilogue under the next tile's loads
        {
            const int cur = flat & 1;
            const int akb1 = (int)(p.a_koff(1) * 2);
            __builtin_amdgcn_sched_barrier(0);
            phase_a(smem + cur * STAGE_BYTES, next_t{});
            sync_all();
            // next tile's K-tile 1 goes into the stage this K-tile just drained, piece by piece under the
            // MFMAs; all of it is issued BEFORE the stores, so vmcnt(N_STORES) at the next barrier covers it
            phase_b_rows(cur, ns_a, ns_w, 1, akb1, dma_t{}, std::integral_constant<bool, false>{});
            GSTAMP(4);
            epilogue(cs_m0, cs_n0);
            GSTAMP(6);
#ifdef LTXMI_GEMM_STAMPS
            gst_acc[7] += 1;
#endif
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) {
                cs_a = ns_a; cs_w = ns_w; cs_m0 = ns_m0; cs_n0 = ns_n0;
                // k-step-0 fragments of the next tile are fetched only now: holding them across the
                // epilogue would not fit the register file next to the 128 accumulators
                read_f0(smem + (cur ^ 1) * STAGE_BYTES);
            }
            ++flat;
        }
    }
    };
    if (dma_wave) walk(std::integral_constant<bool, true>{});
    else walk(std::integral_constant<bool, false>{});
#ifdef LTXMI_GEMM_STAMPS
    if (g_gemm_stamps && lane == 0 && blockIdx.x < 256) {
        // [0] phase A, [1] vmcnt wait, [2] barrier, [3] phase B (middle K-tiles); [4] everything else in the K loops;
        // [5] middle K-tiles counted; [6] epilogues; [7] tiles
        unsigned long long* o = g_gemm_stamps + (blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 8; ++i) o[i] = gst_acc[i];
    }
#endif
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int MODE>
static int launch_tile(const GemmParams& p0, int epi, hipStream_t stream, const char* what) {
    GemmParams p = p0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const int grid = p.tiles_m * p.tiles_n;
    constexpr int threads = WAVES_M * WAVES_N * 64;
    constexpr int smem = 2 * (BM + BN) * BK * 2;
#define LTXMI_GEMM_LAUNCH(E)                                                                          \
    {                                                                                                 \
        auto kern = gemm_bf16_nt_kernel<BM, BN, WAVES_M, WAVES_N, E, MODE>;                                \
        static unsigned long long lds_done = 0;                                                       \
        if (const int rc_ = reserve_lds((const void*)kern, smem, &lds_done, what)) return rc_;        \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), smem, stream, p);                        \
    }
    if constexpr (MODE == 0) {
        switch (epi) {
            case LTXMI_EPI_NONE: LTXMI_GEMM_LAUNCH(LTXMI_EPI_NONE) break;
            case EPI_SUMSQ: LTXMI_GEMM_LAUNCH(EPI_SUMSQ) break;
            case LTXMI_EPI_GELU_TANH: LTXMI_GEMM_LAUNCH(LTXMI_EPI_GELU_TANH) break;
            case LTXMI_EPI_SILU: LTXMI_GEMM_LAUNCH(LTXMI_EPI_SILU) break;
            case LTXMI_EPI_GATE_RESIDUAL: LTXMI_GEMM_LAUNCH(LTXMI_EPI_GATE_RESIDUAL) break;
            case EPI_RESIDUAL: LTXMI_GEMM_LAUNCH(EPI_RESIDUAL) break;
            default: set_error("%s: bad epilogue %d", what, epi); return LTXMI_ERR_INVALID_ARG;
        }
    } else {
        if (epi == EPI_D2S) LTXMI_GEMM_LAUNCH(EPI_D2S)
        else if (epi == EPI_RESIDUAL) LTXMI_GEMM_LAUNCH(EPI_RESIDUAL)
        else LTXMI_GEMM_LAUNCH(LTXMI_EPI_NONE)
    }
#undef LTXMI_GEMM_LAUNCH
    return check_launch(what);
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_persistent(const GemmParams& p0, int epi, hipStream_t stream, const char* what) {
    GemmParams p = p0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int n_cu = device_cu_count(what);
    if (n_cu <= 0) return LTXMI_ERR_LAUNCH;
    const int grid = ntiles < n_cu ? ntiles : n_cu;      // one 8-wave workgroup per CU (all 160 KB of LDS)
    constexpr int threads = WAVES_M * WAVES_N * 64;
    constexpr int smem = 2 * (BM + BN) * BK * 2 + WAVES_M * WAVES_N * 4096;   // 2 stages + epilogue scratch
#define LTXMI_GEMM_LAUNCH_P(E)                                                                        \
    {                                                                                                 \
        auto kern = gemm_bf16_nt_persistent_kernel<BM, BN, WAVES_M, WAVES_N, E>;                      \
        static unsigned long long lds_done = 0;                                                       \
        if (const int rc_ = reserve_lds((const void*)kern, smem, &lds_done, what)) return rc_;        \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), smem, stream, p);                        \
    }
    switch (epi) {
        // the plain epilogue runs on the row-sums instance with the sums switched off (sumsq_cols = 0: its extra stores
        // carry an out-of-range offset and are dropped): same bits, and measured 3-6 % faster than a dedicated instance,
        // which hipcc happens to allocate worst of all (55 spilled VGPRs)
        case LTXMI_EPI_NONE:
        case EPI_SUMSQ: LTXMI_GEMM_LAUNCH_P(EPI_SUMSQ) break;
        case LTXMI_EPI_GELU_TANH: LTXMI_GEMM_LAUNCH_P(LTXMI_EPI_GELU_TANH) break;
        case LTXMI_EPI_SILU: LTXMI_GEMM_LAUNCH_P(LTXMI_EPI_SILU) break;
        case LTXMI_EPI_GATE_RESIDUAL: LTXMI_GEMM_LAUNCH_P(LTXMI_EPI_GATE_RESIDUAL) break;
        case EPI_RESIDUAL: LTXMI_GEMM_LAUNCH_P(EPI_RESIDUAL) break;
        default: set_error("%s: bad epilogue %d", what, epi); return LTXMI_ERR_INVALID_ARG;
    }
#undef LTXMI_GEMM_LAUNCH_P
    return check_launch(what);
}

// device address of the zero page of the CURRENT device (stands in for a missing bias with stride 0); a __device__
// symbol has one address per device, so the cache is per device like the other launcher state (common.h)
static const uint16_t* zero_page_ptr() {
    static const uint16_t* ptrs[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    const bool tracked = dev >= 0 && dev < 64;
    if (tracked) {
        const uint16_t* c = __atomic_load_n(&ptrs[dev], __ATOMIC_ACQUIRE);
        if (c) return c;
    }
    void* d = nullptr;
    if (hipGetSymbolAddress(&d, HIP_SYMBOL(g_zero_page)) != hipSuccess) return nullptr;
    if (tracked) __atomic_store_n(&ptrs[dev], (const uint16_t*)d, __ATOMIC_RELEASE);
    return (const uint16_t*)d;
}

}  // namespace ltxmi

using namespace ltxmi;

extern "C" int ltxmi_gemm_bf16(const ltxmi_gemm_args* a, void* stream) {
    LTXMI_REQUIRE(a && a->A && a->W && a->C, LTXMI_ERR_INVALID_ARG, "ltxmi_gemm_bf16: NULL argument");
    LTXMI_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_gemm_bf16: non-positive shape M=%d N=%d K=%d", a->M, a->N, a->K);
    LTXMI_REQUIRE(a->K % 64 == 0, LTXMI_ERR_UNSUPPORTED, "ltxmi_gemm_bf16: K=%d must be a multiple of 64", a->K);
    LTXMI_REQUIRE(a->N % 8 == 0, LTXMI_ERR_UNSUPPORTED, "ltxmi_gemm_bf16: N=%d must be a multiple of 8", a->N);
    LTXMI_REQUIRE(a->lda % 8 == 0 && a->ldw % 8 == 0 && a->ldc % 4 == 0 && (a->lda >= a->K || a->a_kblock > 0) && a->ldw >= a->K &&
                      a->ldc >= a->N,
                  LTXMI_ERR_UNSUPPORTED, "ltxmi_gemm_bf16: bad leading dimensions lda=%lld ldw=%lld ldc=%lld",
                  (long long)a->lda, (long long)a->ldw, (long long)a->ldc);
    LTXMI_REQUIRE((((uintptr_t)a->A | (uintptr_t)a->W) & 15) == 0 && (((uintptr_t)a->C) & 7) == 0 &&
                      (((uintptr_t)a->bias) & 7) == 0,
                  LTXMI_ERR_UNSUPPORTED, "ltxmi_gemm_bf16: pointers must be 16-byte (A, W) / 8-byte (C, bias) aligned");
    if (a->epilogue == LTXMI_EPI_GATE_RESIDUAL) {
        LTXMI_REQUIRE(a->residual && a->ldr >= a->N && a->ldr % 4 == 0, LTXMI_ERR_INVALID_ARG,
                      "ltxmi_gemm_bf16: GATE_RESIDUAL needs a residual with ldr >= N");
        if (a->gate_table)
            LTXMI_REQUIRE(a->gate_temb && a->rows_per_group > 0 && a->gate_ld % 4 == 0, LTXMI_ERR_INVALID_ARG,
                          "ltxmi_gemm_bf16: gate_table given without gate_temb / rows_per_group");
    }
    GemmParams p;
    p.A = (const uint16_t*)a->A; p.lda = a->lda;
    p.W = (const uint16_t*)a->W; p.ldw = a->ldw;
    p.bias = a->bias ? (const uint16_t*)a->bias : zero_page_ptr();
    p.bias_stride = a->bias ? 1 : 0;
    LTXMI_REQUIRE(p.bias, LTXMI_ERR_LAUNCH, "ltxmi_gemm_bf16: cannot resolve the zero page");
    p.C = (uint16_t*)a->C; p.ldc = a->ldc;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.R = (const uint16_t*)a->residual; p.ldr = a->ldr;
    p.gate_table = (const uint16_t*)a->gate_table;
    p.gate_temb = (const uint16_t*)a->gate_temb;
    p.gate_ld = a->gate_ld;
    p.rows_per_group = a->rows_per_group > 0 ? a->rows_per_group : 1;
    p.tiles_m = p.tiles_n = 0;
    p.cB = p.cT = p.cH = p.cW = p.cCin = 1; p.oT = p.oH = p.oW = 1; p.sT = p.sHW = 1; p.tpad = 0; p.tzero = 0; p.pad_replicate = 0; p.res = nullptr; p.res_ch = 0;
    // without a rowsumsq pointer the cols / ld fields are NOT read: the plain epilogue runs on the row-sums instance and
    // must see cols = 0 (its extra stores are then dropped) whatever a caller left in those fields
    p.sumsq = a->rowsumsq;
    p.sumsq_cols = a->rowsumsq ? a->rowsumsq_cols : 0;
    p.sumsq_ld = a->rowsumsq ? a->rowsumsq_ld : 0;
    p.a_kblk = a->a_kblock; p.a_kblk_stride = a->a_kblock_stride;
    if (a->a_kblock) {
        LTXMI_REQUIRE(a->a_kblock > 0 && a->a_kblock % 64 == 0 && a->K % a->a_kblock == 0 && a->a_kblock_stride % 8 == 0 &&
                          a->lda >= a->a_kblock,
                      LTXMI_ERR_INVALID_ARG, "ltxmi_gemm_bf16: a_kblock=%d must be a multiple of 64 that divides K, lda >= a_kblock",
                      a->a_kblock);
        LTXMI_REQUIRE(((int64_t)(a->K / a->a_kblock - 1) * a->a_kblock_stride + (int64_t)256 * a->lda) * 2 < (1ll << 31),
                      LTXMI_ERR_UNSUPPORTED, "ltxmi_gemm_bf16: K-blocked A spans more than 2 GiB per tile");
    }
    if (a->rowsumsq) {
        LTXMI_REQUIRE(a->epilogue == LTXMI_EPI_NONE, LTXMI_ERR_UNSUPPORTED, "ltxmi_gemm_bf16: rowsumsq needs the plain epilogue");
        LTXMI_REQUIRE(a->rowsumsq_cols > 0 && a->rowsumsq_cols % 64 == 0 && a->rowsumsq_cols <= a->N &&
                          a->rowsumsq_ld >= a->rowsumsq_cols / 64 && (((uintptr_t)a->rowsumsq) & 3) == 0,
                      LTXMI_ERR_INVALID_ARG, "ltxmi_gemm_bf16: rowsumsq_cols=%d must be a multiple of 64 within N, ld >= cols/64",
                      a->rowsumsq_cols);
        LTXMI_REQUIRE((int64_t)a->M * a->rowsumsq_ld * 4 < (1ll << 32), LTXMI_ERR_UNSUPPORTED, "ltxmi_gemm_bf16: rowsumsq too large");
    }
    hipStream_t s = (hipStream_t)stream;
    const int epi = a->rowsumsq ? EPI_SUMSQ
                                : ((a->epilogue == LTXMI_EPI_GATE_RESIDUAL && !a->gate_table) ? EPI_RESIDUAL : a->epilogue);
    // Tile choice: 256x256 (8 waves) when it still fills the 256 CUs, else 128x128 (4 waves,
    // 2 blocks/CU); skinny problems (adaLN tables, text K/V) take the 128x128 path too.
    const long t256 = (long)((a->M + 255) / 256) * ((a->N + 255) / 256);
    // a->algo (diagnostics): 0 = by shape, 128 = the 128x128 tile kernel, 256 = the non-persistent 256x256 one
    const int force_tile = a->algo;
    LTXMI_REQUIRE(force_tile == 0 || force_tile == 128 || force_tile == 256, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_gemm_bf16: algo %d not in {0, 128, 256}", force_tile);
    if (force_tile == 128) return launch_tile<128, 128, 2, 2, 0>(p, epi, s, "ltxmi_gemm_bf16");
    constexpr long persist_min = 128;      // measured: 128 > 256 > 384 tiles for M = 4992 .. 9984
    // (M >= 768: the stacked text K/V projection of a forward -- 3 x 256 text rows against every layer's [to_k; to_v])
    if (a->M >= 768 && a->N >= 256 && t256 >= persist_min) {
        const bool fits32 = ((int64_t)a->M * a->ldc * 2 < (1ll << 32)) && ((int64_t)256 * a->lda * 2 < (1ll << 31)) &&
                            ((int64_t)256 * a->ldw * 2 < (1ll << 31));
        // (the persistent kernel reads the residual through a buffer descriptor, in 16-byte pieces)
        const bool res16 = !a->residual || (a->ldr % 8 == 0 && (((uintptr_t)a->residual) & 15) == 0 &&
                                            (int64_t)a->M * a->ldr * 2 < (1ll << 32));
        if (a->K >= 128 && fits32 && res16 && force_tile != 256)
            return launch_persistent<256, 256, 2, 4>(p, epi, s, "ltxmi_gemm_bf16");
        return launch_tile<256, 256, 2, 4, 0>(p, epi, s, "ltxmi_gemm_bf16");
    }
    return launch_tile<128, 128, 2, 2, 0>(p, epi, s, "ltxmi_gemm_bf16");
}

#ifdef LTXMI_GEMM_STAMPS
extern "C" int ltxmi_debug_set_gemm_stamps(void* buf) {
    unsigned long long* b = (unsigned long long*)buf;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(ltxmi::g_gemm_stamps), &b, sizeof(b));
}
#endif

extern "C" int ltxmi_conv3d_fuses_post_norm(const ltxmi_conv3d_args* a) {
    return a != nullptr && a->bias != nullptr && a->B > 0 && a->T > 0 && a->H > 0 && a->W > 0 && a->Cin > 0 && a->Cout > 0 &&
                   conv3d_direct_fuses_post_norm(a) ? 1 : 0;
}

extern "C" int64_t ltxmi_conv3d_workspace_bytes(const ltxmi_conv3d_args* a) {
    return a != nullptr && a->bias != nullptr && a->B > 0 && a->T > 0 && a->H > 0 && a->W > 0 && a->Cin > 0 && a->Cout > 0
               ? conv3d_direct_workspace_bytes(a) : 0;
}

extern "C" int ltxmi_conv3d_ndhwc_bf16(const ltxmi_conv3d_args* a, void* stream) {
    LTXMI_REQUIRE(a && a->x && a->w && a->y, LTXMI_ERR_INVALID_ARG, "ltxmi_conv3d_ndhwc_bf16: NULL argument");
    LTXMI_REQUIRE(a->B > 0 && a->T > 0 && a->H > 0 && a->W > 0 && a->Cin > 0 && a->Cout > 0, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_conv3d_ndhwc_bf16: non-positive shape");
    LTXMI_REQUIRE(a->Cin % 64 == 0, LTXMI_ERR_UNSUPPORTED, "ltxmi_conv3d_ndhwc_bf16: Cin=%d must be a multiple of 64", a->Cin);
    LTXMI_REQUIRE(a->Cout % 8 == 0, LTXMI_ERR_UNSUPPORTED, "ltxmi_conv3d_ndhwc_bf16: Cout=%d must be a multiple of 8", a->Cout);
    const int sT = a->stride_t > 0 ? a->stride_t : 1, sHW = a->stride_hw > 0 ? a->stride_hw : 1;
    LTXMI_REQUIRE((sT == 1 || sT == 2) && (sHW == 1 || sHW == 2), LTXMI_ERR_UNSUPPORTED,
                  "ltxmi_conv3d_ndhwc_bf16: strides must be 1 or 2");
    // output grid: nn.Conv3d arithmetic on the padded input (time padded by tpad frames in front, and by
    // one replicated frame behind when not causal; space padded by 1): floor((L + pad - 3) / s) + 1
    const int kt = a->kernel_t > 0 ? a->kernel_t : 3;       // 1: a 3x3 nn.Conv2d applied to every frame
    LTXMI_REQUIRE(kt == 3 || (kt == 1 && sT == 1 && a->tpad == 0 && a->out_T == 0), LTXMI_ERR_UNSUPPORTED,
                  "ltxmi_conv3d_ndhwc_bf16: kernel_t must be 3, or 1 without time stride/padding");
    const int tpad_front = kt == 1 ? 0 : (a->tpad > 0 ? a->tpad : (a->causal ? 2 : 1));
    const int tpad_back = (kt == 1 || a->tpad > 0 || a->causal) ? 0 : 1;
    const int oT = a->out_T > 0 ? a->out_T : (a->T + tpad_front + tpad_back - kt) / sT + 1;
    const int oH = (a->H + 2 - 3) / sHW + 1, oW = (a->W + 2 - 3) / sHW + 1;
    LTXMI_REQUIRE(!(a->d2s && (sT != 1 || sHW != 1 || oT != a->T)), LTXMI_ERR_UNSUPPORTED,
                  "ltxmi_conv3d_ndhwc_bf16: depth-to-space store needs a stride-1, same-size convolution");
    const int64_t M = (int64_t)a->B * oT * oH * oW;
    LTXMI_REQUIRE(M < (1ll << 31) && (int64_t)a->B * (2 * a->T) * (2 * a->H) * (2 * a->W) < (1ll << 31),
                  LTXMI_ERR_UNSUPPORTED, "ltxmi_conv3d_ndhwc_bf16: too many positions");
    if (a->d2s) {
        LTXMI_REQUIRE(a->Cout % 32 == 0, LTXMI_ERR_UNSUPPORTED,
                      "ltxmi_conv3d_ndhwc_bf16: depth-to-space needs Cout %% 32 == 0 (got %d)", a->Cout);
        if (a->residual)
            LTXMI_REQUIRE(a->res_channels > 0 && a->res_channels % 8 == 0, LTXMI_ERR_INVALID_ARG,
                          "ltxmi_conv3d_ndhwc_bf16: bad residual channel count %d", a->res_channels);
    }
    LTXMI_REQUIRE((((uintptr_t)a->x | (uintptr_t)a->w) & 15) == 0 && (((uintptr_t)a->y | (uintptr_t)a->bias) & 7) == 0,
                  LTXMI_ERR_UNSUPPORTED, "ltxmi_conv3d_ndhwc_bf16: misaligned pointer");
    LTXMI_REQUIRE(a->algo >= 0 && a->algo <= 4, LTXMI_ERR_INVALID_ARG, "ltxmi_conv3d_ndhwc_bf16: algo %d not in {0 .. 4}", a->algo);
    LTXMI_REQUIRE(a->workspace_bytes >= 0 && (a->workspace != nullptr || a->workspace_bytes == 0), LTXMI_ERR_INVALID_ARG,
                  "ltxmi_conv3d_ndhwc_bf16: workspace_bytes without a workspace");
    if (a->post_norm) {
        LTXMI_REQUIRE(a->post_norm == 1 && (a->post_scale != nullptr) == (a->post_shift != nullptr) && a->post_eps >= 0.f,
                      LTXMI_ERR_INVALID_ARG, "ltxmi_conv3d_ndhwc_bf16: post_norm must be 0 or 1, post_scale / post_shift both given or both NULL");
        LTXMI_REQUIRE((((uintptr_t)a->post_scale | (uintptr_t)a->post_shift) & 15) == 0, LTXMI_ERR_UNSUPPORTED,
                      "ltxmi_conv3d_ndhwc_bf16: misaligned post_scale / post_shift");
        LTXMI_REQUIRE(((uintptr_t)a->y_norm & 15) == 0 && a->y_norm != a->y, LTXMI_ERR_INVALID_ARG,
                      "ltxmi_conv3d_ndhwc_bf16: y_norm must be 16-byte aligned and distinct from y");
        LTXMI_REQUIRE(conv3d_direct_fuses_post_norm(a), LTXMI_ERR_UNSUPPORTED,
                      "ltxmi_conv3d_ndhwc_bf16: post_norm is applied by the four-wave direct convolution where a wave holds all "
                      "channels of a position (ask ltxmi_conv3d_fuses_post_norm first)");
    } else {
        LTXMI_REQUIRE(a->y_norm == nullptr, LTXMI_ERR_INVALID_ARG, "ltxmi_conv3d_ndhwc_bf16: y_norm without post_norm");
    }
    if (a->algo != 1) {
        const int rc = launch_conv3d_direct(a, (hipStream_t)stream);      // narrow stride-1 layers: direct convolution
        if (rc >= 0) return rc;
        LTXMI_REQUIRE(a->algo < 2, LTXMI_ERR_UNSUPPORTED,
                      "ltxmi_conv3d_ndhwc_bf16: algo = %d (direct convolution) does not take this shape", a->algo);
    }
    GemmParams p;
    p.A = (const uint16_t*)a->x; p.lda = a->Cin;
    p.W = (const uint16_t*)a->w; p.ldw = 9ll * kt * a->Cin;
    p.bias = a->bias ? (const uint16_t*)a->bias : zero_page_ptr();
    p.bias_stride = a->bias ? 1 : 0;
    LTXMI_REQUIRE(p.bias, LTXMI_ERR_LAUNCH, "ltxmi_conv3d_ndhwc_bf16: cannot resolve the zero page");
    p.C = (uint16_t*)a->y; p.ldc = a->Cout;
    p.M = (int)M; p.N = a->Cout; p.K = 9 * kt * a->Cin;
    p.R = nullptr; p.ldr = 0; p.gate_table = nullptr; p.gate_temb = nullptr; p.gate_ld = 0; p.rows_per_group = 1;
    p.sumsq = nullptr; p.sumsq_cols = 0; p.sumsq_ld = 0; p.a_kblk = 0; p.a_kblk_stride = 0;
    p.tiles_m = p.tiles_n = 0;
    p.cB = a->B; p.cT = a->T; p.cH = a->H; p.cW = a->W; p.cCin = a->Cin;
    p.oT = oT; p.oH = oH; p.oW = oW; p.sT = sT; p.sHW = sHW;
    p.tpad = tpad_front;
    p.tzero = a->time_pad_zeros ? 1 : 0;
    p.pad_replicate = a->pad_replicate;
    p.res = a->d2s ? (const uint16_t*)a->residual : nullptr;
    p.res_ch = a->res_channels;
    hipStream_t s = (hipStream_t)stream;
    LTXMI_REQUIRE(!(a->d2s && a->add), LTXMI_ERR_INVALID_ARG, "ltxmi_conv3d_ndhwc_bf16: `add` is for the plain store only");
    if (a->add) { p.R = (const uint16_t*)a->add; p.ldr = a->Cout; }
    const int epi = a->d2s ? EPI_D2S : (a->add ? EPI_RESIDUAL : LTXMI_EPI_NONE);
    const long t256 = (long)((M + 255) / 256) * ((a->Cout + 255) / 256);
    if (a->Cout >= 256 && t256 >= 384) return launch_tile<256, 256, 2, 4, 1>(p, epi, s, "ltxmi_conv3d_ndhwc_bf16");
    return launch_tile<128, 128, 2, 2, 1>(p, epi, s, "ltxmi_conv3d_ndhwc_bf16");
}
