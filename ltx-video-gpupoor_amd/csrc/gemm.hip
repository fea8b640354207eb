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
#include "common.h"

namespace ltxmi {

struct GemmParams {
    const uint16_t* A; int64_t lda;
    const uint16_t* W; int64_t ldw;
    const uint16_t* bias;
    uint16_t* C; int64_t ldc;
    int M, N, K;
    const uint16_t* R; int64_t ldr;
    const uint16_t* gate_table;
    const uint16_t* gate_temb;
    int64_t gate_ld;
    int rows_per_group;
    int tiles_m, tiles_n;
    // implicit-GEMM convolution (MODE == 1): A rows are gathered from x [B,T,H,W,Cin]
    int cB, cT, cH, cW, cCin;
    int tpad;            // frames replicated in front (2 causal, 1 otherwise)
    int pad_replicate;   // spatial padding mode
    // depth-to-space epilogue (EPI == EPI_D2S)
    const uint16_t* res; int res_ch;
};

constexpr int EPI_D2S = 4;   // internal: conv + pixel-shuffle(2,2,2) scatter (+ residual)

// 64 zero bytes: LDS-DMA source for zero-padded taps
__device__ __attribute__((aligned(16))) uint32_t g_zero_page[16];

constexpr int BK = 64;

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
            cv_x[j] = g % p.cW;
            const int r1 = g / p.cW;
            cv_y[j] = r1 % p.cH;
            const int r2 = r1 / p.cH;
            cv_t[j] = r2 % p.cT;
            const int bb = r2 / p.cT;
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

    auto stage = [&](int buf, int kt) {
        char* sa = smem + buf * STAGE_BYTES;
        char* sb = sa + A_BYTES;
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < A_INSTR; ++j)
                glds16(a_src[j] + kt * BK, sa + (wave * A_INSTR + j) * 1024);
        } else {
            // K index = tap * Cin + cin; a 64-wide K-tile never straddles taps (Cin % 64 == 0)
            const int kk = kt * BK;
            const int tap = kk / p.cCin, c0 = kk - tap * p.cCin;
            const int dt = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
#pragma unroll
            for (int j = 0; j < A_INSTR; ++j) {
                int tt = cv_t[j] + dt - p.tpad;
                tt = tt < 0 ? 0 : (tt >= p.cT ? p.cT - 1 : tt);          // time: always replicate
                int yy = cv_y[j] + dy - 1, xx = cv_x[j] + dx - 1;
                const bool oob = (yy < 0) | (yy >= p.cH) | (xx < 0) | (xx >= p.cW);
                yy = yy < 0 ? 0 : (yy >= p.cH ? p.cH - 1 : yy);
                xx = xx < 0 ? 0 : (xx >= p.cW ? p.cW - 1 : xx);
                const uint16_t* src = a_src[j] + ((int64_t)(tt * p.cH + yy) * p.cW + xx) * p.cCin + c0;
                if (oob && !p.pad_replicate) src = (const uint16_t*)g_zero_page;
                glds16(src, sa + (wave * A_INSTR + j) * 1024);
            }
        }
#pragma unroll
        for (int j = 0; j < B_INSTR; ++j)
            glds16(b_src[j] + kt * BK, sb + (wave * B_INSTR + j) * 1024);
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
        // ---- phase B
        if (kt + 2 < nk) stage(cur, kt + 2);
        if (kt + 1 < nk) {
            const char* sn = smem + (cur ^ 1) * STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < MI; ++i) af0[i] = *(const bf16x8*)(sn + a_off[i]);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf0[j] = *(const bf16x8*)(sn + b_off[j]);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf1[j], af1[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue.  acc[i][j][e]: row m = m0 + wm*WM + i*16 + (lane&15),
    //                               col n = n0 + wn*WN + j*16 + (lane>>4)*4 + e
    const int erow = lane & 15;
    const int ecol = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * WM + i * 16 + erow;
        if (m >= p.M) continue;
        const uint16_t* gate_row = nullptr;
        if (EPI == LTXMI_EPI_GATE_RESIDUAL && p.gate_table)
            gate_row = p.gate_temb + (int64_t)(m / p.rows_per_group) * p.gate_ld;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wn * WN + j * 16 + ecol;
            if (n >= p.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.bias) {
                const u32x2 b = *(const u32x2*)(p.bias + n);
                v[0] += bf_lo(b[0]); v[1] += bf_hi(b[0]); v[2] += bf_lo(b[1]); v[3] += bf_hi(b[1]);
            }
            if (EPI == LTXMI_EPI_GELU_TANH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(v[e]);
            } else if (EPI == LTXMI_EPI_SILU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
            } else if (EPI == LTXMI_EPI_GATE_RESIDUAL) {
                if (gate_row) {
                    const u32x2 gt = *(const u32x2*)(p.gate_table + n);
                    const u32x2 ge = *(const u32x2*)(gate_row + n);
                    v[0] *= bf_lo(gt[0]) + bf_lo(ge[0]);
                    v[1] *= bf_hi(gt[0]) + bf_hi(ge[0]);
                    v[2] *= bf_lo(gt[1]) + bf_lo(ge[1]);
                    v[3] *= bf_hi(gt[1]) + bf_hi(ge[1]);
                }
                const u32x2 rr = *(const u32x2*)(p.R + (int64_t)m * p.ldr + n);
                v[0] += bf_lo(rr[0]); v[1] += bf_hi(rr[0]); v[2] += bf_lo(rr[1]); v[3] += bf_hi(rr[1]);
            }
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
        }
    }
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
        static bool attr_set = false;                                                                 \
        if (!attr_set) {                                                                              \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); \
            attr_set = true;                                                                          \
        }                                                                                             \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), smem, stream, p);                        \
    }
    if constexpr (MODE == 0) {
        switch (epi) {
            case LTXMI_EPI_NONE: LTXMI_GEMM_LAUNCH(LTXMI_EPI_NONE) break;
            case LTXMI_EPI_GELU_TANH: LTXMI_GEMM_LAUNCH(LTXMI_EPI_GELU_TANH) break;
            case LTXMI_EPI_SILU: LTXMI_GEMM_LAUNCH(LTXMI_EPI_SILU) break;
            case LTXMI_EPI_GATE_RESIDUAL: LTXMI_GEMM_LAUNCH(LTXMI_EPI_GATE_RESIDUAL) break;
            default: set_error("%s: bad epilogue %d", what, epi); return LTXMI_ERR_INVALID_ARG;
        }
    } else {
        if (epi == EPI_D2S) LTXMI_GEMM_LAUNCH(EPI_D2S)
        else if (epi == LTXMI_EPI_GATE_RESIDUAL) LTXMI_GEMM_LAUNCH(LTXMI_EPI_GATE_RESIDUAL)
        else LTXMI_GEMM_LAUNCH(LTXMI_EPI_NONE)
    }
#undef LTXMI_GEMM_LAUNCH
    return check_launch(what);
}

}  // namespace ltxmi

using namespace ltxmi;

extern "C" int ltxmi_gemm_bf16(const ltxmi_gemm_args* a, void* stream) {
    LTXMI_REQUIRE(a && a->A && a->W && a->C, LTXMI_ERR_INVALID_ARG, "ltxmi_gemm_bf16: NULL argument");
    LTXMI_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_gemm_bf16: non-positive shape M=%d N=%d K=%d", a->M, a->N, a->K);
    LTXMI_REQUIRE(a->K % 64 == 0, LTXMI_ERR_UNSUPPORTED, "ltxmi_gemm_bf16: K=%d must be a multiple of 64", a->K);
    LTXMI_REQUIRE(a->N % 8 == 0, LTXMI_ERR_UNSUPPORTED, "ltxmi_gemm_bf16: N=%d must be a multiple of 8", a->N);
    LTXMI_REQUIRE(a->lda % 8 == 0 && a->ldw % 8 == 0 && a->ldc % 4 == 0 && a->lda >= a->K && a->ldw >= a->K &&
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
    p.bias = (const uint16_t*)a->bias;
    p.C = (uint16_t*)a->C; p.ldc = a->ldc;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.R = (const uint16_t*)a->residual; p.ldr = a->ldr;
    p.gate_table = (const uint16_t*)a->gate_table;
    p.gate_temb = (const uint16_t*)a->gate_temb;
    p.gate_ld = a->gate_ld;
    p.rows_per_group = a->rows_per_group > 0 ? a->rows_per_group : 1;
    p.tiles_m = p.tiles_n = 0;
    p.cB = p.cT = p.cH = p.cW = p.cCin = 1; p.tpad = 0; p.pad_replicate = 0; p.res = nullptr; p.res_ch = 0;
    hipStream_t s = (hipStream_t)stream;
    // Tile choice: 256x256 (8 waves) when it still fills the 256 CUs, else 128x128 (4 waves,
    // 2 blocks/CU); skinny problems (adaLN tables, text K/V) take the 128x128 path too.
    const long t256 = (long)((a->M + 255) / 256) * ((a->N + 255) / 256);
    if (a->M >= 1024 && a->N >= 256 && t256 >= 384)
        return launch_tile<256, 256, 2, 4, 0>(p, a->epilogue, s, "ltxmi_gemm_bf16");
    return launch_tile<128, 128, 2, 2, 0>(p, a->epilogue, s, "ltxmi_gemm_bf16");
}

extern "C" int ltxmi_conv3d_ndhwc_bf16(const ltxmi_conv3d_args* a, void* stream) {
    LTXMI_REQUIRE(a && a->x && a->w && a->y, LTXMI_ERR_INVALID_ARG, "ltxmi_conv3d_ndhwc_bf16: NULL argument");
    LTXMI_REQUIRE(a->B > 0 && a->T > 0 && a->H > 0 && a->W > 0 && a->Cin > 0 && a->Cout > 0, LTXMI_ERR_INVALID_ARG,
                  "ltxmi_conv3d_ndhwc_bf16: non-positive shape");
    LTXMI_REQUIRE(a->Cin % 64 == 0, LTXMI_ERR_UNSUPPORTED, "ltxmi_conv3d_ndhwc_bf16: Cin=%d must be a multiple of 64", a->Cin);
    LTXMI_REQUIRE(a->Cout % 8 == 0, LTXMI_ERR_UNSUPPORTED, "ltxmi_conv3d_ndhwc_bf16: Cout=%d must be a multiple of 8", a->Cout);
    const int64_t M = (int64_t)a->B * a->T * a->H * a->W;
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
    GemmParams p;
    p.A = (const uint16_t*)a->x; p.lda = a->Cin;
    p.W = (const uint16_t*)a->w; p.ldw = 27ll * a->Cin;
    p.bias = (const uint16_t*)a->bias;
    p.C = (uint16_t*)a->y; p.ldc = a->Cout;
    p.M = (int)M; p.N = a->Cout; p.K = 27 * a->Cin;
    p.R = nullptr; p.ldr = 0; p.gate_table = nullptr; p.gate_temb = nullptr; p.gate_ld = 0; p.rows_per_group = 1;
    p.tiles_m = p.tiles_n = 0;
    p.cB = a->B; p.cT = a->T; p.cH = a->H; p.cW = a->W; p.cCin = a->Cin;
    p.tpad = a->causal ? 2 : 1;
    p.pad_replicate = a->pad_replicate;
    p.res = a->d2s ? (const uint16_t*)a->residual : nullptr;
    p.res_ch = a->res_channels;
    hipStream_t s = (hipStream_t)stream;
    LTXMI_REQUIRE(!(a->d2s && a->add), LTXMI_ERR_INVALID_ARG, "ltxmi_conv3d_ndhwc_bf16: `add` is for the plain store only");
    if (a->add) { p.R = (const uint16_t*)a->add; p.ldr = a->Cout; }
    const int epi = a->d2s ? EPI_D2S : (a->add ? LTXMI_EPI_GATE_RESIDUAL : LTXMI_EPI_NONE);
    const long t256 = (long)((M + 255) / 256) * ((a->Cout + 255) / 256);
    if (a->Cout >= 256 && t256 >= 384) return launch_tile<256, 256, 2, 4, 1>(p, epi, s, "ltxmi_conv3d_ndhwc_bf16");
    return launch_tile<128, 128, 2, 2, 1>(p, epi, s, "ltxmi_conv3d_ndhwc_bf16");
}
