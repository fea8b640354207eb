// conv_direct.hip -- 3x3x3 stride-1 convolution (CausalConv3d / nn.Conv3d(padding=1)) as a DIRECT convolution:
// the input halo of an output tile is staged in LDS once per 64-channel chunk and all 27 taps are formed from it,
// so the CU's LDS-DMA path carries the weights (16 KB per tap) plus the halo once instead of an im2col A tile per
// tap (the implicit-GEMM form of gemm.hip fetches every input element 27 times through that path, which bounds
// the narrow layers of the VAE decoder: DESIGN.md section 5).
//
//   output tile  : 2 (t) x 8 (y) x 16 (x) = 256 positions x 128 output channels, 8 waves (4 position groups x
//                  2 channel halves, 64 x 64 per wave, 4 x 4 MFMA 16x16x32 blocks; a half beyond Cout is not computed)
//   LDS          : halo 4 x 10 x 18 = 720 rows x 128 B (64 channels) = 90 KB, 16-byte slots XOR-swizzled by
//                  row & 7 (on the DMA source side); weights 2 stages x 128 rows x 128 B = 32 KB
//   per chunk    : halo DMA (90 pieces), then per tap: next tap's weight DMA (2 pieces per wave) || fragment
//                  reads (A rows = 16 x-consecutive halo rows at a wave-uniform tap offset) || 32 MFMAs; barrier
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace ltxmi {

struct ConvDirectP {
    const uint16_t* x; const uint16_t* w; const uint16_t* bias; uint16_t* y; const uint16_t* add;
    const uint16_t* res; int res_ch;            // depth-to-space residual (x itself) or NULL
    int B, T, H, W, Cin, Cout;
    int tpad, pad_replicate, tzero;
    int tiles_t, tiles_y, tiles_x, tiles_n;
    const float* post_scale; const float* post_shift; float post_eps;      // EPI >= 3 only
    uint16_t* y2;                                                          // EPI 4 / 5: the activated second output
    // four-wave form only.  swap_hw: the tile's 16-position rows run along H and its 8 rows along W (W = 24 is 1.5 tiles of 16,
    // H = 16 exactly one: 21 tiles instead of 28 at the 1024-channel stage).  ksplit > 1: the input channels are cut into ksplit
    // ranges, one workgroup each (grid x ksplit), whose fp32 partial sums go to `part` [ksplit][B T H W][Cout] (EPI 6, no bias);
    // conv_split_finalize_kernel adds them up and applies the epilogue
    int swap_hw, ksplit; float* part;
};

__device__ __attribute__((aligned(16))) uint32_t g_zero_page_cd[16];

// Diagnostic build only (-DLTXMI_CONV_STAMPS, tools/conv_stamps.py): s_memtime stamps around the sections of a tap,
// summed per wave into a debug buffer.  No stamp executes in the product build.
#ifdef LTXMI_CONV_STAMPS
#define CSTAMP(i)                                                                              \
    do {                                                                                       \
        unsigned long long t_;                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        cst_acc[i] += t_ - cst_prev;                                                           \
        cst_prev = t_;                                                                         \
    } while (0)
__device__ unsigned long long* g_conv_stamps = nullptr;
#else
#define CSTAMP(i) do { } while (0)
#endif

constexpr int CD_TT = 2, CD_TY = 8, CD_TX = 16;
constexpr int CD_HT = CD_TT + 2, CD_HY = CD_TY + 2, CD_HX = CD_TX + 2;
constexpr int CD_HALO_ROWS = CD_HT * CD_HY * CD_HX;            // 720
constexpr int CD_HALO_BYTES = CD_HALO_ROWS * 128;               // 92160
constexpr int CD_W_BYTES = 128 * 128;                           // one tap: 128 output channels x 64 input channels
constexpr int CD_WSTAGES = 3;
constexpr int CD_ROWTAB_BYTES = CD_HALO_ROWS * 4;               // per halo row: byte offset of its source row (or the 'zero' sentinel)
constexpr int CD_CTLTAB_BYTES = 8 * 27 * 8;                     // per (wave, tap): two packed control words
constexpr int CD_SMEM = CD_HALO_BYTES + CD_WSTAGES * CD_W_BYTES + CD_ROWTAB_BYTES + CD_CTLTAB_BYTES; 

// EPI: 0 plain store, 1 y = conv + add, 2 depth-to-space store (+ residual), as the implicit-GEMM kernel's epilogues
template <int EPI>
__global__ __launch_bounds__(512) void conv3d_direct_kernel(ConvDirectP p) {
    constexpr bool ADD = (EPI == 1), D2S = (EPI == 2);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;
    char* wst = smem + CD_HALO_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // waves 0..3 (one per SIMD) take the first 64 output channels of the block, waves 4..7 the second 64: when the block's
    // second half lies beyond Cout (conv_out: 48 channels) the waves that still work are spread over all four SIMDs
    const int wm = wave & 3, wn = wave >> 2;

    // ---- tile: blockIdx.x -> (n block fastest, then x, y, t, b): the n blocks of a position tile are neighbours
    int id = blockIdx.x;
    const int nb = id % p.tiles_n; id /= p.tiles_n;
    const int tx = id % p.tiles_x; id /= p.tiles_x;
    const int ty = id % p.tiles_y; id /= p.tiles_y;
    const int tt = id % p.tiles_t;
    const int b = id / p.tiles_t;
    const int t0 = tt * CD_TT, y0 = ty * CD_TY, x0 = tx * CD_TX, n0 = nb * 128;

    // ---- halo loader: piece q = rows 8q .. 8q+7; lane -> (row 8q + lane>>3, LDS slot lane&7, source slot ^ row&7).
    // Where a halo row comes from depends on the tile only, not on the 64-channel chunk: the byte offset of every halo row's
    // source row is computed ONCE per tile into a 720-entry table in LDS (stamps of round 2: the div/mod/clamp arithmetic per
    // piece made the issue of a chunk's halo take 4500 cycles, with no MFMA running; round 2 kept twelve offsets per lane in
    // registers, and picking one of them by a run-time piece number became an indexed load from scratch memory).  A piece is
    // then one ds_read_b32 (well ahead of its use) + one add + one buffer_load ... lds with the chunk's channel offset as the
    // scalar offset.  Rows that are zero padding get an offset past the descriptor's range: the hardware delivers zeros.
    const int64_t x_bytes = (int64_t)p.B * p.T * p.H * p.W * p.Cin * 2;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.x, 0, (int)(x_bytes < 0x7ffffff0ll ? x_bytes : 0x7ffffff0ll), 0x00020000);
    uint32_t* rowtab = (uint32_t*)(smem + CD_HALO_BYTES + CD_WSTAGES * CD_W_BYTES);
    {
        const int64_t xb = (int64_t)b * p.T * p.H * p.W * p.Cin;
        for (int r = tid; r < CD_HALO_ROWS; r += 512) {
            const int hx = r % CD_HX, hy = (r / CD_HX) % CD_HY, ht = r / (CD_HX * CD_HY);
            int ti = t0 + ht - p.tpad, yi = y0 + hy - 1, xi = x0 + hx - 1;
            const bool toob = (ti < 0) | (ti >= p.T);
            const bool oob = (yi < 0) | (yi >= p.H) | (xi < 0) | (xi >= p.W);
            ti = ti < 0 ? 0 : (ti >= p.T ? p.T - 1 : ti);
            yi = yi < 0 ? 0 : (yi >= p.H ? p.H - 1 : yi);
            xi = xi < 0 ? 0 : (xi >= p.W ? p.W - 1 : xi);
            const int64_t e = xb + ((int64_t)(ti * p.H + yi) * p.W + xi) * p.Cin;
            const bool zero = (oob && !p.pad_replicate) | (toob && p.tzero);
            rowtab[r] = zero ? 0x7ffffff0u : (uint32_t)(e * 2);
        }
    }
    // (a lane's row within a piece is lane >> 3 whatever the piece, so its 16-byte slot term is a per-lane constant; added to
    // the sentinel it stays out of range)
    const uint32_t hslot = (uint32_t)(((lane & 7) ^ ((lane >> 3) & 7)) << 4);
    // The halo is four t-planes of 180 rows, and the taps run dt-major: after tap 8 (dt = 0 done) plane 0 is dead, after
    // tap 17 plane 1 is, and planes 2 and 3 are first read by tap 9 (whose fragments are fetched under tap 8).  So the halo
    // STREAMS, at most one piece per wave and tap: the chunk's own planes 2 and 3 (pieces 45..89) under its taps 0..6, the
    // NEXT chunk's rows 0..175 (pieces 0..21) under taps 9..11 and its rows 176..359 (pieces 22..44) under taps 18..21.  A
    // chunk boundary has no load phase of its own (round 2: 1950 cycles of issue + 2200 of waiting per chunk with the matrix
    // pipe idle); only the first chunk's planes 0 and 1 are loaded in front of the tap stream.
    constexpr int Q_PLANE0 = 22, Q_PLANE1 = 45, Q_END = CD_HALO_ROWS / 8;
    // Source row offset of a tap's piece: read from the table at the end of the tap before (unconditionally), turned into the
    // lane's offset behind that tap's second MFMA block, used at its end.  The read is issued from inline asm and waited for by
    // hand: as a C++ load hipcc put s_waitcnt lgkmcnt(0) in front of it (at the end of every tap, behind the fragment reads
    // just issued).  An LDS read hipcc does not know about only makes its own counted waits more conservative: LDS
    // operations complete in order.
    uint32_t hoff_nx = 0;
    const uint32_t rowtab_lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)rowtab) + (uint32_t)(lane >> 3) * 4;
    auto read_hoff = [&](int q) __attribute__((always_inline)) {
        const uint32_t a = rowtab_lds + (uint32_t)(q * 32);
        asm volatile("ds_read_b32 %0, %1" : "=v"(hoff_nx) : "v"(a) : "memory");
    };
    // ---- weights of one tap and chunk: rows n0 .. n0+127 of w [Cout, 27*Cin], 64 channels at (tap*Cin + c0), through a
    // descriptor over the block's rows (rows past Cout are out of range and arrive as zeros: never stored), so that a piece is
    // one VALU add + one buffer_load ... lds with the (tap, chunk) offset in a scalar.
    const int w_row_bytes = 27 * p.Cin * 2;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.w + (int64_t)n0 * 27 * p.Cin), 0, min(128, p.Cout - n0) * w_row_bytes, 0x00020000);
    constexpr int WPP = 2;                                       // weight pieces per wave and tap: pieces 2 wave, 2 wave + 1
    const int wp0 = wave * 2;
    const uint32_t woff0 = (uint32_t)((wp0 * 8 + (lane >> 3)) * w_row_bytes + (((lane & 7) ^ ((lane >> 3) & 7)) << 4));
    auto load_w_piece = [&](int stage, int soff, int j) __attribute__((always_inline)) {   // piece j of this wave
        blds16(w_rsrc, wst + stage * CD_W_BYTES + (wp0 + j) * 1024, woff0 + (uint32_t)(j * 8 * w_row_bytes), soff);
    };

    // ---- what a tap of the (chunk, tap) stream does besides its MFMAs -- all wave-uniform, all a function of (tap, chunk).
    // Scalar instructions are NOT free beside MFMAs here: both waves of a SIMD pair run the same code at the same time, and a
    // wave that issues a scalar instruction issues no MFMA (tools/ubench/conv_loop.hip: a bare loop of this tap's 2 x 32 MFMAs
    // runs at 1032 cycles per tap; with the fragment reads, their address arithmetic and the barrier 1196; with ~140 dependent
    // scalar instructions behind block 2 it takes 1780).  So the control of the 27 taps is worked out ONCE per tile into a table
    // in LDS (two packed words per (wave, tap)); a tap reads the entry of the tap after the next with one hidden ds_read_b64 at
    // its end, and the tap in between unpacks it behind its third MFMA block: ~20 scalar instructions per tap instead of the ~85
    // that recomputing it from (tap, chunk) took (measured on the way: recomputed at the top of the tap -5 %, behind block 2
    // +2.0..2.7 % over round 2's kernel, behind block 0 / 1 / 3 -3.5 / -2.2 / -2.0 %: profiles/r03_conv_stream.log).
    struct Ctl {
        int w_soff, w_stage;     // weights two taps ahead in the stream: scalar byte offset (< 0: nothing to issue), stage
        int h_q, h_soff;         // this tap's halo piece (-1: none) and its chunk's byte offset
        int n_off, n_stage;      // next tap: halo row offset of its (dt, dy, dx), weight stage
        int n_q;                 // next tap's piece slot in the row table (clamped; whether there is a piece: its own h_q)
    };
    // word 0: byte offset of tap + 2's weights inside a weight row, relative to its chunk; bit 31: that tap belongs to the NEXT chunk
    // word 1: n_off [0,9) | w_stage [9,11) | n_stage [11,13) | halo piece [13,21) (0xff: none) | piece of the chunk itself [21] | n_q [22,29)
    uint32_t* ctltab = (uint32_t*)(smem + CD_HALO_BYTES + CD_WSTAGES * CD_W_BYTES + CD_ROWTAB_BYTES);
    if (tid < 8 * 27) {
        const int w = tid / 27, tap = tid - w * 27;
        auto piece_slot = [](int t) { return t <= 6 ? t + 5 : t <= 11 ? t - 9 : t - 16; };     // (meaningful inside the three windows)
        const bool wrap2 = tap >= 25;                                              // tap + 2 belongs to the next chunk
        const int t2 = wrap2 ? tap - 25 : tap + 2;
        const bool own = tap <= 6;                                                 // the chunk's own planes 2 and 3
        const bool win = own | ((tap >= 9) & (tap <= 11)) | ((tap >= 18) & (tap <= 21));
        const int lo = own ? Q_PLANE1 : tap <= 11 ? 0 : Q_PLANE0, hi = own ? Q_END : tap <= 11 ? Q_PLANE0 : Q_PLANE1;
        const int q = w + 8 * piece_slot(tap);
        const int hq = (win & (q >= lo) & (q < hi)) ? q : 0xff;
        const int tn = tap < 26 ? tap + 1 : 0;                                     // behind a chunk's last tap: the next chunk's tap 0
        const int n_off = ((tn / 9) * CD_HY + (tn / 3) % 3) * CD_HX + tn % 3;
        const int qn = w + 8 * piece_slot(tn);
        const int n_q = qn < 0 ? 0 : qn < Q_END ? qn : Q_END - 1;
        ctltab[tid * 2] = (uint32_t)(t2 * p.Cin * 2) | (wrap2 ? 0x80000000u : 0u);
        ctltab[tid * 2 + 1] = (uint32_t)n_off | (uint32_t)(t2 % CD_WSTAGES) << 9 | (uint32_t)(tn % CD_WSTAGES) << 11 | (uint32_t)hq << 13 |
                              (own ? 1u << 21 : 0u) | (uint32_t)n_q << 22;
    }
    u32x2 cw_nx = {0u, 0u};                                                       // the packed entry in flight
    const uint32_t ctltab_lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)ctltab) + (uint32_t)(wave * 27 * 8);
    auto read_ctl = [&](int tap) __attribute__((always_inline)) {
        const uint32_t a = ctltab_lds + (uint32_t)(tap * 8);
        asm volatile("ds_read_b64 %0, %1" : "=v"(cw_nx) : "v"(a) : "memory");
    };
    auto unpack_ctl = [&](uint32_t d0, uint32_t d1, int c0) __attribute__((always_inline)) -> Ctl {
        Ctl k;
        const int c_next = c0 + 64 < p.Cin ? c0 + 64 : -1;
        const int c2 = (int)d0 < 0 ? c_next : c0;
        k.w_soff = c2 >= 0 ? (int)(d0 & 0x7fffffffu) + c2 * 2 : -1;
        k.n_off = (int)(d1 & 0x1ffu);
        k.w_stage = (int)((d1 >> 9) & 3u);
        k.n_stage = (int)((d1 >> 11) & 3u);
        const int hq = (int)((d1 >> 13) & 0xffu);
        const int c = ((d1 >> 21) & 1u) ? c0 : c_next;
        k.h_q = ((hq != 0xff) & (c >= 0)) ? hq : -1;
        k.h_soff = c * 2;
        k.n_q = (int)((d1 >> 22) & 0x7fu);
        return k;
    };

    // ---- fragment geometry.  A block i of this wave = positions wm*64 + i*16 + (lane & 15): one (t, y) row of
    // the tile, x = lane & 15, so its halo row at tap (dt, dy, dx) is R(i, tap) + (lane & 15), R wave-uniform.
    const int frow = lane & 15, fchunk = lane >> 4;
    int b_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b_off[j] = (wn * 64 + j * 16 + frow) * 128 + ((fchunk ^ (frow & 7)) << 4);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Fragments of tap + 1 are fetched while the MFMAs of tap issue (two register sets, the tap loop is
    // unrolled by two); with three weight stages the weights of tap + 1 have landed before the barrier that
    // ends tap - 1, and the LDS-DMA of tap + 2 has a whole tap to land.
    bf16x8 af[2][4][2], bfr[2][4][2];
    auto read_frags = [&](auto set_tag, int tap) {
        constexpr int S = decltype(set_tag)::value;
        const int dt = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
        const char* ws = wst + (tap % CD_WSTAGES) * CD_W_BYTES;     // (only called for tap 0 of the stream)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int blk = wm * 4 + i;                                  // (t, y) row of the tile
            const int row = (((blk >> 3) + dt) * CD_HY + ((blk & 7) + dy)) * CD_HX + dx + frow;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                af[S][i][ks] = *(const bf16x8*)(halo + row * 128 + (((fchunk + 4 * ks) ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bfr[S][j][ks] = *(const bf16x8*)(ws + (b_off[j] ^ (ks << 6)));
    };
    auto mfmas = [&](auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[S][j][ks], af[S][i][ks], acc[i][j], 0, 0, 0);
    };
    using s0_t = std::integral_constant<int, 0>;
    using s1_t = std::integral_constant<int, 1>;
#ifdef LTXMI_CONV_STAMPS
    unsigned long long cst_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cst_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(cst_prev)::"memory");
#endif
    // keep1: the youngest request of this wave is a prefetched halo piece of the NEXT chunk -- it may stay in flight
    // across the barrier (the wait of the following tap covers it: requests complete in order)
    auto sync_all = [&](bool keep1) {
        CSTAMP(0);                                               // [0] table read + issue of the halo piece
        if (keep1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CSTAMP(1);                                               // [1] wait for this wave's LDS-DMA
        // a bare s_barrier: __syncthreads() would first wait (lgkmcnt(0)) for the fragment reads of the next tap issued a
        // moment ago -- ~290 cycles per tap with the matrix pipe idle (stamps).  Those reads target registers and read
        // LDS regions that nothing overwrites before the NEXT barrier (the weight stage of tap + 1, halo planes still in
        // use), so they may stay in flight across this one; hipcc waits for them where the MFMAs consume them.
        asm volatile("s_barrier" ::: "memory");
        CSTAMP(2);                                               // [2] barrier
    };
    // One tap of the (chunk, tap) stream: the weights two taps ahead start streaming (this chunk's, or the next chunk's first
    // two), the fragments of the next tap are read between this tap's MFMA blocks, at most one halo piece goes out at the end.
    // A wave whose 64 output channels all lie beyond Cout only helps with the loads and barriers.
    // (the stream's very last tap reads "next" fragments too -- of a tap 0 that does not exist, from LDS that nothing writes
    // any more; they are never used: no branch sits between the MFMA blocks)
    const bool active = n0 + wn * 64 < p.Cout;
    int s_tap = 0, s_c0 = 0;                                     // the stream position
    Ctl cur;                                                     // (set in front of the stream, below)
    auto tap_body = [&](auto cur_tag, auto nxt_tag) __attribute__((always_inline)) {
        CSTAMP(3);
        if (cur.w_soff >= 0) {
#pragma unroll
            for (int j = 0; j < WPP; ++j) load_w_piece(cur.w_stage, cur.w_soff, j);
        }
        CSTAMP(6);                                               // [6] issue of the weight pieces
        // the stream position after this tap
        int tap1 = s_tap + 1, c01 = s_c0;
        if (tap1 == 27) { tap1 = 0; c01 += 64; }
        Ctl nxt;
        uint32_t hfin;
        if (active) {
            // The 16 fragment reads of tap + 1 are NOT issued as a burst in front of this tap's MFMAs: right after the
            // barrier all 8 waves would queue 128 ds_read_b128 (512 LDS cycles, and a wave can have only 15 in flight)
            // with the MFMAs stuck behind them in program order -- stamps: ~600 cycles per tap before the first MFMA.
            // Each position block's 8 MFMAs go first, the reads of the next tap's same block (and one weight block) follow.
            constexpr int S = decltype(cur_tag)::value;          // register sets of this tap's / the next tap's fragments
            constexpr int N = decltype(nxt_tag)::value;
            const char* ws = wst + cur.n_stage * CD_W_BYTES;
            u32x2 cw;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[S][j][ks], af[S][i][ks], acc[i][j], 0, 0, 0);
                const int blk = wm * 4 + i;                                      // (t, y) row of the tile
                const int row = ((blk >> 3) * CD_HY + (blk & 7)) * CD_HX + cur.n_off + frow;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    af[N][i][ks] = *(const bf16x8*)(halo + row * 128 + (((fchunk + 4 * ks) ^ (row & 7)) << 4));
                    bfr[N][i][ks] = *(const bf16x8*)(ws + (b_off[i] ^ (ks << 6)));
                }
                if (i == 1) {
                    // the two hidden reads of the last tap's end (row-table entry, control entry) are older than the 8 fragment
                    // reads this tap has issued so far (LDS reads return in order)
                    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                    hfin = hoff_nx + hslot;
                    cw = cw_nx;
                    asm volatile("" : "+v"(hfin), "+v"(cw));
                }
                if (i == 2) {
                    int d0 = __builtin_amdgcn_readfirstlane((int)cw[0]), d1 = __builtin_amdgcn_readfirstlane((int)cw[1]);
                    asm volatile("" : "+s"(d0), "+s"(d1), "+s"(c01));
                    nxt = unpack_ctl((uint32_t)d0, (uint32_t)d1, c01);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            CSTAMP(7);                                           // [7] MFMAs with the next tap's reads in their shadow
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            uint32_t t = hoff_nx;
            u32x2 cw = cw_nx;
            asm volatile("" : "+v"(t), "+v"(cw));
            hfin = t + hslot;
            int d0 = __builtin_amdgcn_readfirstlane((int)cw[0]), d1 = __builtin_amdgcn_readfirstlane((int)cw[1]);
            asm volatile("" : "+s"(d0), "+s"(d1), "+s"(c01));
            nxt = unpack_ctl((uint32_t)d0, (uint32_t)d1, c01);
        }
        read_hoff(cur.n_q);
        read_ctl(tap1 == 26 ? 0 : tap1 + 1);                     // the entry of the tap after the next
        // this tap's halo piece goes out as the wave's youngest request: it may stay in flight across the barrier (the wait of
        // the following tap covers it)
        if (cur.h_q >= 0) blds16(x_rsrc, halo + cur.h_q * 1024, hfin, cur.h_soff);
        sync_all(cur.h_q >= 0);
        cur = nxt;
        s_tap = tap1;
        s_c0 = c01;
    };

    // ---- in front of the stream: planes 0 and 1 of the first chunk's halo, the weights of its taps 0 and 1
    const int nchunks = p.Cin >> 6;
    CSTAMP(3);
    __syncthreads();                                             // the row table is complete
#pragma unroll
    for (int k = 0; k < (Q_PLANE1 + 7) / 8; ++k) {
        const int q = wave + 8 * k;
        if (q < Q_PLANE1) blds16(x_rsrc, halo + q * 1024, rowtab[q * 8 + (lane >> 3)] + hslot, 0);
    }
#pragma unroll
    for (int j = 0; j < WPP; ++j) {
        load_w_piece(0, 0, j);
        load_w_piece(1, p.Cin * 2, j);
    }
    CSTAMP(4);                                                   // [4] issue of the first half halo + first weights
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    CSTAMP(5);                                                   // [5] ... until they have landed everywhere
    if (active) read_frags(s0_t{}, 0);
    // tap 0's control (read the ordinary way: nothing is in flight yet), then the hidden reads tap 0 consumes: tap 0's row-table
    // entry and tap 1's control
    cur = unpack_ctl((uint32_t)__builtin_amdgcn_readfirstlane((int)ctltab[wave * 54]),
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)ctltab[wave * 54 + 1]), 0);
    read_hoff(wave + 40 < Q_END ? wave + 40 : Q_END - 1);        // (tap 0's piece slot: wave + 8 * 5)
    read_ctl(1);
    // The (chunk, tap) stream is walked two taps at a time (the fragment register sets alternate per tap; a chunk has 27 taps,
    // so the pairs straddle the chunk boundaries)
    {
        const int total = 27 * nchunks;
        for (int g = 0; g + 1 < total; g += 2) {
            tap_body(s0_t{}, s1_t{});
            tap_body(s1_t{}, s0_t{});
        }
        if (total & 1) tap_body(s0_t{}, s1_t{});
    }

    // ---- epilogue: bias (+ add), bf16, through a 4 KB per-wave LDS scratch (the halo is free after the last
    // barrier) so that every store instruction writes whole 128-byte rows (16 bytes per lane)
    CSTAMP(3);
#ifdef LTXMI_CONV_STAMPS
    if (g_conv_stamps && lane == 0 && blockIdx.x < 256) {
        // (the first 256 workgroups = the first round) [7] = everything between taps and chunks is in [3]
        unsigned long long* o = g_conv_stamps + (blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 8; ++i) o[i] = cst_acc[i];
    }
#endif
    if (!active) return;                                   // (the last barrier of the tap loop is behind every wave)
    char* scr = smem + wave * 4096;
    const int ecol = (lane >> 4) * 4;
    u32x2 bias_v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bias_v[j] = *(const u32x2*)(p.bias + min(n0 + wn * 64 + j * 16 + ecol, p.Cout - 4));
    // y = conv + add: the eight 16-byte pieces of `add` this lane needs are requested up front, from clamped (always valid)
    // addresses -- loaded where they are used, each sat behind a branch and an s_waitcnt vmcnt(0) that also waited for the
    // store before it: eight serialised memory round trips per tile, ~7 % of a 128-channel layer's time
    u32x4 add_v[2][4];
    if (ADD) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const int pos = wm * 64 + c * 32 + t4 * 8 + (lane >> 3);
                const int t = min(t0 + (pos >> 7), p.T - 1), yy = min(y0 + ((pos >> 4) & 7), p.H - 1), xx = min(x0 + (pos & 15), p.W - 1);
                const int n = min(n0 + wn * 64 + (lane & 7) * 8, p.Cout - 8);
                add_v[c][t4] = *(const u32x4*)(p.add + ((((int64_t)b * p.T + t) * p.H + yy) * p.W + xx) * p.Cout + n);
            }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * c + ii;
            const int row_l = ii * 16 + frow;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                v[0] += bf_lo(bias_v[j][0]); v[1] += bf_hi(bias_v[j][0]);
                v[2] += bf_lo(bias_v[j][1]); v[3] += bf_hi(bias_v[j][1]);
                if (D2S && p.res) {
                    // x_in = repeat(pixel_shuffle(x)): channel c' <- x[(c' mod (Cres/8)) * 8 + pp]   (gemm.hip, EPI_D2S)
                    const int pos = wm * 64 + i * 16 + frow;
                    const int t = min(t0 + (pos >> 7), p.T - 1), yy = min(y0 + ((pos >> 4) & 7), p.H - 1);
                    const int xx = min(x0 + (pos & 15), p.W - 1);
                    const int Cp = p.Cout >> 3, pp = n0 / Cp, cp = n0 - pp * Cp + wn * 64 + j * 16 + ecol;
                    const uint16_t* rrow = p.res + ((((int64_t)b * p.T + t) * p.H + yy) * p.W + xx) * p.res_ch + pp;
                    const int cm = p.res_ch >> 3;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += bf2f(rrow[((cp + e) % cm) * 8]);
                }
                u32x2 o;
                o[0] = pack_bf16(v[0], v[1]);
                o[1] = pack_bf16(v[2], v[3]);
                const int chunk = j * 2 + (lane >> 5);
                *(u32x2*)(scr + row_l * 128 + ((chunk ^ (row_l & 7)) << 4) + ((lane >> 4) & 1) * 8) = o;
            }
        }
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const int row_l = t4 * 8 + (lane >> 3), chunk = lane & 7;
            u32x4 w = *(const u32x4*)(scr + row_l * 128 + ((chunk ^ (row_l & 7)) << 4));
            const int pos = wm * 64 + c * 32 + row_l;                   // position within the tile
            const int t = t0 + (pos >> 7), yy = y0 + ((pos >> 4) & 7), xx = x0 + (pos & 15);
            if (D2S) {
                // weight rows are packed (p1 p2 p3)-major: the 128 columns of this block are one pp
                const int Cp = p.Cout >> 3, pp = n0 / Cp, cp = n0 - pp * Cp + wn * 64 + chunk * 8;
                const int to = 2 * t + (pp >> 2) - 1, yo = 2 * yy + ((pp >> 1) & 1), xo = 2 * xx + (pp & 1);
                if (t < p.T && yy < p.H && xx < p.W && to >= 0) {          // the first upsampled frame is dropped
                    const int64_t opos = (((int64_t)b * (2 * p.T - 1) + to) * (2 * p.H) + yo) * (2 * p.W) + xo;
                    *(u32x4*)(p.y + opos * Cp + cp) = w;
                }
            } else if (t < p.T && yy < p.H && xx < p.W && n0 + wn * 64 + chunk * 8 < p.Cout) {
                const int64_t off = ((((int64_t)b * p.T + t) * p.H + yy) * p.W + xx) * p.Cout + n0 + wn * 64 + chunk * 8;
                if (ADD) {
                    const u32x4 r = add_v[c][t4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        w[e] = pack_bf16(bf_lo(w[e]) + bf_lo(r[e]), bf_hi(w[e]) + bf_hi(r[e]));
                }
                *(u32x4*)(p.y + off) = w;
            }
        }
    }
}


// =====================================================================================================================
// Two workgroups per CU (round 3).  In the kernel above the two waves of a SIMD are waves w and w + 4 of ONE workgroup: they
// run the same code at the same time, so every instruction that is not an MFMA -- the issue of the LDS-DMA pieces, the table
// reads, the unpacking of the control, the address arithmetic, the barrier, a tile's load phase and epilogue -- is a hole in
// the matrix pipe for both (tools/ubench/conv_loop.hip; the non-MFMA work of a tap takes ~1500 cycles by itself).  Here a
// workgroup is FOUR waves (one per SIMD) on the same 256-position x 128-channel tile, its LDS footprint is halved by 32-channel
// chunks (halo rows of 64 B), and two workgroups share a CU: a SIMD's two waves belong to different workgroups at different
// points of their tiles, and one's holes are the other's MFMAs (the attention kernel's arrangement).
//   wave tile    : 64 positions x 128 channels = 4 x 8 MFMA 16x16x32 blocks, ONE k-step (32 input channels) per tap; 128
//                  accumulator registers.  Channel blocks outermost: a channel block's fragment is dead once its 4 MFMAs have
//                  issued and the next tap's is read into it in place; the 4 position fragments are double-buffered
//   per tap, wave: 32 MFMAs, 12 fragment reads, two 1-KB weight pieces, at most one halo piece
//   LDS          : halo 4 planes x 192 rows (10 x 18 = 180, padded to whole 16-row pieces) x 64 B = 48 KB; weights 3 stages x
//                  128 rows x 64 B = 24 KB; row table 3 KB; control table 0.9 KB  => 75.8 KB per workgroup.  64-byte rows:
//                  16-byte slot s of row r holds source chunk s ^ 2 ((r >> 2) & 1), conflict-free for the 16-lane groups of
//                  ds_read_b128 at every row alignment (exhaustive search over the swizzles of that form)
// The stream and the tables are the kernel above with these constants.  Round 4 added, here only: one epilogue code path for
// seven forms (0 plain, 1 + add, 2 depth-to-space + residual, 3 PixelNorm -> AdaLN -> SiLU of the result, 4 / 5 = 1 / 2 with that
// as a SECOND output, 6 fp32 partial sums of a range of input channels), tiles whose 16-position rows run along H instead of W,
// and the split over the input channels (ConvDirectP::swap_hw / ksplit, conv3d_direct_plan, conv_split_finalize_kernel below).
namespace v3 {
constexpr int TT = 2, TY = 8, TX = 16;
constexpr int HT = TT + 2, HY = TY + 2, HX = TX + 2;
constexpr int PLANE_ROWS = HY * HX;                    // 180
constexpr int PLANE_STRIDE = 192;                      // ... in whole 16-row pieces
constexpr int HALO_ROWS = HT * PLANE_STRIDE;           // 768
constexpr int ROW_B = 64;                              // 32 input channels
constexpr int HALO_BYTES = HALO_ROWS * ROW_B;          // 49152
constexpr int W_BYTES = 128 * ROW_B;                   // one tap: 128 output channels x 32 input channels
constexpr int WSTAGES = 3;
constexpr int ROWTAB_BYTES = HALO_ROWS * 4;
constexpr int CTLTAB_BYTES = 4 * 27 * 8;
constexpr int SMEM = HALO_BYTES + WSTAGES * W_BYTES + ROWTAB_BYTES + CTLTAB_BYTES;   // 77664
constexpr int Q_PLANE0 = PLANE_STRIDE / 16, Q_PLANE1 = 2 * Q_PLANE0, Q_END = HALO_ROWS / 16;   // 12, 24, 48 pieces of 16 rows

__device__ __forceinline__ int swz(int c, int r) { return c ^ (((r >> 2) & 1) << 1); }       // slot of source chunk c in LDS row r

template <int EPI>
__global__ __launch_bounds__(256, 2) void conv3d_direct_v3_kernel(ConvDirectP p) {
    // EPI 4 / 5 = EPI 1 / 2 with TWO outputs: y as EPI 1 / 2 store it, and y2 = silu(pixelnorm(y) (1 + scale) + shift) computed from
    // the bf16 values of y -- the next ResnetBlock3D's norm1 -> AdaLN -> SiLU (causal_video_autoencoder.py:1197-1224), or the
    // decoder's tail (:771-795), without the launch that would read y back.  Needs all channels of an output position in one
    // wave: Cout == 128 (EPI 4), Cout == 8 * 128 (EPI 5: a 128-column block is one (p1 p2 p3)).
    constexpr bool PART = (EPI == 6);                                 // fp32 partial sums of a channel range (see ConvDirectP::ksplit)
    constexpr bool DUAL = (EPI == 4 || EPI == 5);
    constexpr bool ADD = (EPI == 1 || EPI == 4), D2S = (EPI == 2 || EPI == 5), PNORM = (EPI == 3 || DUAL);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;
    char* wst = smem + HALO_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // = the wave's position group (64 positions)

    int id = blockIdx.x;
    const int nb = id % p.tiles_n; id /= p.tiles_n;
    const int tx = id % p.tiles_x; id /= p.tiles_x;
    const int ty = id % p.tiles_y; id /= p.tiles_y;
    const int tt = id % p.tiles_t; id /= p.tiles_t;
    const int b = id % p.B;
    const int ks = id / p.B;                                          // channel range of this workgroup (0 unless ksplit > 1)
    const int t0 = tt * TT, y0 = ty * TY, x0 = tx * TX, n0 = nb * 128;     // y0: the tile's 8-row direction, x0: its 16-position one
    // tile coordinates -> image coordinates (swap_hw: the 16-position direction is H)
    const bool swp = p.swap_hw != 0;
    auto img_y = [&](int i8, int i16) __attribute__((always_inline)) { return swp ? x0 + i16 : y0 + i8; };
    auto img_x = [&](int i8, int i16) __attribute__((always_inline)) { return swp ? y0 + i8 : x0 + i16; };
    // this workgroup's input channels: chunks of 32, dealt as evenly as they go
    const int nch_all = p.Cin >> 5, ch_base = nch_all / p.ksplit, ch_rem = nch_all - ch_base * p.ksplit;
    const int c_first = (ks * ch_base + min(ks, ch_rem)) << 5;
    const int nchunks = ch_base + (ks < ch_rem ? 1 : 0);
    const int c_end = c_first + (nchunks << 5);

    // ---- row table (rows of zero padding, and the 12 rows that pad a plane to whole pieces: an offset past the descriptor)
    const int64_t x_bytes = (int64_t)p.B * p.T * p.H * p.W * p.Cin * 2;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.x, 0, (int)(x_bytes < 0x7ffffff0ll ? x_bytes : 0x7ffffff0ll), 0x00020000);
    uint32_t* rowtab = (uint32_t*)(smem + HALO_BYTES + WSTAGES * W_BYTES);
    {
        const int64_t xb = (int64_t)b * p.T * p.H * p.W * p.Cin;
        for (int r = tid; r < HALO_ROWS; r += 256) {
            const int ht = r / PLANE_STRIDE, rr = r - ht * PLANE_STRIDE;
            const int hy = rr / HX, hx = rr - hy * HX;
            int ti = t0 + ht - p.tpad, yi = img_y(hy - 1, hx - 1), xi = img_x(hy - 1, hx - 1);
            const bool toob = (ti < 0) | (ti >= p.T);
            const bool oob = (yi < 0) | (yi >= p.H) | (xi < 0) | (xi >= p.W);
            ti = ti < 0 ? 0 : (ti >= p.T ? p.T - 1 : ti);
            yi = yi < 0 ? 0 : (yi >= p.H ? p.H - 1 : yi);
            xi = xi < 0 ? 0 : (xi >= p.W ? p.W - 1 : xi);
            const int64_t e = xb + ((int64_t)(ti * p.H + yi) * p.W + xi) * p.Cin;
            const bool zero = (oob && !p.pad_replicate) | (toob && p.tzero) | (rr >= PLANE_ROWS);
            rowtab[r] = zero ? 0x7ffffff0u : (uint32_t)(e * 2);
        }
    }
    // ---- control table, per (wave, tap) -- see the kernel above
    uint32_t* ctltab = (uint32_t*)(smem + HALO_BYTES + WSTAGES * W_BYTES + ROWTAB_BYTES);
    if (tid < 4 * 27) {
        const int w = tid / 27, tap = tid - w * 27;
        // piece slots of a wave: its own planes 2 / 3 (24 pieces, 6 per wave) under taps 0..5, the next chunk's plane 0 (12
        // pieces) under taps 9..11, its plane 1 under taps 18..20
        auto piece_slot = [](int t) { return t <= 5 ? t + 6 : t <= 11 ? t - 9 : t - 15; };
        const bool wrap2 = tap >= 25;
        const int t2 = wrap2 ? tap - 25 : tap + 2;
        const bool own = tap <= 5;
        const bool win = own | ((tap >= 9) & (tap <= 11)) | ((tap >= 18) & (tap <= 20));
        const int lo = own ? Q_PLANE1 : tap <= 11 ? 0 : Q_PLANE0, hi = own ? Q_END : tap <= 11 ? Q_PLANE0 : Q_PLANE1;
        const int q = w + 4 * piece_slot(tap);
        const int hq = (win & (q >= lo) & (q < hi)) ? q : 0xff;
        const int tn = tap < 26 ? tap + 1 : 0;
        // (the tap's dy runs along H, its dx along W: with swap_hw those are the halo's 16-position / 8-row directions)
        const int n_off = (tn / 9) * PLANE_STRIDE + (swp ? (tn % 3) * HX + (tn / 3) % 3 : ((tn / 3) % 3) * HX + tn % 3);
        const int qn = w + 4 * piece_slot(tn);
        const int n_q = qn < 0 ? 0 : qn < Q_END ? qn : Q_END - 1;
        ctltab[tid * 2] = (uint32_t)(t2 * p.Cin * 2) | (wrap2 ? 0x80000000u : 0u);
        ctltab[tid * 2 + 1] = (uint32_t)n_off | (uint32_t)(t2 % WSTAGES) << 9 | (uint32_t)(tn % WSTAGES) << 11 | (uint32_t)hq << 13 |
                              (own ? 1u << 21 : 0u) | (uint32_t)n_q << 22;
    }
    struct Ctl {
        int w_soff, w_stage;
        int h_q, h_soff;
        int n_off, n_stage;
        int n_q;
    };
    u32x2 cw_nx = {0u, 0u};
    const uint32_t ctltab_lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)ctltab) + (uint32_t)(wave * 27 * 8);
    auto read_ctl = [&](int tap) __attribute__((always_inline)) {
        const uint32_t a = ctltab_lds + (uint32_t)(tap * 8);
        asm volatile("ds_read_b64 %0, %1" : "=v"(cw_nx) : "v"(a) : "memory");
    };
    auto unpack_ctl = [&](uint32_t d0, uint32_t d1, int c0) __attribute__((always_inline)) -> Ctl {
        Ctl k;
        const int c_next = c0 + 32 < c_end ? c0 + 32 : -1;
        const int c2 = (int)d0 < 0 ? c_next : c0;
        k.w_soff = c2 >= 0 ? (int)(d0 & 0x7fffffffu) + c2 * 2 : -1;
        k.n_off = (int)(d1 & 0x1ffu);
        k.w_stage = (int)((d1 >> 9) & 3u);
        k.n_stage = (int)((d1 >> 11) & 3u);
        const int hq = (int)((d1 >> 13) & 0xffu);
        const int c = ((d1 >> 21) & 1u) ? c0 : c_next;
        k.h_q = ((hq != 0xff) & (c >= 0)) ? hq : -1;
        k.h_soff = c * 2;
        k.n_q = (int)((d1 >> 22) & 0x7fu);
        return k;
    };
    // a piece = 16 rows x 64 B: lane -> (row lane >> 2, LDS slot lane & 3); (row >> 2) & 1 of a lane's row is (lane >> 4) & 1
    const uint32_t hslot = (uint32_t)(((lane & 3) ^ (((lane >> 4) & 1) << 1)) << 4);
    uint32_t hoff_nx = 0;
    const uint32_t rowtab_lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)rowtab) + (uint32_t)(lane >> 2) * 4;
    auto read_hoff = [&](int q) __attribute__((always_inline)) {
        const uint32_t a = rowtab_lds + (uint32_t)(q * 64);
        asm volatile("ds_read_b32 %0, %1" : "=v"(hoff_nx) : "v"(a) : "memory");
    };
    // ---- weights: two pieces per wave and tap (rows 32 wave .. 32 wave + 31 of the block's 128)
    const int w_row_bytes = 27 * p.Cin * 2;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.w + (int64_t)n0 * 27 * p.Cin), 0, min(128, p.Cout - n0) * w_row_bytes, 0x00020000);
    const uint32_t woff0 = (uint32_t)((wave * 32 + (lane >> 2)) * w_row_bytes) + hslot;
    auto load_w = [&](int stage, int soff) __attribute__((always_inline)) {
        blds16(w_rsrc, wst + stage * W_BYTES + (wave * 2) * 1024, woff0, soff);
        blds16(w_rsrc, wst + stage * W_BYTES + (wave * 2 + 1) * 1024, woff0 + (uint32_t)(16 * w_row_bytes), soff);
    };

    // ---- fragment geometry: position block i of this wave = positions wave*64 + i*16 + (lane & 15) = one (t, y) row
    const int frow = lane & 15, fchunk = lane >> 4;
    const int b_off0 = frow * ROW_B + (swz(fchunk, frow) << 4);            // channel block j: + j * 1024
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[2][4], bfr[8];
    auto a_addr = [&](int i, int n_off) __attribute__((always_inline)) -> const char* {
        const int blk = wave * 4 + i;
        const int row = (blk >> 3) * PLANE_STRIDE + (blk & 7) * HX + n_off + frow;
        return halo + row * ROW_B + (swz(fchunk, row) << 4);
    };
    using s0_t = std::integral_constant<int, 0>;
    using s1_t = std::integral_constant<int, 1>;
    auto sync_all = [&](bool keep1) {
        if (keep1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
    };
    int s_tap = 0, s_c0 = c_first;
    Ctl cur;
    auto tap_body = [&](auto cur_tag, auto nxt_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(cur_tag)::value, N = decltype(nxt_tag)::value;
        if (cur.w_soff >= 0) load_w(cur.w_stage, cur.w_soff);
        int tap1 = s_tap + 1, c01 = s_c0;
        if (tap1 == 27) { tap1 = 0; c01 += 32; }
        Ctl nxt;
        uint32_t hfin;
        u32x2 cw;
        const char* ws = wst + cur.n_stage * W_BYTES;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[S][i], acc[i][j], 0, 0, 0);
            bfr[j] = *(const bf16x8*)(ws + b_off0 + j * 1024);                      // the next tap's channel block j, in place
            if (j < 4) af[N][j] = *(const bf16x8*)a_addr(j, cur.n_off);
            if (j == 3) {
                // the two hidden reads of the last tap's end are older than the 8 fragment reads this tap has issued so far
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                hfin = hoff_nx + hslot;
                cw = cw_nx;
                asm volatile("" : "+v"(hfin), "+v"(cw));
            }
            if (j == 5) {
                int d0 = __builtin_amdgcn_readfirstlane((int)cw[0]), d1 = __builtin_amdgcn_readfirstlane((int)cw[1]);
                asm volatile("" : "+s"(d0), "+s"(d1), "+s"(c01));
                nxt = unpack_ctl((uint32_t)d0, (uint32_t)d1, c01);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        read_hoff(cur.n_q);
        read_ctl(tap1 == 26 ? 0 : tap1 + 1);
        if (cur.h_q >= 0) blds16(x_rsrc, halo + cur.h_q * 1024, hfin, cur.h_soff);
        sync_all(cur.h_q >= 0);
        cur = nxt;
        s_tap = tap1;
        s_c0 = c01;
    };

    // ---- in front of the stream: planes 0 and 1 of the first chunk's halo, the weights of its taps 0 and 1
    // Two workgroups share a CU and each SIMD has one wave of either.  With equal priorities the two waves of a SIMD drift into
    // phase (both in their MFMAs, at half rate each, then both in everything else with the pipe idle); a static priority for one of
    // them keeps them apart.  Workgroups are dealt to the CUs' first slots, then to their second slots, 256 at a time, so bit 8
    // of the workgroup number tells a CU's two residents apart (exactly in the first round, mostly afterwards): +1 % over a
    // decode's convolutions, +4.6 % on the 896-workgroup layer; by parity of the number instead: nothing.
    if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(1);
    __syncthreads();                                             // the tables are complete
#pragma unroll
    for (int k = 0; k < Q_PLANE1 / 4; ++k) {
        const int q = wave + 4 * k;
        blds16(x_rsrc, halo + q * 1024, rowtab[q * 16 + (lane >> 2)] + hslot, c_first * 2);
    }
    load_w(0, c_first * 2);
    load_w(1, (p.Cin + c_first) * 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) af[0][i] = *(const bf16x8*)a_addr(i, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) bfr[j] = *(const bf16x8*)(wst + b_off0 + j * 1024);
    cur = unpack_ctl((uint32_t)__builtin_amdgcn_readfirstlane((int)ctltab[wave * 54]),
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)ctltab[wave * 54 + 1]), c_first);
    read_hoff(wave + 24);                                        // tap 0's piece slot: wave + 4 * 6
    read_ctl(1);
    {
        const int total = 27 * nchunks;
        for (int g = 0; g + 1 < total; g += 2) {
            tap_body(s0_t{}, s1_t{});
            tap_body(s1_t{}, s0_t{});
        }
        if (total & 1) tap_body(s0_t{}, s1_t{});
    }

    // ---- epilogue.  A lane holds, per position block i, position wave * 64 + i * 16 + frow and, per channel block j, the channels
    // j * 16 + ecol .. + 3: a position's 128 channels live in the four lanes frow, frow + 16, + 32, + 48.  Everything the
    // epilogue reads from memory is requested up front (each dependent round trip -- a load behind a store behind a load -- costs
    // the tile ~1 % of its time; round 3's form had one per 32 positions x 64 channels), the arithmetic runs on the accumulators in
    // place, and 32 positions x 64 channels at a time go through the wave's 4-KB LDS scratch (the halo is free) so that every
    // store instruction writes whole 128-byte rows.
    char* scr = smem + wave * 4096;
    const int ecol = (lane >> 4) * 4;
    if (PART) {
        // this channel range's fp32 sums, as they are (no bias): 16 bytes per lane and channel block, 64 contiguous bytes per position
        float* dst = p.part + (int64_t)ks * p.B * p.T * p.H * p.W * p.Cout;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pos = wave * 64 + i * 16 + frow;
            const int t = t0 + (pos >> 7), yy = img_y((pos >> 4) & 7, pos & 15), xx = img_x((pos >> 4) & 7, pos & 15);
            if (t < p.T && yy < p.H && xx < p.W) {
                float* row = dst + ((((int64_t)b * p.T + t) * p.H + yy) * p.W + xx) * p.Cout + n0 + ecol;
#pragma unroll
                for (int j = 0; j < 8; ++j) *(f32x4*)(row + j * 16) = acc[i][j];
            }
        }
        return;
    }
    const int Cp = D2S ? (p.Cout >> 3) : p.Cout, pp = D2S ? n0 / Cp : 0, cbase = n0 - pp * Cp;   // D2S: this block lies inside one (p1 p2 p3)
    // per-channel operands (the block's 128 bias values; EPI >= 3: 128 scales and shifts) go through a 1.25-KB table in the wave's
    // own LDS: one cooperative load, then 8- / 16-byte reads where they are used instead of registers held across the epilogue
    char* tbl = smem + 16384 + wave * 2048;                               // [0, 512) scale, [512, 1024) shift, [1024, 1280) bias
    if (PNORM && p.post_scale) {
        const float* src = (lane < 32 ? p.post_scale : p.post_shift) + (int64_t)b * 128 + (lane & 31) * 4;
        *(f32x4*)(tbl + lane * 16) = *(const f32x4*)src;
    }
    if (lane < 16) *(u32x4*)(tbl + 1024 + lane * 16) = *(const u32x4*)(p.bias + n0 + lane * 8);
    int64_t prow[4];                                                       // input-grid position of (i, frow), clamped
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pos = wave * 64 + i * 16 + frow;
        const int t = min(t0 + (pos >> 7), p.T - 1), yy = min(img_y((pos >> 4) & 7, pos & 15), p.H - 1), xx = min(img_x((pos >> 4) & 7, pos & 15), p.W - 1);
        prow[i] = (((int64_t)b * p.T + t) * p.H + yy) * p.W + xx;
    }
    u32x4 add_v[2][2][4];                                                  // EPI 1: `add` in the stores' row-major layout
    if (ADD && !DUAL) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const int pos = wave * 64 + c * 32 + t4 * 8 + (lane >> 3);
                    const int t = min(t0 + (pos >> 7), p.T - 1), yy = min(img_y((pos >> 4) & 7, pos & 15), p.H - 1), xx = min(img_x((pos >> 4) & 7, pos & 15), p.W - 1);
                    const int n = min(n0 + h * 64 + (lane & 7) * 8, p.Cout - 8);
                    add_v[h][c][t4] = *(const u32x4*)(p.add + ((((int64_t)b * p.T + t) * p.H + yy) * p.W + xx) * p.Cout + n);
                }
    }
    u32x2 add_f[4][8];                                                     // EPI 4: `add` in the accumulators' layout
    if (ADD && DUAL) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) add_f[i][j] = *(const u32x2*)(p.add + prow[i] * 128 + j * 16 + ecol);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const u32x2 bv = *(const u32x2*)(tbl + 1024 + (j * 16 + ecol) * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i][j][0] += bf_lo(bv[0]); acc[i][j][1] += bf_hi(bv[0]);
            acc[i][j][2] += bf_lo(bv[1]); acc[i][j][3] += bf_hi(bv[1]);
        }
    }
    if (D2S && p.res) {
        // x_in = repeat(pixel_shuffle(x)): channel c' <- x[(c' mod (Cres/8)) * 8 + pp]   (gemm.hip, EPI_D2S); (conv + bias) + x_in
        // in this order (-ffast-math may re-associate otherwise).  32-bit element offsets from the uniform base (launch: the
        // tensor is below 2 GiB, Cres / 8 a power of two), one position block's 32 values in flight at a time
        const uint32_t cmask = (uint32_t)(p.res_ch >> 3) - 1u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t rbase = (uint32_t)(prow[i] * p.res_ch + pp);
            uint16_t rv[8][4];
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) rv[j][e] = p.res[rbase + (((uint32_t)(cbase + j * 16 + ecol + e) & cmask) << 3)];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("" : "+v"(acc[i][j]));
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] += bf2f(rv[j][e]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (DUAL) {
        // y exactly as EPI 1 / 2 store it, kept in the accumulators as bf16-representable floats: bf16(conv + bias) then + add in
        // bf16 (EPI 1 rounds twice), or bf16(conv + bias + residual) (EPI 2 rounds once)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint32_t r0 = pack_bf16(acc[i][j][0], acc[i][j][1]), r1 = pack_bf16(acc[i][j][2], acc[i][j][3]);
                if (ADD) {
                    const u32x2 a2 = add_f[i][j];
                    r0 = pack_bf16(bf_lo(r0) + bf_lo(a2[0]), bf_hi(r0) + bf_hi(a2[0]));
                    r1 = pack_bf16(bf_lo(r1) + bf_lo(a2[1]), bf_hi(r1) + bf_hi(a2[1]));
                }
                acc[i][j][0] = bf_lo(r0); acc[i][j][1] = bf_hi(r0);
                acc[i][j][2] = bf_lo(r1); acc[i][j][3] = bf_hi(r1);
            }
    }
    // EPI >= 3 (128 channels per output position, all in this wave): PixelNorm -> (1 + scale) x + shift -> SiLU.  The statistic
    // is a lane-local sum and two cross-lane adds, no LDS exchange
    float rstd_i[4];
    if (PNORM) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) s2 += acc[i][j][e] * acc[i][j][e];
            s2 += __shfl_xor(s2, 16, 64);
            s2 += __shfl_xor(s2, 32, 64);
            rstd_i[i] = rsqrtf(s2 * (1.0f / 128.0f) + p.post_eps);
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int pass = 0; pass < (DUAL ? 2 : 1); ++pass) {
                const bool activate = PNORM && (!DUAL || pass == 1);
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int i = 2 * c + ii;
                    const int row_l = ii * 16 + frow;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 a4 = acc[i][h * 4 + j];
                        float v[4] = {a4[0], a4[1], a4[2], a4[3]};
                        if (activate) {
                            f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
                            if (p.post_scale) {
                                sc = *(const f32x4*)(tbl + (h * 64 + j * 16 + ecol) * 4);
                                sh = *(const f32x4*)(tbl + 512 + (h * 64 + j * 16 + ecol) * 4);
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e] * rstd_i[i] * (1.0f + sc[e]) + sh[e]);
                        }
                        u32x2 o;
                        o[0] = pack_bf16(v[0], v[1]);
                        o[1] = pack_bf16(v[2], v[3]);
                        const int chunk = j * 2 + (lane >> 5);
                        *(u32x2*)(scr + row_l * 128 + ((chunk ^ (row_l & 7)) << 4) + ((lane >> 4) & 1) * 8) = o;
                    }
                }
                uint16_t* dst = (DUAL && pass == 1) ? p.y2 : p.y;
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const int row_l = t4 * 8 + (lane >> 3), chunk = lane & 7;
                    u32x4 w = *(const u32x4*)(scr + row_l * 128 + ((chunk ^ (row_l & 7)) << 4));
                    const int pos = wave * 64 + c * 32 + row_l;
                    const int t = t0 + (pos >> 7), yy = img_y((pos >> 4) & 7, pos & 15), xx = img_x((pos >> 4) & 7, pos & 15);
                    if (D2S) {
                        // weight rows are packed (p1 p2 p3)-major: the 128 columns of this block are one pp
                        const int cp = cbase + h * 64 + chunk * 8;
                        const int to = 2 * t + (pp >> 2) - 1, yo = 2 * yy + ((pp >> 1) & 1), xo = 2 * xx + (pp & 1);
                        if (t < p.T && yy < p.H && xx < p.W && to >= 0) {          // the first upsampled frame is dropped
                            const int64_t opos = (((int64_t)b * (2 * p.T - 1) + to) * (2 * p.H) + yo) * (2 * p.W) + xo;
                            *(u32x4*)(dst + opos * Cp + cp) = w;
                        }
                    } else if (t < p.T && yy < p.H && xx < p.W && n0 + h * 64 + chunk * 8 < p.Cout) {
                        const int64_t off = ((((int64_t)b * p.T + t) * p.H + yy) * p.W + xx) * p.Cout + n0 + h * 64 + chunk * 8;
                        if (ADD && !DUAL) {
                            const u32x4 r = add_v[h][c][t4];
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                w[e] = pack_bf16(bf_lo(w[e]) + bf_lo(r[e]), bf_hi(w[e]) + bf_hi(r[e]));
                        }
                        *(u32x4*)(dst + off) = w;
                    }
                }
            }
        }
    }
}
}  // namespace v3

// =====================================================================================================================
// Split over the input channels (round 4).  At the 1024-channel stage of the decoder (4992 positions) a layer is 168-224 tiles:
// one partial round of the chip whatever the kernel.  There the four-wave form cuts the input channels into ksplit ranges (three:
// 504 workgroups for its 512 slots), every workgroup writes the fp32 sums of its range (EPI 6), and this kernel -- one workgroup
// per output position of the convolution grid, a thread per CPT consecutive channels -- adds the ranges up IN RANGE ORDER, adds the
// bias and applies the epilogue of the layer: plain store, + add, or the depth-to-space store (+ residual), each with the
// optional PixelNorm -> (1 + scale) x + shift -> SiLU as the only or as a second output (a row of the partial sums holds every
// channel of its position(s), so the norm rides here at any width).  The arithmetic of the values mirrors the fused epilogues:
// y = bf16(sum + bias [+ residual]), `add` added to the rounded value and rounded again, the statistic from the bf16 values when
// the raw result is kept and from the fp32 ones when only the activated result is (EPI 3's form).
struct ConvFinalizeP {
    const float* part; int ksplit; int64_t rows;             // [ksplit][rows][Cout]
    const uint16_t* bias; const uint16_t* add; const uint16_t* res; int res_ch;
    uint16_t* y; uint16_t* y2;                               // y2: the activated output (with y = NULL: the only one)
    int B, T, H, W, Cout, d2s;
    const float* post_scale; const float* post_shift; float post_eps; int post_norm;
};

template <int CPT, int NT = 256>       // NT threads x CPT channels = Cout
__global__ __launch_bounds__(NT) void conv_split_finalize_kernel(ConvFinalizeP f) {
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = tid * CPT;
    float v[CPT];
    {
        const float* src = f.part + row * f.Cout + c0;
#pragma unroll
        for (int q = 0; q < CPT / 4; ++q) {
            const f32x4 a = *(const f32x4*)(src + q * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[q * 4 + e] = a[e];
        }
        for (int s = 1; s < f.ksplit; ++s) {
            src += f.rows * f.Cout;
#pragma unroll
            for (int q = 0; q < CPT / 4; ++q) {
                const f32x4 a = *(const f32x4*)(src + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[q * 4 + e] += a[e];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < CPT / 4; ++q) {
        const u32x2 bv = *(const u32x2*)(f.bias + c0 + q * 4);
        v[q * 4 + 0] += bf_lo(bv[0]); v[q * 4 + 1] += bf_hi(bv[0]);
        v[q * 4 + 2] += bf_lo(bv[1]); v[q * 4 + 3] += bf_hi(bv[1]);
    }
    const int Cp = f.d2s ? (f.Cout >> 3) : f.Cout;           // channels of an output position = the norm's width
    const int pp = c0 / Cp, cp = c0 - pp * Cp;               // (CPT divides Cp: a thread's channels lie inside one (p1 p2 p3))
    if (f.d2s && f.res) {
        const uint32_t cmask = (uint32_t)(f.res_ch >> 3) - 1u;
        const uint16_t* rrow = f.res + row * f.res_ch + pp;
#pragma unroll
        for (int e = 0; e < CPT; ++e) {
            asm volatile("" : "+v"(v[e]));                   // (conv + bias) + residual, in this order
            v[e] += bf2f(rrow[(((uint32_t)(cp + e)) & cmask) << 3]);
        }
    }
    const bool keep_raw = f.y != nullptr;
    if (keep_raw) {
        // the raw result as the fused epilogues round it
#pragma unroll
        for (int q = 0; q < CPT / 2; ++q) {
            uint32_t r = pack_bf16(v[2 * q], v[2 * q + 1]);
            if (f.add) {
                const uint32_t a = *(const uint32_t*)(f.add + row * f.Cout + c0 + 2 * q);
                r = pack_bf16(bf_lo(r) + bf_lo(a), bf_hi(r) + bf_hi(a));
            }
            v[2 * q] = bf_lo(r); v[2 * q + 1] = bf_hi(r);
        }
    }
    float rstd = 0.f;
    if (f.post_norm) {
        float s2 = 0.f;
#pragma unroll
        for (int e = 0; e < CPT; ++e) s2 += v[e] * v[e];
        const int tpg = Cp / CPT;                            // threads per norm group: a power of two, 32 .. 256
        for (int off = 1; off < (tpg < 64 ? tpg : 64); off <<= 1) s2 += __shfl_xor(s2, off, 64);
        if (tpg > 64) {
            __shared__ float red[NT / 64];
            if (lane == 0) red[wave] = s2;
            __syncthreads();
            const int gw = tpg >> 6, base = (wave / gw) * gw;
            s2 = 0.f;
            for (int k = 0; k < gw; ++k) s2 += red[base + k];
        }
        rstd = rsqrtf(s2 / (float)Cp + f.post_eps);
    }
    // where the position's channels go
    int64_t out_off;
    bool store = true;
    if (f.d2s) {
        int64_t r = row;
        const int x_ = (int)(r % f.W); r /= f.W;
        const int y_ = (int)(r % f.H); r /= f.H;
        const int t_ = (int)(r % f.T);
        const int b_ = (int)(r / f.T);
        const int to = 2 * t_ + (pp >> 2) - 1, yo = 2 * y_ + ((pp >> 1) & 1), xo = 2 * x_ + (pp & 1);
        store = to >= 0;                                     // the first upsampled frame is dropped
        out_off = ((((int64_t)b_ * (2 * f.T - 1) + to) * (2 * f.H) + yo) * (2 * f.W) + xo) * Cp + cp;
    } else {
        out_off = row * f.Cout + c0;
    }
    if (!store) return;
    if (keep_raw) {
#pragma unroll
        for (int q = 0; q < CPT / 4; ++q) {
            u32x2 o;
            o[0] = pack_bf16(v[4 * q], v[4 * q + 1]);
            o[1] = pack_bf16(v[4 * q + 2], v[4 * q + 3]);
            *(u32x2*)(f.y + out_off + 4 * q) = o;
        }
    }
    if (f.post_norm) {
        const int b_ = (int)(row / ((int64_t)f.T * f.H * f.W));
#pragma unroll
        for (int q = 0; q < CPT / 4; ++q) {
            f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (f.post_scale) {
                sc = *(const f32x4*)(f.post_scale + (int64_t)b_ * Cp + cp + 4 * q);
                sh = *(const f32x4*)(f.post_shift + (int64_t)b_ * Cp + cp + 4 * q);
            }
            float u[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = silu_f(v[4 * q + e] * rstd * (1.0f + sc[e]) + sh[e]);
            u32x2 o;
            o[0] = pack_bf16(u[0], u[1]);
            o[1] = pack_bf16(u[2], u[3]);
            *(u32x2*)(f.y2 + out_off + 4 * q) = o;
        }
    }
}

}  // namespace ltxmi
#ifdef LTXMI_CONV_STAMPS
extern "C" int ltxmi_debug_set_conv_stamps(void* buf) {
    unsigned long long* b = (unsigned long long*)buf;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(ltxmi::g_conv_stamps), &b, sizeof(b));
}
#endif
namespace ltxmi {

// the shapes the direct convolution takes at all (either form)
static bool conv3d_direct_takes(const ltxmi_conv3d_args* a) {
    const int st = a->stride_t > 0 ? a->stride_t : 1, sh = a->stride_hw > 0 ? a->stride_hw : 1;
    const int kt = a->kernel_t > 0 ? a->kernel_t : 3;
    if (st != 1 || sh != 1 || kt != 3 || a->out_T > 0 || a->tpad > 0) return false;
    if (a->Cin % 64 != 0 || a->Cout % 8 != 0 || !a->bias) return false;
    if ((int64_t)a->B * a->T * a->H * a->W * a->Cin * 2 >= 0x7ffffff0ll) return false;   // halo rows are addressed with 32-bit byte offsets
    if (a->d2s && (a->Cout % 1024 != 0 || a->add)) return false;       // a 128-column block must be one (p1 p2 p3)
    // the residual's channel wrap (c' mod Cres/8) is a mask in the four-wave form
    if (a->d2s && a->residual && (a->res_channels < 8 || ((a->res_channels >> 3) & ((a->res_channels >> 3) - 1)) != 0)) return false;
    return true;
}
// whole 128-channel blocks: the four-wave form, two workgroups per CU (algo 3 asks for it, algo 4 for the eight-wave form)
// (from 768 workgroups = 1.5 rounds of the chip's 512 slots; measured: 896 workgroups +5.8 %, 600 -0.6 %, 224 -21 %)
static bool conv3d_direct_four_wave_form(const ltxmi_conv3d_args* a, int64_t grid) {
    return a->Cout % 128 == 0 && grid < (1ll << 31) && ((a->algo != 4 && grid >= 768) || a->algo == 3);
}

// How a call is laid out on the chip.  Tiles are 2 (t) x 8 x 16 positions; the 16-position direction is W, or H (swap: the
// four-wave form only).  ksplit > 1: the input channels in ksplit ranges with fp32 partial sums and a finalising pass -- for the
// wide, short layers (Cin >= 1024: the partial sums are Cin / (4 ksplit) times smaller than the halo traffic they replace) whose
// tiles do not fill the chip: 1024 -> 1024 at 13 x 16 x 24 positions is 224 tiles of which 70 % of the positions exist (W = 24
// is 1.5 tiles); swapped it is 168 tiles at 93 %, and three channel ranges make 504 workgroups for the 512 slots.
struct ConvPlan {
    bool four_wave; int swap, ksplit;
    int tiles_t, tiles_8, tiles_16, tiles_n;      // tiles along t, the 8-row direction, the 16-position direction, 128-channel blocks
    int64_t grid;                                 // workgroups of ONE channel range
    int64_t workspace_bytes;                      // ksplit > 1: [ksplit][B T H W][Cout] fp32
};
static ConvPlan conv3d_direct_plan(const ltxmi_conv3d_args* a, bool have_workspace) {
    ConvPlan pl;
    pl.tiles_t = (a->T + CD_TT - 1) / CD_TT; pl.tiles_n = (a->Cout + 127) / 128;
    const int64_t per = (int64_t)a->B * pl.tiles_t * pl.tiles_n;
    const int64_t g_n = per * ((a->H + 7) / 8) * ((a->W + 15) / 16), g_s = per * ((a->W + 7) / 8) * ((a->H + 15) / 16);
    pl.swap = 0; pl.ksplit = 1; pl.workspace_bytes = 0;
    pl.four_wave = conv3d_direct_four_wave_form(a, g_n);
    const double positions = (double)a->B * pl.tiles_t * CD_TT * a->H * a->W * pl.tiles_n;       // (x 128 channels each, t rounded up)
    auto eff4 = [&](int64_t g, int S) {           // useful share of the tiles x fill of the last round of 512 slots - the split's price
        const double rounds = (double)g * S / 512.0;
        return positions / ((double)g * 256.0) * rounds / (double)(int64_t)(rounds + 0.999999) - 0.03 * (S - 1);
    };
    const double eff_now = pl.four_wave ? eff4(g_n, 1)
                                        : positions / ((double)g_n * 256.0) * ((double)g_n / 256.0) / (double)((g_n + 255) / 256) * 0.93;
    // The channel split: the product's own choice (algo 0 / 2 / 3), output rows the finalising pass takes (512 or n x 1024
    // channels).  Cin >= 1024: wherever it buys more than 5 % of the launch at 3 % per extra range.  512 <= Cin < 1024 (the partial
    // sums cost twice as much per FLOP): two ranges only, and only for a call that asks for a norm the unsplit form could not
    // fuse -- the finalising pass replaces that launch (0.084 ms beside a 0.55-ms convolution at the decoder's 512-channel stage,
    // whose 624 workgroups are 2.44 rounds of the eight-wave form), which the efficiency figure does not see.
    const int nch = a->Cin / 32;
    const bool rows_ok = (a->Cout == 512 || a->Cout % 1024 == 0) && a->Cout <= 4096;
    const bool wants_unfusable_norm = a->post_norm && !(a->Cout == 128 || (a->d2s && a->Cout == 1024));
    if (rows_ok && a->Cin % 32 == 0 && a->algo != 1 && a->algo != 4 && (a->Cin >= 1024 || (a->Cin >= 512 && wants_unfusable_norm))) {
        const int64_t g = g_s < g_n ? g_s : g_n;
        const bool wide = a->Cin >= 1024;
        const double price = wide ? 0.03 : 0.06;
        int best = 1;
        double best_eff = wide ? eff_now + 0.05 : eff_now - 0.05;
        for (int S = 2; S <= (wide ? 4 : 2) && S * 4 <= nch; ++S) {
            const double e = eff4(g, S) + 0.03 * (S - 1) - price * (S - 1);
            if (g * S < (1ll << 31) && e > best_eff) { best = S; best_eff = e; }
        }
        if (best > 1) {
            pl.workspace_bytes = (int64_t)best * a->B * a->T * a->H * a->W * a->Cout * 4;
            if (have_workspace && a->workspace && a->workspace_bytes >= pl.workspace_bytes && (((uintptr_t)a->workspace) & 15) == 0) {
                pl.ksplit = best; pl.four_wave = true; pl.swap = g_s < g_n;
            }
        }
    }
    if (pl.ksplit == 1 && pl.four_wave && (g_s + 511) / 512 < (g_n + 511) / 512) pl.swap = 1;    // fewer rounds of the chip
    pl.tiles_8 = pl.swap ? (a->W + 7) / 8 : (a->H + 7) / 8;
    pl.tiles_16 = pl.swap ? (a->H + 15) / 16 : (a->W + 15) / 16;
    pl.grid = per * pl.tiles_8 * pl.tiles_16;
    return pl;
}
int64_t conv3d_direct_workspace_bytes(const ltxmi_conv3d_args* a) {
    if (a->algo == 1 || !conv3d_direct_takes(a)) return 0;
    return conv3d_direct_plan(a, false).workspace_bytes;
}
// ltxmi_conv3d_fuses_post_norm: the four-wave form where a wave holds every channel of its output positions -- ONE 128-channel
// block (plain store; with `add` only as the second output y_norm beside the raw y), or the depth-to-space store to 128 channels
// (second output only: a 128-column block is one (p1 p2 p3)) -- and every call that runs split over its input channels (the
// finalising pass holds whole rows)
bool conv3d_direct_fuses_post_norm(const ltxmi_conv3d_args* a) {
    if (a->algo == 1 || !conv3d_direct_takes(a)) return false;
    if (!a->y_norm && (a->add || a->d2s)) return false;      // the activated result as the ONLY output: the plain store only
    const ConvPlan pl = conv3d_direct_plan(a, true);
    if (pl.ksplit > 1) return true;
    if (!pl.four_wave) return false;
    if (a->y_norm) return (a->d2s && a->Cout == 1024) || (!a->d2s && a->add && a->Cout == 128);
    return a->Cout == 128 && !a->d2s && !a->add;
}

// Returns -1 when the shape is not one this kernel takes (the caller then uses the implicit GEMM).
int launch_conv3d_direct(const ltxmi_conv3d_args* a, hipStream_t stream) {
    if (!conv3d_direct_takes(a)) return -1;
    const ConvPlan pl = conv3d_direct_plan(a, true);
    ConvDirectP p;
    p.x = (const uint16_t*)a->x; p.w = (const uint16_t*)a->w; p.bias = (const uint16_t*)a->bias;
    p.y = (uint16_t*)a->y; p.add = (const uint16_t*)a->add;
    p.res = a->d2s ? (const uint16_t*)a->residual : nullptr; p.res_ch = a->res_channels;
    p.B = a->B; p.T = a->T; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout;
    p.tpad = a->causal ? 2 : 1; p.pad_replicate = a->pad_replicate; p.tzero = a->time_pad_zeros ? 1 : 0;
    p.tiles_t = pl.tiles_t; p.tiles_y = pl.tiles_8; p.tiles_x = pl.tiles_16; p.tiles_n = pl.tiles_n;
    p.swap_hw = pl.swap; p.ksplit = pl.ksplit; p.part = (float*)a->workspace;
    const int64_t grid = pl.grid * pl.ksplit;
    p.post_scale = a->post_scale; p.post_shift = a->post_shift; p.post_eps = a->post_eps;
    p.y2 = (uint16_t*)a->y_norm;
    if (a->post_norm && !conv3d_direct_fuses_post_norm(a)) return -1;
    if (pl.four_wave) {
#define LTXMI_CDV3_LAUNCH(E)                                                                                   \
        {                                                                                                      \
            static unsigned long long lds_done = 0;                                                            \
            if (const int rc_ = reserve_lds((const void*)v3::conv3d_direct_v3_kernel<E>, v3::SMEM, &lds_done,  \
                                            "ltxmi_conv3d_ndhwc_bf16"))                                        \
                return rc_;                                                                                    \
            hipLaunchKernelGGL(v3::conv3d_direct_v3_kernel<E>, dim3((unsigned)grid), dim3(256), v3::SMEM,      \
                               stream, p);                                                                     \
        }
        if (pl.ksplit > 1) {
            LTXMI_CDV3_LAUNCH(6)
            if (const int rc_ = check_launch("ltxmi_conv3d_ndhwc_bf16")) return rc_;
            ConvFinalizeP f;
            f.part = (const float*)a->workspace; f.ksplit = pl.ksplit; f.rows = (int64_t)a->B * a->T * a->H * a->W;
            f.bias = p.bias; f.add = p.add; f.res = p.res; f.res_ch = p.res_ch;
            // post_norm without y_norm: the activated result is the only output (to y); with y_norm: raw to y, activated to y_norm
            f.y = (a->post_norm && !a->y_norm) ? nullptr : p.y;
            f.y2 = a->post_norm ? (a->y_norm ? (uint16_t*)a->y_norm : p.y) : nullptr;
            f.B = a->B; f.T = a->T; f.H = a->H; f.W = a->W; f.Cout = a->Cout; f.d2s = a->d2s;
            f.post_scale = a->post_scale; f.post_shift = a->post_shift; f.post_eps = a->post_eps; f.post_norm = a->post_norm;
            const dim3 fg((unsigned)f.rows);
            switch (a->Cout / 256) {
                case 2: hipLaunchKernelGGL((conv_split_finalize_kernel<4, 128>), fg, dim3(128), 0, stream, f); break;
                case 4: hipLaunchKernelGGL(conv_split_finalize_kernel<4>, fg, dim3(256), 0, stream, f); break;
                case 8: hipLaunchKernelGGL(conv_split_finalize_kernel<8>, fg, dim3(256), 0, stream, f); break;
                case 12: hipLaunchKernelGGL(conv_split_finalize_kernel<12>, fg, dim3(256), 0, stream, f); break;
                default: hipLaunchKernelGGL(conv_split_finalize_kernel<16>, fg, dim3(256), 0, stream, f); break;
            }
            return check_launch("ltxmi_conv3d_ndhwc_bf16");
        }
        if (a->d2s && a->post_norm) LTXMI_CDV3_LAUNCH(5)
        else if (a->add && a->post_norm) LTXMI_CDV3_LAUNCH(4)
        else if (a->d2s) LTXMI_CDV3_LAUNCH(2)
        else if (a->add) LTXMI_CDV3_LAUNCH(1)
        else if (a->post_norm) LTXMI_CDV3_LAUNCH(3)
        else LTXMI_CDV3_LAUNCH(0)
#undef LTXMI_CDV3_LAUNCH
        return check_launch("ltxmi_conv3d_ndhwc_bf16");
    }
    // one workgroup per CU is resident: below ~half the CUs the implicit GEMM's smaller tiles fill the chip better
    // (algo >= 2 asks for the direct convolution whatever the grid)
    constexpr int min_grid = 128;
    if (grid >= (1ll << 31) || (grid < min_grid && a->algo < 2)) return -1;
#define LTXMI_CD_LAUNCH(E)                                                                                     \
    {                                                                                                          \
        static unsigned long long lds_done = 0;                                                                \
        if (const int rc_ = reserve_lds((const void*)conv3d_direct_kernel<E>, CD_SMEM, &lds_done,              \
                                        "ltxmi_conv3d_ndhwc_bf16"))                                            \
            return rc_;                                                                                        \
        hipLaunchKernelGGL(conv3d_direct_kernel<E>, dim3((unsigned)grid), dim3(512), CD_SMEM, stream, p);      \
    }
    if (a->d2s) LTXMI_CD_LAUNCH(2)
    else if (a->add) LTXMI_CD_LAUNCH(1)
    else LTXMI_CD_LAUNCH(0)
#undef LTXMI_CD_LAUNCH
    return check_launch("ltxmi_conv3d_ndhwc_bf16");
}

}  // namespace ltxmi
