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
};

__device__ __attribute__((aligned(16))) uint32_t g_zero_page_cd[16];

constexpr int CD_TT = 2, CD_TY = 8, CD_TX = 16;
constexpr int CD_HT = CD_TT + 2, CD_HY = CD_TY + 2, CD_HX = CD_TX + 2;
constexpr int CD_HALO_ROWS = CD_HT * CD_HY * CD_HX;            // 720
constexpr int CD_HALO_BYTES = CD_HALO_ROWS * 128;               // 92160
constexpr int CD_W_BYTES = 128 * 128;                           // one tap: 128 output channels x 64 input channels
constexpr int CD_WSTAGES = 3;
constexpr int CD_SMEM = CD_HALO_BYTES + CD_WSTAGES * CD_W_BYTES; // 141312

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

    // ---- halo loader: piece q = rows 8q .. 8q+7; lane -> (row 8q + lane>>3, LDS slot lane&7, source slot ^ row&7)
    const int64_t xb = (int64_t)b * p.T * p.H * p.W * p.Cin;
    auto load_halo = [&](int c0) {
        for (int q = wave; q < CD_HALO_ROWS / 8; q += 8) {
            const int r = q * 8 + (lane >> 3);
            const int hx = r % CD_HX, hy = (r / CD_HX) % CD_HY, ht = r / (CD_HX * CD_HY);
            int ti = t0 + ht - p.tpad, yi = y0 + hy - 1, xi = x0 + hx - 1;
            const bool toob = (ti < 0) | (ti >= p.T);
            const bool oob = (yi < 0) | (yi >= p.H) | (xi < 0) | (xi >= p.W);
            ti = ti < 0 ? 0 : (ti >= p.T ? p.T - 1 : ti);
            yi = yi < 0 ? 0 : (yi >= p.H ? p.H - 1 : yi);
            xi = xi < 0 ? 0 : (xi >= p.W ? p.W - 1 : xi);
            const uint16_t* src = p.x + xb + ((int64_t)(ti * p.H + yi) * p.W + xi) * p.Cin + c0 + (((lane & 7) ^ (r & 7)) << 3);
            if ((oob && !p.pad_replicate) | (toob && p.tzero)) src = (const uint16_t*)g_zero_page_cd;
            glds16(src, halo + q * 1024);
        }
    };
    // ---- weights of one tap and chunk: rows n0 .. n0+127 of w [Cout, 27*Cin], 64 channels at (tap*Cin + c0)
    const int wrow0 = (wave * 2) * 8 + (lane >> 3);
    auto load_w = [&](int stage, int tap, int c0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = wrow0 + j * 8;
            const int n = min(n0 + row, p.Cout - 1);                      // rows past Cout: recomputed, never stored
            const uint16_t* src = p.w + (int64_t)n * (27 * p.Cin) + tap * p.Cin + c0 + (((lane & 7) ^ (row & 7)) << 3);
            glds16(src, wst + stage * CD_W_BYTES + (wave * 2 + j) * 1024);
        }
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
        const char* ws = wst + (tap % CD_WSTAGES) * CD_W_BYTES;
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
    auto sync_all = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    // one tap: weights of tap + 2 start streaming, fragments of tap + 1 are read, MFMAs of tap
    // a wave whose 64 output channels all lie beyond Cout only helps with the loads and barriers
    const bool active = n0 + wn * 64 < p.Cout;
    auto tap_body = [&](int tap, int c0, auto cur_tag, auto nxt_tag) {
        if (tap + 2 < 27) load_w((tap + 2) % CD_WSTAGES, tap + 2, c0);
        if (active) {
            if (tap + 1 < 27) read_frags(nxt_tag, tap + 1);
            mfmas(cur_tag);
        }
        sync_all();
    };

    const int nchunks = p.Cin >> 6;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int c0 = ch * 64;
        load_halo(c0);
        load_w(0, 0, c0);
        load_w(1, 1, c0);
        sync_all();
        if (active) read_frags(s0_t{}, 0);
        for (int tap = 0; tap < 26; tap += 2) {
            tap_body(tap, c0, s0_t{}, s1_t{});
            tap_body(tap + 1, c0, s1_t{}, s0_t{});
        }
        tap_body(26, c0, s0_t{}, s1_t{});
    }

    // ---- epilogue: bias (+ add), bf16, through a 4 KB per-wave LDS scratch (the halo is free after the last
    // barrier) so that every store instruction writes whole 128-byte rows (16 bytes per lane)
    if (!active) return;                                   // (the last barrier of the tap loop is behind every wave)
    char* scr = smem + wave * 4096;
    const int ecol = (lane >> 4) * 4;
    u32x2 bias_v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bias_v[j] = *(const u32x2*)(p.bias + min(n0 + wn * 64 + j * 16 + ecol, p.Cout - 4));
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
                    const u32x4 r = *(const u32x4*)(p.add + off);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        w[e] = pack_bf16(bf_lo(w[e]) + bf_lo(r[e]), bf_hi(w[e]) + bf_hi(r[e]));
                }
                *(u32x4*)(p.y + off) = w;
            }
        }
    }
}

// Returns -1 when the shape is not one this kernel takes (the caller then uses the implicit GEMM).
int launch_conv3d_direct(const ltxmi_conv3d_args* a, hipStream_t stream) {
    const int st = a->stride_t > 0 ? a->stride_t : 1, sh = a->stride_hw > 0 ? a->stride_hw : 1;
    const int kt = a->kernel_t > 0 ? a->kernel_t : 3;
    if (st != 1 || sh != 1 || kt != 3 || a->out_T > 0 || a->tpad > 0) return -1;
    if (a->Cin % 64 != 0 || a->Cout % 8 != 0 || !a->bias) return -1;
    if (a->d2s && (a->Cout % 1024 != 0 || a->add)) return -1;          // a 128-column block must be one (p1 p2 p3)
    ConvDirectP p;
    p.x = (const uint16_t*)a->x; p.w = (const uint16_t*)a->w; p.bias = (const uint16_t*)a->bias;
    p.y = (uint16_t*)a->y; p.add = (const uint16_t*)a->add;
    p.res = a->d2s ? (const uint16_t*)a->residual : nullptr; p.res_ch = a->res_channels;
    p.B = a->B; p.T = a->T; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout;
    p.tpad = a->causal ? 2 : 1; p.pad_replicate = a->pad_replicate; p.tzero = a->time_pad_zeros ? 1 : 0;
    p.tiles_t = (a->T + CD_TT - 1) / CD_TT; p.tiles_y = (a->H + CD_TY - 1) / CD_TY;
    p.tiles_x = (a->W + CD_TX - 1) / CD_TX; p.tiles_n = (a->Cout + 127) / 128;
    const int64_t grid = (int64_t)a->B * p.tiles_t * p.tiles_y * p.tiles_x * p.tiles_n;
    // one workgroup per CU is resident: below ~half the CUs the implicit GEMM's smaller tiles fill the chip better
    // (algo = 2 asks for this kernel whatever the grid)
    constexpr int min_grid = 128;
    if (grid >= (1ll << 31) || (grid < min_grid && a->algo != 2)) return -1;
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
