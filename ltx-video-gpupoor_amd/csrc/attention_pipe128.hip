// attention_pipe128.hip -- software-pipelined flash attention forward at head_dim 128 (no key bias): the large
// self-attention shapes of the models with 128-wide heads -- Wan (wan/modules/model.py:175-240 of the reference calls
// pay_attention with [1, 32760, 12, 128] at config 4) and the 13B LTX model the reference ships (ltxv.py:171-194) -- behind
// pay_attention()'s eager branch (wan/modules/attention.py:99-116,344-347).
//
// Same data flow as attention_pipe.hip (swapped product S^T = K Q^T with the query on the lane, online softmax in registers,
// O^T += V^T P^T with P straight from the accumulator registers, row sums on the matrix pipe, K/V by LDS-DMA into two rings
// of four tiles behind counted waits, one s_barrier per key tile, two 32-row query blocks per wave running half a key tile
// apart).  What head_dim 128 changes:
//   * twice the MFMAs per score (16 QK^T + 16 PV per 32 x 64 block-tile) for the same softmax work: a chunk = ONE MFMA with
//     ONE score's fma + exp2 (a v_cvt_pk every other chunk) -- the matrix pipe, not the VALU, paces the loop;
//   * the state of two blocks (2 x 64 O^T + 2 x 32 S^T accumulators, 2 x 32 Q^T fragment registers ...) is ~300 registers:
//     ONE workgroup of four waves per CU, one wave per SIMD with the whole 512-register file (launch_bounds(256, 1)), 128 KB
//     of LDS for the two rings of 16-KB tiles.
//   * with nobody else on the SIMD to cover a wave's stalls, nothing is left in front of a segment's chunk stream: every
//     chunk requests the LDS operand of the MFMA FOUR chunks later (K fragments, then V^T fragments, then the first K
//     fragments of the next segment's slot), the tile maximum of the freshly produced scores is folded two per chunk as a
//     running v_max3 chain behind the PV MFMAs, and the ones-MFMA row sums of a P fragment are issued in the block's own
//     softmax segment as soon as the fragment is packed.  (At head_dim 64, two waves per SIMD, the same restructuring was
//     measured 2 % SLOWER -- profiles/r03_attn_headless_variants.log: a partner wave already covers the heads there and the
//     chunks are issue-bound.  Here the VALU has slack and there is no partner.)
#include <type_traits>

#include "attention.h"

namespace ltxmi {

namespace pipe128 {

constexpr int DH = 128;
constexpr int KV_TILE = 64;
constexpr int ROW_BYTES = DH * 2;                // 256
constexpr int TILE_BYTES = KV_TILE * DH * 2;     // 16 KiB: one K or V tile
constexpr int RING = 4;
constexpr int SMEM = 2 * RING * TILE_BYTES;      // K ring | V ring = 128 KiB (+ 16 B: the redo flag)
constexpr int Q_PER_WG = 256;                    // 4 waves x 2 blocks x 32 rows
constexpr int NS = DH / 16;                      // k-steps of a QK^T product (8)
constexpr int ND = DH / 32;                      // 32-row blocks of O^T (4)
constexpr int NCH = 2 * NS + 4 * ND;             // chunks (MFMAs) per segment: 16 QK^T + 16 PV
constexpr int PIECES = TILE_BYTES / 1024 / 4;    // LDS-DMA pieces of a tile per wave (4)

struct Blk {
    f32x16 s[2];     // S^T accumulators: keys 0..31 / 32..63 of the tile (rows), query on the lane
    f32x16 o[ND];    // O^T accumulators: head-dim rows 32 d .. 32 d + 31
    u32x4 pf[4];     // P^T fragments of the last finished softmax (B operand of the PV product), packed bf16 pairs
    bf16x8 q[NS];    // Q^T fragments (B operand of the QK^T product), k-steps of 16
    f32x4 l;         // row sums (ones-MFMA accumulator): lanes 0..15, registers 0 / 1 = query n / n + 16
    float m;         // running row max (raw scores)
    float mt;        // maximum of the scores in s (this block's pending tile), over both lane halves
};

struct Lane {
    int lane, r, hh;
    int koff[8];     // byte offset inside a K slot of this lane's 16 bytes of row r for k-step s (row 32 + r: + 32 * ROW_BYTES):
                     // r * 256 + ((2 s + hh) ^ (r & 15)) * 16 -- kept in registers: every read is then base + offset + immediate
    int v_rd;        // byte offset of this lane's transposed-read address inside a V sub-tile
};

// K image: row r = 256 B, 16-byte chunk c at slot c ^ (r & 15): the four 16-lane groups of a ds_read_b128 (lanes
// {0-3,12-15,20-27} ...) then touch 16 distinct slots each.  V image: [8 key][32 col] sub-tiles of 512 B, sub-tile
// (key >> 3) * 4 + (col >> 5) -- attention.hip's image with four column blocks instead of two.
__device__ __forceinline__ bf16x8 kread(const char* slot, const Lane& L, int i) {
    // fragment i: key block kb = i >> 3 (rows 32 kb + r), k-step s = i & 7 (head-dim 16 s + 8 hh ..)
    return *(const bf16x8*)(slot + L.koff[i & 7] + (i >> 3) * (32 * ROW_BYTES));
}
__device__ __forceinline__ bf16x8 vread(const char* slot, const Lane& L, int i) {
    // V^T fragment of PV MFMA i: k-step sp = i >> 2 (keys 16 sp ..), head-dim block d = i & 3
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const char* base = slot + L.v_rd + ((2 * (i >> 2)) * 4 + (i & 3)) * 512;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * 512));
    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

// Register placement is pinned by hand.  The state of two blocks is ~300 registers; what only the matrix pipe touches --
// the O^T accumulators (2 x 64) and the Q^T fragments (2 x 32) -- lives in the accumulation registers (AGPRs), everything the
// VALU works on (scores, P, LDS operands, softmax state) in the 256 architectural VGPRs.  hipcc cannot be talked into that
// split through the MFMA builtins (it put the Q^T fragments and LDS offsets into AGPRs and copied four registers in front
// of every QK^T MFMA: ~110 v_accvgpr moves per key tile on a loop whose one wave per SIMD is issue-bound), so the loop's
// MFMAs are issued from inline asm with "a" / "v" constraints.  hipcc does not know these statements are MFMAs and pads no
// hazards around them: every use of their results by the VALU is a segment away (the scores), and operands written by the
// VALU (P fragments) are a chunk or more old when an MFMA reads them.
//
// The running-max RESCALE of O^T is VALU work on the accumulators, and ANY such code inside the loop -- however rarely
// executed -- makes hipcc route one block's 64 accumulators through VGPRs on the hot path (64 v_accvgpr_write per key tile).
// So the loop exists twice: the EXACT form (rescale whenever a row maximum grows) runs the first two key tiles, where the
// maxima are still settling; the STEADY form then keeps each row's reference fixed -- P = 2^((s - m) c) may exceed 1, which
// costs nothing in fp32 / bf16 floating point as long as it cannot overflow -- and contains no rescale and no row maximum
// at all.  A score ~100 bits (69 nats) or more above everything in the row's first 128 keys leaves a row sum or an
// accumulator of magnitude >= 2^100 (up to non-finite); a workgroup that finds one after its last key tile redoes its item
// with the exact form throughout.
__device__ __forceinline__ void mfma_s0(f32x16& acc, const bf16x8& k, const bf16x8& q) {      // acc = K Q^T (first k-step)
    asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(k), "a"(q));
}
__device__ __forceinline__ void mfma_s(f32x16& acc, const bf16x8& k, const bf16x8& q) {       // acc += K Q^T
    asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(k), "a"(q));
}
__device__ __forceinline__ void mfma_o(f32x16& acc, const bf16x8& vt, const bf16x8& pt) {     // acc += V^T P^T
    asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(vt), "v"(pt));
}
template <int N>
__device__ __forceinline__ void settle_o(f32x16 (&o)[N]) {
    // 16-pass MFMA result -> v_accvgpr_read / VALU: 18 wait states, software-inserted; the asm "uses" every accumulator so
    // that no read of them can be scheduled in front of it
    static_assert(N == 4, "four O^T blocks");
    asm volatile("s_nop 15\n\ts_nop 7" : "+a"(o[0]), "+a"(o[1]), "+a"(o[2]), "+a"(o[3]));
}
// One segment: VALU = softmax of X's pending scores (X.s -> X.pf, X.m, rescale of X.o / X.l) and the running maximum of the
// scores Y produces; matrix pipe = Y's next scores (kq: fragments 0..3 of this segment's K slot, read during the previous
// segment; the rest from ks_cur), PV with Y's pending P against the V slot vs, and the row sums of the P fragments as they
// complete.  On return kq holds fragments 0..3 of ks_next.  key0_y: first key of the tile whose scores Y receives; a
// ragged tile is masked where its scores are produced (a wave-uniform branch).  hook(j): the kernel's LDS-DMA issue points.
template <bool EXACT, typename Hook>
__device__ __forceinline__ void segment(Blk& X, Blk& Y, bf16x8 (&kq)[4], const char* ks_cur, const char* ks_next, const char* vs,
                                        const Lane& L, const bf16x8& ones, float c, int key0_y, int Lk, Hook&& hook) {
    if (EXACT) {
        const float m_new = fmaxf(X.m, X.mt);
        // the O-wide rescale: a real wave-uniform branch
        if (__any(m_new != X.m)) {
            asm volatile("; rescale branch (kept a real branch: not if-converted)" ::: "memory");
            const float alpha = fast_exp2((X.m - m_new) * c);
            X.l[0] *= alpha;
            X.l[1] *= __shfl(alpha, (L.lane + 16) & 63, 64);
            settle_o(X.o);
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) X.o[d][e] *= alpha;
            X.m = m_new;
        }
    }
    // (steady form: the reference stays, nothing to do here -- and no tile maximum is computed at all: an overflow against
    // the fixed reference shows as a non-finite row sum / output at the end of the item)
    const float nmoff = -X.m * c;
    __builtin_amdgcn_sched_barrier(0);

    bf16x8 f[NCH];                                  // f[j] = the LDS operand of chunk j's MFMA (static indices: registers)
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = kq[i];
    float pa = 0.f, pb = 0.f;
    float mt = -INFINITY;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        if (j < 2 * NS) {
            const int kb = j / NS, s = j % NS;
            if (s == 0) mfma_s0(Y.s[kb], f[j], Y.q[s]);
            else mfma_s(Y.s[kb], f[j], Y.q[s]);
        } else {
            const int jj = j - 2 * NS, sp = jj / ND, d = jj % ND;
            mfma_o(Y.o[d], f[j], __builtin_bit_cast(bf16x8, Y.pf[sp]));
        }
        {
            const int cn = j + 4;                   // the chunk whose operand is requested now
            if (cn < 2 * NS) f[cn] = kread(ks_cur, L, cn);
            else if (cn < NCH) f[cn] = vread(vs, L, cn - 2 * NS);
            else kq[cn - NCH] = kread(ks_next, L, cn - NCH);
        }
        // row sums on the matrix pipe (one 16x16x32 against the masked all-ones operand per P fragment): Y's last fragment
        // of its previous softmax, then X's fragments 0..2 right after the chunk that packed their last pair
        if (j == 1) Y.l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, __builtin_bit_cast(bf16x8, Y.pf[3]), Y.l, 0, 0, 0);
        if (j == 9 || j == 17 || j == 25)
            X.l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, __builtin_bit_cast(bf16x8, X.pf[(j - 9) >> 3]), X.l, 0, 0, 0);
        hook(j);
        // softmax of one score per chunk; a pair (even score in pa, odd score in pb) is packed at the next EVEN chunk, i.e.
        // one chunk after its second exp2 (a v_cvt_pk right behind the v_exp it reads costs an s_nop: transcendental ->
        // VALU hazard)
        const float p = fast_exp2(__builtin_fmaf(X.s[j >> 4][j & 15], c, nmoff));
        if ((j & 1) == 0) {
            if (j >= 2) {
                const int jp = j / 2 - 1;           // scores 2 jp, 2 jp + 1
                // (the empty asm pins the conversion to this chunk: instruction selection otherwise gathers the
                // conversions of a segment behind its last MFMA)
                uint32_t pw = pack_bf16(pa, pb);
                asm volatile("" : "+v"(pw));
                X.pf[2 * (jp >> 3) + ((jp & 7) >> 2)][jp & 3] = pw;
            }
            pa = p;
        } else {
            pb = p;
        }
        // running maximum of Y's new scores, two per chunk (Y.s[0] is complete since chunk 7, Y.s[1] since chunk 15)
        if (j >= 2 * NS) {
            const int jm = j - 2 * NS, kbm = jm >> 3, em = 2 * (jm & 7);
            if (em == 0 && key0_y + KV_TILE > Lk) {
                asm volatile("; ragged key tile (kept a real branch)" ::: "memory");
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = key0_y + 32 * kbm + (e & 3) + 8 * (e >> 2) + 4 * L.hh;
                    if (key >= Lk) Y.s[kbm][e] = -INFINITY;
                }
            }
            if (EXACT) {
                mt = fmaxf(fmaxf(mt, Y.s[kbm][em]), Y.s[kbm][em + 1]);
                asm volatile("" : "+v"(mt));        // (pinned to this chunk, like the conversions)
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    X.pf[3][3] = pack_bf16(pa, pb);
    if (EXACT) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
        Y.mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
}

__global__ __launch_bounds__(256, 1) void attn_pipe128_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kring = smem;
    char* vring = smem + RING * TILE_BYTES;

    const int tid = threadIdx.x;
    Lane L;
    L.lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    L.r = L.lane & 31;
    L.hh = L.lane >> 5;

    // ---- XCD-aware work id (bijective chunking): an XCD walks whole (batch, head) pairs
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, qn = nwg >> 3, rn = nwg & 7;
    const int work = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (orig >> 3);
    const int bh = work / p.q_tiles, qt = work % p.q_tiles;
    const int b = bh / p.H, head = bh % p.H;

    const uint16_t* qb = p.q + (int64_t)b * p.q_sb + head * DH;
    const uint16_t* kb_ = p.k + (int64_t)b * p.k_sb + head * DH;
    const uint16_t* vb = p.v + (int64_t)b * p.v_sb + head * DH;
    uint16_t* ob = p.o + (int64_t)b * p.o_sb + head * DH;

    // ---- LDS-DMA sources (attention_pipe.hip): one descriptor per operand, key rows past Lk arrive as zeros; issued from
    // inline asm so that hipcc does not drain the ring in front of every transposed LDS read
    auto make_desc = [](const void* base, int64_t bytes) {
        const uint64_t a = (uint64_t)base;
        return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, (uint32_t)bytes, 0x00020000u};
    };
    const u32x4 k_desc = make_desc(kb_, ((int64_t)(p.Lk - 1) * p.k_sl + DH) * 2);
    const u32x4 v_desc = make_desc(vb, ((int64_t)(p.Lk - 1) * p.v_sl + DH) * 2);
    // wave w moves pieces 4w .. 4w+3 (1 KiB each) of every K and V tile.
    //   K piece P: rows 4P .. 4P+3; lane l writes row l >> 4, slot l & 15 <- source chunk (l & 15) ^ (row & 15)
    //   V piece P: keys 8 (P >> 1) .. + 7, columns 64 (P & 1) .. + 63 = two adjacent sub-tiles (1 KiB, contiguous in the
    //              image: sub-tile index (P >> 1) * 4 + (P & 1) * 2 -> byte offset P * 1024); lane l writes sub-tile l >> 5,
    //              key (l >> 2) & 7, 16-byte chunk l & 3
    uint32_t k_voff[PIECES], v_voff[PIECES];
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
        const int piece = PIECES * wave + j;
        const int krow = 4 * piece + (L.lane >> 4);
        const int kchunk = (L.lane & 15) ^ (krow & 15);
        k_voff[j] = (uint32_t)((krow * (int)p.k_sl + 8 * kchunk) * 2);
        const int vrow = 8 * (piece >> 1) + ((L.lane >> 2) & 7);
        const int vchunk = 8 * (piece & 1) + 4 * (L.lane >> 5) + (L.lane & 3);
        v_voff[j] = (uint32_t)((vrow * (int)p.v_sl + 8 * vchunk) * 2);
    }
    const uint32_t k_tile_step = (uint32_t)(KV_TILE * (int)p.k_sl * 2), v_tile_step = (uint32_t)(KV_TILE * (int)p.v_sl * 2);
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    auto dma1 = [&](const u32x4& desc, uint32_t lds_addr, uint32_t voff) {
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "buffer_load_dwordx4 %1, %3, 0 offen lds\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(lds_addr), "s"(desc)
            : "memory");
    };
    // piece i (0..3) of this wave's share of K tile t / V tile t
    auto dma_k1 = [&](int t, int i) {
        const uint32_t dst = lds0 + (uint32_t)((t & (RING - 1)) * TILE_BYTES + (PIECES * wave + i) * 1024);
        dma1(k_desc, dst, k_voff[i] + (uint32_t)t * k_tile_step);
    };
    auto dma_v1 = [&](int t, int i) {
        const uint32_t dst = lds0 + (uint32_t)((RING + (t & (RING - 1))) * TILE_BYTES + (PIECES * wave + i) * 1024);
        dma1(v_desc, dst, v_voff[i] + (uint32_t)t * v_tile_step);
    };
    auto dma_k = [&](int t) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) dma_k1(t, i);
    };
    auto dma_v = [&](int t) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) dma_v1(t, i);
    };

    const int nt = (p.Lk + KV_TILE - 1) / KV_TILE;
    // ---- per-lane LDS read offsets
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) L.koff[s8] = L.r * ROW_BYTES + (((2 * s8 + L.hh) ^ (L.r & 15)) << 4);   // swz(32 + r) == swz(r)
    {
        const int g16 = L.lane >> 4, i16 = L.lane & 15;
        L.v_rd = (4 * (g16 >> 1) + (i16 >> 2)) * 64 + (16 * (g16 & 1) + 4 * (i16 & 3)) * 2;
    }

    // ---- state.  Block B's "pending" P of tile -1 is zero and multiplies V slot 3, which is zero-filled.
    Blk A, Bk;
    bf16x8 ones;
    {
        const bool on = ((L.lane & 15) == 0 && ((L.lane >> 4) & 1) == 0) || ((L.lane & 15) == 1 && ((L.lane >> 4) & 1) == 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = on ? (__bf16)1.0f : (__bf16)0.0f;
    }
    auto init = [&](Blk& X, int blk) {
        const int row = qt * Q_PER_WG + wave * 64 + 32 * blk + L.r;
        const int q_ld = row < p.Lq ? row : p.Lq - 1;
#pragma unroll
        for (int s = 0; s < NS; ++s) X.q[s] = *(const bf16x8*)(qb + (int64_t)q_ld * p.q_sl + 16 * s + 8 * L.hh);
        if (p.q_on_load()) {
            // fused q_norm (+ RoPE) on load: the arithmetic of rmsnorm_rope_kernel (rowops.hip), see attention_pipe.hip
            const float rstd = p.q_row_rstd(b, q_ld, p.H * DH);
            const int64_t trow = (int64_t)b * p.rope_sb + (int64_t)q_ld * p.rope_sl;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int col = head * DH + 16 * s + 8 * L.hh;
                const bf16x8 wv = *(const bf16x8*)(p.q_w + col);
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (float)X.q[s][e] * rstd * (float)wv[e];
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
                for (int e = 0; e < 8; ++e) X.q[s][e] = (__bf16)o[e];
            }
        }
        X.m = -INFINITY;
        X.mt = -INFINITY;
        X.l = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) X.pf[i] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) X.o[d][e] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) X.s[kb][e] = 0.f;
    };
    const float c = p.scale_log2e;
    volatile int* redo_flag = (volatile int*)(smem + SMEM);         // one word behind the rings
    if (tid == 0) *redo_flag = 0;
    // attempt 0: exact form for the first two key tiles, steady form for the rest; attempt 1 (only if some wave of the
    // workgroup saw a score outgrow its reference): the whole item again in the exact form
    for (int attempt = p.force_exact ? 1 : 0; attempt < 2; ++attempt) {
        if (attempt == 1 && !p.force_exact && p.redo_count != nullptr && tid == 0) atomicAdd(p.redo_count, 1u);
        dma_k(0); dma_v(0); dma_k(1); dma_v(1); dma_k(2);
        init(A, 0);
        init(Bk, 1);
        // (Q^T fragments pinned to the accumulation registers: every use below is an "a" operand)
#pragma unroll
        for (int s8 = 0; s8 < NS; ++s8) {
            asm volatile("" : "+a"(A.q[s8]));
            asm volatile("" : "+a"(Bk.q[s8]));
        }
        {
            const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < TILE_BYTES / (256 * 16); ++i) *(u32x4*)(vring + 3 * TILE_BYTES + (i * 256 + tid) * 16) = z;
        }
        // (the builtin, not asm: hipcc must KNOW the Q loads have landed, or it waits for them with a counted vmcnt at their
        // first use inside the loop -- which then drains the LDS-DMA ring every iteration)
        __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);

        // S_A(0), its maximum, and fragments 0..3 of K slot 0 kept for segment 1 of iteration 0 (block B's first scores)
        bf16x8 kq[4];
#pragma unroll
        for (int j = 0; j < 2 * NS; ++j) {
            const bf16x8 kfj = kread(kring, L, j);
            if (j % NS == 0) mfma_s0(A.s[j / NS], kfj, A.q[j % NS]);
            else mfma_s(A.s[j / NS], kfj, A.q[j % NS]);
            if (j < 4) kq[j] = kfj;
        }
        {
            float mt = -INFINITY;
            // (MFMA results read by the VALU: the hazard window is not padded for asm MFMAs)
            asm volatile("s_nop 15\n\ts_nop 7" : "+v"(A.s[0]), "+v"(A.s[1]));
#pragma unroll
            for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (p.Lk < KV_TILE) {
                        const int key = 32 * kb2 + (e & 3) + 8 * (e >> 2) + 4 * L.hh;
                        if (key >= p.Lk) A.s[kb2][e] = -INFINITY;
                    }
                    mt = fmaxf(mt, A.s[kb2][e]);
                }
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
            A.mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }

        // iteration t: segment 1 produces block B's scores of tile t, segment 2 block A's of tile t + 1 (tile nt: every key
        // out of range, never used).  This iteration's eight LDS-DMA pieces -- K(t+3) into the slot K(t-1) left at this
        // barrier, V(t+2) into V(t-2)'s -- go out one at a time behind an MFMA, in this order (the counted wait relies on it).
        auto iteration = [&](int t, auto exact_tag) __attribute__((always_inline)) {
            constexpr bool EXACT = decltype(exact_tag)::value;
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const char* k_t = kring + (t & 3) * TILE_BYTES;
            const char* k_t1 = kring + ((t + 1) & 3) * TILE_BYTES;
            segment<EXACT>(A, Bk, kq, k_t, k_t1, vring + ((t + 3) & 3) * TILE_BYTES, L, ones, c, t * KV_TILE, p.Lk,
                           [&](int j) { if ((j & 7) == 3) dma_k1(t + 3, j >> 3); });
            segment<EXACT>(Bk, A, kq, k_t1, k_t1, vring + (t & 3) * TILE_BYTES, L, ones, c, (t + 1) * KV_TILE, p.Lk,
                           [&](int j) { if ((j & 7) == 3) dma_v1(t + 2, j >> 3); });
            // everything issued before this iteration's eight pieces has landed: K(t+2), V(t+1)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        const int t_exact = attempt == 0 ? (nt < 2 ? nt : 2) : nt;
        int t = 0;
        for (; t < t_exact; ++t) iteration(t, std::integral_constant<bool, true>{});
        for (; t < nt; ++t) iteration(t, std::integral_constant<bool, false>{});

        // ---- drain: block B's last P fragment's row sums and its PV
        {
            const char* vs = vring + ((nt - 1) & 3) * TILE_BYTES;
            Bk.l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, __builtin_bit_cast(bf16x8, Bk.pf[3]), Bk.l, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4 * ND; ++i) mfma_o(Bk.o[i % ND], vread(vs, L, i), __builtin_bit_cast(bf16x8, Bk.pf[i / ND]));
        }
        // the out-of-range pieces of tiles >= nt are still landing (as zeros): drain them before the rings are reused (as
        // the output scratch, or by the redo's stream); and agree on the redo
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        bool outgrown = false;
        if (attempt == 0 && t_exact < nt) {
            // did a score outgrow its row's fixed reference?  Then a row sum or an accumulator has magnitude >= 2^100 (or is
            // inf / NaN: the test is on the exponent bits, the file is built with -fno-honor-nans).  Not only overflow:
            // 1 / l for l > 2^126 is a denormal and flushes to zero; a legitimate l is at most (keys) x 2^(a few bits).
            constexpr uint32_t OUTGROWN_EXP = (127u + 100u) << 23;
            settle_o(A.o);
            settle_o(Bk.o);
            uint32_t worst = 0;
            auto scan = [&](const Blk& X) {
                worst |= (uint32_t)((__float_as_uint(X.l[0]) & 0x7f800000u) >= OUTGROWN_EXP);
                worst |= (uint32_t)((__float_as_uint(X.l[1]) & 0x7f800000u) >= OUTGROWN_EXP);
#pragma unroll
                for (int d = 0; d < ND; ++d)
#pragma unroll
                    for (int e = 0; e < 16; ++e) worst |= (uint32_t)((__float_as_uint(X.o[d][e]) & 0x7f800000u) >= OUTGROWN_EXP);
            };
            scan(A);
            scan(Bk);
            outgrown = __any(worst != 0);
        }
        if (outgrown && L.lane == 0) *redo_flag = 1;
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int redo = *redo_flag;
        __builtin_amdgcn_s_barrier();
        if (attempt == 1 || redo == 0) break;
    }


    // ---- epilogue: O = O^T / l, through a per-wave LDS scratch so that rows leave whole (attention.hip): 32 rows of 256 B,
    // 16-byte chunks XOR-swizzled by the row
    auto store = [&](Blk& X, int blk) {
        settle_o(X.o);
        const float l0 = __shfl(X.l[0], L.r & 15, 64), l1 = __shfl(X.l[1], L.r & 15, 64);
        const float inv = 1.0f / ((L.r & 16) ? l1 : l0);
        char* scr = smem + (wave * 2 + blk) * (32 * ROW_BYTES);
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 w;
                w[0] = pack_bf16(X.o[d][4 * g + 0] * inv, X.o[d][4 * g + 1] * inv);
                w[1] = pack_bf16(X.o[d][4 * g + 2] * inv, X.o[d][4 * g + 3] * inv);
                const int chunk = 4 * d + g;
                *(u32x2*)(scr + L.r * ROW_BYTES + ((chunk ^ (L.r & 15)) << 4) + L.hh * 8) = w;
            }
        const int q0 = qt * Q_PER_WG + wave * 64 + 32 * blk;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int row = t * 4 + (L.lane >> 4), chunk = L.lane & 15;
            const u32x4 w = *(const u32x4*)(scr + row * ROW_BYTES + ((chunk ^ (row & 15)) << 4));
            if (q0 + row < p.Lq) *(u32x4*)(ob + p.o_row(q0 + row) + chunk * 8) = w;
        }
    };
    store(A, 0);
    store(Bk, 1);
}

}  // namespace pipe128

bool attn_pipe128_takes(int B, int H, int Lq, int Lk, int head_dim, bool has_bias) {
    // one 256-row workgroup per CU: from about half a chip's worth of workgroups on, and with enough key tiles for the
    // ring's prologue / drain to amortise (the 512-key text cross-attention of config 4 stays on the register-staged kernel)
    return head_dim == 128 && !has_bias && (int64_t)B * H * ((Lq + 255) / 256) >= 128 && Lk >= 1024;
}

int launch_attn_pipe128(AttnParams p, hipStream_t stream) {
    // the buffer descriptors address a (batch, head)'s K / V rows with 32-bit byte offsets
    if (!attn_pipe_span_ok(p.Lk, p.k_sl, p.v_sl, pipe128::DH)) return -1;
    auto kern = pipe128::attn_pipe128_kernel;
    static unsigned long long lds_done = 0;
    if (const int rc = reserve_lds((const void*)kern, pipe128::SMEM + 16, &lds_done, "ltxmi_attention_fwd_bf16")) return rc;
    p.q_tiles = (p.Lq + pipe128::Q_PER_WG - 1) / pipe128::Q_PER_WG;
    const int64_t grid = (int64_t)p.B * p.H * p.q_tiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), pipe128::SMEM + 16, stream, p);
    return check_launch("ltxmi_attention_fwd_bf16");
}

}  // namespace ltxmi
