// attention_pipe.hip -- software-pipelined flash attention forward for the large self-attention
// shapes of the DiT (head_dim 64, no key bias): the kernel behind pay_attention()'s eager branch
// (wan/modules/attention.py:99-116,344-347 of the reference) when there is enough work to fill the chip.
//
// Same arithmetic as attention.hip (swapped product S^T = K Q^T with the query on the lane, online
// softmax in registers, O^T += V^T P^T with P straight from the accumulator registers, row sums on
// the matrix pipe) -- what changes is WHEN things are issued:
//
//   * A wave owns two 32-row query blocks A and B and runs them half a key tile apart.  One key
//     tile is two segments; in each the VALU does the softmax of one block while the matrix pipe
//     runs the other block's products:
//         segment 1:  softmax A(t)      ||  row sums B(t-1), QK^T B(t),   PV B(t-1)
//         segment 2:  softmax B(t)      ||  row sums A(t),   QK^T A(t+1), PV A(t)
//     At head_dim 64 the textbook softmax (exp2 + fma + cvt per score, tile maxima) needs MORE VALU issue
//     cycles than the MFMAs of the same scores need pipe cycles, so (a) every MFMA is issued with its share
//     of the VALU work behind it and neither pipe ever waits for a whole phase of the other -- the
//     interleave is written out in the source, chunk by chunk, and pinned with sched_barrier -- and
//     (b) the normal run does no VALU work per score beyond exp2 + cvt: q arrives scaled by
//     softmax_scale * log2(e), and P = 2^s is taken against the reference 0 for every row (see the kernel).
//   * K / V tiles go HBM -> LDS by LDS-DMA (buffer_load ... lds, 1 KiB per wave-instruction) into
//     two rings of four 8-KiB slots, K three tiles ahead and V two, behind a counted vmcnt: no
//     staging registers, no ds_write, and no VMEM/LDS-store issue slots taken from the softmax.
//     The swizzles of both LDS images (attention.hip) are applied to the per-lane SOURCE address.
//   * One s_barrier per key tile.
#include <type_traits>

#include "attention.h"

namespace ltxmi {

namespace pipe {

// Diagnostic build only (-DLTXMI_ATTN_STAMPS, tools/attn_stamps.py): s_memtime stamps around the sections of
// an iteration, summed per wave into a debug buffer.  No stamp executes in the product build.
#ifdef LTXMI_ATTN_STAMPS
#define STAMP(i)                                                                               \
    do {                                                                                       \
        unsigned long long t_;                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        st.acc[i] += t_ - st.prev;                                                             \
        st.prev = t_;                                                                          \
    } while (0)
struct Stamps { unsigned long long prev; unsigned long long acc[8]; };
__device__ unsigned long long* g_attn_stamps = nullptr;
#else
#define STAMP(i) do { } while (0)
struct Stamps {};
#endif

constexpr int DH = 64;
constexpr int KV_TILE = 64;
constexpr int ROW_BYTES = DH * 2;
constexpr int TILE_BYTES = KV_TILE * DH * 2;     // 8 KiB: one K or V tile
constexpr int RING = 4;
constexpr int SMEM = 2 * RING * TILE_BYTES;      // K ring | V ring = 64 KiB (+ 16 B behind them: the redo flag)
constexpr int Q_PER_WG = 256;                    // 4 waves x 2 blocks x 32 rows

struct Blk {
    f32x16 s[2];     // S^T accumulators: keys 0..31 / 32..63 of the tile (rows), query on the lane
    f32x16 o[2];     // O^T accumulators: head-dim rows 0..31 / 32..63
    u32x4 pf[4];     // P^T fragments of the last finished softmax (B operand of the PV product), packed bf16 pairs
    bf16x8 q[4];     // Q^T fragments (B operand of the QK^T product), k-steps of 16
    f32x4 l;         // row sums (ones-MFMA accumulator): lanes 0..15, registers 0 / 1 = query n / n + 16
    float m;         // running row max (raw scores)
};

struct Lane {
    int lane, r, hh;
    int k_rd[2];     // byte offset of K row 32 kb + r inside a K slot
    int k_sw0;       // swizzle of that row
    int v_rd;        // byte offset of this lane's transposed-read address inside a V slot
};

// K fragment j of a slot: key block j >> 2 (32 keys), k-step j & 3 (16 head-dim columns)
__device__ __forceinline__ bf16x8 k_fragment(const char* slot, const Lane& L, int j) {
    return *(const bf16x8*)(slot + L.k_rd[j >> 2] + (((2 * (j & 3) + L.hh) ^ L.k_sw0) << 4));
}

// One segment: VALU = softmax of X's pending scores (X.s -> X.pf, X.m, rescale of X.o / X.l);
// matrix pipe = Y's row sums and PV with Y's pending P against the V slot `vs`, and Y's next
// scores against the K slot `ks`.
// hook(j) runs once per chunk j, right behind the chunk's MFMA: the kernel uses it to issue its LDS-DMA pieces one at a
// time in the shadow of an MFMA instead of as a burst of four behind the barrier
// QSCALED: the scores arrive in bits (q was scaled by c = softmax_scale * log2(e) before its rounding to bf16, see the
// kernel); otherwise they are raw and c is applied here.
// STEADY: P = 2^(c s) against the fixed reference 0 (see the kernel): no tile maximum, no rescale, no subtraction.
template <bool STEADY, bool QSCALED, typename Hook>
__device__ __forceinline__ void segment(Blk& X, Blk& Y, bf16x8 (&kf)[8], const char* ks_next, const char* vs, const Lane& L,
                                        const bf16x8& ones, float c, int key0, int Lk, Stamps& st, int st0, Hook&& hook) {
    // ---- head: Y's row sums, X's row max.  kf = the K fragments of this segment's slot, requested one per chunk by the
    // second half of the PREVIOUS segment (whose K registers are free from its chunk 8 on); this segment does the same for the
    // next one's slot `ks_next` -- a whole-slot burst of 8 ds_read_b128 at the head had every wave wait out the LDS latency here
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) Y.l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, __builtin_bit_cast(bf16x8, Y.pf[sp]), Y.l, 0, 0, 0);

    if (key0 + KV_TILE > Lk) {
        asm volatile("; ragged key tile (a wave-uniform branch, not a template instance)" ::: "memory");
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = key0 + 32 * kb + (e & 3) + 8 * (e >> 2) + 4 * L.hh;
                if (key >= Lk) X.s[kb][e] = -INFINITY;
            }
    }
    if (!STEADY) {
        float mt;
        {
            auto max3 = [](float a, float b, float c3) { return fmaxf(fmaxf(a, b), c3); };
            float l1[11];
#pragma unroll
            for (int g = 0; g < 5; ++g) {
                l1[g] = max3(X.s[0][3 * g], X.s[0][3 * g + 1], X.s[0][3 * g + 2]);
                l1[5 + g] = max3(X.s[1][3 * g], X.s[1][3 * g + 1], X.s[1][3 * g + 2]);
            }
            l1[10] = max3(X.s[0][15], X.s[1][15], l1[0]);
            const float a = max3(l1[1], l1[2], l1[3]), b2 = max3(l1[4], l1[5], l1[6]), c2 = max3(l1[7], l1[8], l1[9]);
            mt = fmaxf(max3(a, b2, c2), l1[10]);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
            mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const float m_new = fmaxf(X.m, mt);
        // the O-wide rescale is a real wave-uniform branch
        if (__any(m_new != X.m)) {
            asm volatile("; rescale branch (kept a real branch: not if-converted)" ::: "memory");
            const float alpha = fast_exp2(QSCALED ? X.m - m_new : (X.m - m_new) * c);
            X.l[0] *= alpha;
            X.l[1] *= __shfl(alpha, (L.lane + 16) & 63, 64);
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) X.o[d][e] *= alpha;
            X.m = m_new;
        }
    }
    const float nmoff = STEADY ? 0.f : QSCALED ? -X.m : -X.m * c;
    __builtin_amdgcn_sched_barrier(0);
    STAMP(st0);

    // ---- 16 chunks: one 32x32x16 MFMA, the LDS reads of a later MFMA, and two scores' worth of
    // softmax (2 exp2, 1 cvt_pk; + 2 v_mul for raw scores; the exact form: + 2 v_sub / v_fma) each
    bf16x8 vf[8];
    float pp0 = 0.f, pp1 = 0.f;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j < 8) {
            const int kb = j >> 2, s = j & 3;
            if (s == 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) Y.s[kb][e] = 0.f;
            }
            Y.s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[j], Y.q[s], Y.s[kb], 0, 0, 0);
            // V^T fragment of PV MFMA j (k-step sp = j >> 1, head-dim block d = j & 1)
            const char* base = vs + L.v_rd + (2 * (j >> 1) * 2 + (j & 1)) * 512;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 2 * 512));
            const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            vf[j] = __builtin_bit_cast(bf16x8, both);
        } else {
            const int jj = j - 8, sp = jj >> 1, d = jj & 1;
            Y.o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[jj], __builtin_bit_cast(bf16x8, Y.pf[sp]), Y.o[d], 0, 0, 0);
            kf[jj] = k_fragment(ks_next, L, jj);
        }
        hook(j);
        // softmax of two scores; the pair is packed one chunk later (a v_cvt_pk right behind the v_exp
        // it reads costs an s_nop: transcendental -> VALU hazard)
        const int kb = j >> 3, e0 = 2 * (j & 7);
        auto bits = [&](float sc) { return QSCALED ? (STEADY ? sc : sc + nmoff) : (STEADY ? sc * c : __builtin_fmaf(sc, c, nmoff)); };
        const float p0 = fast_exp2(bits(X.s[kb][e0]));
        const float p1 = fast_exp2(bits(X.s[kb][e0 + 1]));
        if (j > 0) {
            const int jp = j - 1, kbp = jp >> 3, ep = 2 * (jp & 7);
            // (the empty asm pins the conversion to this chunk: instruction selection otherwise gathers
            // all sixteen v_cvt_pk of a segment behind its last MFMA)
            uint32_t pw = pack_bf16(pp0, pp1);
            asm volatile("" : "+v"(pw));
            X.pf[2 * kbp + (ep >> 3)][(ep & 7) >> 1] = pw;
        }
        pp0 = p0;
        pp1 = p1;
        __builtin_amdgcn_sched_barrier(0);
    }
    X.pf[3][3] = pack_bf16(pp0, pp1);
    STAMP(st0 + 1);
}

// One work item = one (batch, head, 256-row query tile).  REDO = false: the normal run -- EVERY key tile in the steady form
// (P = 2^s against the fixed reference 0: no tile maximum, no running maximum, no rescale; t_exact = 0 below), then the
// range check on row sums and accumulators: a row whose scores left about +-100 bits (69 nats) has a sum or accumulator
// outside [2^-100, 2^100) -- the item then returns true WITHOUT having stored anything and the workgroup goes through it
// again.  REDO = true: the exact (textbook online-softmax) form for every key tile, always stores.
// QSCALED = q is produced on load (the fused K1 below): softmax_scale * log2(e) goes into it before its one rounding to bf16
// and the scores leave the matrix pipe in bits.  A q that arrives as bf16 is left as it is (scaling it would round it a second
// time, an error proportional to the score: visible once logits reach tens of nats) and its scores take one v_mul each.
template <bool REDO, bool QSCALED>
__device__ __forceinline__ bool attn_pipe_item(const AttnParams& p, const int tid, char* smem) {
    char* kring = smem;
    char* vring = smem + RING * TILE_BYTES;

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

    // ---- LDS-DMA sources.  One descriptor per operand (base = this (batch, head)'s first row,
    // num_records = up to the end of its last row): key rows past Lk are out of range and arrive as zeros.
    // The DMA is issued from inline asm: for a builtin LDS-DMA hipcc puts s_waitcnt vmcnt(0) in front of the
    // next transposed LDS read (it cannot tell the slots apart), which would drain the ring every segment.
    // Completion is counted by hand instead (vmcnt(4) at the end of an iteration, then the barrier).
    auto make_desc = [](const void* base, int64_t bytes) {
        const uint64_t a = (uint64_t)base;
        return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, (uint32_t)bytes, 0x00020000u};
    };
    const u32x4 k_desc = make_desc(kb_, ((int64_t)(p.Lk - 1) * p.k_sl + DH) * 2);
    const u32x4 v_desc = make_desc(vb, ((int64_t)(p.Lk - 1) * p.v_sl + DH) * 2);
    // wave w moves pieces 2w, 2w+1 (8 key rows = 1 KiB each) of every K and V tile.
    //   K image: row r, 16-byte chunk c at slot c ^ ((r >> 1) & 7); lane l of a piece writes row l >> 3, slot l & 7
    //   V image: [8 key][32 col] sub-tiles of 512 B; lane l writes sub-tile l >> 5 (column half), key (l >> 2) & 7, chunk l & 3
    uint32_t k_voff[2], v_voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int piece = 2 * wave + j;
        const int krow = 8 * piece + (L.lane >> 3);
        const int kchunk = (L.lane & 7) ^ ((krow >> 1) & 7);
        k_voff[j] = (uint32_t)((krow * (int)p.k_sl + 8 * kchunk) * 2);
        const int vrow = 8 * piece + ((L.lane >> 2) & 7);
        const int vchunk = 4 * (L.lane >> 5) + (L.lane & 3);
        v_voff[j] = (uint32_t)((vrow * (int)p.v_sl + 8 * vchunk) * 2);
    }
    const uint32_t k_tile_step = (uint32_t)(KV_TILE * (int)p.k_sl * 2), v_tile_step = (uint32_t)(KV_TILE * (int)p.v_sl * 2);
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    // two pieces (this wave's share of one tile) per statement; M0 = LDS byte address of the piece
    auto dma2 = [&](const u32x4& desc, uint32_t lds_addr, uint32_t voff0, uint32_t voff1) {
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "buffer_load_dwordx4 %1, %4, 0 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "s_nop 0\n\t"
            "buffer_load_dwordx4 %2, %4, 0 offen lds\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff0), "v"(voff1), "s"(lds_addr), "s"(desc)
            : "memory", "scc");
    };
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
    // piece i (0 / 1) of this wave's share of K tile t / V tile t
    auto dma_k1 = [&](int t, int i) {
        const uint32_t dst = lds0 + (uint32_t)((t & (RING - 1)) * TILE_BYTES + (2 * wave + i) * 1024);
        dma1(k_desc, dst, k_voff[i] + (uint32_t)t * k_tile_step);
    };
    auto dma_v1 = [&](int t, int i) {
        const uint32_t dst = lds0 + (uint32_t)((RING + (t & (RING - 1))) * TILE_BYTES + (2 * wave + i) * 1024);
        dma1(v_desc, dst, v_voff[i] + (uint32_t)t * v_tile_step);
    };
    auto dma_k = [&](int t) {
        const uint32_t dst = lds0 + (uint32_t)((t & (RING - 1)) * TILE_BYTES + 2 * wave * 1024);
        const uint32_t toff = (uint32_t)t * k_tile_step;
        dma2(k_desc, dst, k_voff[0] + toff, k_voff[1] + toff);
    };
    auto dma_v = [&](int t) {
        const uint32_t dst = lds0 + (uint32_t)((RING + (t & (RING - 1))) * TILE_BYTES + 2 * wave * 1024);
        const uint32_t toff = (uint32_t)t * v_tile_step;
        dma2(v_desc, dst, v_voff[0] + toff, v_voff[1] + toff);
    };

    const int nt = (p.Lk + KV_TILE - 1) / KV_TILE;

    // ---- per-lane LDS read offsets (attention.hip's images)
    L.k_rd[0] = L.r * ROW_BYTES;
    L.k_rd[1] = (32 + L.r) * ROW_BYTES;
    L.k_sw0 = (L.r >> 1) & 7;                       // swz(32 + r) == swz(r)
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
    const float c = p.scale_log2e;
    auto init = [&](Blk& X, int blk) {
        const int row = qt * Q_PER_WG + wave * 64 + 32 * blk + L.r;
        const int q_ld = row < p.Lq ? row : p.Lq - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) X.q[s] = *(const bf16x8*)(qb + (int64_t)q_ld * p.q_sl + 16 * s + 8 * L.hh);
        if (QSCALED) {                     // == p.q_on_load(): the launcher picks the instance by it
            // fused K1: q is the raw projection output.  q_norm (RMSNorm over all H * dh channels, attention.py:478-479,
            // 1040-1041) from the row's factor (finalised per row by k's pass, or the projection GEMM's partial sums of
            // squares), x weight, then the interleaved-pair RoPE on the flat channel axis (:960-975, 1053-1055) -- the
            // arithmetic of rmsnorm_rope_kernel (rowops.hip), one rounding to bf16 at the end.
            const float rstd = p.q_row_rstd(b, q_ld, p.H * DH);
            const int64_t trow = (int64_t)b * p.rope_sb + (int64_t)q_ld * p.rope_sl;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
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
                // (x softmax_scale * log2(e) before the one rounding: the scores leave the matrix pipe in bits)
#pragma unroll
                for (int e = 0; e < 8; ++e) X.q[s][e] = (__bf16)(QSCALED ? o[e] * c : o[e]);
            }
        }
        X.m = -INFINITY;
        X.l = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) X.pf[i] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) { X.o[d][e] = 0.f; X.s[d][e] = 0.f; }
    };
    // The loop exists in two forms.  STEADY, the normal run: P = 2^s with every row's reference at 0 -- softmax is invariant
    // under the choice of reference, and fp32 / bf16 carry the same RELATIVE precision at every magnitude, so as long as
    // nothing leaves the exponent range the result is the textbook one to rounding.  No tile maximum, no running maximum,
    // no subtraction, no rescale branch: per score the VALU issues an exp2 and half a cvt_pk, on a loop whose VALU port was
    // fuller than its matrix pipe.  What can go wrong is range: a score above ~ +100 bits (69 nats) overflows a row sum /
    // accumulator, a row whose scores ALL lie below ~ -100 bits loses its sum to underflow (1 / l overflows).  Both leave a
    // row sum or accumulator outside [2^-100, 2^100), which the workgroup checks for after its last key tile -- and then
    // redoes the item in the EXACT form (REDO): the textbook online softmax, tile maximum, running maximum, rescale of O / l
    // whenever a row's maximum grows.
    volatile int* redo_flag = (volatile int*)(smem + SMEM);           // one word behind the rings
    if (!REDO && tid == 0) *redo_flag = 0;
    const bool wave_idle = qt * Q_PER_WG + wave * 64 >= p.Lq;
    // the two workgroups of a CU otherwise fall into step (both in their segment heads / at their barriers together): a static
    // priority for every other dispatch round breaks the symmetry (+0.9 % at the workload shape, nothing at long sequences)
    if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(1);
    Stamps st;
#ifdef LTXMI_ATTN_STAMPS
    unsigned long long rt0_ = 0;
#endif
    dma_k(0); dma_v(0); dma_k(1); dma_v(1); dma_k(2);
    init(A, 0);
    init(Bk, 1);
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
        *(u32x4*)(vring + 3 * TILE_BYTES + tid * 32) = z;
        *(u32x4*)(vring + 3 * TILE_BYTES + tid * 32 + 16) = z;
    }
    // (the builtin, not asm: hipcc must KNOW the Q loads have landed, or it waits for them with a counted
    // vmcnt at their first use inside the loop -- which then drains the LDS-DMA ring every iteration)
    __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // A wave whose 64 query rows all lie past Lq (the half-empty last query tile of a (batch, head): N = 4992 = 19.5 x 256)
    // keeps the workgroup's K/V stream going -- its LDS-DMA pieces, the barriers, the counted waits -- and computes nothing,
    // so its SIMD partner (a wave of the other resident workgroup) gets the pipes to itself.
#ifdef LTXMI_ATTN_STAMPS
    for (int i = 0; i < 8; ++i) st.acc[i] = 0;
    { unsigned long long t0_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_)::"memory"); st.prev = t0_; }
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0_)::"memory");
#endif
    const int t_exact = REDO ? nt : 0;
    if (wave_idle) {
        for (int t = 0; t < nt; ++t) {
            __builtin_amdgcn_s_barrier();
            dma_k(t + 3);
            dma_v(t + 2);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
    } else {
        // S_A(0); the fragments of K slot 0 stay in kf for segment 1 of iteration 0 (S_B(0))
        bf16x8 kf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) kf[j] = k_fragment(kring, L, j);
#pragma unroll
        for (int j = 0; j < 8; ++j) A.s[j >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[j], A.q[j & 3], A.s[j >> 2], 0, 0, 0);

        using exact_form = std::integral_constant<bool, false>;
        using steady_form = std::integral_constant<bool, true>;
        auto iteration = [&](int t, auto steady_tag) {
            constexpr bool STEADY = decltype(steady_tag)::value;
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            STAMP(0);
            STAMP(1);
            // this iteration's four LDS-DMA pieces -- K(t+3) into the slot K(t-1) left at this barrier, V(t+2) into V(t-2)'s --
            // go out one at a time behind an MFMA (chunks 3 and 11 of each segment), in this order (the counted wait below
            // relies on it): a burst of four behind the barrier cost ~200 cycles per iteration with the wave issuing nothing else
            segment<STEADY, QSCALED>(A, Bk, kf, kring + ((t + 1) & 3) * TILE_BYTES, vring + ((t + 3) & 3) * TILE_BYTES, L, ones, c, t * KV_TILE, p.Lk, st, 2,
                          [&](int j) { if (j == 3) dma_k1(t + 3, 0); else if (j == 11) dma_k1(t + 3, 1); });
            segment<STEADY, QSCALED>(Bk, A, kf, kring + ((t + 1) & 3) * TILE_BYTES, vring + (t & 3) * TILE_BYTES, L, ones, c, t * KV_TILE, p.Lk, st, 4,
                          [&](int j) { if (j == 3) dma_v1(t + 2, 0); else if (j == 11) dma_v1(t + 2, 1); });
            // everything issued before this iteration's four pieces has landed: K(t+2), V(t+1)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            STAMP(6);
        };
        int t = 0;
        for (; t < t_exact; ++t) iteration(t, exact_form{});
        if (!REDO)
            for (; t < nt; ++t) iteration(t, steady_form{});
    }

#ifdef LTXMI_ATTN_STAMPS
    unsigned long long rt1_;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1_)::"memory");
    if (g_attn_stamps && L.lane == 0 && blockIdx.x < 4096 && !wave_idle) {
        for (int i = 0; i < 7; ++i) g_attn_stamps[(blockIdx.x * 4 + wave) * 8 + i] = st.acc[i];
        // low 32 bits: key tiles; high 32 bits: elapsed s_memrealtime ticks (100 MHz) of the loop
        g_attn_stamps[(blockIdx.x * 4 + wave) * 8 + 7] = (unsigned long long)nt | ((rt1_ - rt0_) << 32);
    }
#endif
    // ---- drain: block B's last row sums and PV
    if (!wave_idle) {
        const char* vs = vring + ((nt - 1) & 3) * TILE_BYTES;
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) Bk.l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, __builtin_bit_cast(bf16x8, Bk.pf[sp]), Bk.l, 0, 0, 0);
        typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll
        for (int sp = 0; sp < 4; ++sp)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const char* base = vs + L.v_rd + (2 * sp * 2 + d) * 512;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 2 * 512));
                const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                Bk.o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, both), __builtin_bit_cast(bf16x8, Bk.pf[sp]), Bk.o[d], 0, 0, 0);
            }
    }
    // the out-of-range pieces of tiles >= nt are still landing (as zeros): drain them before the rings are reused (as the
    // output scratch, or by the redo's stream); then agree on the redo
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (!REDO) {
    bool outgrown = false;
    if (t_exact < nt && !wave_idle) {
        // a row sum or an accumulator of magnitude >= 2^100 (or inf / NaN: the test is on the exponent bits, the file is
        // built with -fno-honor-nans) = a score too large for the fixed reference (not only overflow: 1 / l for l > 2^126 is
        // a denormal and flushes to zero); a row sum below 2^-100 (or 0) = a row whose scores all underflowed.
        constexpr uint32_t OUTGROWN_EXP = (127u + 100u) << 23, VANISHED_EXP = (127u - 100u) << 23;
        uint32_t worst = 0;
        auto scan = [&](const Blk& X) {
            worst |= (uint32_t)((__float_as_uint(X.l[0]) & 0x7f800000u) >= OUTGROWN_EXP);
            worst |= (uint32_t)((__float_as_uint(X.l[1]) & 0x7f800000u) >= OUTGROWN_EXP);
            // (the row sums live in lanes 0..15, registers 0 / 1; a sum of 0 has exponent bits 0)
            if (L.lane < 16) {
                worst |= (uint32_t)((__float_as_uint(X.l[0]) & 0x7f800000u) < VANISHED_EXP);
                worst |= (uint32_t)((__float_as_uint(X.l[1]) & 0x7f800000u) < VANISHED_EXP);
            }
#pragma unroll
            for (int d = 0; d < 2; ++d)
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
    if (redo != 0) return true;
    } else {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: O = O^T / l, through a per-wave LDS scratch so that rows leave whole (attention.hip)
    auto store = [&](Blk& X, int blk) {
        const float l0 = __shfl(X.l[0], L.r & 15, 64), l1 = __shfl(X.l[1], L.r & 15, 64);
        const float inv = 1.0f / ((L.r & 16) ? l1 : l0);
        char* scr = smem + (wave * 2 + blk) * (32 * ROW_BYTES);
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 w;
                w[0] = pack_bf16(X.o[d][4 * g + 0] * inv, X.o[d][4 * g + 1] * inv);
                w[1] = pack_bf16(X.o[d][4 * g + 2] * inv, X.o[d][4 * g + 3] * inv);
                const int chunk = 4 * d + g;
                *(u32x2*)(scr + L.r * ROW_BYTES + ((chunk ^ (L.r & 7)) << 4) + L.hh * 8) = w;
            }
        const int q0 = qt * Q_PER_WG + wave * 64 + 32 * blk;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = t * 8 + (L.lane >> 3), chunk = L.lane & 7;
            const u32x4 w = *(const u32x4*)(scr + row * ROW_BYTES + ((chunk ^ (row & 7)) << 4));
            if (q0 + row < p.Lq) *(u32x4*)(ob + p.o_row(q0 + row) + chunk * 8) = w;
        }
    };
    store(A, 0);
    store(Bk, 1);
    return false;
}

// FORCE_EXACT (diagnostic, ltxmi_attn_args.force_exact): every item straight in the exact form -- an instance of its own,
// chosen by the launcher.  (As a run-time branch in front of the normal run it made hipcc share state between the two forms of
// the item: 157 spilled registers instead of 19, the launch twice as slow.)
template <int OCC, bool QSCALED, bool FORCE_EXACT>
__global__ __launch_bounds__(256, OCC) void attn_pipe_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    if (FORCE_EXACT) {
        attn_pipe_item<true, QSCALED>(p, tid, smem);
        return;
    }
    if (!attn_pipe_item<false, QSCALED>(p, tid, smem)) return;
    // (rare) the item again, exact form throughout.  The thread id is laundered through an empty asm so that nothing the first
    // run derived from it is kept alive -- i.e. spilled -- across its loops for this path's sake: everything is recomputed.
    int tid2 = tid;
    asm volatile("" : "+v"(tid2));
    if (p.redo_count != nullptr && tid2 == 0) atomicAdd(p.redo_count, 1u);
    attn_pipe_item<true, QSCALED>(p, tid2, smem);
}

}  // namespace pipe

#ifdef LTXMI_ATTN_STAMPS
extern "C" int ltxmi_debug_set_attn_stamps(void* buf) {
    unsigned long long* b = (unsigned long long*)buf;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(pipe::g_attn_stamps), &b, sizeof(b));
}
#endif

#ifndef LTXMI_ATTN_PIPE_OCC
#define LTXMI_ATTN_PIPE_OCC 2
#endif

bool attn_pipe_takes(int B, int H, int Lq, int Lk, int head_dim, bool has_bias) {
    // from 192 workgroups of 256 rows (of the chip's 512 slots) this kernel beats attention.hip's: measured +11 % at 240
    // and +20 % at 480 workgroups (B 3, N 4992 with 4 / 8 heads: what a rank sees in the Ulysses mode at P = 8 / 4)
    return head_dim == 64 && !has_bias && (int64_t)B * H * ((Lq + 255) / 256) >= 192 && Lk > 0;
}

int launch_attn_pipe(AttnParams p, hipStream_t stream) {
    // the buffer descriptors address a (batch, head)'s K / V rows with 32-bit byte offsets
    if (!attn_pipe_span_ok(p.Lk, p.k_sl, p.v_sl, pipe::DH)) return -1;
    const bool qscaled = p.q_on_load();
    auto kern = p.force_exact ? (qscaled ? pipe::attn_pipe_kernel<LTXMI_ATTN_PIPE_OCC, true, true> : pipe::attn_pipe_kernel<LTXMI_ATTN_PIPE_OCC, false, true>)
                              : (qscaled ? pipe::attn_pipe_kernel<LTXMI_ATTN_PIPE_OCC, true, false> : pipe::attn_pipe_kernel<LTXMI_ATTN_PIPE_OCC, false, false>);
    static unsigned long long lds_done[4] = {0, 0, 0, 0};
    if (const int rc = reserve_lds((const void*)kern, pipe::SMEM + 16, &lds_done[(p.force_exact ? 2 : 0) + qscaled], "ltxmi_attention_fwd_bf16")) return rc;
    p.q_tiles = (p.Lq + pipe::Q_PER_WG - 1) / pipe::Q_PER_WG;
    const int64_t grid = (int64_t)p.B * p.H * p.q_tiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), pipe::SMEM + 16, stream, p);
    return check_launch("ltxmi_attention_fwd_bf16");
}

}  // namespace ltxmi
