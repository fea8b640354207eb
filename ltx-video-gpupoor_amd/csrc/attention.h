// attention.h -- parameter block shared by the attention kernels (attention.hip, attention_pipe.hip).
#pragma once
#include "common.h"

namespace ltxmi {

struct AttnParams {
    const uint16_t* q; int64_t q_sb, q_sl;
    const uint16_t* k; int64_t k_sb, k_sl;
    const uint16_t* v; int64_t v_sb, v_sl;
    uint16_t* o; int64_t o_sb, o_sl;
    const float* bias; int64_t bias_sb;
    int B, H, Lq, Lk;
    float scale_log2e;   // softmax_scale * log2(e)
    int q_tiles;         // query tiles per (batch, head)
    // optional: q arrives as the raw projection output; RMSNorm over all H * dh channels (row sums of squares
    // given as per-64-column partials by the projection GEMM) x weight, then interleaved RoPE, applied on load
    const float* q_ss; int64_t q_ss_sb, q_ss_sl; int q_ss_n;
    // ... or, finalised (one float per row: rsqrt(mean(x^2) + eps), ltxmi_rmsnorm_rope_rstd_bf16); takes precedence
    const float* q_rstd; int64_t q_rstd_sb, q_rstd_sl;
    const uint16_t* q_w; float q_eps;
    const uint16_t* rope_cos; const uint16_t* rope_sin; int64_t rope_sb, rope_sl;
    // optional: the output's token axis is cut into segments of o_seg tokens, o_sseg elements apart (Ulysses: the
    // return all-to-all's send buffer [P dst][B][N / P][H dh]); 0 = one segment
    int o_seg; int64_t o_sseg;
    // diagnostics of the pipelined kernels' steady (reference-0) form: a device counter that every workgroup whose item had to
    // be redone in the exact form bumps once (nullptr = off), and a switch that sends EVERY item straight to the exact form
    uint32_t* redo_count; int force_exact;

    __host__ __device__ __forceinline__ bool q_on_load() const { return q_ss != nullptr || q_rstd != nullptr; }
    // q's RMSNorm factor of row `row` of batch b (HD = H * head_dim, the normalised width)
    __device__ __forceinline__ float q_row_rstd(int b, int row, int HD) const {
        if (q_rstd) return q_rstd[(int64_t)b * q_rstd_sb + (int64_t)row * q_rstd_sl];
        const float* ss = q_ss + (int64_t)b * q_ss_sb + (int64_t)row * q_ss_sl;
        float s2 = 0.f;
        if ((q_ss_n & 3) == 0 && (((uintptr_t)ss) & 15) == 0) {
            // (one 16-byte load per four partials: the row's partials are contiguous)
            const f32x4* ss4 = (const f32x4*)ss;
            for (int j = 0; j < (q_ss_n >> 2); ++j) {
                const f32x4 v4 = ss4[j];
                s2 += (v4[0] + v4[1]) + (v4[2] + v4[3]);
            }
        } else {
            for (int j = 0; j < q_ss_n; ++j) s2 += ss[j];
        }
        return rsqrtf(s2 / (float)HD + q_eps);
    }

    __device__ __forceinline__ int64_t o_row(int row) const {
        if (o_seg <= 0) return (int64_t)row * o_sl;
        const int sg = row / o_seg;
        return (int64_t)sg * o_sseg + (int64_t)(row - sg * o_seg) * o_sl;
    }
};

// attention_pipe.hip: software-pipelined self-attention (head_dim 64, no key bias).
// Returns -1 when the shape is not taken (the caller then uses attention.hip's kernel).
int launch_attn_pipe(AttnParams p, hipStream_t stream);
// whether launch_attn_pipe takes this shape
bool attn_pipe_takes(int B, int H, int Lq, int Lk, int head_dim, bool has_bias);
// attention_pipe128.hip: the same for head_dim 128 (one 4-wave workgroup per CU, 512 registers per wave)
int launch_attn_pipe128(AttnParams p, hipStream_t stream);
bool attn_pipe128_takes(int B, int H, int Lq, int Lk, int head_dim, bool has_bias);
// attention_cross.hip: short key sequences (<= 256 keys, head_dim 64; the T5 cross-attention): K / V resident in LDS
int launch_attn_cross(AttnParams p, hipStream_t stream);
bool attn_cross_takes(int B, int H, int Lq, int Lk, int head_dim);
// both pipelined kernels address a (batch, head)'s K / V rows through buffer descriptors with 32-bit byte offsets: the rows
// of all key tiles (+ the ring's run-ahead) must span less than 2 GiB
inline bool attn_pipe_span_ok(int Lk, int64_t k_sl, int64_t v_sl, int head_dim) {
    const int64_t k_span = ((int64_t)(Lk + 4 * 64) * k_sl + head_dim) * 2, v_span = ((int64_t)(Lk + 4 * 64) * v_sl + head_dim) * 2;
    return k_span < (1ll << 31) && v_span < (1ll << 31);
}

}  // namespace ltxmi
