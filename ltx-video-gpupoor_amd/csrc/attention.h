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
    const uint16_t* q_w; float q_eps;
    const uint16_t* rope_cos; const uint16_t* rope_sin; int64_t rope_sb, rope_sl;
    // optional: the output's token axis is cut into segments of o_seg tokens, o_sseg elements apart (Ulysses: the
    // return all-to-all's send buffer [P dst][B][N / P][H dh]); 0 = one segment
    int o_seg; int64_t o_sseg;

    __device__ __forceinline__ int64_t o_row(int row) const {
        if (o_seg <= 0) return (int64_t)row * o_sl;
        const int sg = row / o_seg;
        return (int64_t)sg * o_sseg + (int64_t)(row - sg * o_seg) * o_sl;
    }
};

// attention_pipe.hip: software-pipelined self-attention (head_dim 64, no key bias).
// Returns -1 when the shape is not taken (the caller then uses attention.hip's kernel).
int launch_attn_pipe(AttnParams p, hipStream_t stream);
// whether launch_attn_pipe takes this shape (the only kernel that can normalise q on load)
bool attn_pipe_takes(int B, int H, int Lq, int Lk, int head_dim, bool has_bias);

}  // namespace ltxmi
