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
};

// attention_pipe.hip: software-pipelined self-attention (head_dim 64, no key bias).
// Returns -1 when the shape is not taken (the caller then uses attention.hip's kernel).
int launch_attn_pipe(AttnParams p, hipStream_t stream);

}  // namespace ltxmi
