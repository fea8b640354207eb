// abi_host_asan.cpp -- drives the HOST side of every libltxmi entry point under AddressSanitizer.
//
// Built by `make -C ltx-video-gpupoor_amd/csrc asan` against libltxmi_asan.so (host code instrumented, device code
// not: the GPU pool has no sanitizer support) and run by tests/test_native_asan.py with NO device visible
// (HIP_VISIBLE_DEVICES=-1), so nothing is ever launched: a call either is refused by the argument checks
// (LTXMI_ERR_INVALID_ARG / _UNSUPPORTED) or runs the whole launcher -- kernel choice, grid and descriptor
// arithmetic, per-device state -- up to the HIP call that reports the missing device (LTXMI_ERR_LAUNCH).
// ASan aborts the process on any host-side out-of-bounds access on the way; the driver itself checks that
// no call reports success and that every failure left a message in ltxmi_last_error().
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/ltxmi.h"

static int g_calls = 0, g_bad = 0;

static void expect_fail(int rc, const char* what, int want = 0) {
    ++g_calls;
    const char* msg = ltxmi_last_error();
    const bool ok = rc < 0 && (want == 0 || rc == want) && msg && msg[0];
    if (!ok) {
        ++g_bad;
        fprintf(stderr, "UNEXPECTED %s: rc %d (wanted %s), message '%s'\n", what, rc, want ? "that code" : "< 0", msg ? msg : "(null)");
    }
}

int main() {
    // 32 MiB of host memory: every pointer handed over is valid HOST memory, never dereferenced by host code
    std::vector<uint16_t> buf(16u << 20, 0);
    void* p = buf.data();
    float* pf = reinterpret_cast<float*>(buf.data());

    if (strcmp(ltxmi_arch(), "gfx950") != 0 || !strstr(ltxmi_version(), "ltxmi")) {
        fprintf(stderr, "version/arch strings\n");
        return 2;
    }

    // ---- GEMM
    expect_fail(ltxmi_gemm_bf16(nullptr, nullptr), "gemm(NULL)", LTXMI_ERR_INVALID_ARG);
    ltxmi_gemm_args g;
    memset(&g, 0, sizeof g);
    expect_fail(ltxmi_gemm_bf16(&g, nullptr), "gemm(zeroed)");
    g.A = p; g.W = p; g.C = p; g.lda = 2048; g.ldw = 2048; g.ldc = 6144; g.M = 4992; g.N = 6144; g.K = 2048;
    for (int algo : {0, 128, 256, 7}) {
        g.algo = algo;
        expect_fail(ltxmi_gemm_bf16(&g, nullptr), "gemm(valid shape, no device)");
    }
    g.algo = 0;
    g.K = 2047;
    expect_fail(ltxmi_gemm_bf16(&g, nullptr), "gemm(K % 8)");
    g.K = 2048;
    g.epilogue = 99;
    expect_fail(ltxmi_gemm_bf16(&g, nullptr), "gemm(bad epilogue)");
    g.epilogue = LTXMI_EPI_GATE_RESIDUAL;                       // residual missing
    expect_fail(ltxmi_gemm_bf16(&g, nullptr), "gemm(gate without residual)");
    g.epilogue = 0;
    g.rowsumsq = pf; g.rowsumsq_cols = 2048; g.rowsumsq_ld = 32;
    expect_fail(ltxmi_gemm_bf16(&g, nullptr), "gemm(rowsumsq, no device)");
    g.rowsumsq_cols = 2000;
    expect_fail(ltxmi_gemm_bf16(&g, nullptr), "gemm(rowsumsq_cols % 64)");
    // a caller that leaves the tail fields of an older struct unset: NULL pointer, garbage cols / ld -- must be
    // treated as "off" (the launcher zeroes them), never as a store descriptor based at address 0
    g.rowsumsq = nullptr; g.rowsumsq_cols = 0x7fffff00; g.rowsumsq_ld = -12345;
    expect_fail(ltxmi_gemm_bf16(&g, nullptr), "gemm(NULL rowsumsq, garbage cols/ld, no device)", LTXMI_ERR_LAUNCH);
    g.rowsumsq = nullptr; g.rowsumsq_cols = 0; g.rowsumsq_ld = 0;
    g.a_kblock = 1024; g.a_kblock_stride = 4992 * 1024; g.lda = 1024;
    expect_fail(ltxmi_gemm_bf16(&g, nullptr), "gemm(K-blocked A, no device)");
    g.a_kblock = 1000;
    expect_fail(ltxmi_gemm_bf16(&g, nullptr), "gemm(a_kblock not dividing K)");
    for (int M : {1, 63, 128, 9984, 32760}) {                  // every kernel-choice branch
        memset(&g, 0, sizeof g);
        g.A = p; g.W = p; g.C = p; g.lda = 512; g.ldw = 512; g.ldc = 512; g.M = M; g.N = 512; g.K = 512;
        expect_fail(ltxmi_gemm_bf16(&g, nullptr), "gemm(M sweep, no device)");
    }

    // ---- attention
    expect_fail(ltxmi_attention_fwd_bf16(nullptr, nullptr), "attention(NULL)", LTXMI_ERR_INVALID_ARG);
    ltxmi_attn_args a;
    memset(&a, 0, sizeof a);
    expect_fail(ltxmi_attention_fwd_bf16(&a, nullptr), "attention(zeroed)");
    a.q = a.k = a.v = p; a.o = p;
    a.B = 3; a.H = 32; a.Lq = a.Lk = 4992; a.head_dim = 64; a.softmax_scale = 0.125f;
    a.q_stride_l = a.k_stride_l = a.v_stride_l = 6144; a.q_stride_b = a.k_stride_b = a.v_stride_b = 4992 * 6144;
    a.o_stride_l = 2048; a.o_stride_b = 4992 * 2048;
    expect_fail(ltxmi_attention_fwd_bf16(&a, nullptr), "attention(pipelined shape, no device)");
    a.head_dim = 96;
    expect_fail(ltxmi_attention_fwd_bf16(&a, nullptr), "attention(head_dim 96)", LTXMI_ERR_UNSUPPORTED);
    a.head_dim = 128; a.H = 12;
    expect_fail(ltxmi_attention_fwd_bf16(&a, nullptr), "attention(head_dim 128, no device)");
    a.head_dim = 64; a.H = 32; a.Lk = 77; a.key_bias = pf; a.bias_stride_b = 77;
    expect_fail(ltxmi_attention_fwd_bf16(&a, nullptr), "attention(cross, bias, no device)");
    a.key_bias = nullptr; a.Lk = 4992;
    a.q_rowsumsq = pf; a.q_rowsumsq_blocks = 31;                // != H * dh / 64
    a.q_norm_weight = p;
    expect_fail(ltxmi_attention_fwd_bf16(&a, nullptr), "attention(rowsumsq blocks mismatch)");
    a.q_rowsumsq_blocks = 32; a.q_rowsumsq_stride_l = 32; a.q_rowsumsq_stride_b = 4992 * 32;
    a.rope_cos = p; a.rope_sin = p; a.rope_stride_l = 2048;
    expect_fail(ltxmi_attention_fwd_bf16(&a, nullptr), "attention(fused q norm, no device)");
    a.Lq = a.Lk = 100;                                          // the generic kernel
    expect_fail(ltxmi_attention_fwd_bf16(&a, nullptr), "attention(fused q norm on a small shape, no device)");
    a.q_rowsumsq = nullptr; a.q_rstd = pf; a.q_rstd_stride_l = 1; a.q_rstd_stride_b = 100;   // the finalised row factor
    expect_fail(ltxmi_attention_fwd_bf16(&a, nullptr), "attention(q_rstd, no device)");
    a.q_norm_weight = nullptr;
    expect_fail(ltxmi_attention_fwd_bf16(&a, nullptr), "attention(q_rstd without weight)", LTXMI_ERR_INVALID_ARG);
    a.q_rstd = nullptr; a.rope_cos = a.rope_sin = nullptr;
    a.Lq = a.Lk = 4992; a.o_segment_len = 1000;                 // does not divide Lq
    expect_fail(ltxmi_attention_fwd_bf16(&a, nullptr), "attention(o segments not dividing Lq)");
    if (ltxmi_attention_fuses_qnorm(3, 32, 4992, 4992, 64, 0) != 1 || ltxmi_attention_fuses_qnorm(1, 32, 64, 64, 64, 1) != 1 ||
        ltxmi_attention_fuses_qnorm(3, 12, 4992, 4992, 128, 0) != 1 || ltxmi_attention_fuses_qnorm(3, 12, 4992, 4992, 96, 0) != 0) {
        fprintf(stderr, "ltxmi_attention_fuses_qnorm\n");
        ++g_bad;
    }

    if (ltxmi_attention_kernel_id(3, 32, 4992, 4992, 64, 0, 6144, 6144) != 3 || ltxmi_attention_kernel_id(1, 4, 100, 100, 64, 0, 768, 768) != 0 ||
        ltxmi_attention_kernel_id(3, 32, 4992, 256, 64, 1, 4096, 4096) != 7 ||         // short key sequences: K / V resident in LDS
        ltxmi_attention_kernel_id(3, 32, 512, 256, 64, 1, 4096, 4096) != 1 ||          // ... from 1024 query rows
        ltxmi_attention_kernel_id(1, 12, 32760, 32760, 128, 0, 4608, 4608) != 6 ||     // the head_dim-128 pipelined kernel
        ltxmi_attention_kernel_id(1, 2, 520, 520, 128, 0, 768, 768) != 4 || ltxmi_attention_kernel_id(1, 12, 32760, 512, 128, 1, 1536, 1536) != 5 ||
        // a token stride that puts a (batch, head)'s rows 2 GiB apart: the pipelined kernels hand the shape over, and the
        // id says so (the query used to answer 3 regardless)
        ltxmi_attention_kernel_id(3, 32, 4992, 4992, 64, 0, 400000, 6144) != 2 ||
        ltxmi_attention_kernel_id(1, 12, 100, 100, 96, 0, 1152, 1152) != -1) {
        fprintf(stderr, "ltxmi_attention_kernel_id\n");
        ++g_bad;
    }

    // ---- row ops
    expect_fail(ltxmi_norm_modulate_bf16(nullptr, 0, nullptr, 0, 0, 0, 1e-6f, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr), "norm_modulate(NULL)");
    expect_fail(ltxmi_norm_modulate_bf16(p, 2048, p, 2048, 4992, 2048, 1e-6f, LTXMI_NORM_RMS, p, p, p, p, 2048 * 6, 4992, nullptr), "norm_modulate(no device)");
    expect_fail(ltxmi_norm_modulate_bf16(p, 2048, p, 2048, 4992, 2044, 1e-6f, LTXMI_NORM_RMS, p, p, p, p, 2048 * 6, 4992, nullptr), "norm_modulate(D % 8)");
    expect_fail(ltxmi_norm_modulate_bf16(p, 2048, p, 2048, 4992, 2048, 1e-6f, 5, p, p, p, p, 2048 * 6, 4992, nullptr), "norm_modulate(bad kind)");
    expect_fail(ltxmi_rmsnorm_rope_bf16(nullptr, 0, 0, 0, nullptr, 0.f, nullptr, nullptr, 0, 0, nullptr), "rmsnorm_rope(NULL)");
    expect_fail(ltxmi_rmsnorm_rope_bf16(p, 6144, 4992, 2048, p, 1e-6f, p, p, 2048, 4992, nullptr), "rmsnorm_rope(no device)");
    expect_fail(ltxmi_rmsnorm_rope_bf16(p, 6144, 4992, 2048, p, 1e-6f, p, nullptr, 2048, 4992, nullptr), "rmsnorm_rope(cos without sin)");
    expect_fail(ltxmi_rmsnorm_rope_rstd_bf16(p, 6144, 4992, 2048, p, 1e-6f, p, p, 2048, 4992, nullptr, 32, 32, 2048, 1e-6f, pf, nullptr), "rmsnorm_rope_rstd(NULL sums)", LTXMI_ERR_INVALID_ARG);
    expect_fail(ltxmi_rmsnorm_rope_rstd_bf16(p, 6144, 4992, 2048, p, 1e-6f, p, p, 2048, 4992, pf, 16, 32, 2048, 1e-6f, pf, nullptr), "rmsnorm_rope_rstd(ld < blocks)", LTXMI_ERR_INVALID_ARG);
    expect_fail(ltxmi_rmsnorm_rope_rstd_bf16(p, 6144, 4992, 2048, p, 1e-6f, p, p, 2048, 4992, pf, 32, 32, 2048, 1e-6f, pf, nullptr), "rmsnorm_rope_rstd(no device)");
    expect_fail(ltxmi_rowsumsq_rstd_f32(nullptr, 32, 32, 4992, 2048, 1e-6f, pf, nullptr), "rowsumsq_rstd(NULL)", LTXMI_ERR_INVALID_ARG);
    expect_fail(ltxmi_rowsumsq_rstd_f32(pf, 16, 32, 4992, 2048, 1e-6f, pf, nullptr), "rowsumsq_rstd(ld < blocks)", LTXMI_ERR_INVALID_ARG);
    expect_fail(ltxmi_rowsumsq_rstd_f32(pf, 32, 32, 4992, 2048, 1e-6f, pf, nullptr), "rowsumsq_rstd(no device)");
    expect_fail(ltxmi_qkv_norm_rope_pack_bf16(nullptr, 0, 0, 0, 0, 0, nullptr, nullptr, 0.f, nullptr, nullptr, 0, 0, nullptr, nullptr), "pack(NULL)");
    expect_fail(ltxmi_qkv_norm_rope_pack_bf16(p, 6144, 3, 2496, 2048, 2, p, p, 1e-6f, p, p, 2048, 2496, p, nullptr), "pack(no device)");
    expect_fail(ltxmi_qkv_norm_rope_pack_bf16(p, 6144, 3, 2496, 2048, 3, p, p, 1e-6f, p, p, 2048, 2496, p, nullptr), "pack(P not dividing the heads)");
    expect_fail(ltxmi_silu_bf16(nullptr, nullptr, 0, nullptr), "silu(NULL)");
    expect_fail(ltxmi_silu_bf16(p, p, 6144, nullptr), "silu(no device)");
    expect_fail(ltxmi_add_bf16(p, p, p, 1 << 20, nullptr), "add(no device)");
    expect_fail(ltxmi_add_bf16(p, nullptr, p, 1 << 20, nullptr), "add(NULL operand)");
    expect_fail(ltxmi_timestep_embedding_bf16(pf, p, 3, 256, nullptr), "timestep_embedding(no device)");
    expect_fail(ltxmi_timestep_embedding_bf16(pf, p, 3, 0, nullptr), "timestep_embedding(dim 0)");

    // ---- convolution: both implementations, the fused stores, the encoder / upsampler extensions
    expect_fail(ltxmi_conv3d_ndhwc_bf16(nullptr, nullptr), "conv3d(NULL)", LTXMI_ERR_INVALID_ARG);
    ltxmi_conv3d_args c;
    memset(&c, 0, sizeof c);
    expect_fail(ltxmi_conv3d_ndhwc_bf16(&c, nullptr), "conv3d(zeroed)");
    c.x = p; c.w = p; c.bias = p; c.y = p;
    c.B = 1; c.T = 13; c.H = 32; c.W = 48; c.Cin = 512; c.Cout = 512; c.causal = 1;
    for (int algo : {0, 1, 2, 3, 4, 9}) {
        c.algo = algo;
        expect_fail(ltxmi_conv3d_ndhwc_bf16(&c, nullptr), "conv3d(no device)");
    }
    c.algo = 0;
    c.Cin = 100;
    expect_fail(ltxmi_conv3d_ndhwc_bf16(&c, nullptr), "conv3d(Cin % 64)", LTXMI_ERR_UNSUPPORTED);
    c.Cin = 512; c.Cout = 4096; c.d2s = 1; c.residual = p; c.res_channels = 512;
    expect_fail(ltxmi_conv3d_ndhwc_bf16(&c, nullptr), "conv3d(d2s, no device)");
    c.d2s = 0; c.residual = nullptr; c.Cout = 512; c.stride_t = 2; c.stride_hw = 2; c.out_T = 7;
    expect_fail(ltxmi_conv3d_ndhwc_bf16(&c, nullptr), "conv3d(strided, no device)");
    c.algo = 2;
    expect_fail(ltxmi_conv3d_ndhwc_bf16(&c, nullptr), "conv3d(direct asked for a strided shape)", LTXMI_ERR_UNSUPPORTED);
    c.algo = 0; c.stride_t = 3;
    expect_fail(ltxmi_conv3d_ndhwc_bf16(&c, nullptr), "conv3d(stride 3)");
    c.stride_t = c.stride_hw = 0; c.out_T = 0; c.kernel_t = 1; c.causal = 0;
    expect_fail(ltxmi_conv3d_ndhwc_bf16(&c, nullptr), "conv3d(per-frame 3x3, no device)");
    c.kernel_t = 0; c.time_pad_zeros = 1;
    expect_fail(ltxmi_conv3d_ndhwc_bf16(&c, nullptr), "conv3d(zero-padded time, no device)");

    // post_norm (0.4): fused only by the four-wave direct convolution with Cout == 128; refused elsewhere, never launched blind
    memset(&c, 0, sizeof c);
    c.x = p; c.w = p; c.bias = p; c.y = p;
    c.B = 1; c.T = 97; c.H = 128; c.W = 192; c.Cin = 128; c.Cout = 128; c.pad_replicate = 1;
    if (ltxmi_conv3d_fuses_post_norm(&c) != 1) { fprintf(stderr, "FAIL fuses_post_norm(128 -> 128 at the decoder's last stage)\n"); ++g_bad; }
    if (ltxmi_conv3d_fuses_post_norm(nullptr) != 0) { fprintf(stderr, "FAIL fuses_post_norm(NULL)\n"); ++g_bad; }
    c.post_norm = 1; c.post_scale = pf; c.post_shift = pf; c.post_eps = 1e-8f;
    expect_fail(ltxmi_conv3d_ndhwc_bf16(&c, nullptr), "conv3d(post_norm, no device)");
    c.post_shift = nullptr;
    expect_fail(ltxmi_conv3d_ndhwc_bf16(&c, nullptr), "conv3d(post_norm, scale without shift)", LTXMI_ERR_INVALID_ARG);
    c.post_shift = pf; c.Cout = 256;
    if (ltxmi_conv3d_fuses_post_norm(&c) != 0) { fprintf(stderr, "FAIL fuses_post_norm(Cout 256)\n"); ++g_bad; }
    expect_fail(ltxmi_conv3d_ndhwc_bf16(&c, nullptr), "conv3d(post_norm at Cout 256)", LTXMI_ERR_UNSUPPORTED);
    c.Cout = 128; c.algo = 1;
    expect_fail(ltxmi_conv3d_ndhwc_bf16(&c, nullptr), "conv3d(post_norm on the implicit GEMM)", LTXMI_ERR_UNSUPPORTED);
    c.algo = 0; c.add = p;
    expect_fail(ltxmi_conv3d_ndhwc_bf16(&c, nullptr), "conv3d(post_norm with add)", LTXMI_ERR_UNSUPPORTED);
    // y_norm (0.5): the norm as a SECOND output rides on conv2 + skip at Cout 128; y_norm needs post_norm and must not be y
    c.y_norm = pf;
    if (ltxmi_conv3d_fuses_post_norm(&c) != 1) { fprintf(stderr, "FAIL fuses_post_norm(add + y_norm, Cout 128)\n"); ++g_bad; }
    expect_fail(ltxmi_conv3d_ndhwc_bf16(&c, nullptr), "conv3d(add + y_norm, no device)");
    c.y_norm = c.y;
    expect_fail(ltxmi_conv3d_ndhwc_bf16(&c, nullptr), "conv3d(y_norm == y)", LTXMI_ERR_INVALID_ARG);
    c.y_norm = pf; c.post_norm = 0;
    expect_fail(ltxmi_conv3d_ndhwc_bf16(&c, nullptr), "conv3d(y_norm without post_norm)", LTXMI_ERR_INVALID_ARG);
    // workspace (0.5): the plan of the decoder's 1024-channel stage (13 x 16 x 24 positions): three channel ranges; with the
    // workspace the norm is available at that width, without it (or with too little of it) the call is the 0.4 one
    memset(&c, 0, sizeof c);
    c.x = p; c.w = p; c.bias = p; c.y = p;
    c.B = 1; c.T = 13; c.H = 16; c.W = 24; c.Cin = 1024; c.Cout = 1024; c.causal = 1; c.pad_replicate = 1;
    const int64_t want = ltxmi_conv3d_workspace_bytes(&c);
    if (want != 3ll * 13 * 16 * 24 * 1024 * 4) { fprintf(stderr, "FAIL workspace_bytes(1024 -> 1024 at 13x16x24) = %lld\n", (long long)want); ++g_bad; }
    if (ltxmi_conv3d_workspace_bytes(nullptr) != 0) { fprintf(stderr, "FAIL workspace_bytes(NULL)\n"); ++g_bad; }
    c.post_norm = 1; c.post_eps = 1e-8f;
    if (ltxmi_conv3d_fuses_post_norm(&c) != 0) { fprintf(stderr, "FAIL fuses_post_norm(Cout 1024, no workspace)\n"); ++g_bad; }
    c.workspace = pf; c.workspace_bytes = want;
    if (ltxmi_conv3d_fuses_post_norm(&c) != 1) { fprintf(stderr, "FAIL fuses_post_norm(Cout 1024, workspace)\n"); ++g_bad; }
    expect_fail(ltxmi_conv3d_ndhwc_bf16(&c, nullptr), "conv3d(split + post_norm, no device)");
    c.workspace_bytes = want - 4;
    if (ltxmi_conv3d_fuses_post_norm(&c) != 0) { fprintf(stderr, "FAIL fuses_post_norm(workspace too small)\n"); ++g_bad; }
    c.workspace = nullptr; c.workspace_bytes = want;
    expect_fail(ltxmi_conv3d_ndhwc_bf16(&c, nullptr), "conv3d(workspace_bytes without a workspace)", LTXMI_ERR_INVALID_ARG);
    c.workspace_bytes = 0; c.post_norm = 0; c.Cin = 512; c.Cout = 512; c.T = 25; c.H = 32; c.W = 48;
    if (ltxmi_conv3d_workspace_bytes(&c) != 0) { fprintf(stderr, "FAIL workspace_bytes(512 -> 512 without a norm)\n"); ++g_bad; }
    c.post_norm = 1;
    if (ltxmi_conv3d_workspace_bytes(&c) != 2ll * 25 * 32 * 48 * 512 * 4) { fprintf(stderr, "FAIL workspace_bytes(512 -> 512 with a norm)\n"); ++g_bad; }
    c.algo = 4;
    if (ltxmi_conv3d_workspace_bytes(&c) != 0) { fprintf(stderr, "FAIL workspace_bytes(eight-wave form asked for)\n"); ++g_bad; }

    // ---- VAE pointwise / layout kernels, guidance, conditioning, upsampler
    expect_fail(ltxmi_pixelnorm_ada_silu_bf16(nullptr, nullptr, 0, 0, 0, nullptr, nullptr, 0, 0.f, nullptr), "pixelnorm(NULL)");
    expect_fail(ltxmi_pixelnorm_ada_silu_bf16(p, p, 13 * 32 * 48, 512, 13 * 32 * 48, pf, pf, 1, 1e-6f, nullptr), "pixelnorm(no device)");
    expect_fail(ltxmi_pixelnorm_ada_silu_bf16(p, p, 13 * 32 * 48, 509, 13 * 32 * 48, pf, pf, 1, 1e-6f, nullptr), "pixelnorm(C % 8)");
    expect_fail(ltxmi_layernorm_affine_bf16(nullptr, nullptr, 0, 0, nullptr, nullptr, 0.f, nullptr), "layernorm(NULL)");
    expect_fail(ltxmi_layernorm_affine_bf16(p, p, 1000, 512, p, p, 1e-6f, nullptr), "layernorm(no device)");
    expect_fail(ltxmi_ncdhw_to_ndhwc_bf16(nullptr, nullptr, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr), "ncdhw_to_ndhwc(NULL)");
    expect_fail(ltxmi_ncdhw_to_ndhwc_bf16(p, p, 1, 128, 13, 32, 48, pf, pf, nullptr), "ncdhw_to_ndhwc(no device)");
    expect_fail(ltxmi_unpatchify_to_ncdhw_bf16(p, p, 1, 97, 64, 96, 3, 4, nullptr), "unpatchify(no device)");
    expect_fail(ltxmi_unpatchify_to_ncdhw_bf16(p, p, 1, 97, 64, 96, 3, 0, nullptr), "unpatchify(patch 0)");
    expect_fail(ltxmi_patchify_to_ndhwc_bf16(p, p, 1, 3, 9, 64, 64, 4, 64, nullptr), "patchify(no device)");
    expect_fail(ltxmi_patchify_to_ndhwc_bf16(p, p, 1, 3, 9, 63, 64, 4, 64, nullptr), "patchify(H % patch)");
    expect_fail(ltxmi_space_to_depth_skip_bf16(p, p, p, 1, 8, 16, 16, 128, 64, 2, 2, 2, nullptr), "space_to_depth_skip(no device)");
    expect_fail(ltxmi_space_to_depth_skip_bf16(p, nullptr, p, 1, 8, 16, 16, 128, 64, 2, 2, 2, nullptr), "space_to_depth_skip(NULL)");
    expect_fail(ltxmi_ndhwc_to_ncdhw_bf16(p, 256, 0, p, 1, 128, 2, 8, 8, pf, pf, nullptr), "ndhwc_to_ncdhw(no device)");
    expect_fail(ltxmi_ndhwc_to_ncdhw_bf16(nullptr, 256, 0, p, 1, 128, 2, 8, 8, pf, pf, nullptr), "ndhwc_to_ncdhw(NULL)");
    expect_fail(ltxmi_stg_blend_bf16(nullptr, 0, nullptr, 0, nullptr, 0, 0, 0, nullptr), "stg_blend(NULL)");
    expect_fail(ltxmi_stg_blend_bf16(p, 2048, p, 6144, pf, 3, 4992, 2048, nullptr), "stg_blend(no device)");
    expect_fail(ltxmi_stg_blend_grouped_bf16(nullptr, nullptr, 0, 0, 0, nullptr, 0, 0, 0, 0, nullptr), "stg_blend_grouped(NULL)");
    expect_fail(ltxmi_stg_blend_grouped_bf16(p, p, 256, 624 * 6144, 6144, pf, 8, 3, 624, 250, nullptr), "stg_blend_grouped(D % 8)");
    expect_fail(ltxmi_stg_blend_grouped_bf16(p, p, 256, 624 * 6144, 6144, pf, 8, 3, 624, 256, nullptr), "stg_blend_grouped(no device)");
    expect_fail(ltxmi_guidance_step_bf16(p, 4992 * 128, 3, 3.f, 1.f, 0.7f, 1, 1, 1, pf, 0, -0.1f, pf, nullptr), "guidance_step(no device)");
    expect_fail(ltxmi_guidance_step_bf16(p, 4992 * 128, 4, 3.f, 1.f, 0.7f, 1, 1, 1, pf, 0, -0.1f, pf, nullptr), "guidance_step(num_conds 4)");
    expect_fail(ltxmi_guidance_step_bf16(p, 4992 * 128, 3, 3.f, 1.f, 0.7f, 1, 1, 1, pf, 0, -0.1f, nullptr, nullptr), "guidance_step(no workspace)");
    expect_fail(ltxmi_guidance_step_masked_bf16(p, 4992 * 128, 3, 3.f, 1.f, 0.7f, 1, 1, 1, pf, 0, -0.1f, pf, 128, 0.9f, pf, nullptr), "guidance_step_masked(no device)");
    expect_fail(ltxmi_guidance_step_masked_bf16(p, 4992 * 128, 3, 3.f, 1.f, 0.7f, 1, 1, 1, pf, 0, -0.1f, nullptr, 128, 0.9f, pf, nullptr), "guidance_step_masked(no mask)");
    expect_fail(ltxmi_image_cond_noise(p, p, p, 1, pf, 4992, 128, 0.15f, 0.9f, nullptr), "image_cond_noise(no device)");
    expect_fail(ltxmi_image_cond_noise(p, nullptr, p, 1, pf, 4992, 128, 0.15f, 0.9f, nullptr), "image_cond_noise(NULL)");
    expect_fail(ltxmi_groupnorm_silu_bf16(nullptr, nullptr, nullptr, 0, 0, 0, 0, nullptr, nullptr, 0.f, nullptr, nullptr), "groupnorm(NULL)");
    expect_fail(ltxmi_groupnorm_silu_bf16(p, p, nullptr, 1, 2 * 16 * 16, 512, 32, p, p, 1e-5f, pf, nullptr), "groupnorm(no device)");
    expect_fail(ltxmi_groupnorm_silu_bf16(p, p, nullptr, 1, 2 * 16 * 16, 512, 30, p, p, 1e-5f, pf, nullptr), "groupnorm(groups not dividing C)");
    expect_fail(ltxmi_pixel_shuffle2d_ndhwc_bf16(nullptr, nullptr, 0, 0, 0, 0, nullptr), "pixel_shuffle(NULL)");
    expect_fail(ltxmi_pixel_shuffle2d_ndhwc_bf16(p, p, 2, 16, 16, 2048, nullptr), "pixel_shuffle(no device)");
    expect_fail(ltxmi_adain_filter(p, p, p, 1, 128, 2 * 16 * 16, 1 * 8 * 8, 1.0f, nullptr), "adain(no device)");
    expect_fail(ltxmi_adain_filter(p, nullptr, p, 1, 128, 2 * 16 * 16, 1 * 8 * 8, 1.0f, nullptr), "adain(NULL)");
    expect_fail(ltxmi_tile_blend(p, p, 2, 3 * 97, 512, 512, 768, 128, nullptr), "tile_blend(no device)");
    expect_fail(ltxmi_tile_blend(p, p, 1, 3 * 97, 512, 100, 768, 128, nullptr), "tile_blend(extent > length)");
    expect_fail(ltxmi_tile_blend(p, p, 3, 3 * 97, 512, 512, 768, 128, nullptr), "tile_blend(bad dtype)");
    expect_fail(ltxmi_tile_blend(nullptr, p, 1, 3 * 97, 512, 512, 768, 128, nullptr), "tile_blend(NULL)");

    printf("abi_host_asan: %d calls, %d unexpected\n", g_calls, g_bad);
    return g_bad ? 1 : 0;
}
