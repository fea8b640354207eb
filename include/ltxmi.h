/*
 * ltxmi.h -- C ABI of libltxmi.so: the MI355X (gfx950) kernels of the LTX-Video
 * denoise hot path (DiT forward + causal 3-D VAE decode).
 *
 * The reference project (soasme/LTX-Video-GPUPoor) is 100 % Python and has no FFI of
 * its own: every entry point below replaces a PyTorch / third-party-wheel call made
 * by the reference at the cited file:line.  INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add at each of those sites.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless named host_*; caller-allocated;
 *     nothing is allocated, freed or synchronised inside the library;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); kernels are
 *     enqueued on it and the call returns immediately; re-entrant per stream;
 *   - bf16 tensors are raw uint16 payloads (torch.bfloat16 storage), row-major,
 *     innermost dimension contiguous; "ld*" arguments are row strides in ELEMENTS;
 *   - return value: 0 = LTXMI_OK, negative = ltxmi_status; ltxmi_last_error() returns
 *     a thread-local message for the last failing call.  Unsupported shapes are an
 *     error -- there is no fallback path of any kind;
 *   - argument structs (ltxmi_*_args) MUST be zero-initialised before the fields in use are
 *     set (`ltxmi_gemm_args a = {0};` / memset): versions append optional fields at the tail
 *     (0.2: rowsumsq*, a_kblock* of ltxmi_gemm_args; q_rowsumsq*, q_norm*, rope_*, o_segment*
 *     of ltxmi_attn_args; 0.3: q_rstd*; 0.4: conv3d post_*; 0.5: redo_counter, force_exact of ltxmi_attn_args, y_norm, workspace of ltxmi_conv3d_args), and a zero there means "off".  A caller must be
 *     rebuilt against the header of the library it loads.  An optional pointer that is NULL
 *     switches its companion size / stride fields off whatever they hold.
 */
#ifndef LTXMI_H
#define LTXMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ltxmi_status {
    LTXMI_OK = 0,
    LTXMI_ERR_INVALID_ARG = -1,   /* NULL pointer, non-positive size, bad enum            */
    LTXMI_ERR_UNSUPPORTED = -2,   /* shape / alignment outside what the kernels handle    */
    LTXMI_ERR_LAUNCH = -3         /* hipGetLastError() != hipSuccess after the launch     */
} ltxmi_status;

/* Library identification and error text. */
const char* ltxmi_version(void);
const char* ltxmi_last_error(void);
/* Name of the gfx target the kernels were compiled for ("gfx950"). */
const char* ltxmi_arch(void);

/* ---------------------------------------------------------------------------------
 * GEMM  C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias[N])       bf16 in, fp32 acc, bf16 out
 *
 * Replaces nn.Linear (+ the elementwise ops the reference runs right after it):
 *   to_q/to_k/to_v, to_out[0]      ltx_video/models/transformers/attention.py:1040-1059,1147
 *   ff.net[0] (Linear+GELU-tanh)   attention.py:339       ff.net[2]  attention.py:340
 *   gate * x ; hidden += x         attention.py:282-288, 345-351, 310
 *   patchify_proj / proj_out       ltx_video/models/transformers/transformer3d.py:418,503
 *   adaln_single / caption_projection linears   transformer3d.py:428-433,448
 * W is the nn.Linear weight as stored in the checkpoint ([out,in], K contiguous).
 * Requirements: K % 64 == 0, N % 8 == 0, all pointers 16-byte aligned, lda/ldw/ldc % 8 == 0.
 * ------------------------------------------------------------------------------- */
typedef enum ltxmi_epilogue {
    LTXMI_EPI_NONE = 0,       /* C = acc + bias                                              */
    LTXMI_EPI_GELU_TANH = 1,  /* C = gelu_tanh(acc + bias)                                   */
    LTXMI_EPI_SILU = 2,       /* C = silu(acc + bias)                                        */
    LTXMI_EPI_GATE_RESIDUAL = 3 /* C = R + gate * (acc + bias); R may alias C (in place)
                                   gate[r, n] = gate_table[n] + gate_temb[(r / rows_per_group) * gate_ld + n]
                                   (AdaLN-Zero gate, attention.py:239-246); gate_table == NULL -> gate = 1 */
} ltxmi_epilogue;

typedef struct ltxmi_gemm_args {
    const void* A;  int64_t lda;      /* [M,K] bf16                                         */
    const void* W;  int64_t ldw;      /* [N,K] bf16                                         */
    const void* bias;                 /* [N] bf16 or NULL                                   */
    void*       C;  int64_t ldc;      /* [M,N] bf16                                         */
    int32_t M, N, K;
    int32_t epilogue;                 /* ltxmi_epilogue                                     */
    const void* residual; int64_t ldr;/* [M,N] bf16, GATE_RESIDUAL only                     */
    const void* gate_table;           /* [N] bf16 (row of scale_shift_table) or NULL        */
    const void* gate_temb;            /* bf16, row g at gate_temb + g*gate_ld               */
    int64_t     gate_ld;
    int32_t     rows_per_group;       /* tokens sharing one modulation row (N_tok / T1)     */
    int32_t     algo;                 /* 0 = kernel chosen by shape (the product setting); diagnostics:
                                         128 = 128x128-tile kernel, 256 = non-persistent 256x256 kernel */
    /* Optional (plain epilogue only): for the output columns < rowsumsq_cols the epilogue also writes
     * rowsumsq[m * rowsumsq_ld + n / 64] = sum over that 64-column block of (bf16 C[m, n])^2 as fp32.
     * The fused QKV projection asks for it over the q columns: the attention kernel then applies q's
     * RMSNorm (over ALL heads, attention.py:478-479,1040-1041) + RoPE (:960-975,1053-1055) while it loads Q,
     * so q needs no pass of its own between the projection and attention.  NULL = off. */
    float*      rowsumsq; int32_t rowsumsq_cols; int64_t rowsumsq_ld;
    /* Optional: A's K axis cut into blocks of a_kblock elements that lie a_kblock_stride elements apart (element k of
     * row m at A[(k / a_kblock) * a_kblock_stride + m * lda + k % a_kblock]).  The receive buffer of the Ulysses return
     * all-to-all, [P source ranks][rows][D / P], is the A operand of to_out in place.  0 = plain row-major A. */
    int32_t     a_kblock; int64_t a_kblock_stride;
} ltxmi_gemm_args;

int ltxmi_gemm_bf16(const ltxmi_gemm_args* args, void* stream);

/* ---------------------------------------------------------------------------------
 * Fused standardisation + AdaLN modulation
 *   y = norm(x) * (1 + scale) + shift,   norm = RMSNorm (no affine) or LayerNorm (no affine)
 *   scale[r,c] = scale_table[c] + scale_temb[g*temb_ld + c],  g = r / rows_per_group (same for shift)
 * Replaces norm1/norm2 + `*= 1+scale; += shift`  attention.py:233-251, 314-320 and
 * norm_out + modulation transformer3d.py:489-502 (kind = LTXMI_NORM_LAYER).
 * D % 8 == 0, D <= 8192.
 * ------------------------------------------------------------------------------- */
typedef enum ltxmi_norm_kind { LTXMI_NORM_RMS = 0, LTXMI_NORM_LAYER = 1 } ltxmi_norm_kind;

int ltxmi_norm_modulate_bf16(const void* x, int64_t ldx, void* y, int64_t ldy,
                             int32_t rows, int32_t D, float eps, int32_t kind,
                             const void* scale_table, const void* scale_temb,
                             const void* shift_table, const void* shift_temb,
                             int64_t temb_ld, int32_t rows_per_group, void* stream);

/* ---------------------------------------------------------------------------------
 * q/k RMSNorm across heads (+weight, eps) followed by interleaved-pair RoPE on the flat
 * channel axis; in place.   attention.py:1041,1048,1052 (q_norm/k_norm = diffusers RMSNorm
 * over heads*dim_head, attention.py:478-479) and apply_rotary_emb attention.py:960-975.
 * x: [rows, D] (row stride ldx); cos/sin: [rope_rows, D] bf16 with row index
 * (r % rope_period) -- rope_period = tokens per sample when the table is shared over the
 * batch; cos == NULL -> no rotation (cross-attention).  weight: [D] bf16.
 * ------------------------------------------------------------------------------- */
int ltxmi_rmsnorm_rope_bf16(void* x, int64_t ldx, int32_t rows, int32_t D,
                            const void* weight, float eps,
                            const void* cos_tab, const void* sin_tab, int64_t ld_tab,
                            int32_t rope_period, void* stream);
/* The same pass (k of self-attention) and, riding on the launch, the RMSNorm factor of ANOTHER tensor's rows (q):
 * rstd_out[r] = rsqrt(sum_j rowsumsq[r * rowsumsq_ld + j] / norm_dim + norm_eps), j < rowsumsq_blocks -- the per-64-column
 * sums of squares ltxmi_gemm_bf16 wrote for the q columns of the fused QKV projection (attention.py:1040-1041: q_norm is an
 * RMSNorm over all heads).  ltxmi_attention_fwd_bf16 takes the result as q_rstd. */
int ltxmi_rmsnorm_rope_rstd_bf16(void* x, int64_t ldx, int32_t rows, int32_t D, const void* weight, float eps,
                                 const void* cos_tab, const void* sin_tab, int64_t ld_tab, int32_t rope_period,
                                 const float* rowsumsq, int64_t rowsumsq_ld, int32_t rowsumsq_blocks,
                                 int32_t norm_dim, float norm_eps, float* rstd_out, void* stream);

/* ---------------------------------------------------------------------------------
 * Flash attention forward (non-causal, softmax scale given), bf16 in/out, fp32 softmax.
 * Replaces pay_attention(..)'s eager branch = F.scaled_dot_product_attention
 *   wan/modules/attention.py:162-199, 344-347 -> sdpa_wrapper :99-116
 * called from AttnProcessor2_0 at attention.py:1112-1119 (self) and, with the additive
 * key bias built at transformer3d.py:411-415, for T5 cross-attention (attention.py:303-309).
 * Layout NHD like the seam: element (b, l, h, d) at  base + b*stride_b + l*stride_l + h*head_dim + d
 * (strides in elements; lets q,k,v alias slices of one fused [B,L,3*H*dh] projection buffer).
 * key_bias: optional fp32 [B, Lk] added to the scaled scores (broadcast over heads and queries).
 * head_dim in {64, 128}; Lq, Lk >= 1 (ragged tails are masked inside the kernel).
 * Kernels behind the entry point (chosen by shape, ltxmi_attention_kernel_id): the software-pipelined LDS-DMA kernels for
 * large bias-free shapes (head_dim 64: two waves per SIMD; head_dim 128: one wave per SIMD with the whole register file),
 * the short-key-sequence kernel (0.5: head_dim 64, <= 256 keys, >= 1024 queries -- the T5 cross-attention: K / V of a
 * (batch, head) resident in LDS, a query row's scores all in registers, single-pass softmax; id 7)
 * and the register-staged kernel for everything else (key bias with longer key sequences, small shapes).
 * ------------------------------------------------------------------------------- */
typedef struct ltxmi_attn_args {
    const void* q; int64_t q_stride_b, q_stride_l;
    const void* k; int64_t k_stride_b, k_stride_l;
    const void* v; int64_t v_stride_b, v_stride_l;
    void*       o; int64_t o_stride_b, o_stride_l;
    const float* key_bias; int64_t bias_stride_b;   /* NULL = no bias */
    int32_t B, H, Lq, Lk, head_dim;
    float   softmax_scale;
    /* Optional fused q_norm + RoPE on load (the "fused QKV-projection + RoPE" of the path, attention.py:1040-1055):
     * q is the RAW projection output; q_rowsumsq[b * stride_b + l * stride_l + j], j < q_rowsumsq_blocks = H*head_dim/64,
     * are the projection GEMM's per-64-column sums of squares of that row (ltxmi_gemm_args.rowsumsq).  The kernel
     * applies x * rsqrt(mean(x^2) + q_norm_eps) * q_norm_weight[c] and, if rope_cos/rope_sin are given, the
     * interleaved-pair rotation with table row b * rope_stride_b + l * rope_stride_l (strides in elements; stride_b = 0:
     * one table shared by the batch) while it loads Q.  NULL = off. */
    const float* q_rowsumsq; int64_t q_rowsumsq_stride_b, q_rowsumsq_stride_l; int32_t q_rowsumsq_blocks;
    const void*  q_norm_weight; float q_norm_eps;
    const void*  rope_cos; const void* rope_sin; int64_t rope_stride_b, rope_stride_l;
    /* Optional: the output's token axis in segments of o_segment_len tokens lying o_stride_segment elements apart
     * (token l of batch b at o[(l / o_segment_len) * o_stride_segment + b * o_stride_b + (l % o_segment_len) * o_stride_l]):
     * the send buffer of the Ulysses return all-to-all, [P destination ranks][B][N / P][H dh], written in place.  0 = off. */
    int32_t o_segment_len; int64_t o_stride_segment;
    /* 0.3 -- q's RMSNorm factor already finalised: q_rstd[b * stride_b + l * stride_l] = rsqrt(mean(x^2) + eps) of the raw
     * projection row (written by ltxmi_rmsnorm_rope_rstd_bf16 from the GEMM's partial sums, riding on k's pass).  Given,
     * it replaces q_rowsumsq (every workgroup -- one per head and query tile -- then reads 4 bytes per row instead of
     * re-summing H*head_dim/64 partials); q_norm_weight / rope_* as above.  NULL = off. */
    const float* q_rstd; int64_t q_rstd_stride_b, q_rstd_stride_l;
    /* 0.5 -- diagnostics of the pipelined kernels' normal run.  They take P = 2^s against the fixed reference 0 for every row
     * (no running maximum), which is exact as long as a row's scaled scores stay within about +-100 bits (+-69 nats): a
     * (batch, head, 256-query) item whose row sums or accumulators leave [2^-100, 2^100) is detected by its workgroup after the
     * last key tile and run again with the textbook online softmax ("exact form"), so the RESULT never depends on the range --
     * only the time does.  redo_counter: optional device word, incremented once per item that was redone (the caller zeroes
     * it); force_exact != 0: every item takes the exact form at once (what a launch costs when nothing fits the range).
     * Ignored by the register-staged kernel, which has the exact form only. */
    uint32_t* redo_counter; int32_t force_exact;
} ltxmi_attn_args;

int ltxmi_attention_fwd_bf16(const ltxmi_attn_args* args, void* stream);
/* 1 if ltxmi_attention_fwd_bf16 can normalise + rotate q on load for this shape, else 0 (the caller then runs
 * ltxmi_rmsnorm_rope_bf16 on q as a pass of its own).  Since 0.2 every shape the entry point accepts qualifies. */
int ltxmi_attention_fuses_qnorm(int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t head_dim, int32_t has_key_bias);
/* Identifier (>= 0) of the kernel instance ltxmi_attention_fwd_bf16 runs for this shape, -1 if unsupported.  Shapes with
 * the same id get the same arithmetic per (batch, head, query row) -- e.g. B and B - 1 batch rows of one model call.
 * k_stride_l / v_stride_l (0.3): the token strides the call will pass -- the pipelined kernels (ids 3 and 6) address a
 * (batch, head)'s keys with 32-bit byte offsets and hand shapes whose rows span 2 GiB or more to the other kernels. */
int ltxmi_attention_kernel_id(int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t head_dim, int32_t has_key_bias,
                              int64_t k_stride_l, int64_t v_stride_l);

/* The same row factor as a launch of its own (cross-attention's q, which has no k pass of equal row count to ride on):
 * rstd_out[r] = rsqrt(sum_j rowsumsq[r * rowsumsq_ld + j] / norm_dim + norm_eps), j < rowsumsq_blocks, r < rows. */
int ltxmi_rowsumsq_rstd_f32(const float* rowsumsq, int64_t rowsumsq_ld, int32_t rowsumsq_blocks, int32_t rows,
                            int32_t norm_dim, float norm_eps, float* rstd_out, void* stream);

/* Ulysses send buffer in one pass (sequence-parallel self-attention, xdit_context_parallel.py:149-184 of the reference
 * for Wan; here for the LTX DiT): q/k RMSNorm(weight) + interleaved RoPE exactly as ltxmi_rmsnorm_rope_bf16 and v,
 * read from the packed projection qkv [B*Nl, 3 D] (row stride ld, row = b * Nl + n) and written destination-major:
 *   out[P][Nl][B][3][D / P]  (destination rank = head group of the channel).
 * After all_to_all_single, rank r holds [P*Nl][B][3][D/P]: its heads over all tokens, uniform strides. */
int ltxmi_qkv_norm_rope_pack_bf16(const void* qkv, int64_t ld, int32_t B, int32_t Nl, int32_t D, int32_t P,
                                  const void* q_weight, const void* k_weight, float eps, const void* cos_tab,
                                  const void* sin_tab, int64_t ld_tab, int32_t rope_period, void* out, void* stream);

/* ---------------------------------------------------------------------------------
 * Small elementwise helpers on the DiT path.
 * ------------------------------------------------------------------------------- */
/* y = silu(x) (kind 0) -- AdaLayerNormSingle's SiLU between its linears, transformer3d.py:428. */
int ltxmi_silu_bf16(const void* x, void* y, int64_t n, void* stream);
/* Sinusoidal timestep projection (256 ch, flip_sin_to_cos, shift 0) -> bf16 [n,256];
 * t is fp32 [n] ALREADY multiplied by timestep_scale_multiplier.
 * ltx_video/models/transformers/embeddings.py:10-50 as used by AdaLayerNormSingle. */
int ltxmi_timestep_embedding_bf16(const float* t, void* out, int32_t n, int32_t dim, void* stream);
/* STG "attention values" blend: a = a*m[b] + v*(1-m[b]),  attention.py:1134-1141.
 * a: [B, L, D] contiguous rows lda; v rows ldv; m fp32 [B]. */
int ltxmi_stg_blend_bf16(void* a, int64_t lda, const void* v, int64_t ldv,
                         const float* m, int32_t B, int32_t L, int32_t D, void* stream);
/* The same blend (attention.py:1127-1141) over a contiguous a [G, B, L, D] with v addressed by strides in elements:
 * v(g, b, l, :) at v + g*v_stride_g + b*v_stride_b + l*v_stride_l.  One launch over the K-blocked receive buffer of the
 * Ulysses return exchange ([P source ranks][B, N/P][D/P]; the reference's xfuser path has no STG and no counterpart). */
int ltxmi_stg_blend_grouped_bf16(void* a, const void* v, int64_t v_stride_g, int64_t v_stride_b, int64_t v_stride_l,
                                 const float* m, int32_t G, int32_t B, int32_t L, int32_t D, void* stream);

/* ---------------------------------------------------------------------------------
 * VAE decode kernels (channels-last NDHWC activations, bf16).
 * ------------------------------------------------------------------------------- */
/* 3x3x3 stride-1 convolution as implicit GEMM with the reference's padding semantics
 * folded into the address computation:
 *   time:  replicate (k-1) frames in front when causal, else (k-1)/2 on both sides
 *          ltx_video/models/autoencoders/causal_conv3d.py:44-59
 *   space: pad 1, zeros or replicate (nn.Conv3d padding_mode)   causal_conv3d.py:33-42
 * With strides the output grid is nn.Conv3d's: floor((L + pad - 3) / stride) + 1 per axis.
 * x: [B, T, H, W, Cin]; w: [Cout, 27, Cin] (tap-major, K contiguous; repacked from the
 * checkpoint's [Cout,Cin,3,3,3] once at load); bias [Cout]; y: [B, T, H, W, Cout].
 * Optional fused output transform (DepthToSpaceUpsample, causal_video_autoencoder.py:1051-1065):
 *   d2s = 1 -> the Cout = 8*C' channels are scattered as (c p1 p2 p3) into
 *   y: [B, 2T-1, 2H, 2W, C'] (first output frame dropped) and, if residual != NULL,
 *   the pixel-shuffled, channel-repeated input is added (res_repeat = 8 / reduction).
 * Cin % 64 == 0; Cout % 8 == 0.
 * ------------------------------------------------------------------------------- */
typedef struct ltxmi_conv3d_args {
    const void* x; const void* w; const void* bias; void* y;
    int32_t B, T, H, W, Cin, Cout;
    int32_t causal;            /* 1: replicate 2 frames in front; 0: 1 + 1                */
    int32_t pad_replicate;     /* spatial padding mode: 0 zeros, 1 replicate              */
    int32_t d2s;               /* 0 plain NDHWC store, 1 depth-to-space (2,2,2) store     */
    const void* residual;      /* d2s only: x itself (pre-conv block input) or NULL       */
    int32_t res_channels;      /* channels of the residual tensor (Cin of the block)      */
    const void* add;           /* plain store only: y = conv + add, add [B,T,H,W,Cout] or NULL
                                  (ResnetBlock3D skip, causal_video_autoencoder.py:1256)   */
    /* encoder-side extensions (0 = default): strided causal convolutions of the "compress_*"
     * encoder blocks (causal_video_autoencoder.py:395-432) and the extended-in-time convolution of
     * SpaceToDepthDownsample, which runs on the input with its first frame duplicated (:991-1009):
     * tpad = frames replicated in front (default 2 causal / 1 otherwise), out_T = output frames. */
    int32_t stride_t, stride_hw;   /* 1 or 2                                                */
    int32_t tpad, out_T;
    /* plain (non-causal-VAE) convolutions of LatentUpsampler (latent_upsampler.py:15-149), 0 = default:
     * kernel_t = 1: a 3x3 nn.Conv2d applied per frame (w is [Cout, 9*Cin]); time_pad_zeros = 1:
     * nn.Conv3d(padding=1) -- the time axis is padded with zeros instead of replicated frames. */
    int32_t kernel_t, time_pad_zeros;
    int32_t algo;              /* 0 = implementation chosen by shape (the product setting); 1 = implicit GEMM;
                                  2 = direct convolution, form chosen by shape; 3 / 4 = direct convolution in its
                                  four-wave (two workgroups per CU, Cout % 128 == 0) / eight-wave form whatever the
                                  grid (LTXMI_ERR_UNSUPPORTED if the direct convolution does not take the shape).
                                  Used by the parity tests to check the implementations against each other. */
    /* 0.4 (optional, zeros = off): the norm2 -> SiLU that follows conv1 inside a ResnetBlock3D
     * (causal_video_autoencoder.py:1226-1243) applied in the convolution's epilogue, from the fp32 accumulators:
     * y = silu(pixelnorm(conv + bias) * (1 + post_scale[b]) + post_shift[b]), the arithmetic of
     * ltxmi_pixelnorm_ada_silu_bf16 without the bf16 rounding in between.  Only where one wave holds all channels of a
     * position: ltxmi_conv3d_fuses_post_norm() says whether a call would; post_norm = 1 on any other call is
     * LTXMI_ERR_UNSUPPORTED (the caller then runs the PixelNorm launch itself). */
    int32_t post_norm;         /* 0 off, 1 PixelNorm -> AdaLN -> SiLU                       */
    const float* post_scale;   /* fp32 [B, Cout], or NULL together with post_shift          */
    const float* post_shift;
    float post_eps;            /* PixelNorm eps (pixel_norm.py: 1e-8)                       */
    /* 0.5 (optional, NULL = off): with post_norm = 1, a SECOND output beside y.  y keeps the raw result (conv + bias + add,
     * or the depth-to-space store with its residual) bit for bit as without post_norm; y_norm (y's shape) receives
     * silu(pixelnorm(y) * (1 + post_scale[b]) + post_shift[b]) computed from y's bf16 values: the NEXT ResnetBlock3D's
     * norm1 -> AdaLN -> SiLU (causal_video_autoencoder.py:1197-1224) or the decoder's tail (:771-795) without the launch that
     * reads y back.  `add` with Cout == 128, or d2s with Cout == 1024 (post_scale / post_shift are [B, 128] then); ask
     * ltxmi_conv3d_fuses_post_norm(). */
    void* y_norm;
    /* 0.5 (optional, NULL / 0 = off): scratch memory the call may use.  With at least ltxmi_conv3d_workspace_bytes(args) bytes
     * (16-byte aligned) a wide, short layer (Cin >= 1024, or >= 512 with post_norm; Cout 512 or a multiple of 1024 up to 4096) whose tiles do not fill the chip runs
     * split over its input channels: fp32 partial sums of 2 .. 4 channel ranges into the workspace, summed in range order by a
     * finalising pass that applies the epilogue -- and post_norm / y_norm at ANY width (it holds whole rows).  Results are those of
     * the unsplit call to fp32 summation order.  The contents are meaningless before and after the call; the same workspace may
     * serve every call on a stream. */
    void* workspace; int64_t workspace_bytes;
} ltxmi_conv3d_args;

/* Two implementations behind this entry, chosen by shape: a direct convolution with the input halo
 * resident in LDS (stride 1, kernel_t 3, >= 128 workgroups; four waves per workgroup and two workgroups per CU where
 * Cout is a multiple of 128, eight waves per workgroup otherwise) and an implicit GEMM (everything else).
 * Requirements for both: Cin % 64 == 0, Cout % 8 == 0, 16-byte aligned x / w. */
int ltxmi_conv3d_ndhwc_bf16(const ltxmi_conv3d_args* args, void* stream);
/* 1 if ltxmi_conv3d_ndhwc_bf16(args) with post_norm = 1 would apply it in its epilogue (the four-wave direct convolution
 * with Cout == 128, plain store, no `add`; with y_norm set: `add` and Cout == 128, or d2s and Cout == 1024), 0 otherwise.  Reads the fields that choose the implementation (shape, flags, algo,
 * bias != NULL); no launch. */
int ltxmi_conv3d_fuses_post_norm(const ltxmi_conv3d_args* args);
/* Bytes of workspace with which ltxmi_conv3d_ndhwc_bf16(args) would run split over its input channels (see
 * ltxmi_conv3d_args.workspace); 0 when it would not.  Reads the shape, flags, algo and post_norm (at 512 input channels the split
 * pays only when the finalising pass takes a norm along); ignores args->workspace*. */
int64_t ltxmi_conv3d_workspace_bytes(const ltxmi_conv3d_args* args);

/* PixelNorm (pixel_norm.py:5-12, eps 1e-8) -> optional (1+scale)*x+shift per (batch, channel)
 * (ResnetBlock3D AdaLN, causal_video_autoencoder.py:1206-1243, Decoder tail :771-795)
 * -> optional SiLU; NDHWC rows of C channels; scale/shift fp32 [B, C] or NULL.
 * rows_per_batch = T*H*W.  C % 8 == 0, C <= 4096. */
int ltxmi_pixelnorm_ada_silu_bf16(const void* x, void* y, int64_t rows, int32_t C,
                                  int64_t rows_per_batch, const float* scale, const float* shift,
                                  int32_t apply_silu, float eps, void* stream);

/* y = a + b elementwise (ResnetBlock3D skip add, causal_video_autoencoder.py:1256). */
int ltxmi_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream);

/* Channel LayerNorm with affine over NDHWC rows (norm3 of res_x_y blocks,
 * causal_video_autoencoder.py:1068-1077,1170-1174). */
int ltxmi_layernorm_affine_bf16(const void* x, void* y, int64_t rows, int32_t C,
                                const void* gamma, const void* beta, float eps, void* stream);

/* Layout changes at the decoder boundary:
 *  ncdhw_to_ndhwc: latent z [B,C,T,H,W] (any float bf16) * std[c] + mean[c] -> NDHWC bf16
 *                  (un_normalize_latents, vae_encode.py:239-247; std == NULL -> plain copy)
 *  unpatchify:     conv_out result NDHWC [B,T,H,W,3*p*p] -> pixels NCDHW [B,3,T,H*p,W*p]
 *                  "b (c p r q) f h w -> b c (f p) (h q) (w r)" with p=1 (causal_video_autoencoder.py:1282-1299) */
int ltxmi_ncdhw_to_ndhwc_bf16(const void* z, void* y, int32_t B, int32_t C, int32_t T, int32_t H,
                              int32_t W, const float* std, const float* mean, void* stream);
int ltxmi_unpatchify_to_ncdhw_bf16(const void* x, void* y, int32_t B, int32_t T, int32_t H, int32_t W,
                                   int32_t C_out, int32_t patch, void* stream);

/* Encoder side (image / video conditioning; Encoder.forward, causal_video_autoencoder.py:514-557):
 *  patchify:       pixels NCDHW [B,C,T,H,W] -> NDHWC [B,T,H/p,W/p,C_pad], channel (c p r q) =
 *                  "b c (f p) (h q) (w r) -> b (c p r q) f h w" with p=1 (:1261-1279); channels
 *                  >= C*p*p are zero so that conv_in's K axis is a multiple of 64
 *  space_to_depth_skip: tail of SpaceToDepthDownsample.forward (:991-1020).  conv = the stride-1
 *                  causal convolution of the block ([B,T',H,W,Cconv], T' = T+1 when stride_t == 2,
 *                  i.e. run with tpad = 3 / out_T = T+1 on x), x = the block input [B,T,H,W,Cin];
 *                  out [B,T'/st,H/s,W/s,Cconv*st*s*s] = space_to_depth(conv) + group-mean of
 *                  space_to_depth(x with its first frame duplicated), group = Cin / Cconv
 *  ndhwc_to_ncdhw: channels c0..c0+C of NDHWC rows (row stride ldx) -> NCDHW [B,C,T,H,W], optionally
 *                  (v - mean[c]) / std[c] (normalize_latents, vae_encode.py:228-236) */
int ltxmi_patchify_to_ndhwc_bf16(const void* x, void* y, int32_t B, int32_t C, int32_t T, int32_t H,
                                 int32_t W, int32_t patch, int32_t C_pad, void* stream);
int ltxmi_space_to_depth_skip_bf16(const void* conv, const void* x, void* out, int32_t B, int32_t T,
                                   int32_t H, int32_t W, int32_t Cin, int32_t Cconv, int32_t stride_t,
                                   int32_t stride_hw, int32_t group, void* stream);
int ltxmi_ndhwc_to_ncdhw_bf16(const void* x, int64_t ldx, int32_t c0, void* y, int32_t B, int32_t C,
                              int32_t T, int32_t H, int32_t W, const float* std, const float* mean,
                              void* stream);

/* ---------------------------------------------------------------------------------
 * Denoise-loop step math kept on device (no host sync per step).
 *   guidance: CFG-star + STG + std-rescale  pipeline_ltx_video.py:1183-1222
 *   euler   : x <- x - dt * v               ltx_video/schedulers/rf.py:375
 * noise_pred: bf16 [num_conds, n] (one sample, chunk order uncond/text/perturbed as built at
 * pipeline_ltx_video.py:1035-1051); latents: [n], fp32 (latents_bf16 = 0) or bf16 (= 1).
 * workspace: >= LTXMI_GUIDANCE_WORKSPACE_FLOATS floats (per-block partial sums: the reductions use no
 * atomics, so the step is run-to-run deterministic); contents need not be initialised.
 * ------------------------------------------------------------------------------- */
#define LTXMI_GUIDANCE_WORKSPACE_FLOATS 2048
int ltxmi_guidance_step_bf16(const void* noise_pred, int64_t n, int32_t num_conds,
                             float guidance_scale, float stg_scale, float rescaling_scale,
                             int32_t do_cfg, int32_t do_stg, int32_t do_rescale,
                             void* latents, int32_t latents_bf16, float dt, float* workspace,
                             void* stream);

/* The same step when conditioning items are present (image-/video-to-video):
 * denoising_step, pipeline_ltx_video.py:1309-1342 -- token i (of `channels` values each) is
 * advanced only if t - 1e-6 < 1 - cond_mask[i]; hard-conditioned tokens (mask 1) never move.
 * cond_mask: fp32 [n / channels] or NULL (= ltxmi_guidance_step_bf16).  For the tokens that do
 * move, their per-token timestep min(t, 1 - mask) equals t, so dt is the global step. */
int ltxmi_guidance_step_masked_bf16(const void* noise_pred, int64_t n, int32_t num_conds,
                                    float guidance_scale, float stg_scale, float rescaling_scale,
                                    int32_t do_cfg, int32_t do_stg, int32_t do_rescale,
                                    void* latents, int32_t latents_bf16, float dt,
                                    const float* cond_mask, int32_t channels, float t,
                                    float* workspace, void* stream);

/* add_noise_to_image_conditioning_latents, pipeline_ltx_video.py:606-629: tokens with
 * cond_mask > 1 - 1e-6 become init_latents + noise_scale * noise * t^2, others are untouched.
 * latents / init_latents / noise: [tokens, channels], all fp32 (is_bf16 = 0) or all bf16 (= 1). */
int ltxmi_image_cond_noise(void* latents, const void* init_latents, const void* noise, int32_t is_bf16,
                           const float* cond_mask, int64_t tokens, int32_t channels, float noise_scale,
                           float t, void* stream);

/* ---------------------------------------------------------------------------------
 * Multi-scale bridge between pass 1 and pass 2 (LatentUpsampler, latent_upsampler.py:15-149;
 * adain_filter_latent, pipeline_ltx_video.py:1709-1737).  Its convolutions are
 * ltxmi_conv3d_ndhwc_bf16 with kernel_t / time_pad_zeros.
 *   groupnorm_silu: x [samples, S, C] channels-last -> y = silu(GroupNorm(groups)(x) * gamma + beta
 *                   (+ residual)); ResBlock.forward :30-39 and initial_norm/activation :121-123.
 *                   samples = b (dims 3) or b*f (dims 2).  workspace: >= samples*(2C + 2*groups) floats.
 *   pixel_shuffle2d: x [frames, H, W, 4C] with channel (p1*2 + p2)*C + c -> y [frames, 2H, 2W, C]
 *                   (PixelShuffleND(2) :93-96 with the conv rows packed (p1 p2 c)).
 *   adain_filter:   per (b, c) plane: out = lerp(x, (x - mean_x)/std_x * std_ref + mean_ref, factor);
 *                   latents [planes, n], reference [planes, n_ref], fp32 (is_bf16 = 0) or bf16.
 * ------------------------------------------------------------------------------- */
int ltxmi_groupnorm_silu_bf16(const void* x, void* y, const void* residual, int32_t samples, int64_t S,
                              int32_t C, int32_t groups, const void* gamma, const void* beta, float eps,
                              float* workspace, void* stream);
int ltxmi_pixel_shuffle2d_ndhwc_bf16(const void* x, void* y, int64_t frames, int32_t H, int32_t W, int32_t C,
                                     void* stream);
/* Tile cross-fade of the tiled VAE decode / encode (blend_z / blend_v / blend_h, vae.py:193-221), in place in b:
 *   b[o, z, i] = a[o, len_a - extent + z, i] * (1 - z / extent) + b[o, z, i] * (z / extent),   z < extent,
 * both tensors contiguous and viewed as [outer][len][inner] around the blended axis; dtype: 0 fp32, 1 bf16, 2 fp16 (the
 * reference keeps z-tiles in fp16, vae.py:388). */
int ltxmi_tile_blend(const void* a, void* b, int32_t dtype, int64_t outer, int64_t len_a, int64_t len_b,
                     int64_t inner, int32_t extent, void* stream);
int ltxmi_adain_filter(const void* latents, const void* reference, void* out, int32_t is_bf16, int32_t planes,
                       int64_t n, int64_t n_ref, float factor, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LTXMI_H */
