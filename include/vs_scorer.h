/*
 * vs_scorer.h — C ABI of libvsscore.so, the MI355X (gfx950) frame-importance scorer.
 *
 * The reference (BerserkerMother/Video-Summarization) has no FFI: its scorer is the Python
 * nn.Module `SimNet` (reference src/model/simnet.py:8).  This ABI is what a binding for that
 * module's eval forward binds; each entry point cites the reference interface it replaces.
 * All pointers are DEVICE pointers unless stated otherwise; the library keeps no persistent
 * state besides the packed weights a vs_weights handle owns.  Every function is thread-safe
 * with respect to distinct handles; errors are returned as a non-zero status and a
 * thread-local message (vs_last_error).  No function synchronises the device: work is
 * enqueued on the caller's HIP stream.
 */
#ifndef VS_SCORER_H
#define VS_SCORER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VS_ABI_VERSION 3

/* status codes */
#define VS_OK 0
#define VS_ERR_INVALID 1     /* bad argument / unsupported shape  (reference: RuntimeError from ATen) */
#define VS_ERR_WORKSPACE 2   /* workspace too small */
#define VS_ERR_HIP 3         /* a HIP runtime call failed */

/* forward flags */
#define VS_FLAG_SIGMOID 1u   /* scores = sigmoid(logits): the caller-side head of reference
                                train.py:144 / generate_summary_image.py:68, fused.  Default
                                (flag clear) returns raw logits like SimNet.forward (simnet.py:42). */
#define VS_FLAG_BF16_ATTENTION 2u /* opt-in, long videos (BASELINE config 5): the two attention products
                                run on the bf16 matrix pipe (q*scale, k, v, p rounded to bf16; fp32
                                softmax and accumulation).  Head dim 32 or 64.  Scores then differ from
                                the reference's fp32 path by ~1e-3 (tolerance stated in the tests);
                                the default path stays exact fp32.  Logits up to +-32 000 (log2 units) are soaked (tools/fuzz_attn_w64.py; round 4 fixed a NaN
                                for logits beyond 2^14 off the bf16 grid in the 8-wave kernel); beyond the row
                                constant's clamp at 2^15 the bf16 kernels are not specified, the exact path is.  No model whose logits mean anything is near it. */
#define VS_FLAG_BF16_LINEAR 4u /* opt-in: every Linear (embed, q/k/v, feature_projection, fc1, fc2) multiplies
                                bf16-rounded operands on the bf16 matrix pipe; tensors stay fp32 in HBM, and
                                bias, accumulation, residual, LayerNorm and the score head stay fp32.
                                Any supported d_model (<= 256: the fused bf16-storage layer kernels; above: plain bf16
                                GEMMs + the row LayerNorm pass, fp32 storage).  Same tolerance caveat as VS_FLAG_BF16_ATTENTION.  Batches of up
                                to 8192 frames keep the (faster, exact) fp32 latency kernels (bf16 only:
                                fp16x3 has latency kernels of its own). */
#define VS_FLAG_BF16 (VS_FLAG_BF16_ATTENTION | VS_FLAG_BF16_LINEAR)
#define VS_FLAG_SPLITK 32u   /* opt-in LATENCY mode for reference-sized calls (one T = 320 video, train.py:71,139-148): with
                                the exact fp32 kernels and at most VS_SKINNY_ROWS rows, the K >= 512 Linears (embedding, fc2)
                                and the out-projection are split over K across more CUs (fixed slices, partials added in slice
                                order, then bias, then the residual) and the LayerNorm runs as a row pass.  Deterministic and
                                batch-independent, within ~1e-6 of the default kernels' results (a different summation tree;
                                tests hold it to the goldens at 1e-4) - but NOT their bits, so a video scored alone in this
                                mode and the same video scored in a large batch are no longer bit-identical.  Also with
                                VS_FLAG_F16X3_LINEAR (the split-K kernels emulated on the f16 pipe; the attention is then the
                                exact split-key kernel).  d_model 128 / 256 / 512, head dim 32 / 64 / 128.  Ignored (the default
                                kernels run) for larger inputs, other shapes, the bf16 mode and packed / class-token calls. */
#define VS_FLAG_F16X3_ATTENTION 16u /* opt-in: the two attention products emulated on the f16 pipe the same way
                                (q*scale, k, v, p split into hi + lo halves, three MFMAs per product, fp32
                                softmax and accumulation).  Head dim 32 or 64.  Exclusive with
                                VS_FLAG_BF16_ATTENTION.  See VS_FLAG_F16X3_LINEAR. */
#define VS_FLAG_F16X3 (VS_FLAG_F16X3_ATTENTION | 8u)
#define VS_FLAG_F16X3_LINEAR 8u /* opt-in: every Linear EMULATES the fp32 product on the f16 matrix pipe: each
                                operand x is split as f16(x) + f16(x - f16(x)) (22 significant bits) and
                                a*b ~ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, fp32 accumulate - three f16 MFMAs
                                (96 cycles) instead of eight fp32 MFMAs (512 cycles) per 16 k.  Per-product
                                error ~3e-6 relative: results stay inside the 1e-4 bar (tests).  Operand
                                magnitudes must be < 65504 (f16 range).  Any supported d_model.
                                Exclusive with VS_FLAG_BF16_LINEAR. */

/* Model hyper-parameters: the ctor arguments of reference SimNet.__init__ (simnet.py:10-13)
 * that shape the eval forward. */
typedef struct vs_model_desc {
    int32_t d_model;      /* simnet.py:16;  multiple of 64, <= 1024 (other widths: vs_weights_set_norm_width below) */
    int32_t num_heads;    /* simnet.py:15;  d_model/num_heads in {32, 64, 128, 256} (256: exact attention only) */
    int32_t num_layers;   /* simnet.py:17;  len(encoder.module_list), >= 1 */
    int32_t in_features;  /* simnet.py:22 (1024 in the reference); multiple of 32 */
    int32_t max_len;      /* rows of pos_embedding (simnet.py:188: 2000); 0 when use_pos=False */
    int32_t num_classes;  /* simnet.py:21,30 */
} vs_model_desc;

/* One EncoderBlock's parameters (reference simnet.py:93-100), nn.Linear layout [out,in]. */
typedef struct vs_layer_params {
    const float *wq, *bq, *wk, *bk, *wv, *bv;   /* sa.q / sa.k / sa.v            simnet.py:130-132 */
    const float *wo, *bo;                       /* sa.feature_projection         simnet.py:136 */
    const float *ln1_g, *ln1_b;                 /* norm1                         simnet.py:99 */
    const float *w1, *b1, *w2, *b2;             /* mlp.fc1 [4d,d], mlp.fc2 [d,4d] simnet.py:175-176 */
    const float *ln2_g, *ln2_b;                 /* norm2                         simnet.py:100 */
} vs_layer_params;

/* The whole state_dict as raw device pointers. */
typedef struct vs_model_params {
    const float *embed_w, *embed_b;             /* embedding_layer.feature_transform  simnet.py:199 */
    const float *pos_embedding;                 /* [max_len, d] or NULL (use_pos=False) simnet.py:234 */
    const vs_layer_params *layers;              /* HOST array of num_layers entries */
    const float *final_w, *final_b;             /* final_layer [num_classes, d]       simnet.py:30 */
} vs_model_params;

typedef struct vs_weights vs_weights;           /* opaque: packed device copy of the parameters */

/* ABI version of the loaded library (== VS_ABI_VERSION of the header it was built from). */
int vs_abi_version(void);

/* Message of the last failing call on this thread ("" if none). */
const char *vs_last_error(void);

/* Replaces: SimNet.__init__/load_state_dict (simnet.py:10-30; train.py:42-43,76).
 * Copies the parameters into one device blob in the layout the kernels want (Q/K/V weights
 * concatenated to [3d, d]; everything else as is).  The copy is enqueued on `stream`; the
 * source tensors may be modified afterwards.  Call again after any parameter update. */
int vs_weights_pack(const vs_model_desc *desc, const vs_model_params *params,
                    void *stream, vs_weights **out);
void vs_weights_free(vs_weights *w);

/* Round 4 - the reference's envelope is ANY d_model % num_heads == 0 (simnet.py:123); the kernels' is d_model % 64 == 0
 * with head dim 32 / 64 / 128 / 256.  A model outside it is scored / trained EMBEDDED in the next supported shape: pack it with
 * desc.d_model = d' = num_heads * dh' (dh' the next supported head dim that makes d' a multiple of 64) and every parameter
 * zero-padded - residual-stream axes (rows of embed_w / wo / w2, columns of wq / wk / wv / w1 / final_w, biases, LayerNorm
 * gamma / beta, the positional table) keep feature c at index c; head-structured axes (rows of wq / wk / wv, columns of wo)
 * move feature (h, j) from h * dh + j to h * dh' + j; the MLP's hidden axis keeps its index - then declare the true width
 * here.  The pad columns of every activation are then identically zero; what depends on the true d_model and is NOT
 * invariant under the padding - the LayerNorm statistics and the attention scale d_model ** -0.5 (simnet.py:126) - uses
 * norm_width.  Such a handle always takes the plain GEMM + row-LayerNorm kernels.  SimNet (simnet.py of this package) does
 * all of this itself; `hidden` then has d' columns of which the first norm_width are the model's.
 * norm_width: multiple of 4, 0 < norm_width <= desc.d_model (== desc.d_model: an ordinary handle again). */
int vs_weights_set_norm_width(vs_weights *w, int32_t norm_width);

/* Replaces: the parameter writes of an optimizer step / load_state_dict on an existing module (train.py:127,
 * 42-43).  Re-copies every parameter into the handle's existing device storage (same desc, same device as
 * vs_weights_pack; no allocation, no free), stream-ordered on `stream`: one batched copy launch per encoder layer.
 * params->pos_embedding may be NULL here: the table already packed is kept (it is a buffer, not a parameter).  The
 * kernel-layout copies (fragment-major, fp16x3, bf16 images, transposes) are NOT rebuilt here: each family is rebuilt
 * by the first forward / backward that reads it after the update, on that call's stream.  Calls on DIFFERENT streams are
 * ordered by the library on the device (an event is recorded behind every parameter write / image rebuild and a call on
 * another stream waits for it - no host synchronisation); two host THREADS must still not use one handle at the same time. */
int vs_weights_update(vs_weights *w, const vs_model_params *params, void *stream);

/* Bytes of scratch vs_scorer_forward needs for a [B,T] batch (0 on invalid arguments). */
size_t vs_scorer_workspace_bytes(const vs_weights *w, int32_t B, int32_t T);

/* Replaces: SimNet.forward(x, mask) in eval mode (simnet.py:32-45) including
 * process_mask (simnet.py:47-56), Embedding/PositionalEncoding (:208-238), the Encoder block
 * loop (:77-83, :105-114, :138-164, :180-183) and final_layer (:42).
 *   x             [B, T, in_features] fp32, contiguous
 *   key_pad_mask  [B, T] bytes, non-zero = key is padding (reference bool mask), or NULL
 *   scores        [B, T, num_classes] fp32 out: raw logits, or sigmoid with VS_FLAG_SIGMOID
 *   hidden        [B, T, d_model] fp32 out (second return value of forward), or NULL
 *   workspace     >= vs_scorer_workspace_bytes(w, B, T) bytes, 256-byte aligned
 * Errors mirror the reference's: T > max_len (when a positional table is present) -> VS_ERR_INVALID. */
int vs_scorer_forward(const vs_weights *w, const float *x, const uint8_t *key_pad_mask,
                      int32_t B, int32_t T, uint32_t flags,
                      float *scores, float *hidden,
                      void *workspace, size_t workspace_bytes, void *stream);

/* Replaces: SimNet.forward of a module built with use_cls=True (simnet.py:205-206, 214-216, process_mask :47-51; no
 * reference caller enables it): `cls_token` [d_model] fp32 (16-byte aligned) is prepended to every video AFTER the
 * positional encoding, the encoder sees T + 1 positions (the token is never padding), and
 *   scores [B, T+1, num_classes], hidden [B, T+1, d_model] (or NULL)
 * carry the token's row first.  x, key_pad_mask, flags, stream as vs_scorer_forward (x has T frames per video, the
 * mask T entries); workspace >= vs_scorer_workspace_bytes_cls(w, B, T).  Every compute mode of vs_scorer_forward. */
size_t vs_scorer_workspace_bytes_cls(const vs_weights *w, int32_t B, int32_t T);
int vs_scorer_forward_cls(const vs_weights *w, const float *x, const uint8_t *key_pad_mask,
                          const float *cls_token, int32_t B, int32_t T, uint32_t flags,
                          float *scores, float *hidden,
                          void *workspace, size_t workspace_bytes, void *stream);

/* Packed ragged batch (no reference counterpart: the reference pads with the 1000.0 sentinel and masks,
 * data/dataset.py:157-161, train.py:118): the frames of B videos concatenated, x [Mtot, in_features] with
 * Mtot = sum(lengths), scores [Mtot, num_classes], hidden [Mtot, d_model] or NULL.  No padding rows are computed and
 * no mask is needed; every video's scores are bit-identical to scoring it alone.  `lengths` is the HOST copy (it
 * sizes the launches), `lengths_dev` the same B values in device memory.  Head dim 32 / 64; flags:
 * VS_FLAG_SIGMOID and the precision flags of vs_scorer_forward.  Like vs_scorer_forward it only enqueues work on `stream`. */
size_t vs_scorer_workspace_bytes_packed(const vs_weights *w, const int32_t *lengths, int32_t B);
int vs_scorer_forward_packed(const vs_weights *w, const float *x, const int32_t *lengths,
                             const int32_t *lengths_dev, int32_t B, uint32_t flags,
                             float *scores, float *hidden,
                             void *workspace, size_t workspace_bytes, void *stream);

/* Measurement hooks (no reference counterpart): while enabled, every stage of
 * vs_scorer_forward is bracketed by HIP events recorded on the caller's stream.
 * vs_profile_collect waits for the recorded events, returns per-stage summed milliseconds and launch
 * counts (arrays of VS_NUM_STAGES) and clears the records. */
#define VS_STAGE_EMBED 0       /* feature_transform + positional table      gemm_nt_128<PE>   */
#define VS_STAGE_QKV 1         /* q/k/v projections, head-major store       gemm_nt_128<QKV>  */
#define VS_STAGE_ATTENTION 2   /* softmax(q k^T) v                          attn_fwd          */
#define VS_STAGE_OUTPROJ_LN 3  /* feature_projection + residual + norm1     gemm_res_ln       */
#define VS_STAGE_FC1 4         /* mlp.fc1 + ReLU                            gemm_nt_128<RELU> */
#define VS_STAGE_FC2_LN 5      /* mlp.fc2 + residual + norm2 (+ score head) gemm_res_ln       */
#define VS_NUM_STAGES 6
int vs_profile_enable(int32_t on);
int vs_profile_collect(double *ms_sum, int64_t *launches);
const char *vs_stage_name(int32_t stage);

/* A/B and test switches (DESIGN.md "Environment switches"; none selects a fallback).  Their defaults come from
 * environment variables of the same name, read ONCE when the library is first used - no forward calls getenv.
 * value < 0 restores the environment / built-in default.  Names: VS_SKINNY_ROWS, VS_LP_MIN_ROWS, VS_LP_MIN_ROWS_FUSED, VS_TRAIN_LP_MIN_ROWS, VS_ATTN_W64, VS_ATTN_W64_CHECKED, VS_GEMM_NWM2,
 * VS_GEMM_NJ2, VS_ATTN_NW4, VS_ATTN_LP_SIMPLE, VS_LP_STORE32, VS_LP_MLP_UNFUSED, VS_LP_TAIL_UNFUSED, VS_LP_QKV_UNFUSED, VS_LP_EMBED_UNFUSED, VS_LP_TILE256 (+ VS_MLP_FUSION, VS_MLP_ABL, VS_ATTN_LEGACY, which only the
 * diagnostic build of the library acts on).  Process-wide; not meant to be flipped while forwards are in flight. */
int vs_set_option(const char *name, int32_t value);

/* Per-kernel entry points (same stream/pointer conventions), exported so each HIP kernel can be
 * parity-tested against the oracle in isolation.  Not needed by a drop-in binding. */

/* C[M,N] = A[M,K] * W[N,K]^T + bias[N]  (nn.Linear, simnet.py:211 et al.), optional ReLU
 * (simnet.py:181) and optional + pe[(row % T), :] (simnet.py:237-238; pe may be NULL). */
int vs_linear_f32(const float *A, const float *W, const float *bias, float *C,
                  int32_t M, int32_t N, int32_t K, int32_t relu,
                  const float *pe, int32_t T, void *stream);

/* vs_linear_f32 on the bf16 matrix pipe (see VS_FLAG_BF16_LINEAR). */
int vs_linear_bf16(const float *A, const float *W, const float *bias, float *C,
                   int32_t M, int32_t N, int32_t K, int32_t relu,
                   const float *pe, int32_t T, void *stream);

/* vs_linear_bf16 with A already stored as bf16 (A16 [M,K] row-major; W stays fp32 and is rounded on its way into LDS):
 * the form the bf16 training mode runs fc2 and the fc1 input gradient in (pe: optional [M,N] fp32 term added to C). */
int vs_linear_bf16_a16(const void *A16, const float *W, const float *bias, float *C,
                       int32_t M, int32_t N, int32_t K, const float *pe, void *stream);

/* vs_linear_f32 emulated on the f16 matrix pipe (see VS_FLAG_F16X3_LINEAR). */
int vs_linear_f16x3(const float *A, const float *W, const float *bias, float *C,
                    int32_t M, int32_t N, int32_t K, int32_t relu,
                    const float *pe, int32_t T, void *stream);

/* The wide models' bf16 Linear (d_model > 256 under VS_FLAG_BF16_LINEAR): BOTH operands already bf16 in device memory
 * (A16 [M,K], W16 [N,K] row-major, 16-byte aligned; K, N multiples of 32), fp32 accumulation, C fp32 [M,N] (c16 == 0) or
 * bf16 (c16 != 0).  vs_to_bf16 is the rounding (nearest even) the producers apply: dst16[i] = bf16(src[i]), n % 8 == 0. */
int vs_linear_bf16_operands(const void *A16, const void *W16, const float *bias, void *C,
                            int32_t M, int32_t N, int32_t K, int32_t relu, int32_t c16, void *stream);
/* the same product with the q/k/v epilogue: out[3][B][H][T][dh] (fp32, or bf16 with q times scale * log2 e when c16 != 0) */
int vs_qkv_proj_bf16_operands(const void *h16, const void *Wqkv16, const float *bqkv, void *qkv,
                              int32_t B, int32_t T, int32_t d, int32_t H, int32_t c16, void *stream);
int vs_to_bf16(const float *src, void *dst16, size_t n, void *stream);

/* qkv = h * Wqkv^T + b, scattered head-major: out[3][B][H][T][dh]  (simnet.py:148-153). */
int vs_qkv_proj_f32(const float *h, const float *Wqkv, const float *bqkv, float *qkv,
                    int32_t B, int32_t T, int32_t d, int32_t H, void *stream);

/* softmax(q k^T * scale + keymask) v, output [B,T,H*dh] (simnet.py:155-161); q,k,v head-major
 * [B,H,T,dh]; key_pad_mask [B,T] bytes or NULL. */
int vs_attention_f32(const float *q, const float *k, const float *v, const uint8_t *key_pad_mask,
                     float *out, int32_t B, int32_t H, int32_t T, int32_t dh, float scale,
                     void *stream);

/* The same contract on the bf16 matrix pipe (see VS_FLAG_BF16_ATTENTION); dh in {32, 64, 128}. */
int vs_attention_bf16(const float *q, const float *k, const float *v, const uint8_t *key_pad_mask,
                      float *out, int32_t B, int32_t H, int32_t T, int32_t dh, float scale,
                      void *stream);

/* The bf16 mode's storage form of the same contract (what the fused bf16 layer kernels hand to the attention):
 * q16 = bf16(q * scale * log2 e), k16, v16 head-major bf16 planes [B,H,T,dh]; out16 bf16 [B,T,H*dh]; dh in {32, 64, 128}.
 * Head dim 64 runs on the one-wave-per-SIMD kernel (csrc/vs_attention_w64.hip) unless VS_ATTN_W64 = 0. */
int vs_attention_bf16_stored(const void *q16, const void *k16, const void *v16, const uint8_t *key_pad_mask,
                             void *out16, int32_t B, int32_t H, int32_t T, int32_t dh, void *stream);
/* the factor a producer folds into q16: scale * log2 e */
float vs_attention_qscale(float scale);

/* The same contract emulated on the f16 matrix pipe (see VS_FLAG_F16X3_ATTENTION); dh in {32, 64}. */
int vs_attention_f16x3(const float *q, const float *k, const float *v, const uint8_t *key_pad_mask,
                       float *out, int32_t B, int32_t H, int32_t T, int32_t dh, float scale,
                       void *stream);

/* out = LayerNorm(A*W^T + bias + residual) * gamma + beta, eps 1e-5 (simnet.py:107,110,163,182);
 * N = d_model.  If score_w != NULL also scores[row, c] = out[row,:].score_w[c,:] + score_b[c]
 * (final_layer, simnet.py:42), through sigmoid when `sigmoid` != 0. */
int vs_linear_residual_layernorm_f32(const float *A, const float *W, const float *bias,
                                     const float *residual, const float *gamma, const float *beta,
                                     float *out, int32_t M, int32_t N, int32_t K,
                                     const float *score_w, const float *score_b, int32_t num_classes,
                                     int32_t sigmoid, float *scores, void *stream);

/* The same on the bf16 matrix pipe (see VS_FLAG_BF16_LINEAR); N <= 256. */
int vs_linear_residual_layernorm_bf16(const float *A, const float *W, const float *bias,
                                      const float *residual, const float *gamma, const float *beta,
                                      float *out, int32_t M, int32_t N, int32_t K,
                                      const float *score_w, const float *score_b, int32_t num_classes,
                                      int32_t sigmoid, float *scores, void *stream);

/* The same emulated on the f16 matrix pipe (see VS_FLAG_F16X3_LINEAR); N <= 256. */
int vs_linear_residual_layernorm_f16x3(const float *A, const float *W, const float *bias,
                                       const float *residual, const float *gamma, const float *beta,
                                       float *out, int32_t M, int32_t N, int32_t K,
                                       const float *score_w, const float *score_b, int32_t num_classes,
                                       int32_t sigmoid, float *scores, void *stream);

/* The whole MLP block of encoder layer `layer` of a packed model as ONE kernel on the bf16 matrix pipe (the form the
 * bf16 mode runs at d_model == 256; EncoderBlock / MLP, simnet.py:109-110, 180-183):
 *   out = LayerNorm(relu(h W1^T + b1) W2^T + b2 + h) * gamma2 + beta2,   h, out [M, d_model]
 * with_head != 0 also writes scores[M, num_classes] = out . final_w^T + final_b (through sigmoid if `sigmoid`). */
int vs_mlp_block_bf16(const vs_weights *w, int32_t layer, const float *h, float *out, int32_t M,
                      int32_t with_head, int32_t sigmoid, float *scores, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VS_SCORER_H */
