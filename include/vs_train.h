/*
 * vs_train.h — C ABI of the TRAINING path of libvsscore.so: SimNet.forward in train mode
 * (dropout, activations kept) and its backward, as the reference's callers drive it:
 *     pred, _ = model(feature, mask); loss = mse_with_mask_loss(pred, target, mask); loss.backward()
 *                                                        reference src/train.py:111-131, src/pretrain.py:49-86
 * The reference has no FFI; what a binding binds is torch.autograd's contract for this nn.Module: a forward that
 * keeps what the backward needs, and a backward from (d_scores, d_hidden) to the gradients of every parameter
 * (and of the input).  Same conventions as vs_scorer.h: device pointers, work enqueued on the caller's stream, int
 * status + vs_last_error().  ONE allocation exists on this path: the transposed weights of the dgrad GEMMs, made by
 * vs_train_prepare() (or, if that was never called, by the first vs_train_backward).  A handle's calls (forward,
 * backward, vs_weights_update) may use different streams: the lazily rebuilt weight images are ordered on the stream of the
 * call that rebuilds them and a call on another stream waits, on the device, for the event recorded behind that work
 * (vs_scorer.h); one host thread per handle at a time.  All arithmetic is exact fp32 (v_mfma_f32_32x32x2_f32); every
 * reduction runs in a fixed order, so a step is bitwise reproducible for a given dropout seed.
 */
#ifndef VS_TRAIN_H
#define VS_TRAIN_H

#include "vs_scorer.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The dropout modules of the reference (all nn.Dropout, active in train mode only):
 *   p_embed  PositionalEncoding.dropout, p = `sparsity`   simnet.py:224,237   (0 in train.py:33 / simnet_pretrain.py:30)
 *   p        drop_rate of every EncoderBlock: attention weights simnet.py:159, dropout1 :107, mlp.dropout :181,
 *            dropout2 :110
 * The keep decisions are a counter-based hash of (seed, module, row, column) — see csrc/vs_train_device.h — so the
 * backward rebuilds them from `seed` alone; pass the SAME struct to forward and backward.  torch's Philox stream is
 * not reproduced (no re-implementation can): statistics and forward/backward consistency are what the tests pin. */
typedef struct vs_dropout_cfg {
    float p_embed;
    float p;
    uint64_t seed;
    uint32_t flags;      /* VS_TRAIN_FLAG_*; 0 = exact fp32 everywhere */
    uint32_t reserved;   /* 0 */
} vs_dropout_cfg;

/* Low-precision training, the counterpart of the reference's fp16 autocast (`with amp.autocast():` around the forward,
 * train.py:120, pretrain.py:59): every Linear of the forward (embedding, q/k/v, feature_projection, fc1, fc2), every
 * dgrad GEMM and every weight-gradient GEMM of the backward multiplies bf16-ROUNDED operands on the bf16 matrix pipe
 * (v_mfma_f32_32x32x16_bf16) with fp32 accumulation; tensors stay fp32 in HBM, and bias, residual, LayerNorm, softmax,
 * the attention products, dropout and the loss stay exact fp32 - what autocast keeps in fp32 (softmax, layer_norm, mse)
 * plus the attention matmuls.  Applied from 8192 frames per batch up (below, the exact latency kernels are faster);
 * pass the SAME flags to forward and backward.  With this flag the tensors of the activation record that are only ever
 * bf16 matrix operands (the MLP hidden tensor; with VS_TRAIN_FLAG_BF16_ATTENTION also q / k / v) are STORED as bf16 (first
 * half of their fields) - the same values their readers used to round to, so results do not change (the A/B switches
 * VS_LP_STORE32 / VS_LP_MLP_UNFUSED of vs_set_option must not be flipped between a forward and its backward: they decide the
 * record's format).  Gradients then differ from the float64 truth by 1-3e-2 in relative L2
 * norm per tensor (tests/tolerances.py: TRAIN_LP_GRAD_L2), where the exact path is at 1e-6. */
#define VS_TRAIN_FLAG_BF16_LINEAR 1u
/* ... and the attention products too (S = q k^T, P v, and the five products of the backward), head dim 32 / 64 / 128: q * scale,
 * k, v, dO, the probabilities and dS rounded to bf16, fp32 scores / softmax / lse / accumulation. */
#define VS_TRAIN_FLAG_BF16_ATTENTION 2u
#define VS_TRAIN_FLAG_BF16 (VS_TRAIN_FLAG_BF16_LINEAR | VS_TRAIN_FLAG_BF16_ATTENTION)
/* Modifier of the two flags above: the 16-bit type - of every rounded operand and of every 16-bit-stored tensor of the
 * record - is IEEE fp16 (v_mfma_f32_32x32x16_f16), the reference's own autocast type on CUDA (train.py:120), instead of
 * bf16: 11 significant bits against 8, so the gradients sit ~8x closer to the float64 truth (tests/tolerances.py:
 * TRAIN_FP16_GRAD_L2), but the range ends at 65 504 and values under 6e-8 vanish - use it the way the reference does,
 * with a loss scale (torch.amp.GradScaler, train.py:60,126-128): an operand that overflows becomes inf in the operand
 * and inf / NaN in the gradients it reaches, which is exactly what GradScaler's check looks for before it skips the step.
 * Accumulation, bias, residual, LayerNorm, softmax and the loss stay fp32 as in the bf16 mode.  The A-stationary fc1 / QKV
 * kernels are bf16-only: this mode runs the tiled GEMMs everywhere. */
#define VS_TRAIN_FLAG_FP16 4u
#define VS_TRAIN_FLAG_FP16_ALL (VS_TRAIN_FLAG_BF16 | VS_TRAIN_FLAG_FP16)

/* Gradient destinations: the mirror of vs_layer_params / vs_model_params (nn.Linear layout [out, in]); every
 * pointer is a device buffer of the parameter's shape that the backward OVERWRITES (it does not accumulate). */
typedef struct vs_layer_grads {
    float *wq, *bq, *wk, *bk, *wv, *bv;
    float *wo, *bo;
    float *ln1_g, *ln1_b;
    float *w1, *b1, *w2, *b2;
    float *ln2_g, *ln2_b;
} vs_layer_grads;

typedef struct vs_model_grads {
    float *embed_w, *embed_b;
    const vs_layer_grads *layers;               /* HOST array of num_layers entries */
    float *final_w, *final_b;
} vs_model_grads;

/* Allocates (first call per handle: hipMalloc, synchronises the device - call it OUTSIDE the training loop and outside
 * any stream capture) and (re)builds the transposed weight copies the backward's dgrad GEMMs read, stream-ordered on
 * `stream`.  Later calls - and the backward itself - only rebuild them after a vs_weights_update: no allocation. */
int vs_train_prepare(vs_weights *w, void *stream);

/* Bytes of the activation record one forward leaves for its backward, and of the scratch either call needs. */
size_t vs_train_saved_bytes(const vs_weights *w, int32_t B, int32_t T);
size_t vs_train_workspace_bytes(const vs_weights *w, int32_t B, int32_t T);

/* Replaces: SimNet.forward(x, mask) in TRAIN mode (simnet.py:32-45 with every nn.Dropout active).
 *   x [B,T,in_features], key_pad_mask [B,T] bytes or NULL, scores [B,T,num_classes] raw logits,
 *   hidden [B,T,d_model] or NULL, saved >= vs_train_saved_bytes, workspace >= vs_train_workspace_bytes (256-B aligned).
 * `drop` may be NULL (no dropout: eval-mode values, but with the activation record kept). */
int vs_train_forward(const vs_weights *w, const float *x, const uint8_t *key_pad_mask, int32_t B, int32_t T,
                     const vs_dropout_cfg *drop, float *scores, float *hidden, void *saved, size_t saved_bytes,
                     void *workspace, size_t workspace_bytes, void *stream);

/* Replaces: autograd's backward through that forward (loss.backward(), train.py:126).
 *   d_scores [B,T,num_classes] or NULL (zero), d_hidden [B,T,d_model] or NULL (zero): gradients of the two return
 *   values of forward; `saved` as written by vs_train_forward for the same (w, x, mask, B, T, drop);
 *   grads: every parameter's gradient (overwritten); dx [B,T,in_features] or NULL (input gradient not wanted).
 * The handle must hold the parameter values the forward used (no vs_weights_update in between). */
int vs_train_backward(vs_weights *w, const float *x, const uint8_t *key_pad_mask, int32_t B, int32_t T,
                      const vs_dropout_cfg *drop, const float *d_scores, const float *d_hidden, const void *saved,
                      size_t saved_bytes, const vs_model_grads *grads, float *dx, void *workspace,
                      size_t workspace_bytes, void *stream);

/* Replaces: utils.mse_with_mask_loss(output, targets, mask, reduction) (reference src/utils/utils.py:45-56, called
 * at train.py:122): mean (reduction "avg") or sum over ALL n = B*T entries of ((output - target) * scale)^2 with
 * scale = 0 on masked frames.  output/target [n] fp32, mask [n] bytes or NULL; scratch >= 256 floats; loss [1]. */
int vs_mse_mask_loss_forward(const float *output, const float *target, const uint8_t *mask, int32_t n, int32_t mean,
                             float *scratch, float *loss, void *stream);
int vs_mse_mask_loss_backward(const float *output, const float *target, const uint8_t *mask, const float *d_loss,
                              int32_t n, int32_t mean, float *d_output, void *stream);

/* Replaces: PretrainModel.forward below its encoder call (reference src/model/simnet_pretrain.py:79-98 with
 * repelling_loss :49-71, entropy :43-47, cross_entropy_loss :35-41): video_transform Linear on the scorer's hidden
 * state, the repelling loss, the masked score-softmax pooling with its centering penalty and the soft cross-entropy
 * against the video representation - forward and backward.
 *   hidden [B,T,d] and logits [B,T]: the two outputs of the scorer; key_pad_mask [B,T] bytes or NULL;
 *   vid [B,F] the target video representation; vt_w [F,d], vt_b [F]: video_transform (F = 512 in the reference);
 *   temp = sharpening_t (:29);  entropy_penalty != 0: pen_met == "entropy" (:88), else the norm penalty (:92).
 *   feats [B,T,F] (out: video_transform(hidden), kept for the backward), head_state >= vs_pretrain_head_state_bytes
 *   (out: per-video statistics, kept for the backward), losses [3] = (distillation, centering, repelling) batch means.
 * The repelling loss is evaluated as (||sum_t x^_t||^2 - sum_t ||x^_t||^2) / T^2 - the mean of the reference's
 * [T,T] cosine matrix without its diagonal, never materialised. */
size_t vs_pretrain_head_state_bytes(int32_t B, int32_t T, int32_t F);
size_t vs_pretrain_head_workspace_bytes(int32_t B, int32_t T, int32_t d, int32_t F);
int vs_pretrain_head_forward(const float *hidden, const float *logits, const uint8_t *key_pad_mask, const float *vid,
                             const float *vt_w, const float *vt_b, int32_t B, int32_t T, int32_t d, int32_t F,
                             float temp, int32_t entropy_penalty, float *feats, void *head_state, float *losses,
                             void *stream);
/* d_losses [3]: gradients of the three returned losses.  Outputs (all overwritten): d_hidden [B,T,d], d_logits [B,T],
 * d_vt_w [F,d], d_vt_b [F].  workspace >= vs_pretrain_head_workspace_bytes, 256-byte aligned. */
int vs_pretrain_head_backward(const float *hidden, const float *logits, const uint8_t *key_pad_mask, const float *vid,
                              const float *vt_w, const float *feats, void *head_state, const float *d_losses,
                              int32_t B, int32_t T, int32_t d, int32_t F, float temp, int32_t entropy_penalty,
                              float *d_hidden, float *d_logits, float *d_vt_w, float *d_vt_b, void *workspace,
                              size_t workspace_bytes, void *stream);

/* Per-kernel entry points for the parity tests (not needed by a binding). */

/* softmax(q k^T * scale + keymask) with dropout p on the weights, times v; q,k,v head-major [B,H,T,dh]; out [B,T,H*dh];
 * lse2 [B,H,T] = base-2 log-sum-exp of the scaled masked scores (what the backward re-normalises with). */
int vs_train_attention_forward(const float *q, const float *k, const float *v, const uint8_t *key_pad_mask, float *out,
                               float *lse2, int32_t B, int32_t H, int32_t T, int32_t dh, float scale, uint64_t seed,
                               uint32_t site, float p, void *stream);
/* d_out [B,T,H*dh] -> dqkv [B,T,3*H*dh] (columns dq | dk | dv, head h at h*dh); scratch >= B*H*T floats. */
int vs_train_attention_backward(const float *q, const float *k, const float *v, const uint8_t *key_pad_mask,
                                const float *out, const float *d_out, const float *lse2, float *dqkv, float *scratch,
                                int32_t B, int32_t H, int32_t T, int32_t dh, float scale, uint64_t seed, uint32_t site,
                                float p, void *stream);
/* The attention-weight dropout of one layer, bit-packed (what the full training path keeps in its activation record):
 * dbits >= vs_train_attention_dropout_bits_bytes(B, H, T) bytes; same keep decisions as vs_train_dropout_mask_attention. */
size_t vs_train_attention_dropout_bits_bytes(int32_t B, int32_t H, int32_t T);
int vs_train_attention_dropout_bits(void *dbits, int32_t B, int32_t H, int32_t T, uint64_t seed, uint32_t site, float p,
                                    void *stream);
/* vs_train_attention_forward / _backward on the bf16 matrix pipe (VS_TRAIN_FLAG_BF16_ATTENTION); dh in {32, 64, 128}; p > 0 needs
 * dbits (from vs_train_attention_dropout_bits with the same seed / site / p).  in16 != 0: q, k, v point to bf16 [B,H,T,dh]
 * planes, q already multiplied by scale * log2(e) - what the training forward stores when both low-precision flags are set
 * (results are bit-identical to the fp32-stored form, whose values the kernels round to the same bf16). */
int vs_train_attention_forward_bf16(const float *q, const float *k, const float *v, const uint8_t *key_pad_mask, float *out,
                                    float *lse2, int32_t B, int32_t H, int32_t T, int32_t dh, float scale, float p,
                                    const void *dbits, int32_t in16, void *stream);
int vs_train_attention_backward_bf16(const float *q, const float *k, const float *v, const uint8_t *key_pad_mask,
                                     const float *out, const float *d_out, const float *lse2, float *dqkv, float *scratch,
                                     int32_t B, int32_t H, int32_t T, int32_t dh, float scale, float p, const void *dbits,
                                     int32_t in16, void *stream);

/* dW [N,K] = dY[M,N]^T X[M,K], db [N] = column sums of dY; scratch >= vs_train_wgrad_scratch_floats(M,N,K) floats. */
size_t vs_train_wgrad_scratch_floats(int32_t M, int32_t N, int32_t K);
int vs_train_wgrad(const float *dY, const float *X, int32_t M, int32_t N, int32_t K, float *dW, float *db,
                   float *scratch, void *stream);
/* The same on the bf16 matrix pipe (VS_TRAIN_FLAG_BF16_LINEAR): dY and X rounded to bf16, fp32 accumulation; db exact. */
int vs_train_wgrad_bf16(const float *dY, const float *X, int32_t M, int32_t N, int32_t K, float *dW, float *db,
                   float *scratch, void *stream);
/* keep masks (bytes, 1 = kept) exactly as the kernels draw them: attention weights [B,H,T,T]; elementwise [M,cols] */
int vs_train_dropout_mask_attention(uint8_t *keep, int32_t B, int32_t H, int32_t T, uint64_t seed, uint32_t site,
                                    float p, void *stream);
int vs_train_dropout_mask_rows(uint8_t *keep, int32_t M, int32_t cols, uint64_t seed, uint32_t site, float p,
                               void *stream);
/* Test hook: where a field of layer `layer` lives in the activation record of vs_train_forward (byte offset and
 * float count).  field 0: the MLP activation dropout(relu(fc1)) [B*T, 4*d_model] - its sign pattern is the ReLU /
 * dropout gate the backward applies, which a float64 checker must share (a ReLU input within fp32 rounding of zero
 * may switch the other way in float64); 1: attention output [B*T, d]; 2: y1 [B*T, d]; 3: y2 [B*T, d]. */
int vs_train_saved_field(const vs_weights *w, int32_t B, int32_t T, int32_t layer, int32_t field, size_t *offset_bytes,
                         size_t *count);
/* The form of the activation record the LAST vs_train_forward of the calling thread wrote: bit 31 set (valid), bit 0 bf16
 * Linear / dgrad / wgrad products, bit 1 bf16 attention products, bits 2..4 which tensors of the record are bf16 planes
 * (q/k/v; MLP hidden; written by the A-stationary GEMM), bit 5 the 16-bit type is fp16 (VS_TRAIN_FLAG_FP16).  Bits 0 and 1 both clear = the exact fp32 path ran, whatever
 * vs_dropout_cfg.flags asked for (low-precision training applies above VS_TRAIN_LP_MIN_ROWS frames per batch, default 1024).
 * Pass the value to vs_train_backward in vs_dropout_cfg.reserved: the backward then reads the record in the form it was
 * written in even if a library switch (vs_set_option) changed in between; reserved == 0 derives the form again. */
uint32_t vs_train_last_format(void);

/* module numbers (`site`) of the dropouts: embedding = 0; layer l: 1 + 4*l + {0 attention, 1 dropout1, 2 mlp, 3 dropout2} */
uint32_t vs_train_dropout_site(int32_t layer, int32_t which);

#ifdef __cplusplus
}
#endif
#endif /* VS_TRAIN_H */
