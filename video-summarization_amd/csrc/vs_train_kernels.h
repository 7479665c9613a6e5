// Internal C++ launcher interface between the training C ABI (vs_train.cpp) and its kernels
// (vs_train_kernels.hip, vs_train_attention.hip).  Same conventions as vs_kernels.h: every launcher enqueues on
// `st`, never synchronises, and returns 0, a hipError_t (> 0) or -1 (unsupported shape).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "vs_kernels.h"

// z = dropout_{p,site}(a) + res; y = LayerNorm(z) * gamma + beta (y_copy: optional second copy); stats[row] = (mean, rstd);
// optional score head scores[row, c] = y . score_w[c] + score_b[c]
int vst_rows_fwd(const float *a, const float *res, const float *gamma, const float *beta, float *z, float *y,
                 float *y_copy, float *stats, int M, int d, unsigned long long seed, unsigned site, float p,
                 const float *score_w, const float *score_b, int num_classes, float *scores, hipStream_t st, int dn = 0);   // dn: LayerNorm width of an embedded model (0: d)
// LayerNorm backward; part = [vst_ln_bwd_blocks(M)][2][d] partial (d_gamma, d_beta); dbranch (optional) = forward
// dropout mask applied to dz; dy may be NULL (zero) and dsc/score_w add the score head's contribution
int vst_ln_bwd_blocks(int M);
int vst_ln_bwd(const float *dy, const float *dsc, const float *score_w, int num_classes, const float *z,
               const float *stats, const float *gamma, float *dz, float *dbranch, float *part, int M, int d,
               unsigned long long seed, unsigned site, float p, hipStream_t st, int dn = 0);
int vst_dropout_rows(float *x, int M, int cols, unsigned long long seed, unsigned site, float p, hipStream_t st);
int vst_gate_bwd(float *g, const float *act, size_t n, float scale, hipStream_t st);
int vst_head_rowdot(const float *dO, const float *O, float *delta, int M, int T, int H, int dh, hipStream_t st, int do16 = 0);      // do16 1: dO stored as bf16; 2: fp32 dO rounded to bf16 in the dot
// part = [nblk = vst_ln_bwd_blocks(M)][d] partial sums of w[row*ws] * Y[row,:], then [nblk] partial sums of w[row*ws]
int vst_weighted_colsum(const float *w, int ws, const float *Y, float *part, int M, int d, hipStream_t st);
// dW[N,K] = dY[M,N]^T X[M,K] (+ db = column sums of dY when db0 != NULL); rows of the result are dealt to up to three
// destination tensors of rows_per_dest rows; work >= vst_wgrad_workspace_floats(M, N, K) floats
enum { VST_WGRAD_Y16 = 2, VST_WGRAD_X16 = 4 };     // vst_wgrad prec = 1 | flag: dY / X is stored as bf16
int vst_wgrad_splits(int M, int N, int K);
size_t vst_wgrad_workspace_floats(int M, int N, int K);
int vst_wgrad(const float *dY, int ldy, const float *X, int ldx, int M, int N, int K, float *dW0, float *dW1, float *dW2,
              float *db0, float *db1, float *db2, int rows_per_dest, float *work, hipStream_t st, int prec = 0);   // prec 1: bf16 matrix pipe
int vst_reduce_rows(const float *part, int S, int rows, int cols, float *d0, float *d1, float *d2, int rows_per_dest,
                    hipStream_t st);
int vst_transpose_batch(const VskMatJobs &jobs, hipStream_t st);      // each job: out [cols, rows] = in [rows, cols]^T, one launch
int vst_transpose(const float *in, float *out, int rows, int cols, hipStream_t st);
int vst_mse_mask_blocks(int n);
int vst_mse_mask_fwd(const float *out, const float *tgt, const unsigned char *mask, int n, int mean, float *part,
                     float *loss, hipStream_t st);
int vst_mse_mask_bwd(const float *out, const float *tgt, const unsigned char *mask, const float *gout, int n, int mean,
                     float *dout, hipStream_t st);

// ---- attention (vs_train_attention.hip); q, k, v head-major [B,H,T,dh]; out / dO token-major [B*T, H*dh] ----
// forward with dropout on the attention weights; lse2[b,h,t] = log2 sum_j exp(s_ij) (base-2 log-sum-exp of the
// scaled, masked scores) is saved for the backward
// dbits: both bit-packed copies of the layer's keep decisions (vst_attention_dropout_bits) or nullptr (hash per element)
size_t vst_attention_dropout_bits_words(int B, int H, int T);
int vst_attention_dropout_bits(unsigned *dbits, int B, int H, int T, unsigned long long seed, unsigned site, float p,
                               hipStream_t st);
int vst_attention_fwd(const float *q, const float *k, const float *v, const uint8_t *mask, float *out, float *lse2,
                      int B, int H, int T, int dh, float scale, unsigned long long seed, unsigned site, float p,
                      hipStream_t st, const unsigned *dbits = nullptr);
// dqkv token-major [B*T, 3*H*dh] (columns: dq | dk | dv, head h at h*dh); delta from vst_head_rowdot
int vst_attention_bwd(const float *q, const float *k, const float *v, const uint8_t *mask, const float *dO,
                      const float *lse2, const float *delta, float *dqkv, int B, int H, int T, int dh, float scale,
                      unsigned long long seed, unsigned site, float p, hipStream_t st, const unsigned *dbits = nullptr);
// the same two calls on the bf16 matrix pipe (vs_train_attention_bf16.hip; head dim 32 / 64): operands rounded to bf16, fp32
// softmax / accumulation; p > 0 needs the bit-packed masks
bool vst_attention_bf16_supported(int dh);
int vst_attention_fwd_bf16(const float *q, const float *k, const float *v, const uint8_t *mask, float *out, float *lse2,
                           int B, int H, int T, int dh, float scale, float p, const unsigned *dbits, hipStream_t st, int in16 = 0);
int vst_attention_bwd_bf16(const float *q, const float *k, const float *v, const uint8_t *mask, const float *dO,
                           const float *lse2, const float *delta, float *dqkv, int B, int H, int T, int dh, float scale,
                           float p, const unsigned *dbits, hipStream_t st, int in16 = 0, int out16 = 0);      // in16: q (pre-scaled), k, v stored as bf16; out16: dqkv written as bf16
// test hook: keep[b,h,i,j] (bytes) of the attention-weight dropout, exactly as the two kernels above draw it
int vst_attention_dropout_mask(uint8_t *keep, int B, int H, int T, unsigned long long seed, unsigned site, float p,
                               hipStream_t st);
// test hook: keep[row, col] (bytes) of the elementwise dropouts
int vst_rows_dropout_mask(uint8_t *keep, int M, int cols, unsigned long long seed, unsigned site, float p, hipStream_t st);

// ---- PretrainModel head (vs_pretrain_kernels.hip; reference simnet_pretrain.py:49-69, 80-98) ----
// feats [B,T,F] (video_transform output), scores [B,T] (the scorer's logits), mask [B,T] bytes or NULL, vid [B,F].
// stats (device, floats): per video [B][VSP_STATS] (see vs_pretrain_kernels.hip) + pooled / sumx vectors; part: scratch.
size_t vsp_head_scratch_floats(int B, int T, int F);
int vsp_head_forward(const float *feats, const float *scores, const uint8_t *mask, const float *vid, int B, int T, int F,
                     float inv_temp, int entropy_penalty, float *scratch, float *losses /*[3]*/, hipStream_t st);
int vsp_head_backward(const float *feats, const float *scores, const uint8_t *mask, const float *vid, int B, int T, int F,
                      float inv_temp, int entropy_penalty, const float *scratch, const float *g_losses /*[3]*/,
                      float *d_feats, float *d_scores, hipStream_t st);

// ---- vs_train_gemm_rows.hip: A-stationary bf16 GEMM for K = 256 with a bf16 C (fc1 forward, fc2 input gradient) ----
// epi 0: dropout(relu(.)) (seed, site, p); 1: gate (gate16 = the bf16-stored activation, scale); 2: relu; 3: q / k / v planes
bool vst_gemm_rows16_supported(int M, int N, int K, bool any_rows = false);
int vst_gemm_rows16(const float *A, const void *W16, const float *bias, void *C16, const void *gate16, int M, int N, int K, int epi,
                    float scale, unsigned long long seed, unsigned site, float p, hipStream_t st, int T = 0, int H = 0, int dh = 0);
