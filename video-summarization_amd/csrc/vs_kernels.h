// Internal C++ launcher interface between the C ABI (vs_scorer.cpp) and the kernels (vs_kernels.hip).
// Every launcher enqueues on `st`, never synchronises, and returns 0 or a hipError_t / -1 (unsupported shape).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// bf16 != 0: operands rounded to bf16 for the matrix pipe (opt-in; fp32 storage, bias, accumulation, LayerNorm)
// bf16 == (1 | VSK_STORE16): additionally the tensors that are ONLY ever consumed as bf16 matrix operands live in HBM
// as bf16: C of vsk_qkv and of vsk_linear(relu), A of vsk_linear_res_ln, q/k/v/out of the bf16 attention.  The
// rounding happens in the producer instead of the consumer: same bits, half the bytes.
enum { VSK_STORE16 = 16, VSK_A16 = 32, VSK_F16 = 64 };      // VSK_A16 (training path): A of vsk_linear is stored as bf16;
// VSK_F16 (training path's fp16 mode): the 16-bit type of this call - operands, stored C, stored A - is IEEE f16, not bf16
// Wf: the weight in fragment-major order (vsk_pack_fragments) or nullptr; enables the packed latency kernels
int vsk_linear(const float *A, const float *W, const float *Wf, const float *bias, float *C, int M, int N, int K,
               int relu, const float *pe, int T, int bf16, hipStream_t st);
// training-path epilogues on the same GEMM kernels (exact fp32): ReLU/dropout gate of a dgrad; fc1 + ReLU + dropout
int vsk_linear_gate(const float *A, const float *W, const float *Wf, const float *bias, const float *gate, float scale,
                    float *C, int M, int N, int K, hipStream_t st, int bf16 = 0);
int vsk_linear_relu_dropout(const float *A, const float *W, const float *Wf, const float *bias, float *C, int M, int N,
                            int K, unsigned long long seed, unsigned site, float p, hipStream_t st, int bf16 = 0);
int vsk_pack_fragments(const float *W, float *Wf, int N, int K, hipStream_t st);
// the fp16x3 counterpart (hi|lo f16 halves of 2^10 * W, same size): pass it as `Wf` together with bf16 == 2
int vsk_pack_fragments_f16x3(const float *W, float *Wh, int N, int K, hipStream_t st);
int vsk_qkv(const float *h, const float *Wqkv, const float *Wf, const float *bqkv, float *qkv, int B, int T, int d,
            int H, int bf16, hipStream_t st, float qscale = 1.0f);
int vsk_attention(const float *q, const float *k, const float *v, const uint8_t *mask, float *out,
                  int B, int H, int T, int dh, float scale, hipStream_t st);
int vsk_attention_packed(const float *q, const float *k, const float *v, float *out, int H, int Mtot, int dh,
                         float scale, const int *cu, const int *work, int nwork, int nw, int prec, hipStream_t st);
// the factor the bf16-storage QKV epilogue folds into q (the attention kernels' scale * log2 e, computed in one place)
float vsk_attention_qscale(float scale);
// prec 1: bf16 operands; 2: fp32 emulated with f16 hi+lo operand halves ("fp16x3")
int vsk_attention_bf16(const float *q, const float *k, const float *v, const uint8_t *mask, float *out,
                       int B, int H, int T, int dh, float scale, int prec, hipStream_t st);
// vs_attention_w64.hip: bf16 q (pre-scaled) / k / v planes in, bf16 out, head dim 64, one wave per SIMD (4 waves x 64 query
// rows per block).  -1: shape outside this kernel (the caller uses attn_fwd_lp_pipe).
int vsk_attention_bf16_w64(const void *q, const void *k, const void *v, const uint8_t *mask, void *out, int B, int H, int T,
                           hipStream_t st);
int vsk_attention_bf16_w64_packed(const void *q, const void *k, const void *v, void *out, int H, int Mtot, const int *cu,
                                  const int *work, int nwork, hipStream_t st);
int vsk_linear_res_ln(const float *A, const float *W, const float *Wf, const float *bias, const float *res,
                      const float *gamma, const float *beta, float *out, int M, int N, int K,
                      const float *score_w, const float *score_b, int num_classes, int sigmoid,
                      float *scores, int bf16, hipStream_t st);
// out = LayerNorm(a + res) * gamma + beta (+ score head): the row pass behind a plain GEMM for d_model > 256
int vsk_rows_res_ln(const float *a, const float *res, const float *gamma, const float *beta, float *out, int M, int d,
                    const float *score_w, const float *score_b, int num_classes, int sigmoid, float *scores,
                    hipStream_t st, void *out16 = nullptr, int dn = 0,       // dn: LayerNorm width of an embedded model (0: d)
                    int nsplit = 0, const float *pbias = nullptr);           // nsplit > 0: `a` = K-slice partials [nsplit][M][d], pbias the Linear's bias
// A batch of small matrix jobs in ONE launch (round 4: a reference-sized training step rebuilt its weight images with 55 tiny
// launches - 17 transposes, 34 fragment packs, 4 conversions - per optimizer step; the host enqueue alone was ~0.2 ms)
struct VskMatJobs {
    enum { MAX = 24 };
    const float *in[MAX];
    float *out[MAX];
    int rows[MAX], cols[MAX];      // the INPUT's shape [rows, cols] (pack_fragments: W [N = rows, K = cols])
    int n;
};
int vsk_pack_fragments_batch(const VskMatJobs &jobs, hipStream_t st);      // each job: vsk_pack_fragments(in, out, rows, cols)
int vsk_attention_splitkv(const float *q, const float *k, const float *v, const uint8_t *mask, float *out,
                          int B, int H, int T, int dh, float scale, hipStream_t st);       // latency mode: keys split over a block's waves
// latency mode (VS_FLAG_SPLITK): split-K partial products through the fragment-major latency kernel, and the embedding's reduction
int vsk_linear_parts(const float *A, const float *Wf, float *parts, int M, int N, int K, int nsplit, hipStream_t st,
                     int f16x3 = 0);      // f16x3: Wf is the pack_fragments_f16x3 copy, the product emulated on the f16 pipe
int vsk_sum_parts_pe(const float *parts, int nsplit, const float *bias, const float *pe, int T, float *out, int M, int N, hipStream_t st);      // out16: optional bf16 copy of the output rows
int vsk_diag_attention(const float *q, const float *k, const float *v, float *out, int B, int H, int T, float scale,
                       unsigned long long *diag, hipStream_t st);      // diagnostic library only; returns the blocks launched
int vsk_diag_attention_lp(const float *q, const float *k, const float *v, float *out, int B, int H, int T, float scale,
                          int prec, unsigned long long *diag, hipStream_t st);
int vsk_diag_gemm(const float *A, const float *W, const float *bias, float *C, int M, int N, int K,
                  int grid, unsigned long long *diag, hipStream_t st);
// fc1 + ReLU + fc2 + residual + LayerNorm (+ score head) in one kernel; d_model == 256 only (-1 otherwise)
int vsk_mlp_fused(const float *H1, const float *W1, const float *b1, const float *W2, const float *b2,
                  const float *gamma, const float *beta, float *out, int M, int d,
                  const float *score_w, const float *score_b, int num_classes, int sigmoid, float *scores,
                  hipStream_t st);
// bf16 mode, d_model == 256: fc1 + ReLU + fc2 + residual + LayerNorm (+ score head) as one kernel whose hidden
// activations stay in registers (vs_mlp_fused.hip) - and, when the attention output is stored as bf16 (att16 != nullptr),
// the out-projection + residual + norm1 in front of it as well (h is then the layer input, else h = h1).
// img: the per-chunk LDS images of Wo, W1, W2 made by vsk_pack_mlp_bf16 (vsk_mlp_bf16_image_bytes(d) bytes, 256-byte
// aligned; 0 = this d_model has no fused kernel).
bool vsk_mlp_bf16_supported(int d);
size_t vsk_mlp_bf16_image_bytes(int d);
int vsk_pack_mlp_bf16(const float *Wo, const float *W1, const float *W2, void *img, int d, hipStream_t st);
// next != nullptr: the kernel also projects the rows it produced to the NEXT layer's q * qscale / k / v (bf16, head-major:
// three [M / T, H, T, d / H] planes of M * d elements at qkv16; M a multiple of T) - bit-identical to vsk_qkv(.. STORE16)
struct VskNextQkv {
    const void *img;            // vsk_pack_qkv_bf16 image of that layer's Wqkv (vsk_qkv_bf16_image_bytes(d) bytes)
    const float *bqkv;
    void *qkv16;
    int T, H;
    float qscale;
};
// embedding + positional rows (+ the first layer's QKV) on the same design; 0 bytes = shape not supported (d != 256, K % 64)
size_t vsk_embed_bf16_image_bytes(int d, int K);
int vsk_pack_embed_bf16(const float *W, void *img, int d, int K, hipStream_t st);
int vsk_embed_bf16(const float *x, const void *img, const float *bias, const float *pe, int T, float *out, int M, int d, int K,
                   const VskNextQkv *next, hipStream_t st);
size_t vsk_qkv_bf16_image_bytes(int d);
int vsk_pack_qkv_bf16(const float *Wqkv, void *img, int d, hipStream_t st);
int vsk_mlp_bf16(const float *h, const void *att16, const float *bo, const float *gamma1, const float *beta1,
                 const void *img, const float *b1, const float *b2,
                 const float *gamma, const float *beta, float *out, int M, int d,
                 const float *score_w, const float *score_b, int num_classes, int sigmoid, float *scores,
                 const VskNextQkv *next, hipStream_t st);
// packed ragged batches: row offsets cu[B+1] and the (video, query tile) work list from device lengths; gather of
// the positional rows pe[t] of every frame into rows[Mtot, d]
int vsk_plan_packed(const int *lengths_dev, int B, int qb, int *cu, int *work, int work_cap, hipStream_t st);   // work_cap: (video, tile) pairs `work` can hold
int vsk_gather_rows(const float *pe, const int *cu, int B, int tmax, int d, float *rows, hipStream_t st);
// up to VSK_COPY_MAX_SEGS device-to-device float copies in ONE launch (vs_weights_pack / _update)
enum { VSK_COPY_MAX_SEGS = 20 };
struct VskCopySegs {
    const float *src[VSK_COPY_MAX_SEGS];
    float *dst[VSK_COPY_MAX_SEGS];
    unsigned n[VSK_COPY_MAX_SEGS];       // floats
    int count;
};
int vsk_copy_segments(const VskCopySegs &segs, hipStream_t st);
// use_cls=True: h [B, T+1, d] = class token row + the embedded frames e [B, T, d]; mask1 [B, T+1] (or both nullptr)
int vsk_insert_cls(const float *e, const float *cls, const uint8_t *mask, float *h, uint8_t *mask1, int B, int T, int d,
                   hipStream_t st);
int vsk_skinny_max_rows();      // rows up to which the skinny (latency) kernels are used

// A/B and test switches (DESIGN.md "Environment switches").  Read from the environment ONCE, when the library is
// first used; afterwards only vs_set_option() changes them, so no forward pays a getenv.
struct VskOptions {
    int skinny_rows;      // VS_SKINNY_ROWS   (default 16384)
    int lp_min_rows;      // VS_LP_MIN_ROWS   (default 8192)
    int train_lp_min_rows;// VS_TRAIN_LP_MIN_ROWS (default 1024): frames per batch above which a low-precision TRAINING request is honoured
                          // (profiles/r04_lp_min_rows_sweep.txt: break-even at ~1280 frames, 1.1x at 2560, 1.7x at 8192)
    int lp_min_rows_fused;// VS_LP_MIN_ROWS_FUSED (default 256): the same threshold where the fused bf16 layer kernels apply
    int gemm_nwm2;        // VS_GEMM_NWM2     128x128 four-wave GEMM blocks only
    int gemm_nj2;         // VS_GEMM_NJ2      128-column GEMM tiles only
    int attn_nw4;         // VS_ATTN_NW4      4-wave attention blocks only
    int attn_lp_simple;   // VS_ATTN_LP_SIMPLE phase-aligned low-precision attention
    int attn_w64;         // VS_ATTN_W64      (default 1) bf16-stored head-dim-64 attention on the one-wave-per-SIMD kernel (0: the 8-wave kernel, A/B)
    int lp_store32;       // VS_LP_STORE32    bf16 mode keeps q/k/v, the attention output and the MLP hidden tensor fp32 in HBM (A/B)
    int lp_mlp_unfused;   // VS_LP_MLP_UNFUSED bf16 mode runs fc1 and fc2 + LayerNorm as two kernels (A/B)
    int lp_tail_unfused;  // VS_LP_TAIL_UNFUSED bf16 mode runs the out-projection + norm1 as its own kernel in front of the fused MLP (A/B)
    int lp_qkv_unfused;   // VS_LP_QKV_UNFUSED  bf16 mode runs every layer's QKV projection as its own kernel (A/B)
    int lp_tile256;       // VS_LP_TILE256     the fused bf16 layer kernels always use 256-row tiles / 8-wave blocks (A/B)
    int lp_embed_unfused; // VS_LP_EMBED_UNFUSED bf16 mode runs the embedding as the generic GEMM + the first QKV kernel (A/B)
    int mlp_fusion;       // VS_MLP_FUSION    (diagnostic builds only)
    int mlp_abl;          // VS_MLP_ABL       (diagnostic builds only)
    int attn_w64_checked; // VS_ATTN_W64_CHECKED the one-wave-per-SIMD attention skips its optimistic pass (every tile checked; A/B and test pin)
    int attn_legacy;      // VS_ATTN_LEGACY   (diagnostic builds only)
    int attn_w64_abl;     // VS_ATTN_W64_ABL  (diagnostic builds only) timing ablation of the one-wave-per-SIMD attention
};
VskOptions &vsk_options();
// per-device cache of the CU count (one process may drive several GPUs)
int vsk_device_cus();

// ---- vs_gemm_ring.hip: bf16 x bf16 operands from HBM (wide models' bf16 mode) ----
// epi: 0 bias, 1 bias + ReLU, 3 bias + q/k/v head-major scatter (q additionally times qscale when c16)
bool vsk_gemm16_supported(int M, int N, int K);
int vsk_gemm16(const void *A16, const void *W16, const float *bias, void *C, int M, int N, int K, int epi, int c16,
               int T, int H, int dh, float qscale, hipStream_t st);
int vsk_to_bf16(const float *src, void *dst, size_t n, hipStream_t st);
