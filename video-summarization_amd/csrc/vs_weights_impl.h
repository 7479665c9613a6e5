// vs_weights_impl.h — the packed-parameter handle behind `vs_weights` (include/vs_scorer.h), shared by the scoring
// C ABI (vs_scorer.cpp) and the training C ABI (vs_train.cpp).  Internal: not part of the public headers.
#pragma once
#include <stddef.h>

#include <vector>

#include "vs_scorer.h"

struct LayerOff {
    size_t wqkv, bqkv, wo, bo, ln1g, ln1b, w1, b1, w2, b2, ln2g, ln2b;
    size_t f_wqkv, f_wo, f_w1, f_w2;        // fragment-major copies for the latency kernels
    size_t h_wqkv, h_wo, h_w1, h_w2;        // their fp16x3 counterparts (hi|lo f16 halves, same size)
    size_t b_mlp;                           // Wo, W1, W2 as the LDS images of the fused bf16 layer-tail kernel (vsk_pack_mlp_bf16)
    size_t b_qkv;                           // Wqkv as the LDS images of that kernel's QKV epilogue (vsk_pack_qkv_bf16)
    size_t r_wqkv, r_wo, r_w1, r_w2;        // plain row-major bf16 copies (family VSW_ROWS16): the bf16-operand GEMM of d_model > 256
                                            // (vs_gemm_ring.hip) and the training path's A-stationary GEMM (vs_train_gemm_rows.hip)
};

// transposed weights for the dgrad GEMMs of the training backward (dX = dY W is an NT GEMM against W^T); built
// lazily by the first vs_train_backward after each pack / update, never for a scoring-only user
struct LayerOffT {
    size_t t_wqkv, t_wo, t_w1, t_w2;          // W^T, row-major
    size_t t16_w2;                            // W2^T as bf16 (the A-stationary fc2 input-gradient GEMM of the bf16 training mode)
    size_t tf_wqkv, tf_wo, tf_w1, tf_w2;      // the same in fragment-major order (latency kernels at small batch)
};

struct vs_weights {
    vs_model_desc desc;
    float *blob = nullptr;        // one device allocation
    size_t blob_floats = 0;
    int device = 0;               // the device the blob lives on
    size_t embed_w = 0, embed_b = 0, pe = 0, final_w = 0, final_b = 0, f_embed_w = 0, h_embed_w = 0;
    size_t b_embed = 0;           // W_embed as the LDS images of the bf16 embedding kernel (vsk_pack_embed_bf16), if supported
    bool has_b_embed = false;
    bool has_pe = false;
    int norm_width = 0;           // vs_weights_set_norm_width: true d_model of a model embedded in this (wider) shape; 0 = desc.d_model
    int dn() const { return norm_width > 0 ? norm_width : desc.d_model; }
    bool embedded() const { return norm_width > 0 && norm_width != desc.d_model; }
    std::vector<LayerOff> layers;
    const float *p(size_t off) const { return blob + off; }
    // Kernel-layout IMAGES of the parameters, one version stamp per family: built by vsw_ensure() when a forward that
    // reads the family runs, never at pack / update time (a training step never pays for the f16x3 / bf16 images, a
    // large-batch scorer never for the fragment-major latency copies).
    mutable unsigned long long f_version = ~0ull;   // f_*: fragment-major fp32 copies (latency kernels, rows <= VS_SKINNY_ROWS)
    mutable unsigned long long h_version = ~0ull;   // h_*: their fp16x3 counterparts
    mutable unsigned long long b_version = ~0ull;   // b_*: LDS images of the fused bf16 layer kernels
    mutable unsigned long long r_version = ~0ull;   // r_*: plain bf16 copies

    // Stream ordering of everything above and below: the last stream that wrote the parameters or (re)built an
    // image, and an event recorded behind that work.  A call on ANOTHER stream first waits for the event
    // (hipStreamWaitEvent: no host synchronisation), so the host-side version stamps never run ahead of the device.
    mutable void *order_event = nullptr;  // hipEvent_t, created on first use
    mutable void *order_stream = nullptr; // hipStream_t of the recorded work (meaningful only while order_event != nullptr)
    mutable bool order_recorded = false;

    // ---- training side ----
    unsigned long long version = 0;       // bumped by every pack / update
    float *tblob = nullptr;               // second device allocation: transposed weights + a zero vector
    unsigned long long t_version = ~0ull; // version the transposes were built from
    unsigned long long tf_version = ~0ull;// ... and their fragment-major copies (latency kernels only)
    unsigned long long t16_version = ~0ull;// ... and the bf16 copy of W2^T (A-stationary dgrad of the bf16 training mode only)
    size_t t_embed_w = 0, tf_embed_w = 0, zeros = 0;
    std::vector<LayerOffT> tlayers;
    const float *tp(size_t off) const { return tblob + off; }
};

int vs_fail_msg(int code, const char *msg);     // vs_scorer.cpp: sets the thread-local error text

// (re)builds the image families in `families` that are older than the handle's parameters, stream-ordered on `st`.
// A handle's calls must be issued on ONE stream at a time (or be ordered by the caller): include/vs_scorer.h.
enum { VSW_FRAGMENTS = 1, VSW_F16X3 = 2, VSW_BF16 = 4, VSW_ROWS16 = 8 };
int vsw_ensure(const vs_weights *w, unsigned families, void *stream);
// stream ordering helpers (vs_scorer.cpp): vsw_mark() after parameter writes / image rebuilds were enqueued on `stream`;
// vsw_order() before a call on `stream` reads them (a no-op when it is the same stream)
void vsw_mark(const vs_weights *w, void *stream);
void vsw_order(const vs_weights *w, void *stream);
