// vs_train.cpp — the C ABI of include/vs_train.h: the train-mode forward of the scorer with its activation record,
// and the backward pass.  Launch sequences only; the kernels are in vs_kernels.hip (NT GEMMs, reused for every
// forward Linear and — against transposed weights — for every dgrad), vs_train_kernels.hip and
// vs_train_attention.hip.
#include "vs_train.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <mutex>

#include "vs_train_device_sites.h"
#include "vs_train_kernels.h"
#include "vs_weights_impl.h"

namespace {

int failf(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return vs_fail_msg(code, buf);
}

#define VST_LAUNCH(call)                                                                         \
    do {                                                                                         \
        int e_ = (call);                                                                         \
        if (e_ > 0) return failf(VS_ERR_HIP, "%s: %s", #call, hipGetErrorString((hipError_t)e_)); \
        if (e_ < 0) return failf(VS_ERR_INVALID, "%s: unsupported shape", #call);                \
    } while (0)
#define VST_HIP(call)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) return failf(VS_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_));   \
    } while (0)

size_t align_floats(size_t n) { return (n + 63) / 64 * 64; }      // 256-byte granules

// The FORM of an activation record: which tensors of it are bf16 planes / which arithmetic wrote it.  The forward derives it
// from the request (vs_dropout_cfg.flags), the batch size and the library's switches; the backward must read the record in
// the form it was WRITTEN in.  The forward therefore publishes the form (vs_train_last_format(), per calling thread) and the
// backward takes it back through vs_dropout_cfg.reserved - a switch flipped between the two calls (another model, a retained
// graph, a test) can no longer make the backward read bf16 planes as fp32.  reserved == 0: derived again (older callers).
struct RecordForm {
    int lp; bool lpa, qkv16, h16, rows16, f16;      // f16: the 16-bit type of everything above is IEEE f16 (VS_TRAIN_FLAG_FP16)
    uint32_t bits() const {
        return 0x80000000u | (lp ? 1u : 0u) | (lpa ? 2u : 0u) | (qkv16 ? 4u : 0u) | (h16 ? 8u : 0u) | (rows16 ? 16u : 0u) | (f16 ? 32u : 0u);
    }
};
thread_local uint32_t g_last_format = 0;
RecordForm derive_form(const vs_weights *w, const vs_dropout_cfg *drop, int B, int T) {
    RecordForm f{};
    // low-precision training (VS_TRAIN_FLAG_BF16_LINEAR): every Linear / dgrad / wgrad GEMM on the bf16 matrix pipe, from the
    // batch size up where that pays (VS_TRAIN_LP_MIN_ROWS, default 1024 frames: tools/sweep_lp_min_rows.py,
    // profiles/r04_lp_min_rows_sweep.txt - break-even at ~1280 frames, never slower; rounds 2-3 used the scoring path's 8192)
    f.lp = (drop && (drop->flags & VS_TRAIN_FLAG_BF16_LINEAR) && (long long)B * T > vsk_options().train_lp_min_rows) ? 1 : 0;
    // ... and (VS_TRAIN_FLAG_BF16_ATTENTION) the attention products of the forward and the backward (head dim 32 / 64 / 128)
    f.lpa = drop && (drop->flags & VS_TRAIN_FLAG_BF16_ATTENTION) && (long long)B * T > vsk_options().train_lp_min_rows &&
            vst_attention_bf16_supported(w->desc.d_model / w->desc.num_heads);
    // bf16 STORAGE of the tensors that are only ever bf16 matrix operands (VS_LP_STORE32 = 1 keeps them fp32: an A/B switch -
    // the kernels round the fp32-stored values to the same bf16, so every result is bit-identical either way)
    f.qkv16 = f.lp && f.lpa && !vsk_options().lp_store32;     // q / k / v of the record are bf16 planes
    f.h16 = f.lp && !vsk_options().lp_store32;                // ... and so is the MLP hidden tensor (and, in the backward, its gradient)
    // ... which the A-stationary GEMM writes where it applies (K = d_model = 256, batches that fill the chip; VS_LP_MLP_UNFUSED = 1: the
    // tiled kernels, 2: the A-stationary kernel at every batch size - A/B and test switches)
    f.f16 = (f.lp || f.lpa) && (drop->flags & VS_TRAIN_FLAG_FP16) != 0;
    // (the A-stationary kernels and their pre-rounded weight copies exist in bf16 only: the fp16 mode runs the tiled GEMMs)
    f.rows16 = f.h16 && !f.f16 && vsk_options().lp_mlp_unfused != 1 &&
               vst_gemm_rows16_supported(B * T, 4 * w->desc.d_model, w->desc.d_model, vsk_options().lp_mlp_unfused == 2);
    return f;
}

// ---- activation record (floats) ----
struct LayerSaved { size_t qkv, att, lse, z1, st1, y1, ffn, z2, st2, y2, dbits; };
struct SavedLayout {
    size_t h0 = 0, total = 0;
    std::vector<LayerSaved> layers;
};
SavedLayout saved_layout(const vs_model_desc &D, int B, int T) {
    SavedLayout S;
    const size_t M = (size_t)B * T, d = D.d_model;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += align_floats(n); return o; };
    S.h0 = take(M * d);
    S.layers.resize(D.num_layers);
    for (auto &L : S.layers) {
        L.qkv = take(3 * M * d); L.att = take(M * d); L.lse = take((size_t)B * D.num_heads * T);
        L.z1 = take(M * d); L.st1 = take(2 * M); L.y1 = take(M * d);
        L.ffn = take(4 * M * d);
        L.z2 = take(M * d); L.st2 = take(2 * M); L.y2 = take(M * d);
        L.dbits = take(vst_attention_dropout_bits_words(B, D.num_heads, T));     // the layer's attention keep masks, bit-packed
    }
    S.total = off;
    return S;
}

// ---- scratch (floats): the forward uses `a` only ----
struct WorkLayout { size_t a, g0, g1, dz, dbr, gf, dy1, datt, dqkv, delta, part, wg, total; };
WorkLayout work_layout(const vs_model_desc &D, int B, int T) {
    WorkLayout W{};
    const size_t M = (size_t)B * T, d = D.d_model, din = D.in_features;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += align_floats(n); return o; };
    W.a = take(M * d);
    W.g0 = take(M * d); W.g1 = take(M * d);
    W.dz = take(M * d); W.dbr = take(M * d);
    W.gf = take(4 * M * d);
    W.dy1 = take(M * d); W.datt = take(M * d);
    W.dqkv = take(3 * M * d);
    W.delta = take((size_t)B * D.num_heads * T);
    const size_t nblk = (size_t)vst_ln_bwd_blocks((int)M);
    W.part = take(nblk * (2 * d + 2));
    size_t wg = 0;                      // the largest split-partial area of the five weight shapes [N, K]
    const size_t shapes[5][2] = {{d, 4 * d}, {4 * d, d}, {d, d}, {3 * d, d}, {d, din}};
    for (auto &s : shapes) {
        const size_t f = vst_wgrad_workspace_floats((int)M, (int)s[0], (int)s[1]);
        wg = f > wg ? f : wg;
    }
    W.wg = take(wg);
    W.total = off;
    return W;
}

int check_common(const vs_weights *w, const float *x, int B, int T, const vs_dropout_cfg *drop) {
    if (!w || !x) return failf(VS_ERR_INVALID, "weights/x is NULL");
    if (B <= 0 || T <= 0) return failf(VS_ERR_INVALID, "B=%d T=%d", B, T);
    if ((long long)B * T > (1ll << 28)) return failf(VS_ERR_INVALID, "B*T too large");
    if (w->has_pe && T > w->desc.max_len)
        return failf(VS_ERR_INVALID, "T=%d exceeds the positional table (max_len=%d)", T, w->desc.max_len);
    if (drop && (drop->p < 0.f || drop->p >= 1.f || drop->p_embed < 0.f || drop->p_embed >= 1.f))
        return failf(VS_ERR_INVALID, "dropout probabilities must be in [0, 1): p=%g p_embed=%g", drop->p, drop->p_embed);
    if ((uintptr_t)x & 15) return failf(VS_ERR_INVALID, "x must be 16-byte aligned");
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != w->device)
        return failf(VS_ERR_INVALID, "called on device %d, the weights handle lives on device %d", dev, w->device);
    return VS_OK;
}

std::mutex g_tmu;

// W^T copies for the dgrad GEMMs + a zero "bias"; rebuilt when the handle's parameters changed.  `frags`: also the
// fragment-major copies of W^T (only the latency kernels, rows <= VS_SKINNY_ROWS, read them).  The buffer is allocated
// by the first call: vs_train_prepare() lets a caller do that outside its backward pass.
int ensure_transposed(vs_weights *w, hipStream_t st, bool frags) {
    std::lock_guard<std::mutex> lk(g_tmu);
    const size_t d = w->desc.d_model, din = w->desc.in_features;
    if (!w->tblob) {
        size_t off = 0;
        auto take = [&](size_t n) { size_t o = off; off += align_floats(n); return o; };
        w->t_embed_w = take(d * din);
        w->tf_embed_w = take(d * din);
        w->tlayers.resize(w->desc.num_layers);
        for (auto &L : w->tlayers) {
            L.t_wqkv = take(3 * d * d); L.t_wo = take(d * d); L.t_w1 = take(4 * d * d); L.t_w2 = take(4 * d * d);
            L.t16_w2 = take(4 * d * d / 2);
            L.tf_wqkv = take(3 * d * d); L.tf_wo = take(d * d); L.tf_w1 = take(4 * d * d); L.tf_w2 = take(4 * d * d);
        }
        const size_t nz = 4 * d > din ? 4 * d : din;
        w->zeros = take(nz);
        VST_HIP(hipMalloc((void **)&w->tblob, off * sizeof(float)));
        VST_HIP(hipMemsetAsync(w->tblob + w->zeros, 0, nz * sizeof(float), st));
        w->t_version = ~0ull;
        w->tf_version = ~0ull;
        w->t16_version = ~0ull;
    }
    // W [N,K] -> W^T [K,N] row-major, and its fragment-major copy for the latency kernels (as vsw_ensure does for W)
    vsw_order(w, (void *)st);
    const bool do_t = w->t_version != w->version, do_f = frags && w->tf_version != w->version;
    if (!do_t && !do_f) return VS_OK;
    struct Mark { const vs_weights *w; void *s; ~Mark() { vsw_mark(w, s); } } mark{w, (void *)st};
    // one launch per family (was one per matrix: 17 transposes + 17 fragment packs + 4 conversions per optimizer step)
    VskMatJobs tj{}, fj{};
    auto add = [&](const float *W, size_t t_off, size_t tf_off, int N, int K) {
        tj.in[tj.n] = W; tj.out[tj.n] = w->tblob + t_off; tj.rows[tj.n] = N; tj.cols[tj.n] = K; ++tj.n;
        fj.in[fj.n] = w->tblob + t_off; fj.out[fj.n] = w->tblob + tf_off; fj.rows[fj.n] = K; fj.cols[fj.n] = N; ++fj.n;      // W^T is [K, N]
    };
    auto flush = [&]() -> int {
        if (tj.n == 0) return 0;
        if (do_t) if (int rc = vst_transpose_batch(tj, st)) return rc;
        if (do_f) if (int rc = vsk_pack_fragments_batch(fj, st)) return rc;
        tj.n = fj.n = 0;
        return 0;
    };
    add(w->p(w->embed_w), w->t_embed_w, w->tf_embed_w, (int)d, (int)din);                                    // [d,din] -> [din,d]
    for (int l = 0; l < w->desc.num_layers; ++l) {
        const LayerOff &P = w->layers[l];
        const LayerOffT &Q = w->tlayers[l];
        if (tj.n + 4 > VskMatJobs::MAX) VST_LAUNCH(flush());
        add(w->p(P.wqkv), Q.t_wqkv, Q.tf_wqkv, (int)(3 * d), (int)d);                                        // [3d,d] -> [d,3d]
        add(w->p(P.wo), Q.t_wo, Q.tf_wo, (int)d, (int)d);
        add(w->p(P.w1), Q.t_w1, Q.tf_w1, (int)(4 * d), (int)d);                                              // [4d,d] -> [d,4d]
        add(w->p(P.w2), Q.t_w2, Q.tf_w2, (int)d, (int)(4 * d));                                              // [d,4d] -> [4d,d]
    }
    VST_LAUNCH(flush());
    // (the bf16 copy of W2^T feeds the A-stationary dgrad of the bf16 training mode only: built when that form runs - t16_version)
    if (do_t) w->t_version = w->version;
    if (do_f) w->tf_version = w->version;
    return VS_OK;
}

// bf16 copy of every W2^T (vst_gemm_rows16's weight operand in the backward of the bf16 training mode): after ensure_transposed,
// only when that form runs
int ensure_t16(vs_weights *w, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_tmu);
    if (w->t16_version == w->version) return VS_OK;
    vsw_order(w, (void *)st);
    struct Mark { const vs_weights *w; void *s; ~Mark() { vsw_mark(w, s); } } mark{w, (void *)st};
    const size_t d = w->desc.d_model;
    for (int l = 0; l < w->desc.num_layers; ++l) {
        const LayerOffT &Q = w->tlayers[l];
        VST_LAUNCH(vsk_to_bf16(w->tblob + Q.t_w2, w->tblob + Q.t16_w2, 4 * d * d, st));
    }
    w->t16_version = w->version;
    return VS_OK;
}

}  // namespace

extern "C" {

int vs_train_prepare(vs_weights *w, void *stream) {
    if (!w) return failf(VS_ERR_INVALID, "weights is NULL");
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != w->device)
        return failf(VS_ERR_INVALID, "called on device %d, the weights handle lives on device %d", dev, w->device);
    return ensure_transposed(w, (hipStream_t)stream, false);
}

size_t vs_train_saved_bytes(const vs_weights *w, int32_t B, int32_t T) {
    if (!w || B <= 0 || T <= 0) return 0;
    return saved_layout(w->desc, B, T).total * sizeof(float);
}

size_t vs_train_workspace_bytes(const vs_weights *w, int32_t B, int32_t T) {
    if (!w || B <= 0 || T <= 0) return 0;
    return work_layout(w->desc, B, T).total * sizeof(float);
}

int vs_train_saved_field(const vs_weights *w, int32_t B, int32_t T, int32_t layer, int32_t field, size_t *offset_bytes,
                         size_t *count) {
    if (!w || !offset_bytes || !count || B <= 0 || T <= 0 || layer < 0 || layer >= w->desc.num_layers)
        return failf(VS_ERR_INVALID, "saved_field: bad arguments");
    const SavedLayout S = saved_layout(w->desc, B, T);
    const LayerSaved &A = S.layers[layer];
    const size_t Md = (size_t)B * T * w->desc.d_model;
    switch (field) {
        case 0: *offset_bytes = A.ffn * sizeof(float); *count = 4 * Md; break;
        case 1: *offset_bytes = A.att * sizeof(float); *count = Md; break;
        case 2: *offset_bytes = A.y1 * sizeof(float); *count = Md; break;
        case 3: *offset_bytes = A.y2 * sizeof(float); *count = Md; break;
        default: return failf(VS_ERR_INVALID, "saved_field: unknown field %d", field);
    }
    return VS_OK;
}

uint32_t vs_train_last_format(void) { return g_last_format; }

uint32_t vs_train_dropout_site(int32_t layer, int32_t which) { return layer < 0 ? VS_SITE_EMBED : VS_SITE_LAYER(layer, which); }

int vs_train_forward(const vs_weights *w, const float *x, const uint8_t *key_pad_mask, int32_t B, int32_t T,
                     const vs_dropout_cfg *drop, float *scores, float *hidden, void *saved, size_t saved_bytes,
                     void *workspace, size_t workspace_bytes, void *stream) {
    if (int rc = check_common(w, x, B, T, drop)) return rc;
    if (!scores || !saved || !workspace) return failf(VS_ERR_INVALID, "scores/saved/workspace is NULL");
    const vs_model_desc &D = w->desc;
    const SavedLayout S = saved_layout(D, B, T);
    const WorkLayout W = work_layout(D, B, T);
    if (saved_bytes < S.total * sizeof(float))
        return failf(VS_ERR_WORKSPACE, "saved %zu bytes < %zu needed", saved_bytes, S.total * sizeof(float));
    if (workspace_bytes < W.total * sizeof(float))
        return failf(VS_ERR_WORKSPACE, "workspace %zu bytes < %zu needed", workspace_bytes, W.total * sizeof(float));
    if (((uintptr_t)saved & 255) || ((uintptr_t)workspace & 255) || (hidden && ((uintptr_t)hidden & 15)))
        return failf(VS_ERR_INVALID, "saved/workspace must be 256-byte, hidden 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int d = D.d_model, H = D.num_heads, L = D.num_layers, M = B * T;
    const float p = drop ? drop->p : 0.f;
    // the embedding dropout lives INSIDE PositionalEncoding (simnet.py:224,237): a use_pos=False model has none
    const float p_embed = (drop && w->has_pe) ? drop->p_embed : 0.f;
    const RecordForm form = derive_form(w, drop, B, T);
    g_last_format = form.bits();
    const int F = form.f16 ? VSK_F16 : 0;            // fp16 mode: rides on every 16-bit precision word below
    const int lp = form.lp ? (1 | F) : 0;
    const bool lpa = form.lpa, qkv16 = form.qkv16, h16 = form.h16, rows16 = form.rows16;
    const size_t kvs = qkv16 ? (size_t)B * T * w->desc.d_model / 2 : (size_t)B * T * w->desc.d_model;     // floats between the planes
    const unsigned long long seed = drop ? drop->seed : 0ull;
    const int dn = w->dn();                                   // LayerNorm width / attention-scale d_model (== d unless embedded)
    const float scale = 1.0f / sqrtf((float)dn);             // simnet.py:126: d_model ** -0.5
    float *sv = (float *)saved, *ws = (float *)workspace;
    float *a = ws + W.a;

    vsw_order(w, stream);
    if (M <= vsk_skinny_max_rows()) if (int rc = vsw_ensure(w, VSW_FRAGMENTS, stream)) return rc;
    if (rows16) if (int rc = vsw_ensure(w, VSW_ROWS16, stream)) return rc;          // bf16 copy of W1 for the A-stationary fc1
    // Embedding + positional table + dropout(sparsity)   simnet.py:211, 237-238
    float *h0 = sv + S.h0;
    VST_LAUNCH(vsk_linear(x, w->p(w->embed_w), w->p(w->f_embed_w), w->p(w->embed_b), h0, M, d, D.in_features, 0,
                          w->has_pe ? w->p(w->pe) : nullptr, T, lp, st));
    if (p_embed > 0.f) VST_LAUNCH(vst_dropout_rows(h0, M, d, seed, VS_SITE_EMBED, p_embed, st));
    const float *h_in = h0;
    for (int l = 0; l < L; ++l) {
        const LayerOff &P = w->layers[l];
        const LayerSaved &A = S.layers[l];
        const bool last = l == L - 1;
        float *qkv = sv + A.qkv;
        // bf16 GEMMs + bf16 attention: q (times scale * log2 e), k, v are WRITTEN as bf16 by the QKV GEMM's epilogue (the
        // scoring path's form) - the attention kernels' only readers round them to bf16 anyway: same bits, half the
        // saved bytes, nothing to convert per streamed tile
        if (rows16 && qkv16)       // A-stationary form (bit-identical to the tiled one)
            VST_LAUNCH(vst_gemm_rows16(h_in, w->p(P.r_wqkv), w->p(P.bqkv), qkv, nullptr, M, 3 * d, d, 3, vsk_attention_qscale(scale), 0ull, 0u,
                                       0.f, st, T, H, d / H));
        else
        VST_LAUNCH(vsk_qkv(h_in, w->p(P.wqkv), w->p(P.f_wqkv), w->p(P.bqkv), qkv, B, T, d, H, qkv16 ? (1 | VSK_STORE16 | F) : lp, st,
                           qkv16 ? vsk_attention_qscale(scale) : 1.0f));         // :148-153
        unsigned *dbits = p > 0.f ? (unsigned *)(sv + A.dbits) : nullptr;
        if (dbits) VST_LAUNCH(vst_attention_dropout_bits(dbits, B, H, T, seed, VS_SITE_LAYER(l, VS_SITE_ATTN), p, st));
        if (lpa)
            VST_LAUNCH(vst_attention_fwd_bf16(qkv, qkv + kvs, qkv + 2 * kvs, key_pad_mask, sv + A.att,
                                              sv + A.lse, B, H, T, d / H, scale, p, dbits, st, (qkv16 ? 1 : 0) | F));
        else
        VST_LAUNCH(vst_attention_fwd(qkv, qkv + (size_t)M * d, qkv + 2 * (size_t)M * d, key_pad_mask, sv + A.att,
                                     sv + A.lse, B, H, T, d / H, scale, seed, VS_SITE_LAYER(l, VS_SITE_ATTN), p, st, dbits));   // :155-161
        VST_LAUNCH(vsk_linear(sv + A.att, w->p(P.wo), w->p(P.f_wo), w->p(P.bo), a, M, d, d, 0, nullptr, 1, lp, st));      // :163
        VST_LAUNCH(vst_rows_fwd(a, h_in, w->p(P.ln1g), w->p(P.ln1b), sv + A.z1, sv + A.y1, nullptr, sv + A.st1, M, d,
                                seed, VS_SITE_LAYER(l, VS_SITE_DROP1), p, nullptr, nullptr, 0, nullptr, st, dn));              // :107
        // bf16 GEMMs: the MLP hidden tensor (post ReLU, post dropout) is only ever a matrix operand or a sign - it is written
        // and saved as bf16 (h16), and fc2, its weight gradient and the backward's gate read it as such
        if (rows16)         // K = d_model = 256: the A-stationary form (vs_train_gemm_rows.hip), bit-identical to the tiled one
            VST_LAUNCH(vst_gemm_rows16(sv + A.y1, w->p(P.r_w1), w->p(P.b1), sv + A.ffn, nullptr, M, 4 * d, d, p > 0.f ? 0 : 2, 0.f, seed,
                                       VS_SITE_LAYER(l, VS_SITE_MLP), p, st));
        else if (p > 0.f)   // fc1 + ReLU + mlp.dropout in one GEMM epilogue (:181)
            VST_LAUNCH(vsk_linear_relu_dropout(sv + A.y1, w->p(P.w1), w->p(P.f_w1), w->p(P.b1), sv + A.ffn, M, 4 * d, d, seed,
                                               VS_SITE_LAYER(l, VS_SITE_MLP), p, st, h16 ? (1 | VSK_STORE16 | F) : lp));
        else
            VST_LAUNCH(vsk_linear(sv + A.y1, w->p(P.w1), w->p(P.f_w1), w->p(P.b1), sv + A.ffn, M, 4 * d, d, 1, nullptr, 1,
                                  h16 ? (1 | VSK_STORE16 | F) : lp, st));
        VST_LAUNCH(vsk_linear(sv + A.ffn, w->p(P.w2), w->p(P.f_w2), w->p(P.b2), a, M, d, 4 * d, 0, nullptr, 1, h16 ? (1 | VSK_A16 | F) : lp, st));  // :182
        VST_LAUNCH(vst_rows_fwd(a, sv + A.y1, w->p(P.ln2g), w->p(P.ln2b), sv + A.z2, sv + A.y2, last ? hidden : nullptr,
                                sv + A.st2, M, d, seed, VS_SITE_LAYER(l, VS_SITE_DROP2), p,
                                last ? w->p(w->final_w) : nullptr, last ? w->p(w->final_b) : nullptr, D.num_classes,
                                last ? scores : nullptr, st, dn));                                                              // :110, :42
        h_in = sv + A.y2;
    }
    return VS_OK;
}

int vs_train_backward(vs_weights *w, const float *x, const uint8_t *key_pad_mask, int32_t B, int32_t T,
                      const vs_dropout_cfg *drop, const float *d_scores, const float *d_hidden, const void *saved,
                      size_t saved_bytes, const vs_model_grads *grads, float *dx, void *workspace,
                      size_t workspace_bytes, void *stream) {
    if (int rc = check_common(w, x, B, T, drop)) return rc;
    if (!saved || !workspace || !grads || !grads->layers || !grads->embed_w || !grads->embed_b || !grads->final_w ||
        !grads->final_b)
        return failf(VS_ERR_INVALID, "saved/workspace/grads is NULL");
    const vs_model_desc &D = w->desc;
    const SavedLayout S = saved_layout(D, B, T);
    const WorkLayout W = work_layout(D, B, T);
    if (saved_bytes < S.total * sizeof(float))
        return failf(VS_ERR_WORKSPACE, "saved %zu bytes < %zu needed", saved_bytes, S.total * sizeof(float));
    if (workspace_bytes < W.total * sizeof(float))
        return failf(VS_ERR_WORKSPACE, "workspace %zu bytes < %zu needed", workspace_bytes, W.total * sizeof(float));
    if (((uintptr_t)saved & 255) || ((uintptr_t)workspace & 255)) return failf(VS_ERR_INVALID, "saved/workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = ensure_transposed(w, st, B * T <= vsk_skinny_max_rows())) return rc;
    const int d = D.d_model, H = D.num_heads, L = D.num_layers, M = B * T, nc = D.num_classes;
    const float p = drop ? drop->p : 0.f;
    // the embedding dropout lives INSIDE PositionalEncoding (simnet.py:224,237): a use_pos=False model has none
    const float p_embed = (drop && w->has_pe) ? drop->p_embed : 0.f;
    // the form the record was written in: handed back by the caller (vs_dropout_cfg.reserved = vs_train_last_format() of the
    // forward), else derived again from the same inputs
    RecordForm form = derive_form(w, drop, B, T);
    if (drop && (drop->reserved & 0x80000000u)) {
        const uint32_t fb = drop->reserved;
        form.lp = (fb & 1u) ? 1 : 0; form.lpa = (fb & 2u) != 0; form.qkv16 = (fb & 4u) != 0; form.h16 = (fb & 8u) != 0; form.rows16 = (fb & 16u) != 0;
        form.f16 = (fb & 32u) != 0;
        if ((form.qkv16 && !(form.lp && form.lpa)) || (form.h16 && !form.lp) || (form.rows16 && !form.h16) ||
            (form.f16 && (form.rows16 || !(form.lp || form.lpa))) || (fb & 0x7fffffc0u) ||
            (form.lpa && !vst_attention_bf16_supported(w->desc.d_model / w->desc.num_heads)) ||
            (form.rows16 && !vst_gemm_rows16_supported(B * T, 4 * w->desc.d_model, w->desc.d_model, true)))
            return failf(VS_ERR_INVALID, "vs_dropout_cfg.reserved = 0x%08x is not a record form this model / batch can have", fb);
    }
    const int F = form.f16 ? VSK_F16 : 0;            // fp16 mode: rides on every 16-bit precision word below
    const int lp = form.lp ? (1 | F) : 0;
    const bool lpa = form.lpa, qkv16 = form.qkv16, h16 = form.h16, rows16 = form.rows16;
    if (rows16) if (int rc = ensure_t16(w, st)) return rc;
    const size_t kvs = qkv16 ? (size_t)B * T * w->desc.d_model / 2 : (size_t)B * T * w->desc.d_model;     // floats between the planes
    const unsigned long long seed = drop ? drop->seed : 0ull;
    const int dn = w->dn();
    const float scale = 1.0f / sqrtf((float)dn);
    const float *sv = (const float *)saved;
    float *ws = (float *)workspace;
    float *g[2] = {ws + W.g0, ws + W.g1};
    float *dz = ws + W.dz, *dbr = ws + W.dbr, *gf = ws + W.gf, *dy1 = ws + W.dy1, *datt = ws + W.datt;
    float *dqkv = ws + W.dqkv, *delta = ws + W.delta, *part = ws + W.part, *wg = ws + W.wg;
    const float *zeros = w->tp(w->zeros);
    const int nblk = vst_ln_bwd_blocks(M);

    // final_layer (simnet.py:42): d_W = d_scores^T hidden, d_b = column sums of d_scores
    {
        const float *y_last = sv + S.layers[L - 1].y2;
        if (d_scores) {
            for (int c = 0; c < nc; ++c) {
                VST_LAUNCH(vst_weighted_colsum(d_scores + c, nc, y_last, part, M, d, st));
                VST_LAUNCH(vst_reduce_rows(part, nblk, 1, d, grads->final_w + (size_t)c * d, nullptr, nullptr, 1, st));
                VST_LAUNCH(vst_reduce_rows(part + (size_t)nblk * d, nblk, 1, 1, grads->final_b + c, nullptr, nullptr, 1, st));
            }
        } else {
            VST_HIP(hipMemsetAsync(grads->final_w, 0, (size_t)nc * d * sizeof(float), st));
            VST_HIP(hipMemsetAsync(grads->final_b, 0, (size_t)nc * sizeof(float), st));
        }
    }

    int cur = 0;
    for (int l = L - 1; l >= 0; --l) {
        const LayerOff &P = w->layers[l];
        const LayerOffT &Q = w->tlayers[l];
        const LayerSaved &A = S.layers[l];
        const vs_layer_grads &G = grads->layers[l];
        const bool last = l == L - 1;
        const float *h_in = l == 0 ? sv + S.h0 : sv + S.layers[l - 1].y2;
        // norm2 (+ the score head's pull on the last layer); d(fc2 output) = dropout2 mask on dz2
        VST_LAUNCH(vst_ln_bwd(last ? d_hidden : g[cur], last ? d_scores : nullptr, w->p(w->final_w), nc, sv + A.z2, sv + A.st2,
                              w->p(P.ln2g), dz, p > 0.f ? dbr : nullptr, part, M, d, seed, VS_SITE_LAYER(l, VS_SITE_DROP2), p, st, dn));
        VST_LAUNCH(vst_reduce_rows(part, nblk, 2, d, G.ln2_g, G.ln2_b, nullptr, 1, st));
        const float *dm = p > 0.f ? dbr : dz;
        // mlp.fc2: weight/bias gradient, then the gradient of its input
        VST_LAUNCH(vst_wgrad(dm, d, sv + A.ffn, 4 * d, M, d, 4 * d, G.w2, nullptr, nullptr, G.b2, nullptr, nullptr, d, wg, st,
                             h16 ? (1 | VST_WGRAD_X16 | F) : lp));
        // ... through mlp.dropout + ReLU in the GEMM's epilogue: the saved activation is > 0 exactly where both let
        // the value through
        // (h16: the gate tensor is the bf16-stored activation, and the gated gradient gf - again only ever a matrix operand -
        // is written as bf16 too)
        if (rows16)
            VST_LAUNCH(vst_gemm_rows16(dm, w->tp(Q.t16_w2), zeros, gf, sv + A.ffn, M, 4 * d, d, 1, p > 0.f ? 1.0f / (1.0f - p) : 1.0f, 0ull, 0u,
                                       0.f, st));
        else
        VST_LAUNCH(vsk_linear_gate(dm, w->tp(Q.t_w2), w->tp(Q.tf_w2), zeros, sv + A.ffn, p > 0.f ? 1.0f / (1.0f - p) : 1.0f, gf, M, 4 * d, d, st,
                                   h16 ? (1 | VSK_STORE16 | F) : lp));
        VST_LAUNCH(vst_wgrad(gf, 4 * d, sv + A.y1, d, M, 4 * d, d, G.w1, nullptr, nullptr, G.b1, nullptr, nullptr, 4 * d, wg, st,
                             h16 ? (1 | VST_WGRAD_Y16 | F) : lp));
        // d y1 = dz2 (residual) + d(fc1 input): the residual rides in the GEMM epilogue
        VST_LAUNCH(vsk_linear(gf, w->tp(Q.t_w1), w->tp(Q.tf_w1), zeros, dy1, M, d, 4 * d, 0, dz, M, h16 ? (1 | VSK_A16 | F) : lp, st));
        // norm1; d(feature_projection output) = dropout1 mask on dz1
        VST_LAUNCH(vst_ln_bwd(dy1, nullptr, nullptr, 0, sv + A.z1, sv + A.st1, w->p(P.ln1g), dz, p > 0.f ? dbr : nullptr, part,
                              M, d, seed, VS_SITE_LAYER(l, VS_SITE_DROP1), p, st, dn));
        VST_LAUNCH(vst_reduce_rows(part, nblk, 2, d, G.ln1_g, G.ln1_b, nullptr, 1, st));
        const float *da = p > 0.f ? dbr : dz;
        VST_LAUNCH(vst_wgrad(da, d, sv + A.att, d, M, d, d, G.wo, nullptr, nullptr, G.bo, nullptr, nullptr, d, wg, st, lp));
        // (qkv16: d(attention output) is written as bf16 - the attention backward multiplies bf16 operands, and delta is then the
        // row dot of the SAME rounded dO with the saved fp32 output)
        VST_LAUNCH(vsk_linear(da, w->tp(Q.t_wo), w->tp(Q.tf_wo), zeros, datt, M, d, d, 0, nullptr, 1, qkv16 ? (1 | VSK_STORE16 | F) : lp, st));
        // attention
        const float *qkv = sv + A.qkv;
        VST_LAUNCH(vst_head_rowdot(datt, sv + A.att, delta, M, T, H, d / H, st, qkv16 ? (1 | F) : lpa ? (2 | F) : 0));
        if (lpa)
            VST_LAUNCH(vst_attention_bwd_bf16(qkv, qkv + kvs, qkv + 2 * kvs, key_pad_mask, datt, sv + A.lse,
                                              delta, dqkv, B, H, T, d / H, scale, p,
                                              p > 0.f ? (const unsigned *)(sv + A.dbits) : nullptr, st, (qkv16 ? 1 : 0) | F, qkv16));
        else
        VST_LAUNCH(vst_attention_bwd(qkv, qkv + (size_t)M * d, qkv + 2 * (size_t)M * d, key_pad_mask, datt, sv + A.lse, delta,
                                     dqkv, B, H, T, d / H, scale, seed, VS_SITE_LAYER(l, VS_SITE_ATTN), p, st,
                                     p > 0.f ? (const unsigned *)(sv + A.dbits) : nullptr));
        // q / k / v projections: one [3d, d] weight gradient dealt to the three parameters
        // (qkv16: dq | dk | dv arrive as bf16 - only ever matrix operands of the two GEMMs below)
        VST_LAUNCH(vst_wgrad(dqkv, 3 * d, h_in, d, M, 3 * d, d, G.wq, G.wk, G.wv, G.bq, G.bk, G.bv, d, wg, st,
                             qkv16 ? (1 | VST_WGRAD_Y16 | F) : lp));
        // gradient of the layer input = dz1 (residual) + dqkv Wqkv
        VST_LAUNCH(vsk_linear(dqkv, w->tp(Q.t_wqkv), w->tp(Q.tf_wqkv), zeros, g[cur ^ 1], M, d, 3 * d, 0, dz, M, qkv16 ? (1 | VSK_A16 | F) : lp, st));
        cur ^= 1;
    }
    // Embedding (simnet.py:211, 237-238): dropout(sparsity) mask, then the Linear
    float *gh0 = g[cur];
    if (p_embed > 0.f) VST_LAUNCH(vst_dropout_rows(gh0, M, d, seed, VS_SITE_EMBED, p_embed, st));
    VST_LAUNCH(vst_wgrad(gh0, d, x, D.in_features, M, d, D.in_features, grads->embed_w, nullptr, nullptr, grads->embed_b,
                         nullptr, nullptr, d, wg, st, lp));
    if (dx) VST_LAUNCH(vsk_linear(gh0, w->tp(w->t_embed_w), w->tp(w->tf_embed_w), zeros, dx, M, D.in_features, d, 0, nullptr, 1, lp, st));
    return VS_OK;
}

int vs_mse_mask_loss_forward(const float *output, const float *target, const uint8_t *mask, int32_t n, int32_t mean,
                             float *scratch, float *loss, void *stream) {
    if (!output || !target || !scratch || !loss || n <= 0) return failf(VS_ERR_INVALID, "mse_mask_loss: bad arguments");
    VST_LAUNCH(vst_mse_mask_fwd(output, target, mask, n, mean, scratch, loss, (hipStream_t)stream));
    return VS_OK;
}

int vs_mse_mask_loss_backward(const float *output, const float *target, const uint8_t *mask, const float *d_loss,
                              int32_t n, int32_t mean, float *d_output, void *stream) {
    if (!output || !target || !d_loss || !d_output || n <= 0) return failf(VS_ERR_INVALID, "mse_mask_loss: bad arguments");
    VST_LAUNCH(vst_mse_mask_bwd(output, target, mask, d_loss, n, mean, d_output, (hipStream_t)stream));
    return VS_OK;
}

size_t vs_pretrain_head_state_bytes(int32_t B, int32_t T, int32_t F) {
    if (B <= 0 || T <= 0 || F <= 0) return 0;
    return align_floats(vsp_head_scratch_floats(B, T, F)) * sizeof(float);
}

namespace {
struct HeadWork { size_t dfeats, wt, zeros, wg, total; };
HeadWork head_work(int B, int T, int d, int F) {
    HeadWork W{};
    const size_t M = (size_t)B * T;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += align_floats(n); return o; };
    W.dfeats = take(M * F);
    W.wt = take((size_t)d * F);
    W.zeros = take((size_t)(d > F ? d : F));
    W.wg = take(vst_wgrad_workspace_floats((int)M, F, d));
    W.total = off;
    return W;
}
}  // namespace

size_t vs_pretrain_head_workspace_bytes(int32_t B, int32_t T, int32_t d, int32_t F) {
    if (B <= 0 || T <= 0 || d <= 0 || F <= 0) return 0;
    return head_work(B, T, d, F).total * sizeof(float);
}

int vs_pretrain_head_forward(const float *hidden, const float *logits, const uint8_t *key_pad_mask, const float *vid,
                             const float *vt_w, const float *vt_b, int32_t B, int32_t T, int32_t d, int32_t F,
                             float temp, int32_t entropy_penalty, float *feats, void *head_state, float *losses,
                             void *stream) {
    if (!hidden || !logits || !vid || !vt_w || !vt_b || !feats || !head_state || !losses)
        return failf(VS_ERR_INVALID, "pretrain head: NULL pointer");
    if (B <= 0 || T <= 0 || d <= 0 || d % 32 || (F != 256 && F != 512 && F != 768 && F != 1024) || !(temp > 0.f))
        return failf(VS_ERR_INVALID, "pretrain head: B=%d T=%d d=%d F=%d temp=%g unsupported (d %% 32 == 0, F in {256,512,768,1024})", B, T, d, F, temp);
    hipStream_t st = (hipStream_t)stream;
    VST_LAUNCH(vsk_linear(hidden, vt_w, nullptr, vt_b, feats, B * T, F, d, 0, nullptr, 1, 0, st));           // :79
    VST_LAUNCH(vsp_head_forward(feats, logits, key_pad_mask, vid, B, T, F, 1.0f / temp, entropy_penalty,
                                (float *)head_state, losses, st));
    return VS_OK;
}

int vs_pretrain_head_backward(const float *hidden, const float *logits, const uint8_t *key_pad_mask, const float *vid,
                              const float *vt_w, const float *feats, void *head_state, const float *d_losses,
                              int32_t B, int32_t T, int32_t d, int32_t F, float temp, int32_t entropy_penalty,
                              float *d_hidden, float *d_logits, float *d_vt_w, float *d_vt_b, void *workspace,
                              size_t workspace_bytes, void *stream) {
    if (!hidden || !logits || !vid || !vt_w || !feats || !head_state || !d_losses || !d_hidden || !d_logits || !d_vt_w ||
        !d_vt_b || !workspace)
        return failf(VS_ERR_INVALID, "pretrain head: NULL pointer");
    if (B <= 0 || T <= 0 || d <= 0 || d % 32 || (F != 256 && F != 512 && F != 768 && F != 1024) || !(temp > 0.f))
        return failf(VS_ERR_INVALID, "pretrain head: B=%d T=%d d=%d F=%d temp=%g unsupported", B, T, d, F, temp);
    const HeadWork W = head_work(B, T, d, F);
    if (workspace_bytes < W.total * sizeof(float) || ((uintptr_t)workspace & 255))
        return failf(VS_ERR_WORKSPACE, "pretrain head: workspace %zu bytes < %zu needed (256-byte aligned)", workspace_bytes, W.total * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    float *ws = (float *)workspace;
    const int M = B * T;
    VST_LAUNCH(vsp_head_backward(feats, logits, key_pad_mask, vid, B, T, F, 1.0f / temp, entropy_penalty,
                                 (const float *)head_state, d_losses, ws + W.dfeats, d_logits, st));
    // video_transform: weight / bias gradient, then the gradient of the hidden state (NT GEMM against W^T)
    VST_LAUNCH(vst_wgrad(ws + W.dfeats, F, hidden, d, M, F, d, d_vt_w, nullptr, nullptr, d_vt_b, nullptr, nullptr, F, ws + W.wg, st));
    VST_LAUNCH(vst_transpose(vt_w, ws + W.wt, F, d, st));                                     // [F,d] -> [d,F]
    VST_HIP(hipMemsetAsync(ws + W.zeros, 0, (size_t)(d > F ? d : F) * sizeof(float), st));
    VST_LAUNCH(vsk_linear(ws + W.dfeats, ws + W.wt, nullptr, ws + W.zeros, d_hidden, M, d, F, 0, nullptr, 1, 0, st));
    return VS_OK;
}

int vs_train_attention_forward(const float *q, const float *k, const float *v, const uint8_t *key_pad_mask, float *out,
                               float *lse2, int32_t B, int32_t H, int32_t T, int32_t dh, float scale, uint64_t seed,
                               uint32_t site, float p, void *stream) {
    if (!q || !k || !v || !out || !lse2) return failf(VS_ERR_INVALID, "NULL pointer");
    if (B <= 0 || H <= 0 || T <= 0) return failf(VS_ERR_INVALID, "B=%d H=%d T=%d", B, H, T);
    VST_LAUNCH(vst_attention_fwd(q, k, v, key_pad_mask, out, lse2, B, H, T, dh, scale, seed, site, p, (hipStream_t)stream));
    return VS_OK;
}

int vs_train_attention_backward(const float *q, const float *k, const float *v, const uint8_t *key_pad_mask,
                                const float *out, const float *d_out, const float *lse2, float *dqkv, float *scratch,
                                int32_t B, int32_t H, int32_t T, int32_t dh, float scale, uint64_t seed, uint32_t site,
                                float p, void *stream) {
    if (!q || !k || !v || !out || !d_out || !lse2 || !dqkv || !scratch) return failf(VS_ERR_INVALID, "NULL pointer");
    if (B <= 0 || H <= 0 || T <= 0) return failf(VS_ERR_INVALID, "B=%d H=%d T=%d", B, H, T);
    hipStream_t st = (hipStream_t)stream;
    VST_LAUNCH(vst_head_rowdot(d_out, out, scratch, B * T, T, H, dh, st));
    VST_LAUNCH(vst_attention_bwd(q, k, v, key_pad_mask, d_out, lse2, scratch, dqkv, B, H, T, dh, scale, seed, site, p, st));
    return VS_OK;
}

size_t vs_train_attention_dropout_bits_bytes(int32_t B, int32_t H, int32_t T) {
    if (B <= 0 || H <= 0 || T <= 0) return 0;
    return vst_attention_dropout_bits_words(B, H, T) * sizeof(unsigned);
}

int vs_train_attention_dropout_bits(void *dbits, int32_t B, int32_t H, int32_t T, uint64_t seed, uint32_t site, float p,
                                    void *stream) {
    if (!dbits || B <= 0 || H <= 0 || T <= 0) return failf(VS_ERR_INVALID, "dropout_bits: bad arguments");
    VST_LAUNCH(vst_attention_dropout_bits((unsigned *)dbits, B, H, T, seed, site, p, (hipStream_t)stream));
    return VS_OK;
}

int vs_train_attention_forward_bf16(const float *q, const float *k, const float *v, const uint8_t *key_pad_mask, float *out,
                                    float *lse2, int32_t B, int32_t H, int32_t T, int32_t dh, float scale, float p,
                                    const void *dbits, int32_t in16, void *stream) {
    if (!q || !k || !v || !out || !lse2) return failf(VS_ERR_INVALID, "NULL pointer");
    if (B <= 0 || H <= 0 || T <= 0) return failf(VS_ERR_INVALID, "B=%d H=%d T=%d", B, H, T);
    VST_LAUNCH(vst_attention_fwd_bf16(q, k, v, key_pad_mask, out, lse2, B, H, T, dh, scale, p, (const unsigned *)dbits,
                                      (hipStream_t)stream, in16 ? 1 : 0));
    return VS_OK;
}

int vs_train_attention_backward_bf16(const float *q, const float *k, const float *v, const uint8_t *key_pad_mask,
                                     const float *out, const float *d_out, const float *lse2, float *dqkv, float *scratch,
                                     int32_t B, int32_t H, int32_t T, int32_t dh, float scale, float p, const void *dbits,
                                     int32_t in16, void *stream) {
    if (!q || !k || !v || !out || !d_out || !lse2 || !dqkv || !scratch) return failf(VS_ERR_INVALID, "NULL pointer");
    if (B <= 0 || H <= 0 || T <= 0) return failf(VS_ERR_INVALID, "B=%d H=%d T=%d", B, H, T);
    hipStream_t st = (hipStream_t)stream;
    VST_LAUNCH(vst_head_rowdot(d_out, out, scratch, B * T, T, H, dh, st, 2));       // dO enters delta as the kernels see it: bf16
    VST_LAUNCH(vst_attention_bwd_bf16(q, k, v, key_pad_mask, d_out, lse2, scratch, dqkv, B, H, T, dh, scale, p,
                                      (const unsigned *)dbits, st, in16 ? 1 : 0));
    return VS_OK;
}

size_t vs_train_wgrad_scratch_floats(int32_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return vst_wgrad_workspace_floats(M, N, K);
}

int vs_train_wgrad(const float *dY, const float *X, int32_t M, int32_t N, int32_t K, float *dW, float *db,
                   float *scratch, void *stream) {
    if (!dY || !X || !dW || !scratch) return failf(VS_ERR_INVALID, "NULL pointer");
    if (M <= 0 || N <= 0 || K <= 0 || N % 4 || K % 4) return failf(VS_ERR_INVALID, "M=%d N=%d K=%d unsupported (N, K multiples of 4)", M, N, K);
    VST_LAUNCH(vst_wgrad(dY, N, X, K, M, N, K, dW, nullptr, nullptr, db, nullptr, nullptr, N, scratch, (hipStream_t)stream));
    return VS_OK;
}

int vs_train_wgrad_bf16(const float *dY, const float *X, int32_t M, int32_t N, int32_t K, float *dW, float *db,
                        float *scratch, void *stream) {
    if (!dY || !X || !dW || !scratch) return failf(VS_ERR_INVALID, "NULL pointer");
    if (M <= 0 || N <= 0 || K <= 0 || N % 4 || K % 4) return failf(VS_ERR_INVALID, "M=%d N=%d K=%d unsupported (N, K multiples of 4)", M, N, K);
    VST_LAUNCH(vst_wgrad(dY, N, X, K, M, N, K, dW, nullptr, nullptr, db, nullptr, nullptr, N, scratch, (hipStream_t)stream, 1));
    return VS_OK;
}

int vs_train_dropout_mask_attention(uint8_t *keep, int32_t B, int32_t H, int32_t T, uint64_t seed, uint32_t site,
                                    float p, void *stream) {
    if (!keep || B <= 0 || H <= 0 || T <= 0) return failf(VS_ERR_INVALID, "bad arguments");
    VST_LAUNCH(vst_attention_dropout_mask(keep, B, H, T, seed, site, p, (hipStream_t)stream));
    return VS_OK;
}

int vs_train_dropout_mask_rows(uint8_t *keep, int32_t M, int32_t cols, uint64_t seed, uint32_t site, float p,
                               void *stream) {
    if (!keep || M <= 0 || cols <= 0) return failf(VS_ERR_INVALID, "bad arguments");
    VST_LAUNCH(vst_rows_dropout_mask(keep, M, cols, seed, site, p, (hipStream_t)stream));
    return VS_OK;
}

}  // extern "C"
