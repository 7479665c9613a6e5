// vs_scorer.cpp — the C ABI of include/vs_scorer.h: argument checks, weight packing, workspace
// carving and the per-layer launch sequence of the scorer's eval forward.
#include "vs_scorer.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "vs_kernels.h"
#include "vs_train_kernels.h"      // vst_attention_fwd: the head-dim-256 attention of the scoring path
#include "vs_weights_impl.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace

// shared with vs_eval.cpp
int vs_fail_msg(int code, const char *msg) { g_err = msg; return code; }

namespace {

#define VS_HIP(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) return fail(VS_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

#define VS_LAUNCH(call)                                                                     \
    do {                                                                                    \
        int e_ = (call);                                                                    \
        if (e_ > 0) return fail(VS_ERR_HIP, "%s: %s", #call, hipGetErrorString((hipError_t)e_)); \
        if (e_ < 0) return fail(VS_ERR_INVALID, "%s: unsupported shape", #call);            \
    } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- optional per-stage HIP-event timing (bench.py's roofline leg) ----
// Events come from a pool that vs_profile_collect refills, so a profiled forward creates none after the first.
const char *const kStageNames[VS_NUM_STAGES] = {"embed_pe", "qkv_proj", "attention", "outproj_ln", "fc1_relu", "fc2_ln_score"};
struct StageRec { int stage; hipEvent_t a, b; };
std::mutex g_prof_mu;
std::atomic<bool> g_prof_on{false};
std::vector<StageRec> g_prof;
std::vector<hipEvent_t> g_event_pool;

hipEvent_t take_event() {
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    }
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}

struct StageScope {
    hipStream_t st; int stage; hipEvent_t a = nullptr, b = nullptr; bool on;
    StageScope(int stage_, hipStream_t st_) : st(st_), stage(stage_), on(g_prof_on.load(std::memory_order_relaxed)) {
        if (!on) return;
        a = take_event(); b = take_event();
        if (!a || !b) { on = false; return; }
        (void)hipEventRecord(a, st);
    }
    ~StageScope() {
        if (!on) return;
        (void)hipEventRecord(b, st);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back({stage, a, b});
    }
};

int check_desc(const vs_model_desc *d) {
    if (!d) return fail(VS_ERR_INVALID, "desc is NULL");
    if (d->d_model <= 0 || d->d_model % 64 || d->d_model > 1024)
        return fail(VS_ERR_INVALID, "d_model=%d unsupported (multiple of 64, <= 1024)", d->d_model);
    if (d->num_heads <= 0 || d->d_model % d->num_heads)
        return fail(VS_ERR_INVALID, "d_model=%d not divisible by num_heads=%d", d->d_model, d->num_heads);
    const int dh = d->d_model / d->num_heads;
    if (dh != 32 && dh != 64 && dh != 128 && dh != 256)
        return fail(VS_ERR_INVALID, "head_dim=%d unsupported (32, 64, 128 or 256)", dh);
    if (d->num_layers < 1) return fail(VS_ERR_INVALID, "num_layers=%d unsupported (>= 1)", d->num_layers);
    if (d->in_features <= 0 || d->in_features % 32)
        return fail(VS_ERR_INVALID, "in_features=%d unsupported (multiple of 32)", d->in_features);
    if (d->num_classes <= 0) return fail(VS_ERR_INVALID, "num_classes=%d", d->num_classes);
    if (d->max_len < 0) return fail(VS_ERR_INVALID, "max_len=%d", d->max_len);
    return VS_OK;
}

}  // namespace

// ---- A/B switches: environment read once, then only vs_set_option() ----
namespace {
struct OptionName { const char *name; int VskOptions::*field; int dflt; };
const OptionName kOptions[] = {
    {"VS_SKINNY_ROWS", &VskOptions::skinny_rows, 16384}, {"VS_LP_MIN_ROWS", &VskOptions::lp_min_rows, 8192},
    {"VS_GEMM_NWM2", &VskOptions::gemm_nwm2, 0},         {"VS_GEMM_NJ2", &VskOptions::gemm_nj2, 0},
    {"VS_ATTN_NW4", &VskOptions::attn_nw4, 0},           {"VS_ATTN_LP_SIMPLE", &VskOptions::attn_lp_simple, 0},
    {"VS_MLP_FUSION", &VskOptions::mlp_fusion, 0},       {"VS_MLP_ABL", &VskOptions::mlp_abl, 0},
    {"VS_ATTN_LEGACY", &VskOptions::attn_legacy, 0},     {"VS_LP_STORE32", &VskOptions::lp_store32, 0},
    {"VS_LP_MLP_UNFUSED", &VskOptions::lp_mlp_unfused, 0}, {"VS_LP_TAIL_UNFUSED", &VskOptions::lp_tail_unfused, 0},
    {"VS_LP_QKV_UNFUSED", &VskOptions::lp_qkv_unfused, 0}, {"VS_LP_EMBED_UNFUSED", &VskOptions::lp_embed_unfused, 0},
    {"VS_LP_MIN_ROWS_FUSED", &VskOptions::lp_min_rows_fused, 256}, {"VS_LP_TILE256", &VskOptions::lp_tile256, 0},
    {"VS_ATTN_W64_CHECKED", &VskOptions::attn_w64_checked, 0}, {"VS_TRAIN_LP_MIN_ROWS", &VskOptions::train_lp_min_rows, 1024},
    {"VS_ATTN_W64", &VskOptions::attn_w64, 1},           {"VS_ATTN_W64_ABL", &VskOptions::attn_w64_abl, 0},
};
int option_from_env(const OptionName &o) {
    const char *e = getenv(o.name);
    if (!e) return o.dflt;
    return *e ? atoi(e) : 1;      // "VAR=" (set but empty) counts as on, like the old `getenv != nullptr` switches
}
}  // namespace

VskOptions &vsk_options() {
    static VskOptions opts = [] {
        VskOptions o{};
        for (const auto &k : kOptions) o.*(k.field) = option_from_env(k);
        return o;
    }();
    return opts;
}

int vsk_device_cus() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (dev >= 0 && dev < 64) { const int c = cus[dev].load(std::memory_order_relaxed); if (c > 0) return c; }
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return -1;
    if (dev >= 0 && dev < 64) cus[dev].store(n, std::memory_order_relaxed);
    return n;
}

extern "C" {

int vs_abi_version(void) { return VS_ABI_VERSION; }

const char *vs_last_error(void) { return g_err.c_str(); }

// copies the parameters into w->blob (stream-ordered on `st`): one batched copy kernel for the embedding / head and
// one per encoder layer instead of 16 hipMemcpyAsync per layer (an optimizer step re-packs on every iteration).
// params->pos_embedding == NULL on an UPDATE keeps the table already packed (it is a buffer: an optimizer never writes it).
static int fill_weights(vs_weights *w, const vs_model_params *params, hipStream_t st, bool update) {
    const vs_model_desc *desc = &w->desc;
    const size_t d = desc->d_model, din = desc->in_features, nc = desc->num_classes;
    if (!params->embed_w || !params->embed_b || !params->final_w || !params->final_b ||
        (desc->num_layers > 0 && !params->layers))
        return fail(VS_ERR_INVALID, "a required parameter pointer is NULL");
    if ((params->pos_embedding != nullptr) != (desc->max_len > 0) && !(update && !params->pos_embedding))
        return fail(VS_ERR_INVALID, "pos_embedding / max_len mismatch");
    VskCopySegs segs{};
    bool ok = true;
    auto add = [&](size_t dst, const float *src, size_t n) {
        if (!src || segs.count >= VSK_COPY_MAX_SEGS) { ok = false; return; }
        segs.src[segs.count] = src; segs.dst[segs.count] = w->blob + dst; segs.n[segs.count] = (unsigned)n; ++segs.count;
    };
    auto flush = [&]() { if (ok && segs.count) ok = vsk_copy_segments(segs, st) == 0; segs.count = 0; };
    add(w->embed_w, params->embed_w, d * din);
    add(w->embed_b, params->embed_b, d);
    add(w->final_w, params->final_w, nc * d);
    add(w->final_b, params->final_b, nc);
    flush();
    if (w->has_pe && params->pos_embedding)
        ok &= hipMemcpyAsync(w->blob + w->pe, params->pos_embedding, (size_t)desc->max_len * d * sizeof(float),
                             hipMemcpyDeviceToDevice, st) == hipSuccess;
    for (int l = 0; l < desc->num_layers && ok; ++l) {
        const vs_layer_params &P = params->layers[l];
        const LayerOff &L = w->layers[l];
        add(L.wqkv, P.wq, d * d); add(L.wqkv + d * d, P.wk, d * d); add(L.wqkv + 2 * d * d, P.wv, d * d);
        add(L.bqkv, P.bq, d);     add(L.bqkv + d, P.bk, d);         add(L.bqkv + 2 * d, P.bv, d);
        add(L.wo, P.wo, d * d);   add(L.bo, P.bo, d);
        add(L.ln1g, P.ln1_g, d);  add(L.ln1b, P.ln1_b, d);
        add(L.w1, P.w1, 4 * d * d); add(L.b1, P.b1, 4 * d);
        add(L.w2, P.w2, 4 * d * d); add(L.b2, P.b2, d);
        add(L.ln2g, P.ln2_g, d);  add(L.ln2b, P.ln2_b, d);
        flush();
    }
    if (!ok)
        return fail(VS_ERR_HIP, "parameter copy failed (NULL pointer or launch error: %s)",
                    hipGetErrorString(hipGetLastError()));
    return VS_OK;
}

}  // extern "C"

namespace { std::mutex g_wmu; }

// kernel-layout images, built on first use after each pack / update (see vs_weights_impl.h)
void vsw_mark(const vs_weights *w, void *stream) {
    if (!w->order_event) {
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return; }
        w->order_event = (void *)ev;
    }
    if (hipEventRecord((hipEvent_t)w->order_event, (hipStream_t)stream) == hipSuccess) {
        w->order_stream = stream;
        w->order_recorded = true;
    } else {
        (void)hipGetLastError();
    }
}

void vsw_order(const vs_weights *w, void *stream) {
    if (w->order_recorded && w->order_stream != stream)
        if (hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)w->order_event, 0) != hipSuccess) (void)hipGetLastError();
}

int vsw_ensure(const vs_weights *w, unsigned families, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(g_wmu);
    vsw_order(w, stream);       // parameters written / images built on another stream: wait for them on the device
    const bool rebuilt = ((families & VSW_FRAGMENTS) && w->f_version != w->version) || ((families & VSW_F16X3) && w->h_version != w->version) ||
                         ((families & VSW_BF16) && w->b_version != w->version) || ((families & VSW_ROWS16) && w->r_version != w->version);
    struct Mark { const vs_weights *w; void *s; bool on; ~Mark() { if (on) vsw_mark(w, s); } } mark{w, stream, rebuilt};
    const size_t d = w->desc.d_model, din = w->desc.in_features;
    float *blob = w->blob;
    bool pk = true;
    if ((families & VSW_FRAGMENTS) && w->f_version != w->version) {
        // one launch for all matrices (a reference-sized training step rebuilds this family after every optimizer step)
        VskMatJobs fj{};
        auto add = [&](size_t src, size_t dst, int N, int K) {
            if (fj.n == VskMatJobs::MAX) { pk &= vsk_pack_fragments_batch(fj, st) == 0; fj.n = 0; }
            fj.in[fj.n] = blob + src; fj.out[fj.n] = blob + dst; fj.rows[fj.n] = N; fj.cols[fj.n] = K; ++fj.n;
        };
        add(w->embed_w, w->f_embed_w, (int)d, (int)din);
        for (const auto &L : w->layers) {
            add(L.wqkv, L.f_wqkv, (int)(3 * d), (int)d);
            add(L.wo, L.f_wo, (int)d, (int)d);
            add(L.w1, L.f_w1, (int)(4 * d), (int)d);
            add(L.w2, L.f_w2, (int)d, (int)(4 * d));
        }
        pk &= vsk_pack_fragments_batch(fj, st) == 0;
        if (pk) w->f_version = w->version;
    }
    if ((families & VSW_F16X3) && w->h_version != w->version) {
        pk &= vsk_pack_fragments_f16x3(blob + w->embed_w, blob + w->h_embed_w, (int)d, (int)din, st) == 0;
        for (const auto &L : w->layers) {
            pk &= vsk_pack_fragments_f16x3(blob + L.wqkv, blob + L.h_wqkv, (int)(3 * d), (int)d, st) == 0;
            pk &= vsk_pack_fragments_f16x3(blob + L.wo, blob + L.h_wo, (int)d, (int)d, st) == 0;
            pk &= vsk_pack_fragments_f16x3(blob + L.w1, blob + L.h_w1, (int)(4 * d), (int)d, st) == 0;
            pk &= vsk_pack_fragments_f16x3(blob + L.w2, blob + L.h_w2, (int)d, (int)(4 * d), st) == 0;
        }
        if (pk) w->h_version = w->version;
    }
    if ((families & VSW_BF16) && w->b_version != w->version) {
        if (w->has_b_embed) pk &= vsk_pack_embed_bf16(blob + w->embed_w, blob + w->b_embed, (int)d, (int)din, st) == 0;
        if (vsk_mlp_bf16_supported((int)d))
            for (const auto &L : w->layers) {
                pk &= vsk_pack_mlp_bf16(blob + L.wo, blob + L.w1, blob + L.w2, blob + L.b_mlp, (int)d, st) == 0;
                pk &= vsk_pack_qkv_bf16(blob + L.wqkv, blob + L.b_qkv, (int)d, st) == 0;
            }
        if (pk) w->b_version = w->version;
    }
    if ((families & VSW_ROWS16) && w->r_version != w->version) {
        for (const auto &L : w->layers) {
            pk &= vsk_to_bf16(blob + L.wqkv, blob + L.r_wqkv, 3 * d * d, st) == 0;
            pk &= vsk_to_bf16(blob + L.wo, blob + L.r_wo, d * d, st) == 0;
            pk &= vsk_to_bf16(blob + L.w1, blob + L.r_w1, 4 * d * d, st) == 0;
            pk &= vsk_to_bf16(blob + L.w2, blob + L.r_w2, 4 * d * d, st) == 0;
        }
        if (pk) w->r_version = w->version;
    }
    if (!pk) return fail(VS_ERR_HIP, "weight image packing failed: %s", hipGetErrorString(hipGetLastError()));
    return VS_OK;
}

extern "C" {

int vs_weights_pack(const vs_model_desc *desc, const vs_model_params *params, void *stream,
                    vs_weights **out) {
    if (int rc = check_desc(desc)) return rc;
    if (!params || !out) return fail(VS_ERR_INVALID, "params/out is NULL");
    const size_t d = desc->d_model, din = desc->in_features, nc = desc->num_classes;

    vs_weights *w = new vs_weights();
    w->desc = *desc;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += align_up(n, 64); return o; };   // 256-B aligned
    w->embed_w = take(d * din);
    w->embed_b = take(d);
    w->has_pe = params->pos_embedding != nullptr;
    if (w->has_pe) w->pe = take((size_t)desc->max_len * d);
    w->layers.resize(desc->num_layers);
    for (auto &L : w->layers) {
        L.wqkv = take(3 * d * d); L.bqkv = take(3 * d);
        L.wo = take(d * d);       L.bo = take(d);
        L.ln1g = take(d);         L.ln1b = take(d);
        L.w1 = take(4 * d * d);   L.b1 = take(4 * d);
        L.w2 = take(4 * d * d);   L.b2 = take(d);
        L.ln2g = take(d);         L.ln2b = take(d);
    }
    w->final_w = take(nc * d);
    w->final_b = take(nc);
    w->f_embed_w = take(d * din);
    for (auto &L : w->layers) {
        L.f_wqkv = take(3 * d * d); L.f_wo = take(d * d); L.f_w1 = take(4 * d * d); L.f_w2 = take(4 * d * d);
    }
    w->h_embed_w = take(d * din);
    for (auto &L : w->layers) {
        L.h_wqkv = take(3 * d * d); L.h_wo = take(d * d); L.h_w1 = take(4 * d * d); L.h_w2 = take(4 * d * d);
    }
    if (vsk_mlp_bf16_supported((int)d))
        for (auto &L : w->layers) {
            L.b_mlp = take(vsk_mlp_bf16_image_bytes((int)d) / sizeof(float));
            L.b_qkv = take(vsk_qkv_bf16_image_bytes((int)d) / sizeof(float));
        }
    for (auto &L : w->layers) {
        L.r_wqkv = take(3 * d * d / 2); L.r_wo = take(d * d / 2); L.r_w1 = take(4 * d * d / 2); L.r_w2 = take(4 * d * d / 2);
    }
    w->has_b_embed = vsk_embed_bf16_image_bytes((int)d, (int)din) != 0;
    if (w->has_b_embed) w->b_embed = take(vsk_embed_bf16_image_bytes((int)d, (int)din) / sizeof(float));
    w->blob_floats = off;
    if (hipGetDevice(&w->device) != hipSuccess) { delete w; return fail(VS_ERR_HIP, "hipGetDevice failed"); }
    hipError_t e = hipMalloc((void **)&w->blob, off * sizeof(float));
    if (e != hipSuccess) { delete w; return fail(VS_ERR_HIP, "hipMalloc(%zu): %s", off * sizeof(float), hipGetErrorString(e)); }
    if (int rc = fill_weights(w, params, (hipStream_t)stream, false)) {
        (void)hipFree(w->blob);
        delete w;
        return rc;
    }
    w->version = 1;
    vsw_mark(w, stream);
    *out = w;
    return VS_OK;
}

int vs_weights_update(vs_weights *w, const vs_model_params *params, void *stream) {
    if (!w || !params) return fail(VS_ERR_INVALID, "weights/params is NULL");
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != w->device)
        return fail(VS_ERR_INVALID, "vs_weights_update on device %d, handle was packed on device %d", dev, w->device);
    ++w->version;        // every image family (and the training side's transposes) is rebuilt on its next use
    vsw_order(w, stream);   // (images still being read / built on another stream)
    const int rc = fill_weights(w, params, (hipStream_t)stream, true);
    vsw_mark(w, stream);
    return rc;
}

int vs_weights_set_norm_width(vs_weights *w, int32_t norm_width) {
    if (!w) return fail(VS_ERR_INVALID, "weights is NULL");
    if (norm_width <= 0 || norm_width % 4 || norm_width > w->desc.d_model)
        return fail(VS_ERR_INVALID, "norm_width=%d unsupported (multiple of 4, 0 < norm_width <= d_model = %d)", norm_width, w->desc.d_model);
    w->norm_width = norm_width;
    return VS_OK;
}

void vs_weights_free(vs_weights *w) {
    if (!w) return;
    if (w->blob) (void)hipFree(w->blob);
    if (w->tblob) (void)hipFree(w->tblob);
    if (w->order_event) (void)hipEventDestroy((hipEvent_t)w->order_event);
    delete w;
}

size_t vs_scorer_workspace_bytes(const vs_weights *w, int32_t B, int32_t T) {
    if (!w || B <= 0 || T <= 0) return 0;
    const size_t md = align_up((size_t)B * T * w->desc.d_model * sizeof(float), 256);
    return 10 * md;     // h0, h1, q, k, v, att, ffn(4)
}

}  // extern "C"

namespace {
// packed ragged batch (vs_scorer_forward_packed): the frames of all videos concatenated as ONE "video" of Mtot rows
// for every row-wise kernel; only the positional rows and the attention know about the video boundaries
struct PackedInfo {
    const float *pe_rows;    // [Mtot, d] positional rows gathered per frame (or nullptr without a positional table)
    const int *cu;           // device [B+1] row offsets
    const int *work;         // device [nwork][2] (video, query tile)
    int nwork, nw;           // query tile = 32*nw rows
    int prec;                // attention arithmetic: 0 exact fp32, 2 fp16x3
};

// cls != nullptr (use_cls=True, reference simnet.py:205-206, 214-216, 47-51): x has T frames per video, a class token
// [d_model] is prepended AFTER the positional encoding, so the encoder sees T + 1 positions (the token is never
// padding) and scores / hidden have T + 1 rows per video.
int forward_core(const vs_weights *w, const float *x, const uint8_t *key_pad_mask, int32_t B,
                 int32_t T, uint32_t flags, float *scores, float *hidden, void *workspace,
                 size_t workspace_bytes, void *stream, const PackedInfo *pk, const float *cls = nullptr) {
    if (!w || !x || !scores) return fail(VS_ERR_INVALID, "weights/x/scores is NULL");
    if (B <= 0 || T <= 0) return fail(VS_ERR_INVALID, "B=%d T=%d", B, T);
    const vs_model_desc &D = w->desc;
    if (!pk && w->has_pe && T > D.max_len)
        return fail(VS_ERR_INVALID, "T=%d exceeds the positional table (max_len=%d)", T, D.max_len);
    if (cls && pk) return fail(VS_ERR_INVALID, "packed batches have no class-token form");
    const int Tf = T;                   // frames per video (rows of x, positional rows)
    if (cls) T += 1;                    // sequence length the encoder sees
    if ((long long)B * T > (1ll << 30)) return fail(VS_ERR_INVALID, "B*T too large");
    if ((flags & VS_FLAG_BF16_LINEAR) && (flags & VS_FLAG_F16X3_LINEAR))
        return fail(VS_ERR_INVALID, "VS_FLAG_BF16_LINEAR and VS_FLAG_F16X3_LINEAR are exclusive");
    if ((flags & (VS_FLAG_BF16_ATTENTION | VS_FLAG_F16X3_ATTENTION)) && D.d_model / D.num_heads != 32 && D.d_model / D.num_heads != 64 &&
        !((flags & VS_FLAG_BF16_ATTENTION) && D.d_model / D.num_heads == 128))
        return fail(VS_ERR_INVALID, "VS_FLAG_BF16_ATTENTION needs head_dim 32, 64 or 128, VS_FLAG_F16X3_ATTENTION 32 or 64 (got %d)", D.d_model / D.num_heads);
    if ((flags & VS_FLAG_BF16_ATTENTION) && (flags & VS_FLAG_F16X3_ATTENTION))
        return fail(VS_ERR_INVALID, "VS_FLAG_BF16_ATTENTION and VS_FLAG_F16X3_ATTENTION are exclusive");
    const size_t need_core = vs_scorer_workspace_bytes(w, B, T);
    const size_t need = need_core + (cls ? align_up((size_t)B * T, 256) : 0);       // + the [B, T+1] key mask
    if (!workspace || workspace_bytes < need)
        return fail(VS_ERR_WORKSPACE, "workspace %zu bytes < %zu needed", workspace_bytes, need);
    if (((uintptr_t)workspace & 255) || ((uintptr_t)x & 15) || (hidden && ((uintptr_t)hidden & 15)) || (cls && ((uintptr_t)cls & 15)))
        return fail(VS_ERR_INVALID, "workspace must be 256-byte, x/hidden/cls_token 16-byte aligned");

    hipStream_t st = (hipStream_t)stream;
    const int d = D.d_model, H = D.num_heads, L = D.num_layers, M = B * T;
    const size_t md = align_up((size_t)M * d * sizeof(float), 256) / sizeof(float);
    float *ws = (float *)workspace;
    float *h0 = ws, *h1 = ws + md, *qkv = ws + 2 * md, *att = ws + 5 * md, *ffn = ws + 6 * md;
    const int dn = w->dn();                             // LayerNorm width / the d_model of the attention scale (== d unless embedded)
    const bool embedded = w->embedded();
    const float scale = 1.0f / sqrtf((float)dn);       // reference simnet.py:126: d_model ** -0.5
    const int sig = (flags & VS_FLAG_SIGMOID) ? 1 : 0;
    // bf16 Linear kernels exist as LDS-tiled throughput kernels only: up to 8192 rows (measured crossover) the exact
    // fp32 latency kernels are faster and are used whatever that flag says.  fp16x3 has its own latency kernels
    // (same product order as its tiled kernels: a video's scores do not depend on the batch it is scored in).
    // (measured crossover, T=1024: 8192 rows for the stand-alone bf16 GEMMs; where the fused layer kernels of
    // vs_mlp_fused.hip apply - d_model 256 with bf16 attention - they win from a single T=320 video on (128-row tiles on
    // 4-wave blocks below half a chip of 256-row tiles: 0.36 ms at 320 rows, 0.43 ms from 1k to 8k rows): 256 rows)
    const VskOptions &opt = vsk_options();
    const bool fused_ok = !embedded && vsk_mlp_bf16_supported(d) && (pk ? pk->prec == 1 : (flags & VS_FLAG_BF16_ATTENTION) != 0) &&
                          !opt.lp_store32 && !opt.lp_mlp_unfused && !opt.attn_lp_simple;
    const int lp_min_rows = fused_ok && opt.lp_min_rows_fused < opt.lp_min_rows ? opt.lp_min_rows_fused
                                                                               : opt.lp_min_rows;    // tests / tools pin the bf16 tiled kernels with 0
    const int lbf = (flags & VS_FLAG_F16X3_LINEAR) ? 2 : ((flags & VS_FLAG_BF16_LINEAR) && M > lp_min_rows) ? 1 : 0;
    // latency mode (VS_FLAG_SPLITK): exact kernels, latency-sized inputs, plain padded batches only
    // (d_model 128 .. 512 - M-A and the reference's argparse default M-B: K slices that are multiples of 128, partials in the free regions)
    const bool splitk = (flags & VS_FLAG_SPLITK) && (lbf == 0 || lbf == 2) && !pk && !cls && M <= vsk_skinny_max_rows() && (d == 128 || d == 256 || d == 512);
    // the bf16 Linear + LayerNorm kernels stop at d_model 256 (validated above); fp16x3 has a wide variant too
    const int lnbf = lbf;
    {   // kernel-layout weight images this forward reads, (re)built only if the parameters changed since their last use
        const int rows_min = cls ? B * Tf : M;
        const bool wide16 = lbf == 1 && d > 256;       // the bf16-operand GEMM's weight copies (ring path below)
        const unsigned fam = lbf == 1 ? (VSW_BF16 | (wide16 ? VSW_ROWS16 : 0u)) : rows_min > vsk_skinny_max_rows() ? 0u : lbf == 2 ? VSW_F16X3 : VSW_FRAGMENTS;
        vsw_order(w, stream);       // parameters written on another stream (vs_weights_update): ordered on the device
        if (fam) if (int rc = vsw_ensure(w, fam, stream)) return rc;
    }
    // bf16 mode: the tensors that are only ever read as bf16 matrix operands are WRITTEN as bf16 by their producers
    // (q * scale * log2 e, k, v; the attention output; the MLP hidden tensor) - same bits, half the HBM bytes
    const int aprec = pk ? pk->prec : (flags & VS_FLAG_F16X3_ATTENTION) ? 2 : (flags & VS_FLAG_BF16_ATTENTION) ? 1 : 0;
    // (d_model > 256 - M-B and wider - has no bf16-storage consumers: its LayerNorm GEMMs are the plain bf16 GEMM + the
    // row pass, which read fp32, and the head-dim-128 attention reads fp32 q / k / v: fp32 storage there)
    const bool qkv16 = lbf == 1 && aprec == 1 && d <= 256 && !embedded && !vsk_options().lp_store32 && !vsk_options().attn_lp_simple;
    const bool ffn16 = lbf == 1 && d <= 256 && !embedded && !vsk_options().lp_store32;
    const bool mlp16 = lbf == 1 && !embedded && vsk_mlp_bf16_supported(d) && !vsk_options().lp_mlp_unfused && !vsk_options().lp_store32;
    // d_model > 256 in bf16 mode with a bf16 attention: the bf16-OPERAND GEMM (vs_gemm_ring.hip).  Every Linear after the
    // embedding reads bf16 from HBM: q/k/v, the attention output and the MLP hidden tensor are written as bf16 by their
    // producers, and the two LayerNorm passes write a bf16 copy of their rows beside the fp32 residual stream.
    const bool ring = lbf == 1 && aprec == 1 && !pk && d > 256 && !opt.lp_store32 && vsk_gemm16_supported(M, d, d);
    const size_t kv_stride = (qkv16 || ring) ? (size_t)M * d / 2 : (size_t)M * d;      // floats between the q, k and v planes
    void *h16 = ffn + 2 * md;               // [M, d] bf16 copy of the current LayerNorm output (the upper half of ffn: the
                                            // bf16 hidden tensor needs the lower half only)

    // bf16 mode with bf16 q/k/v: every layer's tail kernel also projects its output rows to the NEXT layer's q/k/v, and
    // the embedding kernel to the first layer's
    const bool qkv_fused = mlp16 && qkv16 && !vsk_options().lp_qkv_unfused;
    bool have_qkv = false;                  // q/k/v of the layer about to run were written by the kernel before it
    const float *pe_rows = pk ? pk->pe_rows : (w->has_pe ? w->p(w->pe) : nullptr);
    // Embedding + positional table (simnet.py:211, 237-238)
    if (cls) {
        // frames through the embedding GEMM into a free region (h1), then token row + frames -> h0, mask -> mask'
        StageScope ps(VS_STAGE_EMBED, st);
        VS_LAUNCH(vsk_linear(x, w->p(w->embed_w), w->p(lbf == 2 ? w->h_embed_w : w->f_embed_w), w->p(w->embed_b), h1, B * Tf, d,
                             D.in_features, 0, pe_rows, Tf, lbf, st));
        uint8_t *mask1 = key_pad_mask ? (uint8_t *)workspace + need_core : nullptr;
        VS_LAUNCH(vsk_insert_cls(h1, cls, key_pad_mask, h0, mask1, B, Tf, d, st));
        key_pad_mask = mask1;
    } else if (lbf == 1 && w->has_b_embed && qkv_fused && L > 0 && !vsk_options().lp_embed_unfused) {
        StageScope ps(VS_STAGE_EMBED, st);
        const LayerOff &N = w->layers[0];
        const VskNextQkv nq{w->p(N.b_qkv), w->p(N.bqkv), qkv, T, H, vsk_attention_qscale(scale)};
        VS_LAUNCH(vsk_embed_bf16(x, w->p(w->b_embed), w->p(w->embed_b), pe_rows, T, h0, M, d, D.in_features, &nq, st));
        have_qkv = true;
    } else if (splitk && D.in_features % 1024 == 0) {
        // K = in_features split 8 ways; the partials (8 x [M, d]: the q .. ffn regions, all free here) + bias + positional rows -> h0
        StageScope ps(VS_STAGE_EMBED, st);
        VS_LAUNCH(vsk_linear_parts(x, w->p(lbf == 2 ? w->h_embed_w : w->f_embed_w), qkv, M, d, D.in_features, 8, st, lbf == 2));
        VS_LAUNCH(vsk_sum_parts_pe(qkv, 8, w->p(w->embed_b), pe_rows, T, h0, M, d, st));
    } else {
        StageScope ps(VS_STAGE_EMBED, st);
        VS_LAUNCH(vsk_linear(x, w->p(w->embed_w), w->p(lbf == 2 ? w->h_embed_w : w->f_embed_w), w->p(w->embed_b), h0, M, d, D.in_features, 0,
                             pe_rows, T, lbf, st));
    }
    if (ring && L > 0) VS_LAUNCH(vsk_to_bf16(h0, h16, (size_t)M * d, st));
    for (int l = 0; l < L; ++l) {
        const LayerOff &P = w->layers[l];
        const bool last = l == L - 1;
        if (ring) {
            const float qs = vsk_attention_qscale(scale);
            {
                StageScope ps(VS_STAGE_QKV, st);
                VS_LAUNCH(vsk_gemm16(h16, w->p(P.r_wqkv), w->p(P.bqkv), qkv, M, 3 * d, d, 3, 1, T, H, d / H, qs, st));
            }
            {
                StageScope ps(VS_STAGE_ATTENTION, st);
                VS_LAUNCH(vsk_attention_bf16(qkv, qkv + kv_stride, qkv + 2 * kv_stride, key_pad_mask, att, B, H, T, d / H, scale,
                                             1 | VSK_STORE16, st));
            }
            float *dst = (last && hidden) ? hidden : h0;
            {   // out-projection (bf16 attention output in, fp32 out into the q region, free by now) + norm1 -> h1 (+ bf16 copy)
                StageScope ps(VS_STAGE_OUTPROJ_LN, st);
                VS_LAUNCH(vsk_gemm16(att, w->p(P.r_wo), w->p(P.bo), qkv, M, d, d, 0, 0, 1, 0, 0, 1.0f, st));
                VS_LAUNCH(vsk_rows_res_ln(qkv, h0, w->p(P.ln1g), w->p(P.ln1b), h1, M, d, nullptr, nullptr, 0, 0, nullptr, st, h16, dn));
            }
            {
                StageScope ps(VS_STAGE_FC1, st);
                VS_LAUNCH(vsk_gemm16(h16, w->p(P.r_w1), w->p(P.b1), ffn, M, 4 * d, d, 1, 1, 1, 0, 0, 1.0f, st));
            }
            {   // fc2 (fp32 out into the att region) + norm2 (+ score head) -> h0 / hidden (+ bf16 copy for the next layer)
                StageScope ps(VS_STAGE_FC2_LN, st);
                VS_LAUNCH(vsk_gemm16(ffn, w->p(P.r_w2), w->p(P.b2), att, M, d, 4 * d, 0, 0, 1, 0, 0, 1.0f, st));
                VS_LAUNCH(vsk_rows_res_ln(att, h1, w->p(P.ln2g), w->p(P.ln2b), dst, M, d, last ? w->p(w->final_w) : nullptr,
                                          last ? w->p(w->final_b) : nullptr, D.num_classes, sig, last ? scores : nullptr, st,
                                          last ? nullptr : h16, dn));
            }
            continue;
        }
        VskNextQkv nq{}, *next = nullptr;
        if (qkv_fused && !last) {
            const LayerOff &N = w->layers[l + 1];
            nq = VskNextQkv{w->p(N.b_qkv), w->p(N.bqkv), qkv, T, H, vsk_attention_qscale(scale)};
            next = &nq;
        }
        if (!have_qkv) {
            StageScope ps(VS_STAGE_QKV, st);
            VS_LAUNCH(vsk_qkv(h0, w->p(P.wqkv), w->p(lbf == 2 ? P.h_wqkv : P.f_wqkv), w->p(P.bqkv), qkv, B, T, d, H,
                              qkv16 ? (1 | VSK_STORE16) : lbf, st, qkv16 ? vsk_attention_qscale(scale) : 1.0f));
        }
        {
            StageScope ps(VS_STAGE_ATTENTION, st);
            if (pk)
                VS_LAUNCH(vsk_attention_packed(qkv, qkv + kv_stride, qkv + 2 * kv_stride, att, H, M, d / H, scale,
                                               pk->cu, pk->work, pk->nwork, pk->nw, qkv16 ? (1 | VSK_STORE16) : pk->prec, st));
            else if (d / H == 256)      // head dim 256 (round 4, correctness first): the training path's exact forward kernel without
                                        // dropout; its log-sum-exp output lands in the MLP hidden region, which is free here
                VS_LAUNCH(vst_attention_fwd(qkv, qkv + (size_t)M * d, qkv + 2 * (size_t)M * d, key_pad_mask, att, ffn, B, H, T, 256, scale,
                                            0ull, 0u, 0.f, st, nullptr));
            else if (splitk && aprec != 1 && (d / H == 32 || d / H == 64 || d / H == 128))      // latency mode: the keys split over a block's waves
                VS_LAUNCH(vsk_attention_splitkv(qkv, qkv + (size_t)M * d, qkv + 2 * (size_t)M * d, key_pad_mask, att, B, H, T, d / H, scale, st));
            else if (aprec)
                VS_LAUNCH(vsk_attention_bf16(qkv, qkv + kv_stride, qkv + 2 * kv_stride, key_pad_mask, att,
                                             B, H, T, d / H, scale, qkv16 ? (1 | VSK_STORE16) : aprec, st));
            else
                VS_LAUNCH(vsk_attention(qkv, qkv + (size_t)M * d, qkv + 2 * (size_t)M * d, key_pad_mask, att, B, H,
                                        T, d / H, scale, st));
        }
        float *dst = (last && hidden) ? hidden : h0;
        // bf16 mode, d_model 256, bf16 attention output: out-projection + norm1 + the whole MLP block + norm2 (+ score
        // head) as ONE kernel; h1 never exists in HBM
        if (mlp16 && qkv16 && !vsk_options().lp_tail_unfused) {
            StageScope ps(VS_STAGE_FC2_LN, st);
            VS_LAUNCH(vsk_mlp_bf16(h0, att, w->p(P.bo), w->p(P.ln1g), w->p(P.ln1b), w->p(P.b_mlp), w->p(P.b1), w->p(P.b2),
                                   w->p(P.ln2g), w->p(P.ln2b), dst, M, d, last ? w->p(w->final_w) : nullptr,
                                   last ? w->p(w->final_b) : nullptr, D.num_classes, sig, last ? scores : nullptr, next, st));
            have_qkv = next != nullptr;
            continue;
        }
        have_qkv = false;
        // d_model > 256: plain GEMM + the row LayerNorm pass (faster than the fused wide kernel at every M; the
        // GEMM's output goes to a region of the workspace that is free at that point: q after the attention, att after fc1)
        const bool split_ln = d > 256 || embedded;      // (an embedded model's LayerNorm width is not d: the row pass knows it)
        {
            StageScope ps(VS_STAGE_OUTPROJ_LN, st);
            if (splitk && d >= 256) {     // K = d in two slices (q / k regions are free after the attention), LayerNorm as a row pass
                const int so = 2;
                VS_LAUNCH(vsk_linear_parts(att, w->p(lbf == 2 ? P.h_wo : P.f_wo), qkv, M, d, d, so, st, lbf == 2));
                VS_LAUNCH(vsk_rows_res_ln(qkv, h0, w->p(P.ln1g), w->p(P.ln1b), h1, M, d, nullptr, nullptr, 0, 0, nullptr, st, nullptr, dn,
                                          so, w->p(P.bo)));
            } else if (split_ln) {
                VS_LAUNCH(vsk_linear(att, w->p(P.wo), w->p(lnbf == 2 ? P.h_wo : P.f_wo), w->p(P.bo), qkv, M, d, d, 0, nullptr, 1, lnbf, st));
                VS_LAUNCH(vsk_rows_res_ln(qkv, h0, w->p(P.ln1g), w->p(P.ln1b), h1, M, d, nullptr, nullptr, 0, 0, nullptr, st, nullptr, dn));
            } else
            VS_LAUNCH(vsk_linear_res_ln(att, w->p(P.wo), w->p(lnbf == 2 ? P.h_wo : P.f_wo), w->p(P.bo), h0, w->p(P.ln1g), w->p(P.ln1b), h1, M, d, d,
                                        nullptr, nullptr, 0, 0, nullptr, qkv16 ? (1 | VSK_STORE16) : lnbf, st));
        }
#ifdef VS_WITH_DIAG     // diagnostic library only (tools/): the measured-slower fused MLP kernel
        // Opt-in alternative (VS_MLP_FUSION=1, d_model = 256): fc1 + ReLU + fc2 + residual + norm2 (+ score head)
        // as ONE kernel with the activations kept in registers (reported under the fc2 stage).  Bit-identical
        // to the two-kernel path but measured 6 % slower: its ~506 registers per lane allow one
        // wave per SIMD only, so nothing covers the weight loads' issue stalls.
        const bool fused = d == 256 && vsk_options().mlp_fusion != 0 && M > vsk_skinny_max_rows() && !lbf;
        if (fused) {
            StageScope ps(VS_STAGE_FC2_LN, st);
            VS_LAUNCH(vsk_mlp_fused(h1, w->p(P.w1), w->p(P.b1), w->p(P.w2), w->p(P.b2), w->p(P.ln2g), w->p(P.ln2b), dst, M, d,
                                    last ? w->p(w->final_w) : nullptr, last ? w->p(w->final_b) : nullptr,
                                    D.num_classes, sig, last ? scores : nullptr, st));
            continue;
        }
#endif
        // bf16 mode, d_model 256, throughput batches: the whole MLP block as one kernel (hidden activations in registers)
        if (mlp16) {
            StageScope ps(VS_STAGE_FC2_LN, st);
            VS_LAUNCH(vsk_mlp_bf16(h1, nullptr, nullptr, nullptr, nullptr, w->p(P.b_mlp), w->p(P.b1), w->p(P.b2), w->p(P.ln2g), w->p(P.ln2b), dst, M, d,
                                   last ? w->p(w->final_w) : nullptr, last ? w->p(w->final_b) : nullptr,
                                   D.num_classes, sig, last ? scores : nullptr, next, st));
            have_qkv = next != nullptr;
            continue;
        }
        {
            StageScope ps(VS_STAGE_FC1, st);
            VS_LAUNCH(vsk_linear(h1, w->p(P.w1), w->p(lbf == 2 ? P.h_w1 : P.f_w1), w->p(P.b1), ffn, M, 4 * d, d, 1, nullptr, 1,
                                 ffn16 ? (1 | VSK_STORE16) : lbf, st));
        }
        if (splitk) {      // K = 4 d split four ways (partials in the q / k / v / att regions, free by now), LayerNorm (+ score head) as a row pass
            StageScope ps(VS_STAGE_FC2_LN, st);
            VS_LAUNCH(vsk_linear_parts(ffn, w->p(lbf == 2 ? P.h_w2 : P.f_w2), qkv, M, d, 4 * d, 4, st, lbf == 2));
            VS_LAUNCH(vsk_rows_res_ln(qkv, h1, w->p(P.ln2g), w->p(P.ln2b), dst, M, d, last ? w->p(w->final_w) : nullptr,
                                      last ? w->p(w->final_b) : nullptr, D.num_classes, sig, last ? scores : nullptr, st, nullptr, dn,
                                      4, w->p(P.b2)));
        } else if (split_ln) {
            StageScope ps(VS_STAGE_FC2_LN, st);
            VS_LAUNCH(vsk_linear(ffn, w->p(P.w2), w->p(lnbf == 2 ? P.h_w2 : P.f_w2), w->p(P.b2), att, M, d, 4 * d, 0, nullptr, 1, lnbf, st));
            VS_LAUNCH(vsk_rows_res_ln(att, h1, w->p(P.ln2g), w->p(P.ln2b), dst, M, d, last ? w->p(w->final_w) : nullptr,
                                      last ? w->p(w->final_b) : nullptr, D.num_classes, sig, last ? scores : nullptr, st, nullptr, dn));
        } else {
            StageScope ps(VS_STAGE_FC2_LN, st);
            VS_LAUNCH(vsk_linear_res_ln(ffn, w->p(P.w2), w->p(lnbf == 2 ? P.h_w2 : P.f_w2), w->p(P.b2), h1, w->p(P.ln2g), w->p(P.ln2b), dst, M, d,
                                        4 * d, last ? w->p(w->final_w) : nullptr,
                                        last ? w->p(w->final_b) : nullptr, D.num_classes, sig,
                                        last ? scores : nullptr, ffn16 ? (1 | VSK_STORE16) : lnbf, st));
        }
    }
    return VS_OK;
}
}  // namespace

extern "C" {

int vs_scorer_forward(const vs_weights *w, const float *x, const uint8_t *key_pad_mask, int32_t B,
                      int32_t T, uint32_t flags, float *scores, float *hidden, void *workspace,
                      size_t workspace_bytes, void *stream) {
    return forward_core(w, x, key_pad_mask, B, T, flags, scores, hidden, workspace, workspace_bytes, stream, nullptr);
}

size_t vs_scorer_workspace_bytes_cls(const vs_weights *w, int32_t B, int32_t T) {
    if (!w || B <= 0 || T <= 0) return 0;
    return vs_scorer_workspace_bytes(w, B, T + 1) + align_up((size_t)B * (T + 1), 256);
}

int vs_scorer_forward_cls(const vs_weights *w, const float *x, const uint8_t *key_pad_mask, const float *cls_token,
                          int32_t B, int32_t T, uint32_t flags, float *scores, float *hidden, void *workspace,
                          size_t workspace_bytes, void *stream) {
    if (!cls_token) return fail(VS_ERR_INVALID, "cls_token is NULL");
    return forward_core(w, x, key_pad_mask, B, T, flags, scores, hidden, workspace, workspace_bytes, stream, nullptr, cls_token);
}

// ---- packed ragged batches ----
static int packed_plan(const vs_weights *w, const int32_t *lengths, int32_t B, std::vector<int> &cu,
                       std::vector<int> &work, int &nw, bool narrow_only = false, bool wide_only = false) {
    if (!w || !lengths || B <= 0) return fail(VS_ERR_INVALID, "weights/lengths is NULL or B=%d", B);
    const vs_model_desc &D = w->desc;
    const int dh = D.d_model / D.num_heads;
    if (dh != 32 && dh != 64 && dh != 128) return fail(VS_ERR_INVALID, "packed batches need head_dim 32, 64 or 128 (got %d)", dh);
    cu.assign(B + 1, 0);
    long long r8 = 0, r4 = 0;
    for (int b = 0; b < B; ++b) {
        const int t = lengths[b];
        if (t <= 0) return fail(VS_ERR_INVALID, "lengths[%d]=%d", b, t);
        if (w->has_pe && t > D.max_len)
            return fail(VS_ERR_INVALID, "T=%d exceeds the positional table (max_len=%d)", t, D.max_len);
        if ((long long)cu[b] + t > (1ll << 30)) return fail(VS_ERR_INVALID, "too many frames");
        cu[b + 1] = cu[b] + t;
        r8 += (t + 255) / 256 * 256;
        r4 += (t + 127) / 128 * 128;
    }
    nw = (!narrow_only && r8 * 100 <= r4 * 105) ? 8 : 4;     // 256-row query tiles unless the ragged tails waste too much
    if (dh == 128) nw = wide_only ? 8 : 4;                   // head dim 128 (round 4): the exact kernel has 4-wave blocks, the bf16 one 8-wave blocks only
    const int qb = 32 * nw;
    work.clear();
    for (int b = 0; b < B; ++b)
        for (int q = 0; q < (lengths[b] + qb - 1) / qb; ++q) { work.push_back(b); work.push_back(q); }
    return VS_OK;
}

// (video, tile) pairs the plan area is sized for: ALWAYS the 128-row tiling, the longer of the two lists, so the
// size does not depend on which tiling the forward picks for the flags it is given
static size_t packed_plan_capacity(const int32_t *lengths, int32_t B) {
    size_t n = 0;
    for (int b = 0; b < B; ++b) n += (size_t)(lengths[b] + 127) / 128;
    return n;
}

size_t vs_scorer_workspace_bytes_packed(const vs_weights *w, const int32_t *lengths, int32_t B) {
    std::vector<int> cu, work;
    int nw = 0;
    if (packed_plan(w, lengths, B, cu, work, nw) != VS_OK) return 0;
    const size_t md = align_up((size_t)cu[B] * w->desc.d_model * sizeof(float), 256);
    // + gathered positional rows, plan (row offsets + work list)
    return 11 * md + align_up((cu.size() + 2 * packed_plan_capacity(lengths, B)) * sizeof(int), 256);
}

int vs_scorer_forward_packed(const vs_weights *w, const float *x, const int32_t *lengths, const int32_t *lengths_dev,
                             int32_t B, uint32_t flags, float *scores, float *hidden, void *workspace,
                             size_t workspace_bytes, void *stream) {
    if (!lengths_dev) return fail(VS_ERR_INVALID, "lengths_dev is NULL");
    if (!w) return fail(VS_ERR_INVALID, "weights is NULL");
    if ((flags & VS_FLAG_BF16_ATTENTION) && (flags & VS_FLAG_F16X3_ATTENTION))
        return fail(VS_ERR_INVALID, "VS_FLAG_BF16_ATTENTION and VS_FLAG_F16X3_ATTENTION are exclusive");
    const int aprec = (flags & VS_FLAG_F16X3_ATTENTION) ? 2 : (flags & VS_FLAG_BF16_ATTENTION) ? 1 : 0;
    std::vector<int> cu, work;
    int nw = 0;
    // (the low-precision attention has no 8-wave form for head dim 32)
    if (aprec == 2 && w->desc.d_model / w->desc.num_heads == 128)
        return fail(VS_ERR_INVALID, "VS_FLAG_F16X3_ATTENTION needs head_dim 32 or 64 (got 128)");
    if (int rc = packed_plan(w, lengths, B, cu, work, nw, aprec != 0 && w->desc.d_model / w->desc.num_heads == 32, aprec == 1)) return rc;
    const size_t need = vs_scorer_workspace_bytes_packed(w, lengths, B);
    if (!workspace || workspace_bytes < need)
        return fail(VS_ERR_WORKSPACE, "workspace %zu bytes < %zu needed", workspace_bytes, need);
    if ((uintptr_t)workspace & 255) return fail(VS_ERR_INVALID, "workspace must be 256-byte aligned");
    const int M = cu[B], d = w->desc.d_model;
    const size_t md = align_up((size_t)M * d * sizeof(float), 256);
    hipStream_t st = (hipStream_t)stream;
    float *pe_rows = (float *)((char *)workspace + 10 * md);
    int *plan = (int *)((char *)workspace + 11 * md);
    // the device builds its own copy of the plan from the device lengths (no host memory is read after return,
    // no synchronisation); the host copy above only sized the launch
    const size_t cap = packed_plan_capacity(lengths, B);
    if (work.size() / 2 > cap) return fail(VS_ERR_INVALID, "internal: packed plan %zu > capacity %zu", work.size() / 2, cap);
    VS_LAUNCH(vsk_plan_packed(lengths_dev, B, 32 * nw, plan, plan + cu.size(), (int)cap, st));
    PackedInfo pk{nullptr, plan, plan + cu.size(), (int)(work.size() / 2), nw, aprec};
    if (w->has_pe) {
        int tmax = 0;
        for (int b = 0; b < B; ++b) tmax = lengths[b] > tmax ? lengths[b] : tmax;
        VS_LAUNCH(vsk_gather_rows(w->p(w->pe), pk.cu, B, tmax, d, pe_rows, st));
        pk.pe_rows = pe_rows;
    }
    return forward_core(w, x, nullptr, 1, M, flags & ~(VS_FLAG_F16X3_ATTENTION | VS_FLAG_BF16_ATTENTION), scores, hidden, workspace, 10 * md, stream, &pk);
}

int vs_profile_enable(int32_t on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &r : g_prof) { g_event_pool.push_back(r.a); g_event_pool.push_back(r.b); }
    g_prof.clear();
    if (!on) {      // profiling off: give the events back to the runtime
        for (hipEvent_t e : g_event_pool) (void)hipEventDestroy(e);
        g_event_pool.clear();
    }
    g_prof_on.store(on != 0);
    return VS_OK;
}

int vs_set_option(const char *name, int32_t value) {
    if (!name) return fail(VS_ERR_INVALID, "option name is NULL");
    for (const auto &k : kOptions)
        if (strcmp(k.name, name) == 0) {
            vsk_options().*(k.field) = value < 0 ? option_from_env(k) : value;
            return VS_OK;
        }
    return fail(VS_ERR_INVALID, "unknown option %s", name);
}

int vs_profile_collect(double *ms_sum, int64_t *launches) {
    if (!ms_sum || !launches) return fail(VS_ERR_INVALID, "NULL pointer");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < VS_NUM_STAGES; ++i) { ms_sum[i] = 0.0; launches[i] = 0; }
    for (auto &r : g_prof) {
        VS_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        VS_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        ms_sum[r.stage] += ms;
        launches[r.stage] += 1;
        g_event_pool.push_back(r.a);
        g_event_pool.push_back(r.b);
    }
    g_prof.clear();
    return VS_OK;
}

const char *vs_stage_name(int32_t stage) {
    return (stage >= 0 && stage < VS_NUM_STAGES) ? kStageNames[stage] : "";
}

#ifdef VS_WITH_DIAG
// Diagnostic entry (not part of include/vs_scorer.h): fc1-shaped GEMM with s_memtime phase stamps.
// diag = [grid*4 waves][8] u64: issue, mfma, epilogue, stage, barrier, total cycles, k-tiles, t_begin.
int vs_diag_gemm(const float *A, const float *W, const float *bias, float *C, int32_t M, int32_t N, int32_t K,
                 int32_t grid, unsigned long long *diag, void *stream) {
    VS_LAUNCH(vsk_diag_gemm(A, W, bias, C, M, N, K, grid, diag, (hipStream_t)stream));
    return VS_OK;
}
// Diagnostic entry: exact head-dim-64 attention with per-phase s_memtime stamps; returns the number of blocks (> 0) or
// -status.  diag: [blocks * 8 waves][16] u64 (layout: vs_attention.hip, attn_fwd_pipe DIAG).
int vs_diag_attention(const float *q, const float *k, const float *v, float *out, int32_t B, int32_t H, int32_t T,
                      float scale, unsigned long long *diag, void *stream) {
    const int blocks = vsk_diag_attention(q, k, v, out, B, H, T, scale, diag, (hipStream_t)stream);
    return blocks;
}
int vs_diag_attention_lp(const float *q, const float *k, const float *v, float *out, int32_t B, int32_t H, int32_t T,
                         float scale, int32_t prec, unsigned long long *diag, void *stream) {
    return vsk_diag_attention_lp(q, k, v, out, B, H, T, scale, prec, diag, (hipStream_t)stream);
}
#endif  // VS_WITH_DIAG

static int linear_entry(const float *A, const float *W, const float *bias, float *C, int32_t M, int32_t N,
                        int32_t K, int32_t relu, const float *pe, int32_t T, int bf16, void *stream) {
    if (!A || !W || !bias || !C) return fail(VS_ERR_INVALID, "NULL pointer");
    if (M <= 0 || N <= 0 || N % 32 || K <= 0 || K % 32)
        return fail(VS_ERR_INVALID, "M=%d N=%d K=%d unsupported (N, K multiples of 32)", M, N, K);
    if (pe && (T <= 0 || relu)) return fail(VS_ERR_INVALID, "pe needs T > 0 and relu == 0");
    VS_LAUNCH(vsk_linear(A, W, nullptr, bias, C, M, N, K, relu, pe, T > 0 ? T : 1, bf16, (hipStream_t)stream));
    return VS_OK;
}

int vs_linear_bf16_operands(const void *A16, const void *W16, const float *bias, void *C, int32_t M, int32_t N, int32_t K,
                            int32_t relu, int32_t c16, void *stream) {
    if (!A16 || !W16 || !bias || !C) return fail(VS_ERR_INVALID, "NULL pointer");
    if (!vsk_gemm16_supported(M, N, K)) return fail(VS_ERR_INVALID, "M=%d N=%d K=%d unsupported (N, K multiples of 32)", M, N, K);
    if (((uintptr_t)A16 | (uintptr_t)W16 | (uintptr_t)C | (uintptr_t)bias) & 15) return fail(VS_ERR_INVALID, "operands must be 16-byte aligned");
    VS_LAUNCH(vsk_gemm16(A16, W16, bias, C, M, N, K, relu ? 1 : 0, c16 ? 1 : 0, 1, 0, 0, 1.0f, (hipStream_t)stream));
    return VS_OK;
}

int vs_qkv_proj_bf16_operands(const void *h16, const void *Wqkv16, const float *bqkv, void *qkv, int32_t B, int32_t T,
                              int32_t d, int32_t H, int32_t c16, void *stream) {
    if (!h16 || !Wqkv16 || !bqkv || !qkv) return fail(VS_ERR_INVALID, "NULL pointer");
    if (B <= 0 || T <= 0 || d <= 0 || d % 32 || H <= 0 || d % H || (d / H) % 32)
        return fail(VS_ERR_INVALID, "B=%d T=%d d=%d H=%d unsupported", B, T, d, H);
    VS_LAUNCH(vsk_gemm16(h16, Wqkv16, bqkv, qkv, B * T, 3 * d, d, 3, c16 ? 1 : 0, T, H, d / H,
                         vsk_attention_qscale(1.0f / sqrtf((float)d)), (hipStream_t)stream));
    return VS_OK;
}

int vs_to_bf16(const float *src, void *dst16, size_t n, void *stream) {
    if (!src || !dst16 || n % 8 || (((uintptr_t)src | (uintptr_t)dst16) & 15)) return fail(VS_ERR_INVALID, "vs_to_bf16: n %% 8 == 0, 16-byte aligned pointers");
    VS_LAUNCH(vsk_to_bf16(src, dst16, n, (hipStream_t)stream));
    return VS_OK;
}

int vs_linear_bf16_a16(const void *A16, const float *W, const float *bias, float *C, int32_t M, int32_t N, int32_t K,
                       const float *pe, void *stream) {
    if (!A16 || !W || !bias || !C) return fail(VS_ERR_INVALID, "NULL pointer");
    if (M <= 0 || N <= 0 || N % 32 || K <= 0 || K % 32)
        return fail(VS_ERR_INVALID, "M=%d N=%d K=%d unsupported (N, K multiples of 32)", M, N, K);
    VS_LAUNCH(vsk_linear((const float *)A16, W, nullptr, bias, C, M, N, K, 0, pe, M, 1 | VSK_A16, (hipStream_t)stream));
    return VS_OK;
}

int vs_linear_f32(const float *A, const float *W, const float *bias, float *C, int32_t M, int32_t N,
                  int32_t K, int32_t relu, const float *pe, int32_t T, void *stream) {
    return linear_entry(A, W, bias, C, M, N, K, relu, pe, T, 0, stream);
}

int vs_linear_bf16(const float *A, const float *W, const float *bias, float *C, int32_t M, int32_t N,
                   int32_t K, int32_t relu, const float *pe, int32_t T, void *stream) {
    return linear_entry(A, W, bias, C, M, N, K, relu, pe, T, 1, stream);
}

int vs_mlp_block_bf16(const vs_weights *w, int32_t layer, const float *h, float *out, int32_t M, int32_t with_head,
                      int32_t sigmoid, float *scores, void *stream) {
    if (!w || !h || !out) return fail(VS_ERR_INVALID, "NULL pointer");
    if (layer < 0 || layer >= w->desc.num_layers || M <= 0) return fail(VS_ERR_INVALID, "layer=%d M=%d", layer, M);
    if (!vsk_mlp_bf16_supported(w->desc.d_model))
        return fail(VS_ERR_INVALID, "the fused bf16 MLP kernel needs d_model == 256 (got %d)", w->desc.d_model);
    if (with_head && !scores) return fail(VS_ERR_INVALID, "scores is NULL");
    if (((uintptr_t)h & 15) || ((uintptr_t)out & 15)) return fail(VS_ERR_INVALID, "h/out must be 16-byte aligned");
    const LayerOff &P = w->layers[layer];
    if (int rc = vsw_ensure(w, VSW_BF16, stream)) return rc;
    VS_LAUNCH(vsk_mlp_bf16(h, nullptr, nullptr, nullptr, nullptr, w->p(P.b_mlp), w->p(P.b1), w->p(P.b2), w->p(P.ln2g), w->p(P.ln2b), out, M,
                           w->desc.d_model, with_head ? w->p(w->final_w) : nullptr, with_head ? w->p(w->final_b) : nullptr,
                           w->desc.num_classes, sigmoid, with_head ? scores : nullptr, nullptr, (hipStream_t)stream));
    return VS_OK;
}

int vs_linear_f16x3(const float *A, const float *W, const float *bias, float *C, int32_t M, int32_t N,
                    int32_t K, int32_t relu, const float *pe, int32_t T, void *stream) {
    return linear_entry(A, W, bias, C, M, N, K, relu, pe, T, 2, stream);
}

int vs_qkv_proj_f32(const float *h, const float *Wqkv, const float *bqkv, float *qkv, int32_t B,
                    int32_t T, int32_t d, int32_t H, void *stream) {
    if (!h || !Wqkv || !bqkv || !qkv) return fail(VS_ERR_INVALID, "NULL pointer");
    if (B <= 0 || T <= 0 || d <= 0 || d % 32 || H <= 0 || d % H || (d / H) % 32)
        return fail(VS_ERR_INVALID, "B=%d T=%d d=%d H=%d unsupported", B, T, d, H);
    VS_LAUNCH(vsk_qkv(h, Wqkv, nullptr, bqkv, qkv, B, T, d, H, 0, (hipStream_t)stream));
    return VS_OK;
}

int vs_attention_f32(const float *q, const float *k, const float *v, const uint8_t *key_pad_mask,
                     float *out, int32_t B, int32_t H, int32_t T, int32_t dh, float scale, void *stream) {
    if (!q || !k || !v || !out) return fail(VS_ERR_INVALID, "NULL pointer");
    if (B <= 0 || H <= 0 || T <= 0) return fail(VS_ERR_INVALID, "B=%d H=%d T=%d", B, H, T);
    if (dh != 32 && dh != 64 && dh != 128) return fail(VS_ERR_INVALID, "head_dim=%d unsupported", dh);
    VS_LAUNCH(vsk_attention(q, k, v, key_pad_mask, out, B, H, T, dh, scale, (hipStream_t)stream));
    return VS_OK;
}

static int attention_lp_entry(const float *q, const float *k, const float *v, const uint8_t *key_pad_mask,
                              float *out, int32_t B, int32_t H, int32_t T, int32_t dh, float scale, int prec, void *stream) {
    if (!q || !k || !v || !out) return fail(VS_ERR_INVALID, "NULL pointer");
    if (B <= 0 || H <= 0 || T <= 0) return fail(VS_ERR_INVALID, "B=%d H=%d T=%d", B, H, T);
    if (dh != 32 && dh != 64 && !(dh == 128 && prec == 1))
        return fail(VS_ERR_INVALID, "head_dim=%d unsupported on the %s path", dh, prec == 1 ? "bf16" : "f16x3");
    VS_LAUNCH(vsk_attention_bf16(q, k, v, key_pad_mask, out, B, H, T, dh, scale, prec, (hipStream_t)stream));
    return VS_OK;
}

int vs_attention_bf16(const float *q, const float *k, const float *v, const uint8_t *key_pad_mask,
                      float *out, int32_t B, int32_t H, int32_t T, int32_t dh, float scale, void *stream) {
    return attention_lp_entry(q, k, v, key_pad_mask, out, B, H, T, dh, scale, 1, stream);
}

int vs_attention_bf16_stored(const void *q16, const void *k16, const void *v16, const uint8_t *key_pad_mask,
                             void *out16, int32_t B, int32_t H, int32_t T, int32_t dh, void *stream) {
    if (!q16 || !k16 || !v16 || !out16) return fail(VS_ERR_INVALID, "NULL pointer");
    if (B <= 0 || H <= 0 || T <= 0) return fail(VS_ERR_INVALID, "B=%d H=%d T=%d", B, H, T);
    if (dh != 32 && dh != 64 && dh != 128) return fail(VS_ERR_INVALID, "head_dim=%d unsupported on the bf16 path", dh);
    if ((((uintptr_t)q16 | (uintptr_t)k16 | (uintptr_t)v16 | (uintptr_t)out16) & 15) != 0)
        return fail(VS_ERR_INVALID, "q16 / k16 / v16 / out16 must be 16-byte aligned");
    VS_LAUNCH(vsk_attention_bf16((const float *)q16, (const float *)k16, (const float *)v16, key_pad_mask, (float *)out16, B, H, T, dh,
                                 1.0f, 1 | VSK_STORE16, (hipStream_t)stream));
    return VS_OK;
}

float vs_attention_qscale(float scale) { return vsk_attention_qscale(scale); }

int vs_attention_f16x3(const float *q, const float *k, const float *v, const uint8_t *key_pad_mask,
                       float *out, int32_t B, int32_t H, int32_t T, int32_t dh, float scale, void *stream) {
    return attention_lp_entry(q, k, v, key_pad_mask, out, B, H, T, dh, scale, 2, stream);
}

static int linear_ln_entry(const float *A, const float *W, const float *bias,
                           const float *residual, const float *gamma, const float *beta,
                           float *out, int32_t M, int32_t N, int32_t K, const float *score_w,
                           const float *score_b, int32_t num_classes, int32_t sigmoid,
                           float *scores, int bf16, void *stream) {
    if (!A || !W || !bias || !residual || !gamma || !beta || !out) return fail(VS_ERR_INVALID, "NULL pointer");
    if (M <= 0 || N <= 0 || N % 64 || N > 512 || K <= 0 || K % 16)
        return fail(VS_ERR_INVALID, "M=%d N=%d K=%d unsupported (N multiple of 64 <= 512, K multiple of 16)", M, N, K);
    if (score_w && (!score_b || !scores || num_classes <= 0))
        return fail(VS_ERR_INVALID, "score head needs score_b, scores and num_classes > 0");
    if (bf16 && N > 256) return fail(VS_ERR_INVALID, "N=%d unsupported on the bf16 path (N <= 256)", N);
    VS_LAUNCH(vsk_linear_res_ln(A, W, nullptr, bias, residual, gamma, beta, out, M, N, K, score_w, score_b, num_classes,
                                sigmoid, scores, bf16, (hipStream_t)stream));
    return VS_OK;
}

int vs_linear_residual_layernorm_f32(const float *A, const float *W, const float *bias,
                                     const float *residual, const float *gamma, const float *beta,
                                     float *out, int32_t M, int32_t N, int32_t K, const float *score_w,
                                     const float *score_b, int32_t num_classes, int32_t sigmoid,
                                     float *scores, void *stream) {
    return linear_ln_entry(A, W, bias, residual, gamma, beta, out, M, N, K, score_w, score_b, num_classes, sigmoid,
                           scores, 0, stream);
}

int vs_linear_residual_layernorm_bf16(const float *A, const float *W, const float *bias,
                                      const float *residual, const float *gamma, const float *beta,
                                      float *out, int32_t M, int32_t N, int32_t K, const float *score_w,
                                      const float *score_b, int32_t num_classes, int32_t sigmoid,
                                      float *scores, void *stream) {
    return linear_ln_entry(A, W, bias, residual, gamma, beta, out, M, N, K, score_w, score_b, num_classes, sigmoid,
                           scores, 1, stream);
}

int vs_linear_residual_layernorm_f16x3(const float *A, const float *W, const float *bias,
                                       const float *residual, const float *gamma, const float *beta,
                                       float *out, int32_t M, int32_t N, int32_t K, const float *score_w,
                                       const float *score_b, int32_t num_classes, int32_t sigmoid,
                                       float *scores, void *stream) {
    return linear_ln_entry(A, W, bias, residual, gamma, beta, out, M, N, K, score_w, score_b, num_classes, sigmoid,
                           scores, 2, stream);
}

}  // extern "C"
