// vs_pretrain_kernels.hip — the PretrainModel loss head on gfx950 (reference
// src/model/simnet_pretrain.py:49-69 repelling loss, :80-98 masked score-softmax pooling, centering penalty and the
// soft cross-entropy against the video representation), forward and backward.
//
// Per video b with frame features f_t (the video_transform output, F columns), scorer logits s_t and key mask:
//   w      = softmax_t(s_t / temp) over the unmasked frames                            :84-87
//   center = mean_t [unmasked] (w_t + 1e-9) log(w_t + 1e-9)   (entropy)  |  ||w||_2   (norm)      :88-92, 43-47
//   pooled = sum_t w_t f_t                                                             :93-95
//   loss   = mean_j -softmax(vid)_j log softmax(pooled)_j                              :96-97, 35-41
//   repel  = ( ||sum_t x^_t||^2 - sum_t ||x^_t||^2 ) / T^2,  x^_t = f_t [unmasked] / (||f_t|| + 1e-9)
//            = the mean of the off-diagonal of the reference's materialised [T,T] cosine matrix (:56-69) without the
//            matrix: O(T F) instead of O(T^2 F), forward and backward.
// The three returned losses are the means over the batch.  Everything is HBM-bound row work (one pass over the
// [B,T,F] features forward, two backward): a wave owns a frame, a lane 4 consecutive features per 256, and all sums
// are two-stage with fixed order (chunk partials -> per-video finals -> batch means), so results are reproducible.
#include "vs_train_device.h"
#include "vs_train_kernels.h"

namespace {

constexpr int CHUNK = 64;            // frames per block
constexpr float EPS = 1e-9f;         // the reference's stabilisers (:55, :90)
enum { ST_M = 0, ST_Z, ST_D, ST_CEN, ST_R, ST_W2, ST_N = 8 };     // per-video scalars

struct Layout {                      // scratch carving (floats)
    size_t stats, pooled, sumx, dce, part, rpart, a, total;
    int nc;
};
__host__ __device__ inline Layout layout(int B, int T, int F) {
    Layout L;
    L.nc = (T + CHUNK - 1) / CHUNK;
    size_t off = 0;
    L.stats = off; off += (size_t)B * ST_N;
    L.pooled = off; off += (size_t)B * F;
    L.sumx = off; off += (size_t)B * F;
    L.dce = off; off += (size_t)B * F;
    L.part = off; off += (size_t)B * L.nc * (2 * F + 4);
    L.rpart = off; off += (size_t)B * L.nc;
    L.a = off; off += (size_t)B * T;
    L.total = off;
    return L;
}

// block-wide max and sum-exp of the video's scaled, masked scores (every block recomputes them: T <= a few thousand)
__device__ __forceinline__ void softmax_stats(const float *__restrict__ s, const uint8_t *__restrict__ mk, int T,
                                              float inv_temp, float &m, float &Z, float *red) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -__builtin_inff();
    for (int t = tid; t < T; t += 256)
        if (!mk || !mk[t]) mx = fmaxf(mx, s[t] * inv_temp);
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float z = 0.f;
    for (int t = tid; t < T; t += 256)
        if (!mk || !mk[t]) z += expf(s[t] * inv_temp - m);
    z = wave_sum(z);
    if (lane == 0) red[wave] = z;
    __syncthreads();
    Z = ((red[0] + red[1]) + red[2]) + red[3];
    __syncthreads();
}

// ---- forward, stage 1: per (video, 64-frame chunk) partial sums ----
template <int NV>       // F = 256 * NV
__global__ __launch_bounds__(256) void head_partial(const float *__restrict__ feats, const float *__restrict__ scores,
                                                    const uint8_t *__restrict__ mask, int T, float inv_temp, int entropy,
                                                    float *__restrict__ scratch, int B) {
    constexpr int F = 256 * NV;
    __shared__ float red[4][2 * F + 4];
    __shared__ float sred[4];
    const Layout L = layout(B, T, F);
    const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *s = scores + (size_t)b * T;
    const uint8_t *mk = mask ? mask + (size_t)b * T : nullptr;
    float m, Z;
    softmax_stats(s, mk, T, inv_temp, m, Z, sred);
    f32x4 pool[NV], sx[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) { pool[u] = f32x4{0.f, 0.f, 0.f, 0.f}; sx[u] = pool[u]; }
    float dsum = 0.f, cen = 0.f;
    for (int t = ch * CHUNK + wave; t < min(T, (ch + 1) * CHUNK); t += 4) {
        const bool valid = !mk || !mk[t];
        const float w = valid ? expf(s[t] * inv_temp - m) / Z : 0.f;
        f32x4 f[NV];
        float nn = 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            f[u] = *(const f32x4 *)(feats + ((size_t)b * T + t) * F + 4 * lane + 256 * u);
            nn += f[u][0] * f[u][0] + f[u][1] * f[u][1] + f[u][2] * f[u][2] + f[u][3] * f[u][3];
            pool[u] += f[u] * w;
        }
        if (valid) {
            const float n = sqrtf(wave_sum(nn)), inv = 1.0f / (n + EPS);
#pragma unroll
            for (int u = 0; u < NV; ++u) sx[u] += f[u] * inv;
            dsum += (n * inv) * (n * inv);
            cen += entropy ? (w + EPS) * logf(w + EPS) : w * w;      // masked frames: masked_fill(mask, 0.) (:46) / w = 0
        }
    }
#pragma unroll
    for (int u = 0; u < NV; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[wave][4 * lane + 256 * u + e] = pool[u][e];
            red[wave][F + 4 * lane + 256 * u + e] = sx[u][e];
        }
    if (lane == 0) { red[wave][2 * F] = dsum; red[wave][2 * F + 1] = cen; }
    __syncthreads();
    float *out = scratch + L.part + ((size_t)b * L.nc + ch) * (2 * F + 4);
    for (int i = tid; i < 2 * F + 2; i += 256) out[i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
}

// ---- forward, stage 2: per-video finals ----
template <int NV>
__global__ __launch_bounds__(256) void head_final(const float *__restrict__ scores, const uint8_t *__restrict__ mask,
                                                  const float *__restrict__ vid, int T, float inv_temp, int entropy,
                                                  float *__restrict__ scratch, int B) {
    constexpr int F = 256 * NV;
    __shared__ float sred[4];
    __shared__ float vec[3][F];            // pooled, sumx, vid
    const Layout L = layout(B, T, F);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m, Z;
    softmax_stats(scores + (size_t)b * T, mask ? mask + (size_t)b * T : nullptr, T, inv_temp, m, Z, sred);
    const float *part = scratch + L.part + (size_t)b * L.nc * (2 * F + 4);
    for (int i = tid; i < 2 * F; i += 256) {
        float acc = 0.f;
        for (int c = 0; c < L.nc; ++c) acc += part[(size_t)c * (2 * F + 4) + i];
        vec[i / F][i % F] = acc;
        if (i < F) scratch[L.pooled + (size_t)b * F + i] = acc;
        else scratch[L.sumx + (size_t)b * F + (i - F)] = acc;
    }
    for (int i = tid; i < F; i += 256) vec[2][i] = vid[(size_t)b * F + i];
    __syncthreads();
    // block reductions (fixed order: per-thread stride, wave shuffle tree, 4 waves in order)
    auto block_sum = [&](float v) {
        v = wave_sum(v);
        if (lane == 0) sred[wave] = v;
        __syncthreads();
        const float r = ((sred[0] + sred[1]) + sred[2]) + sred[3];
        __syncthreads();
        return r;
    };
    auto block_max = [&](float v) {
        for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
        if (lane == 0) sred[wave] = v;
        __syncthreads();
        const float r = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
        __syncthreads();
        return r;
    };
    float ss = 0.f, mp = -__builtin_inff(), mv = -__builtin_inff();
    for (int i = tid; i < F; i += 256) { ss += vec[1][i] * vec[1][i]; mp = fmaxf(mp, vec[0][i]); mv = fmaxf(mv, vec[2][i]); }
    const float S2 = block_sum(ss);
    mp = block_max(mp); mv = block_max(mv);
    float zp = 0.f, zv = 0.f;
    for (int i = tid; i < F; i += 256) { zp += expf(vec[0][i] - mp); zv += expf(vec[2][i] - mv); }
    zp = block_sum(zp); zv = block_sum(zv);
    const float lzp = logf(zp);
    float ce = 0.f;
    for (int i = tid; i < F; i += 256) {
        const float p2 = expf(vec[2][i] - mv) / zv, logp1 = vec[0][i] - mp - lzp;
        ce -= p2 * logp1;
        scratch[L.dce + (size_t)b * F + i] = expf(logp1) - p2;        // d CE_b / d pooled_i (before the 1/(B F) of the mean)
    }
    ce = block_sum(ce);
    if (tid == 0) {
        float D = 0.f, cen = 0.f;
        for (int c = 0; c < L.nc; ++c) { D += part[(size_t)c * (2 * F + 4) + 2 * F]; cen += part[(size_t)c * (2 * F + 4) + 2 * F + 1]; }
        float *st = scratch + L.stats + (size_t)b * ST_N;
        st[ST_M] = m; st[ST_Z] = Z; st[ST_D] = D;
        st[ST_W2] = cen;                                       // norm penalty: sum of w^2
        st[ST_CEN] = entropy ? cen / (float)T : sqrtf(cen);    // per-video centering value
        st[ST_R] = (S2 - D) / ((float)T * (float)T);           // per-video repel value
        st[ST_N - 1] = ce / (float)F;                          // per-video distillation loss
    }
}

// losses[0..2] = batch means of (distillation, centering, repel)
__global__ void head_losses(const float *__restrict__ scratch, int B, int T, int F, float *__restrict__ losses) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const Layout L = layout(B, T, F);
    float a = 0.f, c = 0.f, r = 0.f;
    for (int b = 0; b < B; ++b) {
        const float *st = scratch + L.stats + (size_t)b * ST_N;
        a += st[ST_N - 1]; c += st[ST_CEN]; r += st[ST_R];
    }
    losses[0] = a / (float)B; losses[1] = c / (float)B; losses[2] = r / (float)B;
}

// ---- backward, stage 1: a_t = d_pooled . f_t and the chunk partial of R = sum_t w_t * d w_t ----
template <int NV>
__global__ __launch_bounds__(256) void head_bwd_dots(const float *__restrict__ feats, const float *__restrict__ scores,
                                                     const uint8_t *__restrict__ mask, int T, float inv_temp, int entropy,
                                                     float *__restrict__ scratch, const float *__restrict__ g, int B) {
    constexpr int F = 256 * NV;
    __shared__ float sred[4];
    const Layout L = layout(B, T, F);
    const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *st = scratch + L.stats + (size_t)b * ST_N;
    const float m = st[ST_M], Z = st[ST_Z];
    const uint8_t *mk = mask ? mask + (size_t)b * T : nullptr;
    const float gl = g[0] / ((float)B * (float)F), gc = g[1] / (float)B;
    f32x4 dp[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) dp[u] = *(const f32x4 *)(scratch + L.dce + (size_t)b * F + 4 * lane + 256 * u) * gl;
    float racc = 0.f;
    for (int t = ch * CHUNK + wave; t < min(T, (ch + 1) * CHUNK); t += 4) {
        float dot = 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const f32x4 f = *(const f32x4 *)(feats + ((size_t)b * T + t) * F + 4 * lane + 256 * u);
            dot += f[0] * dp[u][0] + f[1] * dp[u][1] + f[2] * dp[u][2] + f[3] * dp[u][3];
        }
        dot = wave_sum(dot);
        const bool valid = !mk || !mk[t];
        const float w = valid ? expf(scores[(size_t)b * T + t] * inv_temp - m) / Z : 0.f;
        float dw = dot;
        if (entropy) { if (valid) dw += gc / (float)T * (logf(w + EPS) + 1.0f); }
        else dw += gc * w / sqrtf(st[ST_W2]);
        if (lane == 0) { scratch[L.a + (size_t)b * T + t] = dw; racc += w * dw; }
    }
    if (lane == 0) sred[wave] = racc;
    __syncthreads();
    if (tid == 0) scratch[L.rpart + (size_t)b * L.nc + ch] = ((sred[0] + sred[1]) + sred[2]) + sred[3];
}

// ---- backward, stage 2: d_scores and d_feats ----
template <int NV>
__global__ __launch_bounds__(256) void head_bwd_final(const float *__restrict__ feats, const float *__restrict__ scores,
                                                      const uint8_t *__restrict__ mask, int T, float inv_temp,
                                                      const float *__restrict__ scratch, const float *__restrict__ g,
                                                      float *__restrict__ d_feats, float *__restrict__ d_scores, int B) {
    constexpr int F = 256 * NV;
    const Layout L = layout(B, T, F);
    const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *st = scratch + L.stats + (size_t)b * ST_N;
    const float m = st[ST_M], Z = st[ST_Z];
    const uint8_t *mk = mask ? mask + (size_t)b * T : nullptr;
    float R = 0.f;
    for (int c = 0; c < L.nc; ++c) R += scratch[L.rpart + (size_t)b * L.nc + c];
    const float gl = g[0] / ((float)B * (float)F), coef = 2.0f * g[2] / ((float)B * (float)T * (float)T);
    f32x4 dp[NV], S[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        dp[u] = *(const f32x4 *)(scratch + L.dce + (size_t)b * F + 4 * lane + 256 * u) * gl;
        S[u] = *(const f32x4 *)(scratch + L.sumx + (size_t)b * F + 4 * lane + 256 * u);
    }
    for (int t = ch * CHUNK + wave; t < min(T, (ch + 1) * CHUNK); t += 4) {
        const bool valid = !mk || !mk[t];
        const float w = valid ? expf(scores[(size_t)b * T + t] * inv_temp - m) / Z : 0.f;
        const float dw = scratch[L.a + (size_t)b * T + t];
        if (lane == 0) d_scores[(size_t)b * T + t] = w * (dw - R) * inv_temp;
        f32x4 f[NV], out[NV];
        float nn = 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            f[u] = *(const f32x4 *)(feats + ((size_t)b * T + t) * F + 4 * lane + 256 * u);
            nn += f[u][0] * f[u][0] + f[u][1] * f[u][1] + f[u][2] * f[u][2] + f[u][3] * f[u][3];
            out[u] = dp[u] * w;
        }
        nn = wave_sum(nn);
        if (valid && nn > 0.f) {
            // x^ = f / (n + eps):  d f = v / (n + eps) - f (f . v) / (n (n + eps)^2),  v = coef * (S - x^)
            const float n = sqrtf(nn), inv = 1.0f / (n + EPS);
            float fv = 0.f;
            f32x4 v[NV];
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                v[u] = (S[u] - f[u] * inv) * coef;
                fv += f[u][0] * v[u][0] + f[u][1] * v[u][1] + f[u][2] * v[u][2] + f[u][3] * v[u][3];
            }
            fv = wave_sum(fv);
            const float k2 = fv * inv * inv / n;
#pragma unroll
            for (int u = 0; u < NV; ++u) out[u] += v[u] * inv - f[u] * k2;
        }
#pragma unroll
        for (int u = 0; u < NV; ++u) *(f32x4 *)(d_feats + ((size_t)b * T + t) * F + 4 * lane + 256 * u) = out[u];
    }
}

}  // namespace

size_t vsp_head_scratch_floats(int B, int T, int F) { return layout(B, T, F).total; }

int vsp_head_forward(const float *feats, const float *scores, const uint8_t *mask, const float *vid, int B, int T, int F,
                     float inv_temp, int entropy_penalty, float *scratch, float *losses, hipStream_t st) {
    if (F != 256 && F != 512 && F != 768 && F != 1024) return -1;
    const Layout L = layout(B, T, F);
    const dim3 grid(L.nc, B);
#define VSP_CASE(NV_)                                                                                                    \
    case NV_:                                                                                                            \
        hipLaunchKernelGGL(head_partial<NV_>, grid, dim3(256), 0, st, feats, scores, mask, T, inv_temp, entropy_penalty, scratch, B); \
        hipLaunchKernelGGL(head_final<NV_>, dim3(B), dim3(256), 0, st, scores, mask, vid, T, inv_temp, entropy_penalty, scratch, B);   \
        break;
    switch (F / 256) { VSP_CASE(1) VSP_CASE(2) VSP_CASE(3) VSP_CASE(4) default: return -1; }
#undef VSP_CASE
    VSK_CHECK_LAUNCH();
    hipLaunchKernelGGL(head_losses, dim3(1), dim3(64), 0, st, scratch, B, T, F, losses);
    VSK_CHECK_LAUNCH();
    return 0;
}

int vsp_head_backward(const float *feats, const float *scores, const uint8_t *mask, const float *vid, int B, int T, int F,
                      float inv_temp, int entropy_penalty, const float *scratch, const float *g_losses, float *d_feats,
                      float *d_scores, hipStream_t st) {
    (void)vid;
    if (F != 256 && F != 512 && F != 768 && F != 1024) return -1;
    const Layout L = layout(B, T, F);
    const dim3 grid(L.nc, B);
    float *sc = const_cast<float *>(scratch);      // the a / rpart regions are backward-only scratch
#define VSP_CASE(NV_)                                                                                                    \
    case NV_:                                                                                                            \
        hipLaunchKernelGGL(head_bwd_dots<NV_>, grid, dim3(256), 0, st, feats, scores, mask, T, inv_temp, entropy_penalty, sc, g_losses, B); \
        hipLaunchKernelGGL(head_bwd_final<NV_>, grid, dim3(256), 0, st, feats, scores, mask, T, inv_temp, scratch, g_losses, d_feats, d_scores, B); \
        break;
    switch (F / 256) { VSP_CASE(1) VSP_CASE(2) VSP_CASE(3) VSP_CASE(4) default: return -1; }
#undef VSP_CASE
    VSK_CHECK_LAUNCH();
    return 0;
}
