// vs_train_kernels.hip — gfx950 kernels of the TRAINING path: everything of
// SimNet.forward in train mode and its backward that is not a plain NT GEMM (those reuse gemm_nt_128 of
// vs_kernels.hip: a dgrad is an NT GEMM against the transposed weight) and not attention (vs_train_attention.hip).
//
//   train_rows_fwd     z = dropout(a) + residual;  y = LayerNorm(z);  saves z and (mean, rstd); optional score head
//                      (reference simnet.py:107,110: norm(dropout(x1) + x); :42 final_layer)
//   ln_bwd_rows        LayerNorm backward per row + per-block partial d_gamma / d_beta; on the last layer the
//                      incoming gradient is d_hidden + d_scores * final_layer.weight
//   dropout_rows       elementwise dropout in place (PositionalEncoding.dropout simnet.py:237, MLP.dropout :181)
//   gate_bwd           ReLU + dropout backward: g = ffn > 0 ? g / (1 - p) : 0
//   head_rowdot        delta[b,h,t] = sum_c dO[t, h*dh + c] * O[t, h*dh + c]   (flash-attention backward)
//   weighted_colsum    partial sums of w[row] * Y[row, :] (final_layer weight / bias gradients)
//   wgrad_tn           dW[n,k] = sum_m dY[m,n] * X[m,k] on the fp32 matrix pipe, split over m, + column sums of dY
//   reduce_partials    deterministic sum of the per-split partials into the parameter-gradient tensors
//   transpose2d        W^T for the dgrad GEMMs
//   mse_mask_*         utils.mse_with_mask_loss forward / backward (reference utils.py:45-56)
//
// All reductions run in a fixed order: the training step is bitwise reproducible for a given seed.
#include "vs_train_device.h"
#include "vs_train_kernels.h"

namespace {

constexpr float LN_EPS = 1e-5f;      // nn.LayerNorm default (simnet.py:99-100)

// ------------------------------------------------------------------------------------------
// Row kernels: one wave per row, lane l owns columns 4l .. 4l+3 (+256 per extra vector, d <= 512)
// ------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void train_rows_fwd(
    const float *__restrict__ a, const float *__restrict__ res, const float *__restrict__ gamma,
    const float *__restrict__ beta, float *__restrict__ z, float *__restrict__ y, float *__restrict__ y_copy,
    float *__restrict__ stats, int M, int d, unsigned long long seed, unsigned site, float p,
    const float *__restrict__ score_w, const float *__restrict__ score_b, int num_classes,
    float *__restrict__ scores, int dn) {
    // dn: the LayerNorm width (== d unless the model is embedded in a wider shape: columns dn .. d-1 are identically zero)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const DropSite ds = drop_site(seed, site, p);
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        f32x4 v[NV];
        float s = 0.f;
        const unsigned rk = drop_rowkey(ds, (unsigned)row);
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = 4 * lane + 256 * u;
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < d) {
                f32x4 av = *(const f32x4 *)(a + (size_t)row * d + c);
                if (p > 0.f) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[e] = drop_keep(ds, rk, (unsigned)(c + e)) ? av[e] * ds.scale : 0.f;
                }
                const f32x4 rv = *(const f32x4 *)(res + (size_t)row * d + c);
                v[u] = av + rv;
                *(f32x4 *)(z + (size_t)row * d + c) = v[u];
                s += v[u][0] + v[u][1] + v[u][2] + v[u][3];
            }
        }
        const float mean = wave_sum(s) / (float)dn;
        float s2 = 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = 4 * lane + 256 * u;
            if (c < dn) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float t = v[u][e] - mean; s2 += t * t; }
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)dn + LN_EPS);
        if (lane == 0) { stats[2 * (size_t)row] = mean; stats[2 * (size_t)row + 1] = rstd; }
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = 4 * lane + 256 * u;
            if (c < d) {
                const f32x4 g = *(const f32x4 *)(gamma + c), b = *(const f32x4 *)(beta + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[u][e] = (v[u][e] - mean) * rstd * g[e] + b[e];
                *(f32x4 *)(y + (size_t)row * d + c) = v[u];
                if (y_copy) *(f32x4 *)(y_copy + (size_t)row * d + c) = v[u];
            }
        }
        if (score_w != nullptr) {
            for (int cls = 0; cls < num_classes; ++cls) {
                float dot = 0.f;
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const int c = 4 * lane + 256 * u;
                    if (c < d) {
                        const f32x4 w = *(const f32x4 *)(score_w + (size_t)cls * d + c);
                        dot += v[u][0] * w[0] + v[u][1] * w[1] + v[u][2] * w[2] + v[u][3] * w[3];
                    }
                }
                dot = wave_sum(dot);
                if (lane == 0) scores[(size_t)row * num_classes + cls] = dot + score_b[cls];
            }
        }
    }
}

// LayerNorm backward.  dy_row = dy[row] (or 0) + sum_cls dsc[row, cls] * score_w[cls]   (the latter on the last layer only)
//   xh = (z - mean) * rstd;  g = dy * gamma;  dz = rstd * (g - mean(g) - xh * mean(g * xh))
//   dbranch (optional) = dropout mask of the forward applied to dz (the gradient of the Linear that fed this norm)
//   part[blockIdx][0][:] += dy * xh (d_gamma),  part[blockIdx][1][:] += dy (d_beta)
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_rows(
    const float *__restrict__ dy, const float *__restrict__ dsc, const float *__restrict__ score_w, int num_classes,
    const float *__restrict__ z, const float *__restrict__ stats, const float *__restrict__ gamma,
    float *__restrict__ dz, float *__restrict__ dbranch, float *__restrict__ part, int M, int d,
    unsigned long long seed, unsigned site, float p, int dn) {
    // dn: the LayerNorm width (see train_rows_fwd); columns dn .. d-1 carry gamma = 0 and get a zero gradient
    __shared__ float red[4][2][256 * NV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const DropSite ds = drop_site(seed, site, p);
    f32x4 ag[NV], ab[NV], gm[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        ag[u] = f32x4{0.f, 0.f, 0.f, 0.f}; ab[u] = ag[u]; gm[u] = ag[u];
        const int c = 4 * lane + 256 * u;
        if (c < d) gm[u] = *(const f32x4 *)(gamma + c);
    }
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float mean = stats[2 * (size_t)row], rstd = stats[2 * (size_t)row + 1];
        f32x4 g[NV], xh[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = 4 * lane + 256 * u;
            g[u] = f32x4{0.f, 0.f, 0.f, 0.f}; xh[u] = g[u];
            if (c < d) {
                f32x4 dv = dy ? *(const f32x4 *)(dy + (size_t)row * d + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (dsc != nullptr) {
                    for (int cls = 0; cls < num_classes; ++cls) {
                        const float w = dsc[(size_t)row * num_classes + cls];
                        const f32x4 sw = *(const f32x4 *)(score_w + (size_t)cls * d + c);
                        dv += sw * w;
                    }
                }
                const f32x4 zv = *(const f32x4 *)(z + (size_t)row * d + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[u][e] = (zv[e] - mean) * rstd;
                    g[u][e] = dv[e] * gm[u][e];
                    s1 += g[u][e];
                    s2 += g[u][e] * xh[u][e];
                    ag[u][e] += dv[e] * xh[u][e];
                    ab[u][e] += dv[e];
                }
            }
        }
        const float m1 = wave_sum(s1) / (float)dn, m2 = wave_sum(s2) / (float)dn;
        const unsigned rk = drop_rowkey(ds, (unsigned)row);
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = 4 * lane + 256 * u;
            if (c < d) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = c < dn ? rstd * (g[u][e] - m1 - xh[u][e] * m2) : 0.f;
                *(f32x4 *)(dz + (size_t)row * d + c) = o;
                if (dbranch != nullptr) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = drop_keep(ds, rk, (unsigned)(c + e)) ? o[e] * ds.scale : 0.f;
                    *(f32x4 *)(dbranch + (size_t)row * d + c) = o;
                }
            }
        }
    }
    // block partial of d_gamma / d_beta: waves summed in wave order
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int c = 4 * lane + 256 * u;
        if (c < d) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { red[wave][0][c + e] = ag[u][e]; red[wave][1][c + e] = ab[u][e]; }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * d; i += 256) {
        const int which = i / d, c = i - which * d;
        part[((size_t)blockIdx.x * 2 + which) * d + c] =
            ((red[0][which][c] + red[1][which][c]) + red[2][which][c]) + red[3][which][c];
    }
}

// x[row, c] = keep(row, c) ? x * scale : 0   (in place; cols = row length)
__global__ __launch_bounds__(256) void dropout_rows(float *__restrict__ x, int M, int cols, unsigned long long seed,
                                                    unsigned site, float p) {
    const DropSite ds = drop_site(seed, site, p);
    const int c4n = cols / 4;
    const size_t total = (size_t)M * c4n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const unsigned row = (unsigned)(i / c4n), c = (unsigned)(i % c4n) * 4;
        const unsigned rk = drop_rowkey(ds, row);
        f32x4 v = *(f32x4 *)(x + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = drop_keep(ds, rk, c + e) ? v[e] * ds.scale : 0.f;
        *(f32x4 *)(x + i * 4) = v;
    }
}

// g = act > 0 ? g * scale : 0   (act = dropout(relu(fc1)) as saved by the forward: > 0 <=> relu passed AND kept)
__global__ __launch_bounds__(256) void gate_bwd(float *__restrict__ g, const float *__restrict__ act, size_t n4, float scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 v = *(f32x4 *)(g + i * 4);
        const f32x4 a = *(const f32x4 *)(act + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = a[e] > 0.f ? v[e] * scale : 0.f;
        *(f32x4 *)(g + i * 4) = v;
    }
}

// delta[(b*H + h)*T + t] = sum_c dO[m, h*dh + c] * O[m, h*dh + c],  m = b*T + t
// DO16 1: dO is stored as bf16 (the bf16 training mode's out-projection input gradient); 2: dO is fp32 but enters the dot
// rounded to bf16 - the value the bf16 attention backward multiplies (dP = dO V^T), so that dS = P (dP - delta) subtracts
// like from like, and the two storage forms of that mode agree bit for bit
template <int NV, int DO16 = 0, int F16 = 0>       // F16: the 16-bit type is IEEE f16 (fp16 training mode)
__global__ __launch_bounds__(256) void head_rowdot(const float *__restrict__ dO, const float *__restrict__ O,
                                                   float *__restrict__ delta, int M, int T, int H, int dh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, d = H * dh, gl = dh / 4;   // gl lanes per head
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const int b = row / T, t = row - b * T;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = 4 * lane + 256 * u;
            float s = 0.f;
            if (c < d) {
                f32x4 x;
                if constexpr (DO16 == 2) {
                    const f32x4 xf = *(const f32x4 *)(dO + (size_t)row * d + c);
                    const f32x2 a = unpack_lp<F16>(pack_lp<F16>(xf[0], xf[1])), b = unpack_lp<F16>(pack_lp<F16>(xf[2], xf[3]));
                    x = f32x4{a[0], a[1], b[0], b[1]};
                } else if constexpr (DO16 == 1) {
                    const u32x2 xb = *(const u32x2 *)((const unsigned short *)dO + (size_t)row * d + c);
                    const f32x2 a = unpack_lp<F16>(xb[0]), b = unpack_lp<F16>(xb[1]);
                    x = f32x4{a[0], a[1], b[0], b[1]};
                } else
                    x = *(const f32x4 *)(dO + (size_t)row * d + c);
                const f32x4 o = *(const f32x4 *)(O + (size_t)row * d + c);
                s = x[0] * o[0] + x[1] * o[1] + x[2] * o[2] + x[3] * o[3];
            }
            for (int off = 1; off < gl; off <<= 1) s += __shfl_xor(s, off);
            if (c < d && (lane % gl) == 0) delta[((size_t)b * H + c / dh) * T + t] = s;
        }
    }
}

// part[blockIdx][0:d] = sum_rows w[row*ws] * Y[row, :];  part[gridDim.x * d + blockIdx] = sum_rows w[row*ws]
template <int NV>
__global__ __launch_bounds__(256) void weighted_colsum(const float *__restrict__ w, int ws, const float *__restrict__ Y,
                                                       float *__restrict__ part, int M, int d) {
    __shared__ float red[4][256 * NV + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[NV];
    float sw = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float wv = w[(size_t)row * ws];
        sw += wv;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = 4 * lane + 256 * u;
            if (c < d) acc[u] += *(const f32x4 *)(Y + (size_t)row * d + c) * wv;
        }
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int c = 4 * lane + 256 * u;
        if (c < d) {
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave][c + e] = acc[u][e];
        }
    }
    if (lane == 0) red[wave][d] = sw;
    __syncthreads();
    for (int i = threadIdx.x; i <= d; i += 256) {
        const float tot = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
        if (i < d) part[(size_t)blockIdx.x * d + i] = tot;
        else part[(size_t)gridDim.x * d + blockIdx.x] = tot;
    }
}

// ------------------------------------------------------------------------------------------
// Weight gradient:  dW[n, k] = sum_m dY[m, n] * X[m, k]   ("TN": both operands are contracted over their ROW index)
//   This orientation is the natural one for v_mfma_f32_32x32x2_f32: the A operand wants A[i][kk] with i on the lane,
//   and consecutive lanes read consecutive n of one dY row - straight, conflict-free LDS rows, no transposition.
//   Block = 4 waves (2 x 2), tile 128 (n) x 128 (k); a wave owns 64 x 64 as 2 x 2 MFMA tiles with the INTERLEAVED
//   column map  n = n0 + 64*wn + 2*i + ib  (i = MFMA row, ib = which of the two tiles): one ds_read_b64 at
//   [m][.. + 2r] then feeds both tiles, and the output store is a float2 per lane (256 contiguous bytes per row).
//   The contraction runs over the frames m, which is the LONG dimension (65 536 at the bench size) while the output is
//   at most 1024 x 1024: the grid is (tiles, splits) and every split writes its own partial; reduce_partials sums
//   them in split order (deterministic; no atomics).  Blocks with tile_k == 0 also produce the column sums of dY
//   (the bias gradient) from the values they stage anyway.
//   16 rows per stage, double-buffered LDS, next stage's global loads in flight under the current stage's 32 MFMAs.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wgrad_tn(const float *__restrict__ dY, int ldy, const float *__restrict__ X, int ldx,
                                                float *__restrict__ partW, float *__restrict__ partB, int M, int N, int K,
                                                int rows_per_split) {
    constexpr int BR = 16, LDP = 128 + 8;
    __shared__ __attribute__((aligned(16))) float Ys[2][BR][LDP];
    __shared__ __attribute__((aligned(16))) float Xs[2][BR][LDP];
    const int tiles_k = (K + 127) / 128;
    const int tile_n = blockIdx.x / tiles_k, tile_k = blockIdx.x - tile_n * tiles_k;
    const int n0 = tile_n * 128, k0 = tile_k * 128;
    const int split = blockIdx.y;
    const int m_begin = split * rows_per_split, m_end = min(M, m_begin + rows_per_split);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;
    const int srow = tid >> 5, sc4 = (tid & 31) * 4;
    const bool n_ok = n0 + sc4 < N, k_ok = k0 + sc4 < K;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    f32x4 py[2], px[2];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto gload = [&](int m0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + srow + 8 * i;
            py[i] = (m < m_end && n_ok) ? *(const f32x4 *)(dY + (size_t)m * ldy + n0 + sc4) : zero4;
            px[i] = (m < m_end && k_ok) ? *(const f32x4 *)(X + (size_t)m * ldx + k0 + sc4) : zero4;
        }
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(f32x4 *)&Ys[buf][srow + 8 * i][sc4] = py[i];
            *(f32x4 *)&Xs[buf][srow + 8 * i][sc4] = px[i];
            bsum += py[i];
        }
    };
    if (m_begin < m_end) {
        gload(m_begin);
        stage(0);
        __syncthreads();
        int buf = 0;
        for (int m0 = m_begin; m0 < m_end; m0 += BR) {
            const bool more = m0 + BR < m_end;
            if (more) gload(m0 + BR);
#pragma unroll
            for (int s = 0; s < BR / 2; ++s) {
                const f32x2 a2 = *(const f32x2 *)&Ys[buf][2 * s + h][64 * wn + 2 * r];
                const f32x2 b2 = *(const f32x2 *)&Xs[buf][2 * s + h][64 * wk + 2 * r];
                acc[0][0] = MFMA32(a2[0], b2[0], acc[0][0]);
                acc[0][1] = MFMA32(a2[0], b2[1], acc[0][1]);
                acc[1][0] = MFMA32(a2[1], b2[0], acc[1][0]);
                acc[1][1] = MFMA32(a2[1], b2[1], acc[1][1]);
            }
            if (more) stage(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    // partial tile: n = n0 + 64*wn + 2*acc_row(t,h) + ib,  k = k0 + 64*wk + 2*r + {0,1}
    float *pw = partW + (size_t)split * N * K;
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int n = n0 + 64 * wn + 2 * acc_row(t, h) + ib, k = k0 + 64 * wk + 2 * r;
            if (n < N && k < K) *(f32x2 *)(pw + (size_t)n * K + k) = f32x2{acc[ib][0][t], acc[ib][1][t]};
        }
    if (partB != nullptr && tile_k == 0) {
        // column sums of the staged dY values: 8 row groups (srow) per column quad, summed in srow order
        __syncthreads();
        float(*red)[LDP] = Ys[0];
        *(f32x4 *)&red[srow][sc4] = bsum;
        __syncthreads();
        if (tid < 128 && n0 + tid < N) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += red[g][tid];
            partB[(size_t)split * N + n0 + tid] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------
// The same weight gradient on the bf16 matrix pipe (low-precision training, VS_TRAIN_FLAG_BF16_LINEAR): dY and X are
// rounded to bf16 on their way into LDS (v_cvt_pk_bf16_f32), products accumulate in fp32, the column sums of dY (bias
// gradient) stay fp32 sums of the unrounded values.  v_mfma_f32_32x32x16_bf16 wants 8 consecutive m per lane for
// ONE n (A) / ONE k (B): a column of the row-major [m][256] tile - read by ds_read_b64_tr_b16 (4 rows x 16 columns per
// 16 lanes, transposed by the LDS).  Rows are 512 B; 16-byte chunk c of row m sits at chunk c ^ ((m & 3) << 2), which
// spreads the 4 rows of a transposed read over the 4 quarters of the banks (conflict-free) and keeps a staging store's
// 16 lanes on one row's contiguous bytes.  32 rows per stage (two 16-m steps), double-buffered; same partial layout,
// split-K reduction and determinism as wgrad_tn.
// Tile: 256 (n) x 256 (k) per 8-wave block (4 x 2 waves of 64 x 128), one block per CU.  The first version used wgrad_tn's
// 128 x 128 tiles: every (n-tile, k-tile) block streams its column slab of dY and of X over all M rows, so the operands
// were read tiles_k + tiles_n times from L2 - 32 flop per byte, ~100 us per call at 65 536 rows whatever the shape, the
// L2 -> CU rate (~10 TB/s) and not HBM being the bound.  256 x 256 tiles halve those bytes (64 flop/B).
// ------------------------------------------------------------------------------------------
// Y16 / X16: that operand is already bf16 in memory (the MLP hidden tensor and its gradient in the bf16 training mode).
template <bool Y16, bool X16, int F16 = 0>       // F16: operands (rounded here or stored) are IEEE f16 (fp16 training mode)
__global__ __launch_bounds__(512, 2) void wgrad_tn_bf16(const float *__restrict__ dY, int ldy, const float *__restrict__ X, int ldx,
                                                        float *__restrict__ partW, float *__restrict__ partB, int M, int N, int K,
                                                        int rows_per_split) {
    constexpr int BR = 32, TW = 256, ROWB = 2 * TW;          // rows per stage, tile width (columns of dY / of X), LDS row bytes
    typedef unsigned short h16;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) h16 Ys[2][BR * TW];
    __shared__ __attribute__((aligned(16))) h16 Xs[2][BR * TW];
    __shared__ float red[8][TW + 4];
    const int tiles_k = (K + TW - 1) / TW;
    // XCD-aware block -> (tile, split) map (round 4).  The tiles of one row split read the SAME rows of one operand (dY for the
    // k-tiles of an n-tile, X for the n-tiles of a k-tile); workgroups go to the 8 XCDs round-robin in launch order, so with
    // the plain map (tile fastest) the 4 tiles of a split land on 4 different XCDs and each L2 fetches the shared rows for
    // itself.  Here the blocks of ONE XCD (linear index congruent mod 8) walk tile-fastest through their own splits, so the
    // shared rows are fetched once per XCD.  Measured: fc2 / fc1 weight gradients 75.8 -> 73.6 us, QKV 78.5 -> 75.8 - the
    // re-reads were being served by the Infinity Cache already; the kernel is bound by its 32-row stages (one barrier per 16
    // MFMAs of a wave), not by that traffic.
    int tile = blockIdx.x, split = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const int L = blockIdx.x + gridDim.x * blockIdx.y, j = L >> 3;
        tile = j % (int)gridDim.x;
        split = (L & 7) + 8 * (j / (int)gridDim.x);
    }
    const int tile_n = tile / tiles_k, tile_k = tile - tile_n * tiles_k;
    const int n0 = tile_n * TW, k0 = tile_k * TW;
    const int m_begin = split * rows_per_split, m_end = min(M, m_begin + rows_per_split);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;                 // 4 waves along n (64 each), 2 along k (128 each)
    const int srow = tid >> 6, sc4 = (tid & 63) * 4;         // staging: 8 rows x 64 float4 per pass, 4 passes
    const bool n_ok = n0 + sc4 < N, k_ok = k0 + sc4 < K;
    // staging store: 4 columns (8 B) of row m = srow + 8 i; physical chunk = ((tid & 63) >> 1) ^ ((m & 3) << 2)
    const int st_off = srow * ROWB + ((((tid & 63) >> 1) ^ ((srow & 3) << 2)) << 4) + 8 * (tid & 1);
    // transposed read: lane l of a 16-lane group gives row q = (l & 15) >> 2, columns 4 p .. 4 p + 3 (p = l & 3)
    const int q = (lane & 15) >> 2, pcol = lane & 3, cl = 2 * ((lane >> 4) & 1) + (pcol >> 1);
    int a_off[2], b_off[4];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) a_off[ib] = (8 * h + q) * ROWB + ((cl + 4 * ((2 * wn + ib) ^ q)) << 4) + 8 * (pcol & 1);
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) b_off[jb] = (8 * h + q) * ROWB + ((cl + 4 * ((4 * wk + jb) ^ q)) << 4) + 8 * (pcol & 1);

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    f32x4 py[4], px[4];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // a bf16-stored operand: 4 elements = 8 bytes, parked in the first two dwords of its float4 slot
    auto ld16 = [](const float *base, size_t off) __attribute__((always_inline)) {
        const f32x2 v = *(const f32x2 *)((const h16 *)base + off);
        return f32x4{v[0], v[1], 0.f, 0.f};
    };
    auto gload = [&](int m0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + srow + 8 * i;
            if constexpr (Y16) py[i] = (m < m_end && n_ok) ? ld16(dY, (size_t)m * ldy + n0 + sc4) : zero4;
            else py[i] = (m < m_end && n_ok) ? *(const f32x4 *)(dY + (size_t)m * ldy + n0 + sc4) : zero4;
            if constexpr (X16) px[i] = (m < m_end && k_ok) ? ld16(X, (size_t)m * ldx + k0 + sc4) : zero4;
            else px[i] = (m < m_end && k_ok) ? *(const f32x4 *)(X + (size_t)m * ldx + k0 + sc4) : zero4;
        }
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x2 uy, ux;
            if constexpr (Y16) {
                const unsigned y0 = f32_bits(py[i][0]), y1 = f32_bits(py[i][1]);
                uy[0] = y0; uy[1] = y1;
                // the bias gradient sums the stored (bf16) values
                const f32x2 ya = unpack_lp<F16>(y0), yb = unpack_lp<F16>(y1);
                bsum += f32x4{ya[0], ya[1], yb[0], yb[1]};
            } else {
                uy[0] = pack_lp<F16>(py[i][0], py[i][1]); uy[1] = pack_lp<F16>(py[i][2], py[i][3]);
                bsum += py[i];
            }
            if constexpr (X16) { ux[0] = f32_bits(px[i][0]); ux[1] = f32_bits(px[i][1]); }
            else { ux[0] = pack_lp<F16>(px[i][0], px[i][1]); ux[1] = pack_lp<F16>(px[i][2], px[i][3]); }
            *(u32x2 *)((char *)&Ys[buf][0] + st_off + 8 * i * ROWB) = uy;
            *(u32x2 *)((char *)&Xs[buf][0] + st_off + 8 * i * ROWB) = ux;
        }
    };
    auto frag = [&](const h16 *tile, int off) __attribute__((always_inline)) -> u32x4 {
        const char *pb = (const char *)tile + off;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)pb);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(pb + 4 * ROWB));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        const u32x4 v = {l2[0], l2[1], h2[0], h2[1]};
        return v;
    };
    if (m_begin < m_end) {
        gload(m_begin);
        stage(0);
        __syncthreads();
        int buf = 0;
        for (int m0 = m_begin; m0 < m_end; m0 += BR) {
            const bool more = m0 + BR < m_end;
            if (more) gload(m0 + BR);
#pragma unroll
            for (int s = 0; s < BR / 16; ++s) {
                u32x4 a[2], b[4];
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) a[ib] = frag(Ys[buf], a_off[ib] + s * 16 * ROWB);
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) b[jb] = frag(Xs[buf], b_off[jb] + s * 16 * ROWB);
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    acc[0][jb] = mfma_lp<F16>(a[0], b[jb], acc[0][jb]);
                    acc[1][jb] = mfma_lp<F16>(a[1], b[jb], acc[1][jb]);
                }
            }
            if (more) stage(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    // partial tile: n = n0 + 64 wn + 32 ib + acc_row(t, h),  k = k0 + 128 wk + 32 jb + r
    float *pw = partW + (size_t)split * N * K;
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int n = n0 + 64 * wn + 32 * ib + acc_row(t, h), k = k0 + 128 * wk + 32 * jb + r;
                if (n < N && k < K) pw[(size_t)n * K + k] = acc[ib][jb][t];
            }
    if (partB != nullptr && tile_k == 0) {
        *(f32x4 *)&red[srow][sc4] = bsum;
        __syncthreads();
        if (tid < TW && n0 + tid < N) {
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) sum += red[g][tid];
            partB[(size_t)split * N + n0 + tid] = sum;
        }
    }
}

// out[row, c] = sum_s part[s][row][c], rows dealt to up to three destination tensors of rows_per_dest rows each
// (q / k / v weight gradients come out of ONE [3d, d] product).  Fixed summation order (bitwise reproducible):
// thread group g of 16 sums s = g, g + 16, ... in ascending order (4 independent loads in flight), then the 16 group
// sums are added in group order.  Block = 16 groups x 16 vectors of VEC floats; cols % VEC == 0.
template <int VEC>
__global__ __launch_bounds__(256) void reduce_partials(const float *__restrict__ part, int S, int rows, int cols,
                                                       float *d0, float *d1, float *d2, int rows_per_dest) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    __shared__ float red[16][16 * VEC];
    const size_t per = (size_t)rows * cols, nvec = per / VEC;
    const int g = threadIdx.x >> 4, v = threadIdx.x & 15;
    const size_t i = (size_t)blockIdx.x * 16 + v;
    float acc[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
    if (i < nvec) {
        const float *p0 = part + i * VEC;
        int s = g;
        for (; s + 48 < S; s += 64) {
            float t[4][VEC];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if constexpr (VEC == 1) t[u][0] = p0[(size_t)(s + 16 * u) * per];
                else {
                    const vec_t w = *(const vec_t *)(p0 + (size_t)(s + 16 * u) * per);
#pragma unroll
                    for (int q = 0; q < VEC; ++q) t[u][q] = w[q];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[q] += t[u][q];
        }
        for (; s < S; s += 16) {
            if constexpr (VEC == 1) acc[0] += p0[(size_t)s * per];
            else {
                const vec_t w = *(const vec_t *)(p0 + (size_t)s * per);
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[q] += w[q];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < VEC; ++q) red[g][v * VEC + q] = acc[q];
    __syncthreads();
    if (g == 0 && i < nvec) {
        const size_t e = i * VEC;
        const int row = (int)(e / cols), c = (int)(e - (size_t)row * cols);
        const int which = row / rows_per_dest;
        float *dst = (which == 0 ? d0 : which == 1 ? d1 : d2) + (size_t)(row - which * rows_per_dest) * cols + c;
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            float tot = red[0][v * VEC + q];
#pragma unroll
            for (int gg = 1; gg < 16; ++gg) tot += red[gg][v * VEC + q];
            dst[q] = tot;
        }
    }
}

// One launch for a wgrad's two reductions: blocks [0, nb_w) sum the weight partials (float4), the rest the bias
// partials (scalars) - at small batch the reductions are launch-latency bound (5 us each, 17 pairs per step).
__global__ __launch_bounds__(256) void reduce_wgrad(const float *__restrict__ partW, const float *__restrict__ partB, int S,
                                                    int N, int K, float *w0, float *w1, float *w2, float *b0, float *b1,
                                                    float *b2, int rows_per_dest, int nb_w) {
    __shared__ float red[16][64];
    const int g = threadIdx.x >> 4, v = threadIdx.x & 15;
    const bool is_w = (int)blockIdx.x < nb_w;
    const int VEC = is_w ? 4 : 1;
    const size_t per = is_w ? (size_t)N * K : (size_t)N, nvec = per / VEC;
    const size_t i = (size_t)(is_w ? blockIdx.x : blockIdx.x - nb_w) * 16 + v;
    const float *part = is_w ? partW : partB;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < nvec) {
        for (int s = g; s < S; s += 16) {
            if (is_w) {
                const f32x4 w = *(const f32x4 *)(part + (size_t)s * per + i * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += w[q];
            } else {
                acc[0] += part[(size_t)s * per + i];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[g][v * 4 + q] = acc[q];
    __syncthreads();
    if (g == 0 && i < nvec) {
        const int cols = is_w ? K : 1;
        const size_t e = i * VEC;
        const int row = (int)(e / cols), c = (int)(e - (size_t)row * cols);
        const int which = row / rows_per_dest;
        float *d0 = is_w ? w0 : b0, *d1 = is_w ? w1 : b1, *d2 = is_w ? w2 : b2;
        float *dst = (which == 0 ? d0 : which == 1 ? d1 : d2) + (size_t)(row - which * rows_per_dest) * cols + c;
        for (int q = 0; q < VEC; ++q) {
            float tot = red[0][v * 4 + q];
#pragma unroll
            for (int gg = 1; gg < 16; ++gg) tot += red[gg][v * 4 + q];
            dst[q] = tot;
        }
    }
}

// out[c, r] = in[r, c]   (32 x 32 tiles through LDS; grid (cols/32, rows/32), block 256)
__global__ __launch_bounds__(256) void transpose2d(const float *__restrict__ in, float *__restrict__ out, int rows, int cols) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = r0 + ty + 8 * i, cc = c0 + tx;
        if (rr < rows && cc < cols) tile[ty + 8 * i][tx] = in[(size_t)rr * cols + cc];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cc = c0 + ty + 8 * i, rr = r0 + tx;
        if (rr < rows && cc < cols) out[(size_t)cc * rows + rr] = tile[tx][ty + 8 * i];
    }
}

// the same for a batch of matrices: blockIdx.y = job, blockIdx.x = the job's 32 x 32 tile (blocks past its last tile exit)
__global__ __launch_bounds__(256) void transpose2d_batch(VskMatJobs jobs) {
    __shared__ float tile[32][33];
    const int job = blockIdx.y;
    const float *__restrict__ in = jobs.in[job];
    float *__restrict__ out = jobs.out[job];
    const int rows = jobs.rows[job], cols = jobs.cols[job];
    const int tcols = (cols + 31) / 32, ntiles = tcols * ((rows + 31) / 32);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        const int c0 = (t % tcols) * 32, r0 = (t / tcols) * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = r0 + ty + 8 * i, cc = c0 + tx;
            if (rr < rows && cc < cols) tile[ty + 8 * i][tx] = in[(size_t)rr * cols + cc];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cc = c0 + ty + 8 * i, rr = r0 + tx;
            if (rr < rows && cc < cols) out[(size_t)cc * rows + rr] = tile[tx][ty + 8 * i];
        }
        __syncthreads();
    }
}

// ---- utils.mse_with_mask_loss (reference utils.py:45-56): mean (or sum) over ALL B*T entries of
//      ((output - target) * scale)^2, scale = 0 on masked frames ----
__global__ __launch_bounds__(256) void mse_mask_partial(const float *__restrict__ out, const float *__restrict__ tgt,
                                                        const unsigned char *__restrict__ mask, int n, float *__restrict__ part) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float sc = (mask && mask[i]) ? 0.f : 1.f;
        const float dlt = out[i] * sc - tgt[i] * sc;
        s += dlt * dlt;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}
__global__ void mse_mask_final(const float *__restrict__ part, int nblk, float inv_n, float *__restrict__ loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < nblk; ++i) s += part[i];
        loss[0] = s * inv_n;
    }
}
// d_out = gout * 2 * scale * (out*scale - tgt*scale) * inv_n
__global__ __launch_bounds__(256) void mse_mask_bwd(const float *__restrict__ out, const float *__restrict__ tgt,
                                                    const unsigned char *__restrict__ mask, const float *__restrict__ gout,
                                                    int n, float inv_n, float *__restrict__ dout) {
    const float g = gout[0] * 2.0f * inv_n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float sc = (mask && mask[i]) ? 0.f : 1.f;
        dout[i] = g * sc * (out[i] * sc - tgt[i] * sc);
    }
}

int row_grid(int M) { const int b = (M + 3) / 4; return b < 512 ? (b < 1 ? 1 : b) : 512; }      // blocks of the partial-sum row kernels

}  // namespace

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
// rows of up to 1024 columns: NV = float4 per lane (256 columns each)
#define VST_NV_DISPATCH(d_, KERN_, ...)                                              \
    do {                                                                             \
        switch (((d_) + 255) / 256) {                                                \
            case 1: hipLaunchKernelGGL(KERN_<1>, __VA_ARGS__); break;                \
            case 2: hipLaunchKernelGGL(KERN_<2>, __VA_ARGS__); break;                \
            case 3: hipLaunchKernelGGL(KERN_<3>, __VA_ARGS__); break;                \
            default: hipLaunchKernelGGL(KERN_<4>, __VA_ARGS__); break;               \
        }                                                                            \
    } while (0)

int vst_rows_fwd(const float *a, const float *res, const float *gamma, const float *beta, float *z, float *y,
                 float *y_copy, float *stats, int M, int d, unsigned long long seed, unsigned site, float p,
                 const float *score_w, const float *score_b, int num_classes, float *scores, hipStream_t st, int dn) {
    if (d % 4 || d > 1024) return -1;
    if (dn <= 0) dn = d;
    if (dn % 4 || dn > d) return -1;
    const dim3 grid((M + 3) / 4 < 4096 ? (M + 3) / 4 : 4096);
    VST_NV_DISPATCH(d, train_rows_fwd, grid, dim3(256), 0, st, a, res, gamma, beta, z, y, y_copy, stats, M, d, seed, site, p,
                    score_w, score_b, num_classes, scores, dn);
    VSK_CHECK_LAUNCH();
    return 0;
}

int vst_ln_bwd_blocks(int M) { return row_grid(M); }

int vst_ln_bwd(const float *dy, const float *dsc, const float *score_w, int num_classes, const float *z,
               const float *stats, const float *gamma, float *dz, float *dbranch, float *part, int M, int d,
               unsigned long long seed, unsigned site, float p, hipStream_t st, int dn) {
    if (d % 4 || d > 1024) return -1;
    if (dn <= 0) dn = d;
    if (dn % 4 || dn > d) return -1;
    const dim3 grid(row_grid(M));
    VST_NV_DISPATCH(d, ln_bwd_rows, grid, dim3(256), 0, st, dy, dsc, score_w, num_classes, z, stats, gamma, dz, dbranch, part,
                    M, d, seed, site, p, dn);
    VSK_CHECK_LAUNCH();
    return 0;
}

int vst_dropout_rows(float *x, int M, int cols, unsigned long long seed, unsigned site, float p, hipStream_t st) {
    if (cols % 4) return -1;
    const size_t total = (size_t)M * (cols / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(dropout_rows, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, st, x, M, cols, seed, site, p);
    VSK_CHECK_LAUNCH();
    return 0;
}

int vst_gate_bwd(float *g, const float *act, size_t n, float scale, hipStream_t st) {
    if (n % 4) return -1;
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(gate_bwd, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, st, g, act, n4, scale);
    VSK_CHECK_LAUNCH();
    return 0;
}

int vst_head_rowdot(const float *dO, const float *O, float *delta, int M, int T, int H, int dh, hipStream_t st, int do16) {
    const int d = H * dh;
    if (d > 1024 || (dh != 32 && dh != 64 && dh != 128 && dh != 256)) return -1;      // (256: one head per 256-column vector, the 64-lane sum)
    const dim3 grid((M + 3) / 4 < 4096 ? (M + 3) / 4 : 4096);
#define VST_HRD(MODE_, F_)                                                                                                   \
    switch ((d + 255) / 256) {                                                                                               \
        case 1: hipLaunchKernelGGL((head_rowdot<1, MODE_, F_>), grid, dim3(256), 0, st, dO, O, delta, M, T, H, dh); break;   \
        case 2: hipLaunchKernelGGL((head_rowdot<2, MODE_, F_>), grid, dim3(256), 0, st, dO, O, delta, M, T, H, dh); break;   \
        case 3: hipLaunchKernelGGL((head_rowdot<3, MODE_, F_>), grid, dim3(256), 0, st, dO, O, delta, M, T, H, dh); break;   \
        default: hipLaunchKernelGGL((head_rowdot<4, MODE_, F_>), grid, dim3(256), 0, st, dO, O, delta, M, T, H, dh); break;  \
    }
    const bool f16 = (do16 & VSK_F16) != 0;        // the 16-bit type of dO (1: stored, 2: rounded on the way in) is f16
    do16 &= ~VSK_F16;
    if (do16 == 1) { if (f16) { VST_HRD(1, 1) } else { VST_HRD(1, 0) } }
    else if (do16 == 2) { if (f16) { VST_HRD(2, 1) } else { VST_HRD(2, 0) } }
    else
    VST_NV_DISPATCH(d, head_rowdot, grid, dim3(256), 0, st, dO, O, delta, M, T, H, dh);
#undef VST_HRD
    VSK_CHECK_LAUNCH();
    return 0;
}

int vst_weighted_colsum(const float *w, int ws, const float *Y, float *part, int M, int d, hipStream_t st) {
    if (d % 4 || d > 1024) return -1;
    const dim3 grid(row_grid(M));
    VST_NV_DISPATCH(d, weighted_colsum, grid, dim3(256), 0, st, w, ws, Y, part, M, d);
    VSK_CHECK_LAUNCH();
    return 0;
}

// splits for the wgrad of an [N, K] weight over M rows: enough blocks for ~2 per CU, at least 64 rows per split
int vst_wgrad_splits(int M, int N, int K) {
    const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
    // the split count sizes the caller's workspace (vs_train_workspace_bytes) AND the launch, possibly queried with
    // different devices current: it must not depend on the device.  256 = the CUs of an MI355X (the only target).
    constexpr int cus = 256;
    int S = (2 * cus + tiles - 1) / tiles;
    const int maxS = (M + 63) / 64;
    if (S > maxS) S = maxS;
    return S < 1 ? 1 : S;
}

// the bf16 kernel's: 256 x 256 tiles on 8-wave blocks, one per CU
static int wgrad_splits_bf16(int M, int N, int K) {
    const int tiles = ((N + 255) / 256) * ((K + 255) / 256);
    constexpr int cus = 256;
    int S = (cus + tiles - 1) / tiles;
    const int maxS = (M + 63) / 64;
    if (S > maxS) S = maxS;
    return S < 1 ? 1 : S;
}

size_t vst_wgrad_workspace_floats(int M, int N, int K) {       // room for either kernel's partial tiles
    const size_t S0 = (size_t)vst_wgrad_splits(M, N, K), S1 = (size_t)wgrad_splits_bf16(M, N, K);
    return (S0 > S1 ? S0 : S1) * ((size_t)N * K + N);
}

int vst_wgrad(const float *dY, int ldy, const float *X, int ldx, int M, int N, int K, float *dW0, float *dW1, float *dW2,
              float *db0, float *db1, float *db2, int rows_per_dest, float *work, hipStream_t st, int prec) {
    if (N % 4 || K % 4 || ldy % 4 || ldx % 4) return -1;
    const int S = (prec & 1) ? wgrad_splits_bf16(M, N, K) : vst_wgrad_splits(M, N, K);
    int rps = (M + S - 1) / S;
    rps = (rps + 31) / 32 * 32;
    float *partW = work, *partB = work + (size_t)S * N * K;
    const dim3 grid(((N + 127) / 128) * ((K + 127) / 128), S);
    if (prec & 1) {       // + VST_WGRAD_Y16 / VST_WGRAD_X16: that operand is stored as bf16 (ld counted in elements)
        const dim3 g16(((N + 255) / 256) * ((K + 255) / 256), S);
        float *pb = db0 ? partB : nullptr;
        if ((prec & VST_WGRAD_Y16) && (prec & VST_WGRAD_X16)) return -1;
        if (prec & VSK_F16) {      // fp16 training mode
            if (prec & VST_WGRAD_Y16) hipLaunchKernelGGL((wgrad_tn_bf16<true, false, 1>), g16, dim3(512), 0, st, dY, ldy, X, ldx, partW, pb, M, N, K, rps);
            else if (prec & VST_WGRAD_X16) hipLaunchKernelGGL((wgrad_tn_bf16<false, true, 1>), g16, dim3(512), 0, st, dY, ldy, X, ldx, partW, pb, M, N, K, rps);
            else hipLaunchKernelGGL((wgrad_tn_bf16<false, false, 1>), g16, dim3(512), 0, st, dY, ldy, X, ldx, partW, pb, M, N, K, rps);
        } else
        if (prec & VST_WGRAD_Y16) hipLaunchKernelGGL((wgrad_tn_bf16<true, false>), g16, dim3(512), 0, st, dY, ldy, X, ldx, partW, pb, M, N, K, rps);
        else if (prec & VST_WGRAD_X16) hipLaunchKernelGGL((wgrad_tn_bf16<false, true>), g16, dim3(512), 0, st, dY, ldy, X, ldx, partW, pb, M, N, K, rps);
        else hipLaunchKernelGGL((wgrad_tn_bf16<false, false>), g16, dim3(512), 0, st, dY, ldy, X, ldx, partW, pb, M, N, K, rps);
    } else
        hipLaunchKernelGGL(wgrad_tn, grid, dim3(256), 0, st, dY, ldy, X, ldx, partW, db0 ? partB : nullptr, M, N, K, rps);
    VSK_CHECK_LAUNCH();
    {
        const size_t nvec = (size_t)N * K / 4;
        const int nb_w = (int)((nvec + 15) / 16), nb_b = db0 ? (N + 15) / 16 : 0;
        hipLaunchKernelGGL(reduce_wgrad, dim3(nb_w + nb_b), dim3(256), 0, st, partW, partB, S, N, K, dW0, dW1, dW2, db0, db1,
                           db2, rows_per_dest, nb_w);
        VSK_CHECK_LAUNCH();
    }
    return 0;
}

// sums `S` partial rows of `cols` floats into up to 3 destinations (LayerNorm gamma / beta, final_layer gradients)
int vst_reduce_rows(const float *part, int S, int rows, int cols, float *d0, float *d1, float *d2, int rows_per_dest,
                    hipStream_t st) {
    const size_t n = (size_t)rows * cols;
    hipLaunchKernelGGL(reduce_partials<1>, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, part, S, rows, cols, d0, d1, d2,
                       rows_per_dest);
    VSK_CHECK_LAUNCH();
    return 0;
}

int vst_transpose(const float *in, float *out, int rows, int cols, hipStream_t st) {
    hipLaunchKernelGGL(transpose2d, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, st, in, out, rows, cols);
    VSK_CHECK_LAUNCH();
    return 0;
}

int vst_transpose_batch(const VskMatJobs &jobs, hipStream_t st) {
    if (jobs.n < 1 || jobs.n > VskMatJobs::MAX) return -1;
    hipLaunchKernelGGL(transpose2d_batch, dim3(256, jobs.n), dim3(256), 0, st, jobs);
    VSK_CHECK_LAUNCH();
    return 0;
}

int vst_mse_mask_blocks(int n) { const int b = (n + 255) / 256; return b < 256 ? (b < 1 ? 1 : b) : 256; }

int vst_mse_mask_fwd(const float *out, const float *tgt, const unsigned char *mask, int n, int mean, float *part,
                     float *loss, hipStream_t st) {
    const int blocks = vst_mse_mask_blocks(n);
    hipLaunchKernelGGL(mse_mask_partial, dim3(blocks), dim3(256), 0, st, out, tgt, mask, n, part);
    VSK_CHECK_LAUNCH();
    hipLaunchKernelGGL(mse_mask_final, dim3(1), dim3(64), 0, st, part, blocks, mean ? 1.0f / (float)n : 1.0f, loss);
    VSK_CHECK_LAUNCH();
    return 0;
}

int vst_mse_mask_bwd(const float *out, const float *tgt, const unsigned char *mask, const float *gout, int n, int mean,
                     float *dout, hipStream_t st) {
    hipLaunchKernelGGL(mse_mask_bwd, dim3(vst_mse_mask_blocks(n)), dim3(256), 0, st, out, tgt, mask, gout, n,
                       mean ? 1.0f / (float)n : 1.0f, dout);
    VSK_CHECK_LAUNCH();
    return 0;
}
