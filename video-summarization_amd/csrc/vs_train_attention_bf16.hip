// vs_train_attention_bf16.hip — the training path's attention on the bf16 matrix pipe (low-precision training,
// VS_TRAIN_FLAG_BF16_ATTENTION; reference simnet.py:155-161 under `amp.autocast()`, train.py:120): forward with dropout
// and a saved log-sum-exp, and the flash-style backward, head dim 32 / 64 (two waves per SIMD) and 128 (one: 256 + registers).
//
// Same decomposition as the exact kernels of vs_train_attention.hip - three kernels, 4 waves x 32 "owner" rows per block,
// the other sequence streamed through LDS, S and dP recomputed in both backward kernels so that every output element is
// written once by one wave (no atomics, bitwise reproducible) - on v_mfma_f32_32x32x16_bf16:
//   * q * scale * log2(e), k, v, dO and the probabilities / dS are rounded to bf16 (round to nearest even) on their way
//     into LDS / the MFMA; scores, softmax statistics, lse, delta and every accumulator stay fp32; tensors stay fp32 in HBM;
//   * a 64-row tile of the streamed sequence is ONE row-major bf16 image [64][DH] per matrix that serves both kinds of
//     read: ROW fragments (ds_read_b128: lane = row, 8 consecutive columns) for the products that contract over the head
//     dimension (S, dP), and TRANSPOSED fragments (ds_read_b64_tr_b16: 4 rows x 16 columns per 16 lanes, delivered
//     column-major) for the products that contract over the streamed rows (P.V, dS.K, P~^T.dO, dS^T.Q).  16-byte chunk c of
//     row m is stored at chunk c ^ swz(m), chosen per head dim so that both reads are bank-conflict-free;
//   * "accumulator as operand": registers 8s .. 8s+7 of a 32x32 fp32 result, packed pairwise, are the B operand of the
//     next product's 16-row step s (row order 16s + 8(j>>2) + 4h + (j&3) - the transposed fragments follow it);
//   * row constants enter as the accumulator's initial value where the row index is the accumulator's row (-lse2, -delta in
//     the key-owner kernel);
//   * dropout keep decisions come bit-packed from attn_dropout_bits (vs_train_attention.hip): one word per tile and lane.
#include <atomic>

#include "vs_train_device.h"
#include "vs_train_kernels.h"

namespace {

constexpr float NEG_INF_B = -__builtin_inff();
typedef unsigned short h16;
typedef short s16x4 __attribute__((ext_vector_type(4)));

// swizzle of a [rows][DH] bf16 image (rows of 2 DH bytes): XOR of the 16-byte chunk index
template <int DH>
__device__ __forceinline__ int img_swz(int row) {
    if constexpr (DH == 32) return (row >> 2) & 3;
    else if constexpr (DH == 64) return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
    else return ((row & 3) << 2) | ((row >> 2) & 3);
}
template <int DH>
__device__ __forceinline__ int img_off(int row, int chunk) { return row * (2 * DH) + ((chunk ^ img_swz<DH>(row)) << 4); }

// row fragment: lane (r, h) <- columns 16 ks + 8 h .. + 7 of row rbase + r
template <int DH>
__device__ __forceinline__ u32x4 row_frag(const unsigned char *img, int rbase, int ks, int r, int h) {
    return __builtin_bit_cast(u32x4, *(const u32x4 *)(img + img_off<DH>(rbase + r, 2 * ks + h)));
}
// transposed fragment: lane (r, h) <- rows r0 + {4h .. 4h+3, 8 + 4h .. 8 + 4h+3} of column 32 db + r
template <int DH>
__device__ __forceinline__ u32x4 tr_frag(const unsigned char *img, int r0, int db, int lane) {
    const int i16 = lane & 15, qq = i16 >> 2, p = i16 & 3, g2 = (lane >> 4) & 1, h = lane >> 5;
    const int row = r0 + 4 * h + qq, chunk = 4 * db + 2 * g2 + (p >> 1), sub = 8 * (p & 1);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(img + img_off<DH>(row, chunk) + sub));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(img + img_off<DH>(row + 8, chunk) + sub));
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    const u32x4 v = {l2[0], l2[1], h2[0], h2[1]};
    return __builtin_bit_cast(u32x4, v);
}

template <int DH>
struct TileB {                                  // one staged 64-row tile pair + its per-row side data
    unsigned char a[64 * 2 * DH];               // K (fwd, dq) | Q * scale * log2e (dkdv)
    unsigned char b[64 * 2 * DH];               // V (fwd, dq) | dO               (dkdv)
    float s0[64];                               // key bias (0 / -inf)  |  -lse2 (-inf on rows >= T)
    float s1[64];                               //                      |  -delta
};

// global -> registers -> bf16 LDS image of a 64 x DH tile pair by 256 threads.  A16 / B16: that matrix already IS bf16 in
// memory (q * scale * log2 e, k, v as the QKV GEMM's bf16 epilogue writes them: nothing to convert, 16-byte chunks go
// straight into the image); otherwise fp32, rounded here (DH / 16 float4 per thread).
template <int DH, bool A16 = false, bool B16 = false, int F16 = 0>     // F16: the 16-bit type is IEEE f16 (fp16 training mode)
struct StagerB {
    static constexpr int NV = DH / 16, NC = DH / 32;       // float4 per thread (fp32 source) | 16-byte chunks per thread (bf16 source)
    f32x4 va[A16 ? 1 : NV], vb[B16 ? 1 : NV];
    u32x4 ca[A16 ? NC : 1], cb[B16 ? NC : 1];
    __device__ __forceinline__ void load(const void *pa_, size_t lda, const void *pb_, size_t ldb, int row0, int nrows, float mul_a) {
        const int tid = threadIdx.x;
        const u32x4 zc = {0u, 0u, 0u, 0u};
        if constexpr (A16 || B16) {
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const int idx = tid + 256 * i, row = idx / (DH / 8), c8 = (idx % (DH / 8)) * 8;
                const bool ok = row0 + row < nrows;
                if constexpr (A16) ca[i] = ok ? *(const u32x4 *)((const h16 *)pa_ + (size_t)(row0 + row) * lda + c8) : zc;
                if constexpr (B16) cb[i] = ok ? *(const u32x4 *)((const h16 *)pb_ + (size_t)(row0 + row) * ldb + c8) : zc;
            }
        }
        if constexpr (!A16 || !B16) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int idx = tid + 256 * i, row = idx / (DH / 4), c4 = (idx % (DH / 4)) * 4;
                const bool ok = row0 + row < nrows;
                if constexpr (!A16) va[i] = ok ? *(const f32x4 *)((const float *)pa_ + (size_t)(row0 + row) * lda + c4) * mul_a : f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (!B16) vb[i] = ok ? *(const f32x4 *)((const float *)pb_ + (size_t)(row0 + row) * ldb + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    __device__ __forceinline__ void store(TileB<DH> &t) const {
        const int tid = threadIdx.x;
        if constexpr (A16 || B16) {
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const int idx = tid + 256 * i, row = idx / (DH / 8), chunk = idx % (DH / 8);
                if constexpr (A16) *(u32x4 *)(t.a + img_off<DH>(row, chunk)) = ca[i];
                if constexpr (B16) *(u32x4 *)(t.b + img_off<DH>(row, chunk)) = cb[i];
            }
        }
        if constexpr (!A16 || !B16) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int idx = tid + 256 * i, row = idx / (DH / 4), c4 = (idx % (DH / 4)) * 4;
                const int off = img_off<DH>(row, c4 >> 3) + ((c4 & 4) << 1);
                if constexpr (!A16) { u32x2 ua; ua[0] = pack_lp<F16>(va[i][0], va[i][1]); ua[1] = pack_lp<F16>(va[i][2], va[i][3]); *(u32x2 *)(t.a + off) = ua; }
                if constexpr (!B16) { u32x2 ub; ub[0] = pack_lp<F16>(vb[i][0], vb[i][1]); ub[1] = pack_lp<F16>(vb[i][2], vb[i][3]); *(u32x2 *)(t.b + off) = ub; }
            }
        }
    }
};

__device__ __forceinline__ f32x16 zero16b() {
    f32x16 z;
#pragma unroll
    for (int t = 0; t < 16; ++t) z[t] = 0.f;
    return z;
}

// the owner's own rows as B-operand fragments: lane (r, h) <- src[16 ks + 8 h .. + 7] * mul, rounded to bf16
template <int DH, int F16>
__device__ __forceinline__ void owner_frags(const float *src, float mul, int h, u32x4 (&f)[DH / 16]) {
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) {
        const f32x4 v0 = *(const f32x4 *)(src + 16 * ks + 8 * h) * mul, v1 = *(const f32x4 *)(src + 16 * ks + 8 * h + 4) * mul;
        const u32x4 u = {pack_lp<F16>(v0[0], v0[1]), pack_lp<F16>(v0[2], v0[3]), pack_lp<F16>(v1[0], v1[1]), pack_lp<F16>(v1[2], v1[3])};
        f[ks] = __builtin_bit_cast(u32x4, u);
    }
}
// ... from a row that already is bf16 in memory: lane (r, h) <- the 16-byte chunk 2 ks + h of the row
template <int DH>
__device__ __forceinline__ void owner_frags16(const h16 *src, int h, u32x4 (&f)[DH / 16]) {
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) f[ks] = __builtin_bit_cast(u32x4, *(const u32x4 *)(src + 16 * ks + 8 * h));
}
// registers 8 s .. 8 s + 7 of a 32x32 result -> the B operand of the next product's 16-row step s
template <int F16>
__device__ __forceinline__ u32x4 pack_step(const f32x16 &x, int s) {
    const u32x4 u = {pack_lp<F16>(x[8 * s], x[8 * s + 1]), pack_lp<F16>(x[8 * s + 2], x[8 * s + 3]),
                     pack_lp<F16>(x[8 * s + 4], x[8 * s + 5]), pack_lp<F16>(x[8 * s + 6], x[8 * s + 7])};
    return __builtin_bit_cast(u32x4, u);
}

// keep decision `bit` of a lane's dropout word as a multiplier: 1 / (1 - p) or 0  (v_bfe_i32 + v_and)
__device__ __forceinline__ float keep_mul(unsigned word, int bit, unsigned ds_bits) {
    return __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_sbfe((int)word, (unsigned)bit, 1u) & ds_bits);
}
// the two dropout words of a 64-row tile, one tile ahead of their use (a load consumed at once would expose the latency of
// every stage load issued before it: vmcnt retires in order)
__device__ __forceinline__ void load_kw(const unsigned *row, int tile, int W, unsigned (&kw)[2]) {
    kw[0] = 2 * tile < W ? row[2 * tile] : 0u;
    kw[1] = 2 * tile + 1 < W ? row[2 * tile + 1] : 0u;
}
// accumulator whose initial value is a per-row constant from LDS (rows 8 tg + 4 h + e of the 32-row block at c)
__device__ __forceinline__ f32x16 rows_init(const float *c, int h) {
    f32x16 a;
#pragma unroll
    for (int tg = 0; tg < 4; ++tg) {
        const f32x4 v = *(const f32x4 *)&c[8 * tg + 4 * h];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[4 * tg + e] = v[e];
    }
    return a;
}

// ------------------------------------------------------------------------------------------
// forward: owner = queries
// ------------------------------------------------------------------------------------------
template <int DH, bool DROP, bool IN16, int F16 = 0>          // F16: 16-bit operands / storage are IEEE f16; IN16: q (pre-multiplied by scale * log2 e), k, v are bf16 in memory
__global__ __launch_bounds__(256, DH == 128 ? 1 : 2) void attn_fwd_train_bf16(
    const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    const uint8_t *__restrict__ mask, float *__restrict__ out, float *__restrict__ lse2, int H, int T, float scale,
    float drop_scale, const unsigned *__restrict__ dbits) {
    constexpr int NS = DH / 16, ND = DH / 32;
    __shared__ __attribute__((aligned(16))) TileB<DH> lds[2];        // static: 66.5 KB at head dim 128, above the dynamic default
    const int nq = (T + 127) / 128;
    const int bh = blockIdx.x / nq, qt = blockIdx.x - bh * nq, b = bh / H, hd = bh - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int qi = qt * 128 + wave * 32 + r, qc = qi < T ? qi : T - 1;
    const float sl2 = scale * 1.4426950408889634f;
    constexpr int ES = IN16 ? 2 : 4;            // bytes per stored q / k / v element
    const char *qb = (const char *)q + (size_t)bh * T * DH * ES, *kb = (const char *)k + (size_t)bh * T * DH * ES,
               *vb = (const char *)v + (size_t)bh * T * DH * ES;

    u32x4 qf[NS];
    if constexpr (IN16) owner_frags16<DH>((const h16 *)qb + (size_t)qc * DH, h, qf);
    else owner_frags<DH, F16>((const float *)qb + (size_t)qc * DH, sl2, h, qf);
    f32x16 o[ND];
#pragma unroll
    for (int db = 0; db < ND; ++db) o[db] = zero16b();
    float m_run = NEG_INF_B, l_run = 0.f;

    StagerB<DH, IN16, IN16, F16> sg;
    auto side = [&](TileB<DH> &t, int key0) __attribute__((always_inline)) {
        if (tid < 64) {
            const int key = key0 + tid;
            t.s0[tid] = (key >= T || (mask != nullptr && mask[(size_t)b * T + key])) ? NEG_INF_B : 0.f;
        }
    };
    const int nkt = (T + 63) / 64, W = (T + 31) / 32;
    const unsigned *bq = DROP ? dbits + ((size_t)bh * T + qc) * W : nullptr;
    const unsigned dsb = __builtin_bit_cast(unsigned, drop_scale);
    unsigned kwn[2] = {0u, 0u};
    if (DROP) load_kw(bq, 0, W, kwn);
    sg.load(kb, DH, vb, DH, 0, T, 1.0f);
    sg.store(lds[0]);
    side(lds[0], 0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const TileB<DH> &t = lds[kt & 1];
        const unsigned kw[2] = {kwn[0], kwn[1]};
        if (DROP && kt + 1 < nkt) load_kw(bq, kt + 1, W, kwn);
        if (kt + 1 < nkt) sg.load(kb, DH, vb, DH, 64 * (kt + 1), T, 1.0f);
        f32x16 s[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            s[n] = rows_init(&t.s0[32 * n], h);              // the key bias (0 / -inf) is the accumulator's initial value
#pragma unroll
            for (int ks = 0; ks < NS; ++ks) s[n] = mfma_lp<F16>(row_frag<DH>(t.a, 32 * n, ks, r, h), qf[ks], s[n]);
        }
        float mx = s[0][0];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[n][e]);
        mx = pair_max(mx);
        const float m_new = fmaxf(m_run, mx);
        const float m_use = m_new == NEG_INF_B ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        float ls = 0.f;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const unsigned kwh = kw[n] >> (4 * h);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pe = __builtin_amdgcn_exp2f(s[n][e] - m_use);
                ls += pe;
                s[n][e] = !DROP ? pe : pe * keep_mul(kwh, (e & 3) + 8 * (e >> 2), dsb);
            }
        }
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int db = 0; db < ND; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const u32x4 pf = pack_step<F16>(s[n], s2);
#pragma unroll
                for (int db = 0; db < ND; ++db) o[db] = mfma_lp<F16>(tr_frag<DH>(t.b, 32 * n + 16 * s2, db, lane), pf, o[db]);
            }
        if (kt + 1 < nkt) { sg.store(lds[(kt + 1) & 1]); side(lds[(kt + 1) & 1], 64 * (kt + 1)); }
        __syncthreads();
    }
    const float l_tot = pair_sum(l_run);
    const float inv = 1.0f / l_tot;
    if (qi < T) {
        float *op = out + ((size_t)b * T + qi) * (H * DH) + hd * DH;
#pragma unroll
        for (int db = 0; db < ND; ++db)
#pragma unroll
            for (int tg = 0; tg < 4; ++tg) {
                f32x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = o[db][4 * tg + e] * inv;
                *(f32x4 *)(op + 32 * db + 8 * tg + 4 * h) = w;
            }
        if (h == 0) lse2[(size_t)bh * T + qi] = m_run + log2f(l_tot);
    }
}

// ------------------------------------------------------------------------------------------
// backward, queries own: dQ
// ------------------------------------------------------------------------------------------
// OUT16 ("gradient tensors are bf16"): dO [M][d] arrives as bf16 (written so by the out-projection's input-gradient GEMM) and
// dq | dk | dv are written as bf16 [M][3 d] (their only readers - the QKV weight gradient and input gradient - are bf16 GEMMs)
template <int DH, bool DROP, bool IN16, bool OUT16, int F16 = 0>
__global__ __launch_bounds__(256, DH == 128 ? 1 : 2) void attn_bwd_dq_bf16(
    const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    const uint8_t *__restrict__ mask, const float *__restrict__ dO, const float *__restrict__ lse2,
    const float *__restrict__ delta, float *__restrict__ dqkv, int H, int T, float scale, float drop_scale,
    const unsigned *__restrict__ dbits) {
    constexpr int NS = DH / 16, ND = DH / 32;
    __shared__ __attribute__((aligned(16))) TileB<DH> lds[2];        // static: 66.5 KB at head dim 128, above the dynamic default
    const int nq = (T + 127) / 128, d = H * DH;
    const int bh = blockIdx.x / nq, qt = blockIdx.x - bh * nq, b = bh / H, hd = bh - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int qi = qt * 128 + wave * 32 + r, qc = qi < T ? qi : T - 1;
    const float sl2 = scale * 1.4426950408889634f;
    constexpr int ES = IN16 ? 2 : 4;
    const char *qb = (const char *)q + (size_t)bh * T * DH * ES, *kb = (const char *)k + (size_t)bh * T * DH * ES,
               *vb = (const char *)v + (size_t)bh * T * DH * ES;

    u32x4 qf[NS], dof[NS];
    if constexpr (IN16) owner_frags16<DH>((const h16 *)qb + (size_t)qc * DH, h, qf);
    else owner_frags<DH, F16>((const float *)qb + (size_t)qc * DH, sl2, h, qf);
    if constexpr (OUT16) owner_frags16<DH>((const h16 *)dO + ((size_t)b * T + qc) * d + hd * DH, h, dof);
    else owner_frags<DH, F16>(dO + ((size_t)b * T + qc) * d + hd * DH, 1.0f, h, dof);
    const float lq = lse2[(size_t)bh * T + qc], dq_delta = delta[(size_t)bh * T + qc];
    f32x16 acc[ND];
#pragma unroll
    for (int db = 0; db < ND; ++db) acc[db] = zero16b();

    StagerB<DH, IN16, IN16, F16> sg;
    auto side = [&](TileB<DH> &t, int key0) __attribute__((always_inline)) {
        if (tid < 64) {
            const int key = key0 + tid;
            t.s0[tid] = (key >= T || (mask != nullptr && mask[(size_t)b * T + key])) ? NEG_INF_B : 0.f;
        }
    };
    const int nkt = (T + 63) / 64, W = (T + 31) / 32;
    const unsigned *bq = DROP ? dbits + ((size_t)bh * T + qc) * W : nullptr;
    const unsigned dsb = __builtin_bit_cast(unsigned, drop_scale);
    unsigned kwn[2] = {0u, 0u};
    if (DROP) load_kw(bq, 0, W, kwn);
    sg.load(kb, DH, vb, DH, 0, T, 1.0f);
    sg.store(lds[0]);
    side(lds[0], 0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const TileB<DH> &t = lds[kt & 1];
        const unsigned kw[2] = {kwn[0], kwn[1]};
        if (DROP && kt + 1 < nkt) load_kw(bq, kt + 1, W, kwn);
        if (kt + 1 < nkt) sg.load(kb, DH, vb, DH, 64 * (kt + 1), T, 1.0f);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            f32x16 s = rows_init(&t.s0[32 * n], h), dp = zero16b();        // key bias = initial accumulator
#pragma unroll
            for (int ks = 0; ks < NS; ++ks) {
                s = mfma_lp<F16>(row_frag<DH>(t.a, 32 * n, ks, r, h), qf[ks], s);
                dp = mfma_lp<F16>(row_frag<DH>(t.b, 32 * n, ks, r, h), dof[ks], dp);
            }
            const unsigned kwh = kw[n] >> (4 * h);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pe = __builtin_amdgcn_exp2f(s[i] - lq);
                const float g = DROP ? fmaf(dp[i], keep_mul(kwh, (i & 3) + 8 * (i >> 2), dsb), -dq_delta) : dp[i] - dq_delta;
                s[i] = pe * g;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const u32x4 dsf = pack_step<F16>(s, s2);
#pragma unroll
                for (int db = 0; db < ND; ++db) acc[db] = mfma_lp<F16>(tr_frag<DH>(t.a, 32 * n + 16 * s2, db, lane), dsf, acc[db]);
            }
        }
        if (kt + 1 < nkt) { sg.store(lds[(kt + 1) & 1]); side(lds[(kt + 1) & 1], 64 * (kt + 1)); }
        __syncthreads();
    }
    if (qi < T) {
        float *op = dqkv + ((size_t)b * T + qi) * (3 * d) + hd * DH;
        h16 *op16 = (h16 *)dqkv + ((size_t)b * T + qi) * (3 * d) + hd * DH;
#pragma unroll
        for (int db = 0; db < ND; ++db)
#pragma unroll
            for (int tg = 0; tg < 4; ++tg) {
                f32x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = acc[db][4 * tg + e] * scale;
                if constexpr (OUT16) *(u32x2 *)(op16 + 32 * db + 8 * tg + 4 * h) = u32x2{pack_lp<F16>(w[0], w[1]), pack_lp<F16>(w[2], w[3])};
                else *(f32x4 *)(op + 32 * db + 8 * tg + 4 * h) = w;
            }
    }
}

// ------------------------------------------------------------------------------------------
// backward, keys own: dK and dV
// ------------------------------------------------------------------------------------------
template <int DH, bool DROP, bool IN16, bool OUT16, int F16 = 0>
__global__ __launch_bounds__(256, DH == 128 ? 1 : 2) void attn_bwd_dkdv_bf16(
    const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    const uint8_t *__restrict__ mask, const float *__restrict__ dO, const float *__restrict__ lse2,
    const float *__restrict__ delta, float *__restrict__ dqkv, int H, int T, float scale, float drop_scale,
    const unsigned *__restrict__ dbits, int BH) {
    constexpr int NS = DH / 16, ND = DH / 32;
    __shared__ __attribute__((aligned(16))) TileB<DH> lds[2];        // static: 66.5 KB at head dim 128, above the dynamic default
    const int nk = (T + 127) / 128, d = H * DH;
    const int bh = blockIdx.x / nk, ktile = blockIdx.x - bh * nk, b = bh / H, hd = bh - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int ki = ktile * 128 + wave * 32 + r, kc = ki < T ? ki : T - 1;
    const float sl2 = scale * 1.4426950408889634f;
    constexpr int ES = IN16 ? 2 : 4;
    const char *qb = (const char *)q + (size_t)bh * T * DH * ES, *kb = (const char *)k + (size_t)bh * T * DH * ES,
               *vb = (const char *)v + (size_t)bh * T * DH * ES;
    const char *dob = (const char *)dO + ((size_t)b * T * d + hd * DH) * (OUT16 ? 2 : 4);          // row stride d elements

    u32x4 kf[NS], vf[NS];
    if constexpr (IN16) {
        owner_frags16<DH>((const h16 *)kb + (size_t)kc * DH, h, kf);
        owner_frags16<DH>((const h16 *)vb + (size_t)kc * DH, h, vf);
    } else {
        owner_frags<DH, F16>((const float *)kb + (size_t)kc * DH, 1.0f, h, kf);
        owner_frags<DH, F16>((const float *)vb + (size_t)kc * DH, 1.0f, h, vf);
    }
    const bool kmasked = mask != nullptr && mask[(size_t)b * T + kc];
    f32x16 dk[ND], dv[ND];
#pragma unroll
    for (int db = 0; db < ND; ++db) { dk[db] = zero16b(); dv[db] = zero16b(); }

    StagerB<DH, IN16, OUT16> sg;                                  // Q bf16 when IN16; dO bf16 when OUT16
    auto side = [&](TileB<DH> &t, int q0) __attribute__((always_inline)) {
        if (tid < 64) {
            const int qi = q0 + tid;
            const bool ok = qi < T;
            t.s0[tid] = ok ? -lse2[(size_t)bh * T + qi] : NEG_INF_B;      // p = exp2(s - inf) = 0 on rows >= T
            t.s1[tid] = ok ? -delta[(size_t)bh * T + qi] : 0.f;
        }
    };
    const int nqt = (T + 63) / 64, W = (T + 31) / 32;
    const unsigned *bk = DROP ? dbits + (size_t)BH * T * W + ((size_t)bh * T + kc) * W : nullptr;     // the key-major copy
    const unsigned dsb = __builtin_bit_cast(unsigned, drop_scale);
    unsigned kwn[2] = {0u, 0u};
    if (DROP) load_kw(bk, 0, W, kwn);
    sg.load(qb, DH, dob, d, 0, T, sl2);
    sg.store(lds[0]);
    side(lds[0], 0);
    __syncthreads();
    for (int it = 0; it < nqt; ++it) {
        const TileB<DH> &t = lds[it & 1];
        const unsigned kw[2] = {kwn[0], kwn[1]};
        if (DROP && it + 1 < nqt) load_kw(bk, it + 1, W, kwn);
        if (it + 1 < nqt) sg.load(qb, DH, dob, d, 64 * (it + 1), T, sl2);
#pragma unroll
        for (int qblk = 0; qblk < 2; ++qblk) {
            // S[query][key] - lse2[query] and (no dropout) dP[query][key] - delta[query]: the row constants are the initial
            // accumulators.  With dropout the decision multiplies dP BEFORE delta leaves, so -delta joins in the fma below.
            f32x16 s = rows_init(&t.s0[32 * qblk], h), dp = DROP ? zero16b() : rows_init(&t.s1[32 * qblk], h);
#pragma unroll
            for (int ks = 0; ks < NS; ++ks) {
                s = mfma_lp<F16>(row_frag<DH>(t.a, 32 * qblk, ks, r, h), kf[ks], s);
                dp = mfma_lp<F16>(row_frag<DH>(t.b, 32 * qblk, ks, r, h), vf[ks], dp);
            }
            // (a masked owner key needs no bias here: its lane is one COLUMN of every product below, so whatever it
            // computes stays in its own dK / dV row, which is written as zero at the end)
            const unsigned kwh = kw[qblk] >> (4 * h);
            f32x16 pd;          // dropped-out probabilities (the B operand of dV)
#pragma unroll
            for (int tg = 0; tg < 4; ++tg) {
                f32x4 dl = {0.f, 0.f, 0.f, 0.f};
                if (DROP) dl = *(const f32x4 *)&t.s1[32 * qblk + 8 * tg + 4 * h];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * tg + e;
                    const float pe = __builtin_amdgcn_exp2f(s[i]);
                    if (DROP) {
                        const float km = keep_mul(kwh, e + 8 * tg, dsb);
                        pd[i] = pe * km;
                        s[i] = pe * fmaf(dp[i], km, dl[e]);      // dS  (s1 holds -delta)
                    } else {
                        pd[i] = pe;
                        s[i] = pe * dp[i];
                    }
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const u32x4 pdf = pack_step<F16>(pd, s2), dsf = pack_step<F16>(s, s2);
#pragma unroll
                for (int db = 0; db < ND; ++db) {
                    dv[db] = mfma_lp<F16>(tr_frag<DH>(t.b, 32 * qblk + 16 * s2, db, lane), pdf, dv[db]);
                    dk[db] = mfma_lp<F16>(tr_frag<DH>(t.a, 32 * qblk + 16 * s2, db, lane), dsf, dk[db]);
                }
            }
        }
        if (it + 1 < nqt) { sg.store(lds[(it + 1) & 1]); side(lds[(it + 1) & 1], 64 * (it + 1)); }
        __syncthreads();
    }
    if (ki < T) {
        float *op = dqkv + ((size_t)b * T + ki) * (3 * d) + hd * DH;
        h16 *op16 = (h16 *)dqkv + ((size_t)b * T + ki) * (3 * d) + hd * DH;
        const float ln2 = 0.6931471805599453f;       // dK = scale * dS^T Q = (dS^T Qs) / log2(e)
#pragma unroll
        for (int db = 0; db < ND; ++db)
#pragma unroll
            for (int tg = 0; tg < 4; ++tg) {
                f32x4 wk, wv;
#pragma unroll
                for (int e = 0; e < 4; ++e) { wk[e] = kmasked ? 0.f : dk[db][4 * tg + e] * ln2; wv[e] = kmasked ? 0.f : dv[db][4 * tg + e]; }
                if constexpr (OUT16) {
                    *(u32x2 *)(op16 + d + 32 * db + 8 * tg + 4 * h) = u32x2{pack_lp<F16>(wk[0], wk[1]), pack_lp<F16>(wk[2], wk[3])};
                    *(u32x2 *)(op16 + 2 * d + 32 * db + 8 * tg + 4 * h) = u32x2{pack_lp<F16>(wv[0], wv[1]), pack_lp<F16>(wv[2], wv[3])};
                } else {
                *(f32x4 *)(op + d + 32 * db + 8 * tg + 4 * h) = wk;
                *(f32x4 *)(op + 2 * d + 32 * db + 8 * tg + 4 * h) = wv;
                }
            }
    }
}

}  // namespace

#define VSTB_LAUNCH(KERNEL_, DH_, DROP_, IN16_, ...)                                                            \
    do {                                                                                                        \
        if (f16) hipLaunchKernelGGL((KERNEL_<DH_, DROP_, IN16_, 1>), grid, dim3(256), 0, st, __VA_ARGS__);      \
        else hipLaunchKernelGGL((KERNEL_<DH_, DROP_, IN16_, 0>), grid, dim3(256), 0, st, __VA_ARGS__);          \
    } while (0)
#define VSTB_DISPATCH2(KERNEL_, DH_, ...)                                                                       \
    do {                                                                                                        \
        if (drop && in16) VSTB_LAUNCH(KERNEL_, DH_, true, true, __VA_ARGS__);                                   \
        else if (drop) VSTB_LAUNCH(KERNEL_, DH_, true, false, __VA_ARGS__);                                     \
        else if (in16) VSTB_LAUNCH(KERNEL_, DH_, false, true, __VA_ARGS__);                                     \
        else VSTB_LAUNCH(KERNEL_, DH_, false, false, __VA_ARGS__);                                              \
    } while (0)
#define VSTB_DISPATCH(KERNEL_, ...)                                                                             \
    do {                                                                                                        \
        const bool drop = p > 0.f;                                                                              \
        if (dh == 32) VSTB_DISPATCH2(KERNEL_, 32, __VA_ARGS__);                                                 \
        else if (dh == 64) VSTB_DISPATCH2(KERNEL_, 64, __VA_ARGS__);                                            \
        else if (dh == 128) VSTB_DISPATCH2(KERNEL_, 128, __VA_ARGS__);                                          \
        else return -1;                                                                                         \
    } while (0)

bool vst_attention_bf16_supported(int dh) { return dh == 32 || dh == 64 || dh == 128; }

// in16: q (already times scale * log2 e), k, v are bf16 [B, H, T, dh] planes (the QKV GEMM's bf16 epilogue) instead of fp32
int vst_attention_fwd_bf16(const float *q, const float *k, const float *v, const uint8_t *mask, float *out, float *lse2,
                           int B, int H, int T, int dh, float scale, float p, const unsigned *dbits, hipStream_t st, int in16) {
    if (p < 0.f || p >= 1.f || (p > 0.f && dbits == nullptr)) return -1;
    const bool f16 = (in16 & VSK_F16) != 0;          // fp16 training mode: operands (and the stored planes) are IEEE f16
    in16 &= ~VSK_F16;
    const float ds = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    const dim3 grid(B * H * ((T + 127) / 128));
    VSTB_DISPATCH(attn_fwd_train_bf16, q, k, v, mask, out, lse2, H, T, scale, ds, dbits);
    VSK_CHECK_LAUNCH();
    return 0;
}

#define VSTB_LAUNCH_B(KERNEL_, DH_, DROP_, IN16_, OUT16_, ...)                                                  \
    do {                                                                                                        \
        if (f16) hipLaunchKernelGGL((KERNEL_<DH_, DROP_, IN16_, OUT16_, 1>), grid, dim3(256), 0, st, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL_<DH_, DROP_, IN16_, OUT16_, 0>), grid, dim3(256), 0, st, __VA_ARGS__);  \
    } while (0)
#define VSTB_DISPATCH_B3(KERNEL_, DH_, DROP_, ...)                                                              \
    do {                                                                                                        \
        if (in16 && out16) VSTB_LAUNCH_B(KERNEL_, DH_, DROP_, true, true, __VA_ARGS__);                         \
        else if (in16) VSTB_LAUNCH_B(KERNEL_, DH_, DROP_, true, false, __VA_ARGS__);                            \
        else if (out16) VSTB_LAUNCH_B(KERNEL_, DH_, DROP_, false, true, __VA_ARGS__);                           \
        else VSTB_LAUNCH_B(KERNEL_, DH_, DROP_, false, false, __VA_ARGS__);                                     \
    } while (0)
#define VSTB_DISPATCH_B(KERNEL_, ...)                                                                           \
    do {                                                                                                        \
        const bool drop = p > 0.f;                                                                              \
        if (dh == 32) { if (drop) VSTB_DISPATCH_B3(KERNEL_, 32, true, __VA_ARGS__); else VSTB_DISPATCH_B3(KERNEL_, 32, false, __VA_ARGS__); } \
        else if (dh == 64) { if (drop) VSTB_DISPATCH_B3(KERNEL_, 64, true, __VA_ARGS__); else VSTB_DISPATCH_B3(KERNEL_, 64, false, __VA_ARGS__); } \
        else if (dh == 128) { if (drop) VSTB_DISPATCH_B3(KERNEL_, 128, true, __VA_ARGS__); else VSTB_DISPATCH_B3(KERNEL_, 128, false, __VA_ARGS__); } \
        else return -1;                                                                                         \
    } while (0)

// out16: dqkv is written as bf16 [M][3 d] (2-byte elements) instead of fp32
int vst_attention_bwd_bf16(const float *q, const float *k, const float *v, const uint8_t *mask, const float *dO,
                           const float *lse2, const float *delta, float *dqkv, int B, int H, int T, int dh, float scale,
                           float p, const unsigned *dbits, hipStream_t st, int in16, int out16) {
    if (p < 0.f || p >= 1.f || (p > 0.f && dbits == nullptr)) return -1;
    const bool f16 = (in16 & VSK_F16) != 0;
    in16 &= ~VSK_F16;
    const float ds = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    const dim3 grid(B * H * ((T + 127) / 128));
    VSTB_DISPATCH_B(attn_bwd_dkdv_bf16, q, k, v, mask, dO, lse2, delta, dqkv, H, T, scale, ds, dbits, B * H);
    VSK_CHECK_LAUNCH();
    VSTB_DISPATCH_B(attn_bwd_dq_bf16, q, k, v, mask, dO, lse2, delta, dqkv, H, T, scale, ds, dbits);
    VSK_CHECK_LAUNCH();
    return 0;
}
