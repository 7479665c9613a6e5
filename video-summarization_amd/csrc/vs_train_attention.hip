// vs_train_attention.hip — attention of the TRAINING path on gfx950: forward with dropout on the attention weights
// and a saved log-sum-exp, and the flash-style backward (reference simnet.py:155-161 under autograd).
//
// Exact fp32 on v_mfma_f32_32x32x2_f32 like the scoring path.  Conventions (vs_device.h): lane l, r = l & 31,
// h = l >> 5; A operand A[i = r][k = h], B operand B[k = h][j = r], accumulator register t holds
// D[row = acc_row(t, h)][col = r].  Two operand idioms are used throughout:
//   * "row fragment": a lane reads FOUR consecutive contraction indices kk = 8g + 4h + s of its row with one
//     ds_read_b128; MFMA step s contracts {8g + s, 8g + 4 + s}.  Both operands of a product use the same map.
//   * "accumulator as operand": a 32x32 result whose CONTRACTION index for the next product is its row index is used
//     register by register as that product's B operand (step t contracts rows acc_row(t,0), acc_row(t,1)); the other
//     operand is then read column-wise from a row-major LDS tile: ds_read_b32 at [acc_row(t,h)][c0 + r].
//
// Three kernels, all with 4 waves x 32 "owner" rows per block and the other sequence streamed through LDS in
// 32-row tiles (double-buffered, one barrier per tile):
//   attn_fwd_train   owner = queries.  S^T = K Q^T (lane = query), online softmax with the running maximum, dropout
//                    on the normalised-later weights, O^T += V^T P~^T; saves lse2 = log2 sum exp.
//   attn_bwd_dq      owner = queries.  S^T and dP^T = V dO^T recomputed, dS^T = P^T o (dP^T - delta), dQ^T += K^T dS^T.
//   attn_bwd_dkdv    owner = keys.     S = Q K^T and dP = dO V^T (lane = key), dV^T += dO^T P~, dK^T += Q^T dS.
// Recomputing S and dP in both backward kernels costs 7 products instead of 5 but needs no atomics and no
// cross-block hand-off: every output element is written once, by one wave, in a fixed order (bitwise reproducible).
// The dropout keep decisions come from vs_train_device.h's counter hash of (seed, site, (b*H + h)*T + query, key).
// DROP 1: every kernel evaluates the hash per element (13 vector instructions beside a 64-cycle MFMA: 20 % of the
// forward).  DROP 2 (round 3, the full training path): attn_dropout_bits evaluates it ONCE per layer into two bit-packed
// copies - one word per (query, 32 keys) for the kernels whose lanes own queries, one per (key, 32 queries) for the
// kernel whose lanes own keys - which the forward leaves in the activation record for the backward: a kernel reads one
// word per tile and lane and extracts a bit per element (3 instructions).  Same hash, same masks (GPU test: the two
// forms give bit-identical outputs).
#include <atomic>

#include "vs_train_device.h"
#include "vs_train_kernels.h"

namespace {

constexpr float NEG_INF = -__builtin_inff();

template <int DH>
struct TileLds {                       // one staged 32-row tile pair + its per-row side data
    static constexpr int LD = DH + 4;  // 16 distinct rows x b128 then cover all 64 banks
    float a[32][LD];                   // K  (fwd, dq)  |  Q * scale * log2e (dkdv)
    float b[32][LD];                   // V  (fwd, dq)  |  dO               (dkdv)
    float s0[32];                      // key bias (0 / -inf)  |  lse2 (+inf on rows >= T)
    float s1[32];                      //                      |  delta
    unsigned rk[32];                   //                      |  dropout row keys
};

// global -> registers -> LDS staging of a 32 x DH tile pair by 256 threads (DH / 32 float4 per thread and matrix)
template <int DH>
struct Stager {
    static constexpr int NV = DH / 32;          // float4 per thread per matrix
    f32x4 va[NV], vb[NV];
    __device__ __forceinline__ void load(const float *pa, size_t lda, const float *pb, size_t ldb, int row0, int nrows,
                                         float mul_a) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + 256 * i, row = idx / (DH / 4), c4 = (idx % (DH / 4)) * 4;
            const bool ok = row0 + row < nrows;
            va[i] = ok ? *(const f32x4 *)(pa + (size_t)(row0 + row) * lda + c4) * mul_a : f32x4{0.f, 0.f, 0.f, 0.f};
            vb[i] = ok ? *(const f32x4 *)(pb + (size_t)(row0 + row) * ldb + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __device__ __forceinline__ void store(TileLds<DH> &t) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + 256 * i, row = idx / (DH / 4), c4 = (idx % (DH / 4)) * 4;
            *(f32x4 *)&t.a[row][c4] = va[i];
            *(f32x4 *)&t.b[row][c4] = vb[i];
        }
    }
};

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int t = 0; t < 16; ++t) z[t] = 0.f;
    return z;
}

// S^T tile (lane = query): sum_c K[key][c] * Qs[query][c] + key bias; keys in rows
template <int DH>
__device__ __forceinline__ f32x16 st_tile(const TileLds<DH> &t, const f32x4 (&qf)[DH / 8], int r, int h) {
    f32x16 s = zero16();
#pragma unroll
    for (int g = 0; g < DH / 8; ++g) {
        const f32x4 kf = *(const f32x4 *)&t.a[r][8 * g + 4 * h];
#pragma unroll
        for (int e = 0; e < 4; ++e) s = MFMA32(kf[e], qf[g][e], s);
    }
#pragma unroll
    for (int tg = 0; tg < 4; ++tg) {
        const f32x4 bv = *(const f32x4 *)&t.s0[8 * tg + 4 * h];
#pragma unroll
        for (int e = 0; e < 4; ++e) s[4 * tg + e] += bv[e];
    }
    return s;
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int DH, int DROP>        // DROP 0: none, 1: hash per element, 2: bit-packed keep masks (dbits)
__global__ __launch_bounds__(256) void attn_fwd_train(
    const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    const uint8_t *__restrict__ mask, float *__restrict__ out, float *__restrict__ lse2, int H, int T, float scale,
    unsigned long long seed, unsigned site, float p, const unsigned *__restrict__ dbits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];      // 2 x TileLds<DH>: 68 KB at head dim 128
    TileLds<DH> *lds = reinterpret_cast<TileLds<DH> *>(lds_raw);
    const int nq = (T + 127) / 128;
    const int bh = blockIdx.x / nq, qt = blockIdx.x - bh * nq, b = bh / H, hd = bh - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int qi = qt * 128 + wave * 32 + r, qc = qi < T ? qi : T - 1;
    const float sl2 = scale * 1.4426950408889634f;
    const float *qb = q + (size_t)bh * T * DH, *kb = k + (size_t)bh * T * DH, *vb = v + (size_t)bh * T * DH;
    const DropSite ds = drop_site(seed, site, p);
    const unsigned rkq = drop_rowkey(ds, (unsigned)(bh * T + qc));

    f32x4 qf[DH / 8];
#pragma unroll
    for (int g = 0; g < DH / 8; ++g) qf[g] = *(const f32x4 *)(qb + (size_t)qc * DH + 8 * g + 4 * h) * sl2;
    f32x16 o[DH / 32];
#pragma unroll
    for (int cb = 0; cb < DH / 32; ++cb) o[cb] = zero16();
    float m_run = NEG_INF, l_run = 0.f;

    Stager<DH> sg;
    auto side = [&](TileLds<DH> &t, int key0) __attribute__((always_inline)) {
        if (tid < 32) {
            const int key = key0 + tid;
            t.s0[tid] = (key >= T || (mask != nullptr && mask[(size_t)b * T + key])) ? NEG_INF : 0.f;
        }
    };
    const int nkt = (T + 31) / 32;
    const unsigned *bq = DROP == 2 ? dbits + ((size_t)bh * T + qc) * nkt : nullptr;      // this query's keep words
    unsigned kw = DROP == 2 ? bq[0] : 0u;
    sg.load(kb, DH, vb, DH, 0, T, 1.0f);
    sg.store(lds[0]);
    side(lds[0], 0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const TileLds<DH> &t = lds[kt & 1];
        if (kt + 1 < nkt) sg.load(kb, DH, vb, DH, 32 * (kt + 1), T, 1.0f);
        const unsigned kw_cur = kw;
        if (DROP == 2 && kt + 1 < nkt) kw = bq[kt + 1];
        f32x16 s = st_tile<DH>(t, qf, r, h);
        float mx = s[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, s[e]);
        mx = pair_max(mx);
        const float m_new = fmaxf(m_run, mx);
        const float m_use = m_new == NEG_INF ? 0.f : m_new;       // a row with no valid key so far: keep p = 0, not NaN
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        float ls = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = __builtin_amdgcn_exp2f(s[e] - m_use); ls += s[e]; }
        l_run = l_run * alpha + ls;
        m_run = m_new;
        if (DROP == 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                s[e] = drop_keep(ds, rkq, (unsigned)(32 * kt + acc_row(e, h))) ? s[e] * ds.scale : 0.f;
        }
        if (DROP == 2) {
            const unsigned kwh = kw_cur >> (4 * h);                  // bit acc_row(e, h) = (e & 3) + 8 (e >> 2) + 4 h
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = (kwh >> ((e & 3) + 8 * (e >> 2)) & 1u) ? s[e] * ds.scale : 0.f;
        }
#pragma unroll
        for (int cb = 0; cb < DH / 32; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[cb][e] *= alpha;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
#pragma unroll
            for (int cb = 0; cb < DH / 32; ++cb) {
                const float vf = t.b[acc_row(e, h)][32 * cb + r];
                o[cb] = MFMA32(vf, s[e], o[cb]);
            }
        }
        if (kt + 1 < nkt) { sg.store(lds[(kt + 1) & 1]); side(lds[(kt + 1) & 1], 32 * (kt + 1)); }
        __syncthreads();
    }
    const float l_tot = pair_sum(l_run);
    const float inv = 1.0f / l_tot;         // l_tot == 0 (every key masked): NaN rows, like softmax of all -inf
    if (qi < T) {
        float *op = out + ((size_t)b * T + qi) * (H * DH) + hd * DH;
#pragma unroll
        for (int cb = 0; cb < DH / 32; ++cb)
#pragma unroll
            for (int tg = 0; tg < 4; ++tg) {
                f32x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = o[cb][4 * tg + e] * inv;
                *(f32x4 *)(op + 32 * cb + 8 * tg + 4 * h) = w;
            }
        if (h == 0) lse2[(size_t)bh * T + qi] = m_run + log2f(l_tot);
    }
}

// ------------------------------------------------------------------------------------------
// backward, queries own: dQ
// ------------------------------------------------------------------------------------------
template <int DH, int DROP>
__global__ __launch_bounds__(256) void attn_bwd_dq(
    const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    const uint8_t *__restrict__ mask, const float *__restrict__ dO, const float *__restrict__ lse2,
    const float *__restrict__ delta, float *__restrict__ dqkv, int H, int T, float scale, unsigned long long seed,
    unsigned site, float p, const unsigned *__restrict__ dbits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];      // 2 x TileLds<DH>: 68 KB at head dim 128
    TileLds<DH> *lds = reinterpret_cast<TileLds<DH> *>(lds_raw);
    const int nq = (T + 127) / 128, d = H * DH;
    const int bh = blockIdx.x / nq, qt = blockIdx.x - bh * nq, b = bh / H, hd = bh - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int qi = qt * 128 + wave * 32 + r, qc = qi < T ? qi : T - 1;
    const float sl2 = scale * 1.4426950408889634f;
    const float *qb = q + (size_t)bh * T * DH, *kb = k + (size_t)bh * T * DH, *vb = v + (size_t)bh * T * DH;
    const DropSite ds = drop_site(seed, site, p);
    const unsigned rkq = drop_rowkey(ds, (unsigned)(bh * T + qc));

    f32x4 qf[DH / 8], dof[DH / 8];
#pragma unroll
    for (int g = 0; g < DH / 8; ++g) {
        qf[g] = *(const f32x4 *)(qb + (size_t)qc * DH + 8 * g + 4 * h) * sl2;
        dof[g] = *(const f32x4 *)(dO + ((size_t)b * T + qc) * d + hd * DH + 8 * g + 4 * h);
    }
    const float lq = lse2[(size_t)bh * T + qc], dq_delta = delta[(size_t)bh * T + qc];
    f32x16 acc[DH / 32];
#pragma unroll
    for (int cb = 0; cb < DH / 32; ++cb) acc[cb] = zero16();

    Stager<DH> sg;
    auto side = [&](TileLds<DH> &t, int key0) __attribute__((always_inline)) {
        if (tid < 32) {
            const int key = key0 + tid;
            t.s0[tid] = (key >= T || (mask != nullptr && mask[(size_t)b * T + key])) ? NEG_INF : 0.f;
        }
    };
    const int nkt = (T + 31) / 32;
    const unsigned *bq = DROP == 2 ? dbits + ((size_t)bh * T + qc) * nkt : nullptr;
    unsigned kw = DROP == 2 ? bq[0] : 0u;
    sg.load(kb, DH, vb, DH, 0, T, 1.0f);
    sg.store(lds[0]);
    side(lds[0], 0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const TileLds<DH> &t = lds[kt & 1];
        if (kt + 1 < nkt) sg.load(kb, DH, vb, DH, 32 * (kt + 1), T, 1.0f);
        const unsigned kwh = kw >> (4 * h);
        if (DROP == 2 && kt + 1 < nkt) kw = bq[kt + 1];
        f32x16 s = st_tile<DH>(t, qf, r, h);
        f32x16 dp = zero16();
#pragma unroll
        for (int g = 0; g < DH / 8; ++g) {
            const f32x4 vf = *(const f32x4 *)&t.b[r][8 * g + 4 * h];
#pragma unroll
            for (int e = 0; e < 4; ++e) dp = MFMA32(vf[e], dof[g][e], dp);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float pe = __builtin_amdgcn_exp2f(s[e] - lq);
            float g = dp[e];
            if (DROP == 1) g = drop_keep(ds, rkq, (unsigned)(32 * kt + acc_row(e, h))) ? g * ds.scale : 0.f;
            if (DROP == 2) g = (kwh >> ((e & 3) + 8 * (e >> 2)) & 1u) ? g * ds.scale : 0.f;
            s[e] = pe * (g - dq_delta);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
#pragma unroll
            for (int cb = 0; cb < DH / 32; ++cb) {
                const float kf = t.a[acc_row(e, h)][32 * cb + r];
                acc[cb] = MFMA32(kf, s[e], acc[cb]);
            }
        }
        if (kt + 1 < nkt) { sg.store(lds[(kt + 1) & 1]); side(lds[(kt + 1) & 1], 32 * (kt + 1)); }
        __syncthreads();
    }
    if (qi < T) {
        float *op = dqkv + ((size_t)b * T + qi) * (3 * d) + hd * DH;
#pragma unroll
        for (int cb = 0; cb < DH / 32; ++cb)
#pragma unroll
            for (int tg = 0; tg < 4; ++tg) {
                f32x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = acc[cb][4 * tg + e] * scale;
                *(f32x4 *)(op + 32 * cb + 8 * tg + 4 * h) = w;
            }
    }
}

// ------------------------------------------------------------------------------------------
// backward, keys own: dK and dV
// ------------------------------------------------------------------------------------------
template <int DH, int DROP>
__global__ __launch_bounds__(256) void attn_bwd_dkdv(
    const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    const uint8_t *__restrict__ mask, const float *__restrict__ dO, const float *__restrict__ lse2,
    const float *__restrict__ delta, float *__restrict__ dqkv, int H, int T, float scale, unsigned long long seed,
    unsigned site, float p, const unsigned *__restrict__ dbits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];      // 2 x TileLds<DH>: 68 KB at head dim 128
    TileLds<DH> *lds = reinterpret_cast<TileLds<DH> *>(lds_raw);
    const int nk = (T + 127) / 128, d = H * DH;
    const int bh = blockIdx.x / nk, ktile = blockIdx.x - bh * nk, b = bh / H, hd = bh - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int ki = ktile * 128 + wave * 32 + r, kc = ki < T ? ki : T - 1;
    const float sl2 = scale * 1.4426950408889634f;
    const float *qb = q + (size_t)bh * T * DH, *kb = k + (size_t)bh * T * DH, *vb = v + (size_t)bh * T * DH;
    const float *dob = dO + (size_t)b * T * d + hd * DH;          // row stride d
    const DropSite ds = drop_site(seed, site, p);

    f32x4 kf[DH / 8], vf[DH / 8];
#pragma unroll
    for (int g = 0; g < DH / 8; ++g) {
        kf[g] = *(const f32x4 *)(kb + (size_t)kc * DH + 8 * g + 4 * h);
        vf[g] = *(const f32x4 *)(vb + (size_t)kc * DH + 8 * g + 4 * h);
    }
    const float kbias = (ki >= T || (mask != nullptr && mask[(size_t)b * T + kc])) ? NEG_INF : 0.f;
    f32x16 dk[DH / 32], dv[DH / 32];
#pragma unroll
    for (int cb = 0; cb < DH / 32; ++cb) { dk[cb] = zero16(); dv[cb] = zero16(); }

    Stager<DH> sg;
    auto side = [&](TileLds<DH> &t, int q0) __attribute__((always_inline)) {
        if (tid < 32) {
            const int qi = q0 + tid;
            const bool ok = qi < T;
            t.s0[tid] = ok ? lse2[(size_t)bh * T + qi] : __builtin_inff();      // p = exp2(s - inf) = 0 on rows >= T
            t.s1[tid] = ok ? delta[(size_t)bh * T + qi] : 0.f;
            if (DROP == 1) t.rk[tid] = drop_rowkey(ds, (unsigned)(bh * T + (ok ? qi : 0)));
        }
    };
    const int nqt = (T + 31) / 32;
    // DROP 2: the key-major copy (second half of dbits): one word per (key, 32 queries)
    const unsigned *bk = DROP == 2 ? dbits + (size_t)gridDim.x / nk * T * nqt + ((size_t)bh * T + kc) * nqt : nullptr;
    unsigned kw = DROP == 2 ? bk[0] : 0u;
    sg.load(qb, DH, dob, d, 0, T, sl2);
    sg.store(lds[0]);
    side(lds[0], 0);
    __syncthreads();
    for (int it = 0; it < nqt; ++it) {
        const TileLds<DH> &t = lds[it & 1];
        if (it + 1 < nqt) sg.load(qb, DH, dob, d, 32 * (it + 1), T, sl2);
        const unsigned kwh = kw >> (4 * h);
        if (DROP == 2 && it + 1 < nqt) kw = bk[it + 1];
        // S[query][key] - lse2[query]: the row constant is the accumulator's initial value
        f32x16 s, dp = zero16();
#pragma unroll
        for (int tg = 0; tg < 4; ++tg) {
            const f32x4 lv = *(const f32x4 *)&t.s0[8 * tg + 4 * h];
#pragma unroll
            for (int e = 0; e < 4; ++e) s[4 * tg + e] = -lv[e];
        }
#pragma unroll
        for (int g = 0; g < DH / 8; ++g) {
            const f32x4 qv = *(const f32x4 *)&t.a[r][8 * g + 4 * h];
            const f32x4 dov = *(const f32x4 *)&t.b[r][8 * g + 4 * h];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s = MFMA32(qv[e], kf[g][e], s);
                dp = MFMA32(dov[e], vf[g][e], dp);
            }
        }
        f32x16 pd;          // dropped-out probabilities (the B operand of dV)
#pragma unroll
        for (int tg = 0; tg < 4; ++tg) {
            const f32x4 dl = *(const f32x4 *)&t.s1[8 * tg + 4 * h];
            u32x4 rk4 = {0u, 0u, 0u, 0u};
            if (DROP == 1) rk4 = *(const u32x4 *)&t.rk[8 * tg + 4 * h];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * tg + e;
                const float pe = __builtin_amdgcn_exp2f(s[i] + kbias);
                float g = dp[i], pk = pe;
                if (DROP != 0) {
                    const bool keep = DROP == 1 ? drop_keep(ds, rk4[e], (unsigned)ki) : (kwh >> (e + 8 * tg) & 1u) != 0u;
                    g = keep ? g * ds.scale : 0.f;
                    pk = keep ? pe * ds.scale : 0.f;
                }
                pd[i] = pk;
                s[i] = pe * (g - dl[e]);          // dS
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
#pragma unroll
            for (int cb = 0; cb < DH / 32; ++cb) {
                const float dof = t.b[acc_row(e, h)][32 * cb + r];
                const float qf = t.a[acc_row(e, h)][32 * cb + r];
                dv[cb] = MFMA32(dof, pd[e], dv[cb]);
                dk[cb] = MFMA32(qf, s[e], dk[cb]);
            }
        }
        if (it + 1 < nqt) { sg.store(lds[(it + 1) & 1]); side(lds[(it + 1) & 1], 32 * (it + 1)); }
        __syncthreads();
    }
    if (ki < T) {
        float *op = dqkv + ((size_t)b * T + ki) * (3 * d) + hd * DH;
        const float ln2 = 0.6931471805599453f;       // dK = scale * dS^T Q = (dS^T Qs) / log2(e)
#pragma unroll
        for (int cb = 0; cb < DH / 32; ++cb)
#pragma unroll
            for (int tg = 0; tg < 4; ++tg) {
                f32x4 wk, wv;
#pragma unroll
                for (int e = 0; e < 4; ++e) { wk[e] = dk[cb][4 * tg + e] * ln2; wv[e] = dv[cb][4 * tg + e]; }
                *(f32x4 *)(op + d + 32 * cb + 8 * tg + 4 * h) = wk;
                *(f32x4 *)(op + 2 * d + 32 * cb + 8 * tg + 4 * h) = wv;
            }
    }
}

// The keep decisions of one layer's attention dropout, bit-packed twice: words [BH][T queries][W] with bit j of word w =
// keep(query, key 32 w + j), then words [BH][T keys][W] with bit j of word w = keep(query 32 w + j, key); W = ceil(T / 32).
// A wave owns 32 queries x 64 keys: lane l hashes query (l & 31) against the 32 keys of word (l >> 5); the transposed
// words come from 32 ballots (bit j of every lane's word = the column of key j).  Bits beyond T are 0.
__global__ __launch_bounds__(256) void attn_dropout_bits(unsigned *__restrict__ bits, int BH, int T, unsigned long long seed,
                                                         unsigned site, float p) {
    const int W = (T + 31) / 32, W2 = (W + 1) / 2;
    const DropSite ds = drop_site(seed, site, p);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, g = lane >> 5;
    const size_t ntile = (size_t)BH * W * W2;
    unsigned *bitsK = bits + (size_t)BH * T * W;
    for (size_t tile = (size_t)blockIdx.x * 4 + wave; tile < ntile; tile += (size_t)gridDim.x * 4) {
        const int t2 = (int)(tile % W2), qt = (int)((tile / W2) % W), bh = (int)(tile / ((size_t)W2 * W));
        const int q = 32 * qt + r, kw = 2 * t2 + g;
        unsigned word = 0u;
        if (q < T && kw < W) {
            const unsigned rk = drop_rowkey(ds, (unsigned)(bh * T + q));
#pragma unroll 8
            for (int j = 0; j < 32; ++j) {
                const int key = 32 * kw + j;
                word |= (key < T && drop_keep(ds, rk, (unsigned)key)) ? (1u << j) : 0u;
            }
            bits[((size_t)bh * T + q) * W + kw] = word;
        }
        unsigned mine = 0u;
#pragma unroll 8
        for (int j = 0; j < 32; ++j) {
            const unsigned long long bal = __ballot((word >> j) & 1u);
            if (r == j) mine = g == 0 ? (unsigned)bal : (unsigned)(bal >> 32);
        }
        const int key = 64 * t2 + 32 * g + r;       // lane (r, g) holds the queries-of-this-tile word of key 64 t2 + 32 g + r
        if (key < T) bitsK[((size_t)bh * T + key) * W + qt] = mine;
    }
}

__global__ __launch_bounds__(256) void attn_dropout_mask(uint8_t *__restrict__ keep, int BH, int T, unsigned long long seed,
                                                         unsigned site, float p) {
    const DropSite ds = drop_site(seed, site, p);
    const size_t total = (size_t)BH * T * T;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const unsigned j = (unsigned)(i % T), row = (unsigned)(i / T);       // row = bh * T + query
        keep[i] = drop_keep(ds, drop_rowkey(ds, row), j) ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void rows_dropout_mask(uint8_t *__restrict__ keep, int M, int cols, unsigned long long seed,
                                                         unsigned site, float p) {
    const DropSite ds = drop_site(seed, site, p);
    const size_t total = (size_t)M * cols;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const unsigned c = (unsigned)(i % cols), row = (unsigned)(i / cols);
        keep[i] = drop_keep(ds, drop_rowkey(ds, row), c) ? 1 : 0;
    }
}

}  // namespace

// dynamic LDS above 64 KB needs hipFuncAttributeMaxDynamicSharedMemorySize, a per-DEVICE attribute: set once per
// (kernel instantiation, device) - `done` is the call site's own flag array
static int allow_lds(const void *kernel, size_t bytes, std::atomic<unsigned char> *done) {
    if (bytes <= 64 * 1024) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipErrorInvalidDevice;
    if (dev >= 0 && dev < 64 && done[dev].load(std::memory_order_acquire)) return 0;
    const int rc = (int)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (rc == 0 && dev >= 0 && dev < 64) done[dev].store(1, std::memory_order_release);
    return rc;
}

#define VST_ATTN_LAUNCH(KERNEL_, DH_, DROP_, ...)                                                               \
    do {                                                                                                        \
        static std::atomic<unsigned char> done_[64];                                                            \
        constexpr size_t lds_bytes_ = 2 * sizeof(TileLds<DH_>);                                                 \
        if (const int rc_ = allow_lds((const void *)KERNEL_<DH_, DROP_>, lds_bytes_, done_)) return rc_;        \
        hipLaunchKernelGGL((KERNEL_<DH_, DROP_>), grid, dim3(256), lds_bytes_, st, __VA_ARGS__);                \
    } while (0)
#define VST_ATTN_DH(KERNEL_, DH_, ...)                                                                          \
    do {                                                                                                        \
        if (!(p > 0.f)) VST_ATTN_LAUNCH(KERNEL_, DH_, 0, __VA_ARGS__);                                          \
        else if (dbits != nullptr) VST_ATTN_LAUNCH(KERNEL_, DH_, 2, __VA_ARGS__);                               \
        else VST_ATTN_LAUNCH(KERNEL_, DH_, 1, __VA_ARGS__);                                                     \
    } while (0)
#define VST_ATTN_DISPATCH(KERNEL_, ...)                                                                         \
    do {                                                                                                        \
        if (dh == 32) VST_ATTN_DH(KERNEL_, 32, __VA_ARGS__);                                                    \
        else if (dh == 64) VST_ATTN_DH(KERNEL_, 64, __VA_ARGS__);                                               \
        else if (dh == 128) VST_ATTN_DH(KERNEL_, 128, __VA_ARGS__);                                             \
        else if (dh == 256) VST_ATTN_DH(KERNEL_, 256, __VA_ARGS__);      /* round 4: correctness first (1 wave / SIMD, spills) */ \
        else return -1;                                                                                         \
    } while (0)

size_t vst_attention_dropout_bits_words(int B, int H, int T) { return 2 * (size_t)B * H * T * ((T + 31) / 32); }

// dbits (vst_attention_dropout_bits_words(B, H, T) words) <- both bit-packed copies of this layer's keep decisions
int vst_attention_dropout_bits(unsigned *dbits, int B, int H, int T, unsigned long long seed, unsigned site, float p,
                               hipStream_t st) {
    const int W = (T + 31) / 32;
    const size_t ntile = (size_t)B * H * W * ((W + 1) / 2);
    const int blocks = (int)((ntile + 3) / 4 < 16384 ? (ntile + 3) / 4 : 16384);
    hipLaunchKernelGGL(attn_dropout_bits, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, st, dbits, B * H, T, seed, site, p);
    VSK_CHECK_LAUNCH();
    return 0;
}

// dbits: the bit-packed keep masks (vst_attention_dropout_bits) or nullptr (the kernels then hash per element: same masks)
int vst_attention_fwd(const float *q, const float *k, const float *v, const uint8_t *mask, float *out, float *lse2,
                      int B, int H, int T, int dh, float scale, unsigned long long seed, unsigned site, float p,
                      hipStream_t st, const unsigned *dbits) {
    if (p < 0.f || p >= 1.f) return -1;
    const dim3 grid(B * H * ((T + 127) / 128));
    VST_ATTN_DISPATCH(attn_fwd_train, q, k, v, mask, out, lse2, H, T, scale, seed, site, p, dbits);
    VSK_CHECK_LAUNCH();
    return 0;
}

int vst_attention_bwd(const float *q, const float *k, const float *v, const uint8_t *mask, const float *dO,
                      const float *lse2, const float *delta, float *dqkv, int B, int H, int T, int dh, float scale,
                      unsigned long long seed, unsigned site, float p, hipStream_t st, const unsigned *dbits) {
    if (p < 0.f || p >= 1.f) return -1;
    const dim3 grid(B * H * ((T + 127) / 128));
    VST_ATTN_DISPATCH(attn_bwd_dkdv, q, k, v, mask, dO, lse2, delta, dqkv, H, T, scale, seed, site, p, dbits);
    VSK_CHECK_LAUNCH();
    VST_ATTN_DISPATCH(attn_bwd_dq, q, k, v, mask, dO, lse2, delta, dqkv, H, T, scale, seed, site, p, dbits);
    VSK_CHECK_LAUNCH();
    return 0;
}

int vst_attention_dropout_mask(uint8_t *keep, int B, int H, int T, unsigned long long seed, unsigned site, float p,
                               hipStream_t st) {
    const size_t total = (size_t)B * H * T * T;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(attn_dropout_mask, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, st, keep, B * H, T, seed, site, p);
    VSK_CHECK_LAUNCH();
    return 0;
}

int vst_rows_dropout_mask(uint8_t *keep, int M, int cols, unsigned long long seed, unsigned site, float p, hipStream_t st) {
    const size_t total = (size_t)M * cols;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(rows_dropout_mask, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, st, keep, M, cols, seed, site, p);
    VSK_CHECK_LAUNCH();
    return 0;
}
