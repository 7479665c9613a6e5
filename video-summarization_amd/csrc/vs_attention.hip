// vs_attention.hip — gfx950 attention kernels of the frame-importance scorer (reference simnet.py:155-161):
//   attn_fwd            exact fp32, head dim 128 (and the A/B baseline for 32/64)
//   attn_fwd_pipe       exact fp32, software-pipelined, head dim 32/64 (the default path)
//   attn_fwd_lp(_pipe)  opt-in low-precision matrix pipes: bf16 operands, or fp32 emulated with f16 hi+lo halves
#include "vs_device.h"
#include "vs_kernels.h"

namespace {

// ------------------------------------------------------------------------------------------
// Attention: softmax(q k^T * scale + keymask) v without materialising [T,T].
//   grid = (ceil(T/128), B*H); 4 waves, each owns 32 query rows and walks all key tiles.
//   Both products keep the QUERY on the lane: S^T = K * Q^T  (A = K tile from LDS, B = Q in
//   registers) leaves, for query r, 16 keys per register set; O^T = V^T * P^T then takes that
//   accumulator register t directly as its B operand (keys (t&3)+8(t>>2)+4h — exactly the
//   k-pair of MFMA step t) with A = V[key][d-column] read from LDS.  So the softmax row
//   statistics (max, sum, rescale) are lane-local plus one exchange with lane^32, and P never
//   leaves registers.
// ------------------------------------------------------------------------------------------
template <int DH, int NKB, bool VARLEN = false>   // NKB 32-key blocks per tile; VARLEN: packed ragged batches (see attn_fwd_pipe), round 4
__global__ __launch_bounds__(256, 2) void attn_fwd(
    const float *__restrict__ Q, const float *__restrict__ Kg, const float *__restrict__ Vg,
    const uint8_t *__restrict__ mask, float *__restrict__ out, int H, int T, float scale_log2e, int BH,
    const int *__restrict__ cu = nullptr, const int2 *__restrict__ work = nullptr, int Mtot = 0) {
    constexpr int KT = 32 * NKB, LD = DH + 4, NJ = DH / 8, ND = DH / 32;
    constexpr int F4 = KT * DH / 4 / 256;          // float4 per thread per operand tile
    __shared__ __attribute__((aligned(16))) float Ks[KT * LD];
    __shared__ __attribute__((aligned(16))) float Vs[KT * LD];
    __shared__ float mb[KT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    int b, head, qt;
    size_t base, orow0;                              // operand base (floats), first output row of this video
    if constexpr (VARLEN) {
        const int2 wk = work[blockIdx.x];
        b = wk.x; qt = wk.y; head = blockIdx.y;
        const int c0 = cu[b];
        T = cu[b + 1] - c0;
        base = ((size_t)head * Mtot + c0) * DH;
        orow0 = (size_t)c0;
    } else {
        int bh;
        if (!attn_block_map((T + 127) / 128, BH, bh, qt)) return;
        b = bh / H; head = bh - b * H;
        base = (size_t)bh * T * DH;
        orow0 = (size_t)b * T;
    }
    const int q0 = qt * 128 + 32 * wave;
    const float NEG_INF = -__builtin_inff();

    // Q fragment (B operand), pre-multiplied by scale*log2(e) so that p = exp2(s - m)
    float qreg[4 * NJ];
    {
        int qr = q0 + r; qr = qr < T ? qr : T - 1;
        const float *qp = Q + base + (size_t)qr * DH + 4 * h;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const f32x4 v = *(const f32x4 *)(qp + 8 * j);
#pragma unroll
            for (int s = 0; s < 4; ++s) qreg[4 * j + s] = v[s] * scale_log2e;
        }
    }

    f32x16 o[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int t = 0; t < 16; ++t) o[d][t] = 0.f;
    float m_run = NEG_INF, l_run = 0.f;

    const int ntiles = (T + KT - 1) / KT;
    f32x4 pk[F4], pv[F4];
    auto prefetch = [&](int tile) {
#pragma unroll
        for (int i = 0; i < F4; ++i) {
            const int idx = tid + 256 * i;               // float4 index inside the tile
            int row = tile * KT + idx / (DH / 4);
            row = row < T ? row : T - 1;
            const size_t off = base + (size_t)row * DH + (idx % (DH / 4)) * 4;
            pk[i] = *(const f32x4 *)(Kg + off);
            pv[i] = *(const f32x4 *)(Vg + off);
        }
    };
    prefetch(0);

    for (int tile = 0; tile < ntiles; ++tile) {
        const int k0 = tile * KT;
#pragma unroll
        for (int i = 0; i < F4; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / (DH / 4), c = (idx % (DH / 4)) * 4;
            *(f32x4 *)&Ks[row * LD + c] = pk[i];
            *(f32x4 *)&Vs[row * LD + c] = pv[i];
        }
        if (tid < KT) {
            const int key = k0 + tid;
            bool dead = key >= T;
            if (!dead && mask != nullptr) dead = mask[(size_t)b * T + key] != 0;
            mb[tid] = dead ? NEG_INF : 0.f;
        }
        __syncthreads();
        if (tile + 1 < ntiles) prefetch(tile + 1);

        // ---- S^T = K * Q^T ----
        f32x16 s[NKB];
#pragma unroll
        for (int n = 0; n < NKB; ++n)
#pragma unroll
            for (int t = 0; t < 16; ++t) s[n][t] = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f32x4 ka[NKB];
#pragma unroll
            for (int n = 0; n < NKB; ++n) ka[n] = *(const f32x4 *)&Ks[(32 * n + r) * LD + 8 * j + 4 * h];
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int n = 0; n < NKB; ++n) s[n] = MFMA32(ka[n][st], qreg[4 * j + st], s[n]);
        }
        // ---- key mask (padding mask and the ragged tail) ----
        if (mask != nullptr || k0 + KT > T) {
#pragma unroll
            for (int n = 0; n < NKB; ++n)
#pragma unroll
                for (int t = 0; t < 16; ++t) s[n][t] += mb[32 * n + acc_row(t, h)];
        }
        // ---- online softmax, one query per lane pair (l, l^32) ----
        float mx = NEG_INF;
#pragma unroll
        for (int n = 0; n < NKB; ++n)
#pragma unroll
            for (int t = 0; t < 16; ++t) mx = fmaxf(mx, s[n][t]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == NEG_INF) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        float psum = 0.f;
#pragma unroll
        for (int n = 0; n < NKB; ++n)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float p = __builtin_amdgcn_exp2f(s[n][t] - m_use);
                s[n][t] = p;
                psum += p;
            }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int t = 0; t < 16; ++t) o[d][t] *= alpha;
        // ---- O^T += V^T * P^T ----
#pragma unroll
        for (int d = 0; d < ND; ++d) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
#pragma unroll
                for (int n = 0; n < NKB; ++n) {
                    const float va = Vs[(32 * n + acc_row(t, h)) * LD + 32 * d + r];
                    o[d] = MFMA32(va, s[n][t], o[d]);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: O^T[d][q] / l -> out[b, q, head*DH + d]; 4 consecutive d per 16-B store ----
    const int q = q0 + r;
    if (q < T) {
        const float inv = 1.0f / l_run;
        float *op = out + (orow0 + q) * (H * DH) + head * DH;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = o[d][4 * g + e] * inv;
                *(f32x4 *)(op + 32 * d + 8 * g + 4 * h) = v;
            }
    }
}

// ------------------------------------------------------------------------------------------
// Latency mode (VS_FLAG_SPLITK, round 4): exact fp32 attention for ONE reference-sized video per call.  attn_fwd_pipe gives a
// wave 32 query rows and ALL keys: at T = 320 that is 12 blocks on 256 CUs, each wave walking five 64-key tiles one after
// the other (27 us).  Here a block is 32 query rows of one head and its 8 waves SPLIT THE KEYS: wave w takes the 32-key tiles
// w, w + 8, ...; operands come straight from global memory into fragments (no LDS tile, no barrier in the loop), every
// wave keeps its own (m, l, O), and the eight partial results are merged once through LDS in wave order - the flash-decoding
// split, with a fixed tile assignment and a fixed merge order: deterministic and independent of the batch, but not the bits of
// attn_fwd_pipe (tests: goldens at 1e-4).  Same operand trick as attn_fwd (S^T = K Q^T, its accumulator is the B operand of
// O^T = V^T P^T).
// ------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(512) void attn_fwd_splitkv(
    const float *__restrict__ Q, const float *__restrict__ Kg, const float *__restrict__ Vg,
    const uint8_t *__restrict__ mask, float *__restrict__ out, int H, int T, float scale_log2e, int BH) {
    constexpr int NW = 8, NJ = DH / 8, ND = DH / 32;
    __shared__ float ml[NW][2][32];
    __shared__ float linv[32];
    __shared__ __attribute__((aligned(16))) float ob[NW][16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nq = (T + 31) / 32;
    const int qt = blockIdx.x % nq, bh = blockIdx.x / nq;
    if (bh >= BH) return;
    const int b = bh / H, head = bh - b * H;
    const size_t base = (size_t)bh * T * DH;
    const int q0 = qt * 32;
    const float NEG_INF = -__builtin_inff();

    float qreg[4 * NJ];
    {
        int qr = q0 + r; qr = qr < T ? qr : T - 1;
        const float *qp = Q + base + (size_t)qr * DH + 4 * h;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const f32x4 v = *(const f32x4 *)(qp + 8 * j);
#pragma unroll
            for (int st = 0; st < 4; ++st) qreg[4 * j + st] = v[st] * scale_log2e;
        }
    }
    f32x16 o[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int t = 0; t < 16; ++t) o[d][t] = 0.f;
    float m_run = NEG_INF, l_run = 0.f;

    const int ntiles = (T + 31) / 32;
    for (int tile = wave; tile < ntiles; tile += NW) {
        const int k0 = tile * 32;
        // K fragments in halves of <= 8 k-groups and V one 32-column block ahead (head dim 128: 64 + 32 + 32 + 64 + 16 registers)
        constexpr int NJH = NJ < 8 ? NJ : 8;
        int kr = k0 + r; kr = kr < T ? kr : T - 1;
        const float *kp = Kg + base + (size_t)kr * DH + 4 * h;
        f32x4 ka[NJH];
#pragma unroll
        for (int j = 0; j < NJH; ++j) ka[j] = *(const f32x4 *)(kp + 8 * j);
        const float *vbase = Vg + base + r;
        int voff[16];                                  // (offsets inside this (video, head) block: T * DH < 2^31)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            int key = k0 + acc_row(t, h); key = key < T ? key : T - 1;
            voff[t] = key * DH;
        }
        float va[2][16];
#pragma unroll
        for (int t = 0; t < 16; ++t) va[0][t] = vbase[voff[t]];
        f32x16 s;
#pragma unroll
        for (int t = 0; t < 16; ++t) s[t] = 0.f;
#pragma unroll
        for (int jh = 0; jh < NJ; jh += NJH) {
            f32x4 kn[NJH];
            if (jh + NJH < NJ) {
#pragma unroll
                for (int j = 0; j < NJH; ++j) kn[j] = *(const f32x4 *)(kp + 8 * (jh + NJH + j));
            }
#pragma unroll
            for (int j = 0; j < NJH; ++j)
#pragma unroll
                for (int st = 0; st < 4; ++st) s = MFMA32(ka[j][st], qreg[4 * (jh + j) + st], s);
            if (jh + NJH < NJ) {
#pragma unroll
                for (int j = 0; j < NJH; ++j) ka[j] = kn[j];
            }
        }
        if (mask != nullptr || k0 + 32 > T) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int key = k0 + acc_row(t, h);
                bool dead = key >= T;
                if (!dead && mask != nullptr) dead = mask[(size_t)b * T + key] != 0;
                s[t] = dead ? NEG_INF : s[t];
            }
        }
        float mx = NEG_INF;
#pragma unroll
        for (int t = 0; t < 16; ++t) mx = fmaxf(mx, s[t]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == NEG_INF) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float p = __builtin_amdgcn_exp2f(s[t] - m_use);
            s[t] = p;
            psum += p;
        }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            if (d + 1 < ND) {
#pragma unroll
                for (int t = 0; t < 16; ++t) va[(d + 1) & 1][t] = vbase[voff[t] + 32 * (d + 1)];
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) o[d][t] *= alpha;
#pragma unroll
            for (int t = 0; t < 16; ++t) o[d] = MFMA32(va[d & 1][t], s[t], o[d]);
        }
    }

    // ---- merge the eight partial results, wave order ----
    if (h == 0) { ml[wave][0][r] = m_run; ml[wave][1][r] = l_run; }
    __syncthreads();
    float m_all = NEG_INF;
#pragma unroll
    for (int w = 0; w < NW; ++w) m_all = fmaxf(m_all, ml[w][0][r]);
    const float m_use = (m_all == NEG_INF) ? 0.f : m_all;
    float l_tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) l_tot += ml[w][1][r] * __builtin_amdgcn_exp2f(ml[w][0][r] - m_use);
    if (wave == 0 && h == 0) linv[r] = 1.0f / l_tot;
    const float f = __builtin_amdgcn_exp2f(m_run - m_use);
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        if (d) __syncthreads();
#pragma unroll
        for (int t = 0; t < 16; ++t) ob[wave][t * 64 + lane] = o[d][t] * f;
        __syncthreads();
        if (tid < 256) {
            const int g = tid >> 6, ln = tid & 63, rr = ln & 31, hh = ln >> 5;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float acc = ob[0][(4 * g + e) * 64 + ln];
#pragma unroll
                for (int w = 1; w < NW; ++w) acc += ob[w][(4 * g + e) * 64 + ln];
                v[e] = acc * linv[rr];
            }
            const int q = q0 + rr;
            if (q < T) *(f32x4 *)(out + ((size_t)b * T + q) * (H * DH) + head * DH + 32 * d + 8 * g + 4 * hh) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Attention on the bf16 matrix pipe (opt-in, VS_FLAG_BF16_ATTENTION; long videos): the same
// flash-style walk and operand trick as attn_fwd, but both products run as
// v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate).  Q*scale, K, V and the probabilities P are
// rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) on their way into the MFMA; scores,
// softmax statistics and the output accumulate in fp32.  Inputs and output stay fp32 in HBM.
//   S^T = K * Q^T : A = K[key r][d = 16s+8h+j] (one ds_read_b128 of the bf16 tile), B = Q in registers.
//   O^T = V^T * P^T: B = registers 8s..8s+7 of the S^T accumulator packed pairwise — element j of
//   lane half h is key 16s + 8(j>>2) + 4h + (j&3) — and A = V^T[d r][those keys], two ds_read_b64
//   of the TRANSPOSED bf16 V tile, which the staging writes (4 keys of one d packed per b64 store).
// The row sums l are a third "V" block of ones (the matrix pipe has slack, the VALU does not); the running
// max is deferred (raised only on a jump > 2^8, so the O/l rescale almost never runs); K/V tiles arrive by
// buffer loads with scalar offsets, are rounded once into a double-buffered LDS tile, one barrier per tile.
// Measured (T=8192, B=8, M-A): 750 TFLOP/s; per 64-key tile and wave 20 MFMAs (640 cycles) + 32 v_exp_f32
// (quarter rate, 512 cycles) + ~100 VALU - and the SIMD issues them one after the other, so the exp2 of the
// softmax, not the matrix pipe, bounds this kernel at head dim 64.
// NOT within the 1e-4 fp32 bar of the reference: tests/test_hip_parity.py states its tolerance.
// ------------------------------------------------------------------------------------------
// PREC 2 ("fp16x3", VS_FLAG_F16X3_ATTENTION) EMULATES the fp32 products on the f16 pipe instead: every operand
// (q*scale, k, v, p) is split into f16 hi + lo halves (split_f16) kept in two LDS planes / register sets, and
// each product is three MFMAs (lo*hi, hi*lo, hi*hi; fp32 accumulate) - 56 MFMAs of 32 cycles per 64-key tile
// where the fp32 kernel needs 132 of 64 cycles - with results inside the fp32 path's own 1e-4 bar.
template <int DH, int NW, int PREC>       // NW waves per block, 32 query rows each; PREC 1: bf16, 2: f16 hi+lo
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_lp(
    const float *__restrict__ Q, const float *__restrict__ Kg, const float *__restrict__ Vg,
    const uint8_t *__restrict__ mask, float *__restrict__ out, int H, int T, float scale_log2e, int BH) {
    constexpr int KT = 64, NS = DH / 16, ND = DH / 32, NT = 64 * NW, QB = 32 * NW;
    constexpr int NP = PREC == 2 ? 2 : 1;          // operand planes (hi, lo)
    constexpr int LDK = DH + 8;                    // 16-bit elements per K row: 36 (DH 64) / 20 (DH 32) dwords, b128 reads conflict-free
    constexpr int LDV = KT + 4;                    // 16-bit elements per V^T row: 34 dwords, b64 reads conflict-free
    constexpr int D4 = DH / 4;                     // float4 per key row
    constexpr int KPT = KT * D4 / NT;              // keys per thread in the staging, one float4 of d each
    static_assert(KPT == 2 || KPT == 4, "staging packs 2 or 4 keys per V^T store");
    constexpr float THR = 8.0f;                    // deferred max: the applied max is raised only on a jump > 2^8
    typedef unsigned short h16;
    // K tile, V^T tile and key-mask bias, double-buffered: tile t+1 is written while tile t is consumed,
    // one block barrier per tile
    __shared__ __attribute__((aligned(16))) h16 Kb[2][NP][KT * LDK];
    __shared__ __attribute__((aligned(16))) h16 Vt[2][NP][DH * LDV];
    __shared__ __attribute__((aligned(16))) float mb[2][KT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    int bh, qt;
    if (!attn_block_map((T + QB - 1) / QB, BH, bh, qt)) return;
    const int b = bh / H, head = bh - b * H;
    const size_t base = (size_t)bh * T * DH;
    const int q0 = qt * QB + 32 * wave;
    const float NEG_INF = -__builtin_inff();

    // two floats -> one packed pair per plane
    auto pack = [&](float x0, float x1, unsigned (&pl)[NP]) __attribute__((always_inline)) {
        if constexpr (PREC == 2) split_f16(x0, x1, pl[0], pl[1]);
        else pl[0] = pack_bf16(x0, x1);
    };
    auto mma = [&](const u32x4 &a, const u32x4 &bq, const f32x16 &c) __attribute__((always_inline)) -> f32x16 {
        if constexpr (PREC == 2) return MFMA_F16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, bq), c);
        else return MFMA_BF16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bq), c);
    };
    // acc += A * B over the planes: hi*hi (+ lo*hi + hi*lo, small terms first)
    auto mma_planes = [&](const u32x4 (&a)[NP], const u32x4 (&bq)[NP], f32x16 &c) __attribute__((always_inline)) {
        if constexpr (PREC == 2) {
            c = mma(a[1], bq[0], c);
            c = mma(a[0], bq[1], c);
        }
        c = mma(a[0], bq[0], c);
    };

    // Q fragments (B operand): Q[q][16s + 8h + j] * scale*log2(e)
    u32x4 qreg[NS][NP];
    {
        int qr = q0 + r; qr = qr < T ? qr : T - 1;
        const float *qp = Q + base + (size_t)qr * DH + 8 * h;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const f32x4 v0 = *(const f32x4 *)(qp + 16 * s), v1 = *(const f32x4 *)(qp + 16 * s + 4);
            unsigned pl[4][NP];
            pack(v0[0] * scale_log2e, v0[1] * scale_log2e, pl[0]);
            pack(v0[2] * scale_log2e, v0[3] * scale_log2e, pl[1]);
            pack(v1[0] * scale_log2e, v1[1] * scale_log2e, pl[2]);
            pack(v1[2] * scale_log2e, v1[3] * scale_log2e, pl[3]);
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int e = 0; e < 4; ++e) qreg[s][p][e] = pl[e][p];
        }
    }
    // o[0..ND-1] = O^T blocks; o[ND] = the row sums l, as the product of P with a block of ones
    // (the matrix pipe has slack, the VALU does not; and l then sums exactly the rounded P the output uses)
    f32x16 o[ND + 1];
#pragma unroll
    for (int d = 0; d <= ND; ++d)
#pragma unroll
        for (int t = 0; t < 16; ++t) o[d][t] = 0.f;
    float m_run = NEG_INF;                         // the max actually applied to O and l (log2 units)
    constexpr unsigned ONE2 = PREC == 2 ? 0x3C003C00u : 0x3F803F80u;        // (1.0, 1.0) in f16 / bf16
    const u32x4 ones_u = {ONE2, ONE2, ONE2, ONE2};

    // staging: thread (d4 = tid % D4, kq = tid / D4) owns keys KPT*kq .. +KPT-1 at d = 4*d4 .. +3.
    // Buffer loads: the per-tile offset is a scalar, rows beyond T read as zero (and are masked below).
    const int d4 = tid % D4, kq = tid / D4;
    const int ntiles = (T + KT - 1) / KT;
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Kg + base), 0, T * DH * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Vg + base), 0, T * DH * 4, 0x00020000);
    int voff[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) voff[i] = ((KPT * kq + i) * DH + 4 * d4) * 4;
    f32x4 pk[KPT], pv[KPT];
    float pm = 0.f;
    auto gload = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            pk[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, voff[i], tile * (KT * DH * 4), 0));
            pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vrs, voff[i], tile * (KT * DH * 4), 0));
        }
        if (tid < KT) {                             // key-mask bias of key tile*KT + tid: 0 or -inf (also beyond T)
            const int key = tile * KT + tid;
            float pmv = key >= T ? NEG_INF : 0.f;
            if (mask != nullptr) pmv = mask[(size_t)b * T + (key < T ? key : T - 1)] != 0 ? NEG_INF : pmv;
            pm = pmv;
        }
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            unsigned p0[NP], p1[NP];
            pack(pk[i][0], pk[i][1], p0);
            pack(pk[i][2], pk[i][3], p1);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                u32x2 u; u[0] = p0[p]; u[1] = p1[p];
                *(u32x2 *)&Kb[buf][p][(KPT * kq + i) * LDK + 4 * d4] = u;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (KPT == 4) {
                unsigned p0[NP], p1[NP];
                pack(pv[0][e], pv[1][e], p0);
                pack(pv[2][e], pv[3][e], p1);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    u32x2 u; u[0] = p0[p]; u[1] = p1[p];
                    *(u32x2 *)&Vt[buf][p][(4 * d4 + e) * LDV + 4 * kq] = u;
                }
            } else {
                unsigned p0[NP];
                pack(pv[0][e], pv[1][e], p0);
#pragma unroll
                for (int p = 0; p < NP; ++p) *(unsigned *)&Vt[buf][p][(4 * d4 + e) * LDV + 2 * kq] = p0[p];
            }
        }
        if (tid < KT) mb[buf][tid] = pm;
    };
    gload(0);
    stage(0);
    if (ntiles > 1) gload(1);
    __syncthreads();

    for (int tile = 0; tile < ntiles; ++tile) {
        const int cur = tile & 1;
        const bool masked_tile = mask != nullptr || (tile + 1) * KT > T;
        // ---- S^T = K * Q^T ----
        f32x16 s[2];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int t = 0; t < 16; ++t) s[n][t] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NS; ++ks)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                u32x4 ka[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) ka[p] = *(const u32x4 *)&Kb[cur][p][(32 * n + r) * LDK + 16 * ks + 8 * h];
                mma_planes(ka, qreg[ks], s[n]);
            }
        // ---- V^T fragments of this tile (A of O^T): reads in flight under the softmax ----
        u32x4 va[2][2][ND][NP];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int d = 0; d < ND; ++d)
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const h16 *vp = &Vt[cur][p][(32 * d + r) * LDV + 32 * n + 16 * ks + 4 * h];
                        const u32x2 lo = *(const u32x2 *)vp, hi = *(const u32x2 *)(vp + 8);
                        va[n][ks][d][p][0] = lo[0]; va[n][ks][d][p][1] = lo[1]; va[n][ks][d][p][2] = hi[0]; va[n][ks][d][p][3] = hi[1];
                    }
        // ---- tile t+1 into the other LDS buffer, tile t+2 into registers (under the MFMAs / softmax) ----
        if (tile + 1 < ntiles) stage(cur ^ 1);
        if (tile + 2 < ntiles) gload(tile + 2);
        if (masked_tile) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bv = *(const f32x4 *)&mb[cur][32 * n + 8 * g + 4 * h];
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[n][4 * g + e] += bv[e];
                }
        }
        // ---- online softmax with a deferred max, one query per lane pair (l, l^32) ----
        float mx = __builtin_fmaxf(__builtin_fmaxf(s[0][0], s[0][1]), s[1][0]);
        mx = __builtin_fmaxf(mx, s[1][1]);
#pragma unroll
        for (int t = 2; t < 16; t += 2) {
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[0][t]), s[0][t + 1]);
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[1][t]), s[1][t + 1]);
        }
        mx = pair_max(mx);
        const bool raise = mx > m_run + THR || (m_run == NEG_INF && mx != NEG_INF);
        if (__builtin_expect(__any(raise), 0)) {   // first live tile, or a jump > 2^THR: rare, wave-uniform branch
            const float m_new = raise ? mx : m_run;
            const float u_new = (m_new == NEG_INF) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - u_new);    // 1 for lanes that keep their max; 0 from -inf
#pragma unroll
            for (int d = 0; d <= ND; ++d)
#pragma unroll
                for (int t = 0; t < 16; ++t) o[d][t] *= alpha;
            m_run = m_new;
        }
        const float m_use = (m_run == NEG_INF) ? 0.f : m_run;           // p = exp2(s - m_use) <= 2^THR
        const f32x2 mm = {m_use, m_use};
        // ---- O^T += V^T * P^T, l += 1 * P^T ----
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 pf[NP];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2 sv = {s[n][8 * ks + 2 * j], s[n][8 * ks + 2 * j + 1]};
                    const f32x2 dv = sv - mm;
                    unsigned pl[NP];
                    pack(__builtin_amdgcn_exp2f(dv[0]), __builtin_amdgcn_exp2f(dv[1]), pl);
#pragma unroll
                    for (int p = 0; p < NP; ++p) pf[p][j] = pl[p];
                }
#pragma unroll
                for (int d = 0; d < ND; ++d) mma_planes(va[n][ks][d], pf, o[d]);
#pragma unroll
                for (int p = NP - 1; p >= 0; --p) o[ND] = mma(ones_u, pf[p], o[ND]);
            }
        __syncthreads();
    }

    // ---- epilogue: O^T[d][q] / l -> out[b, q, head*DH + d]; 4 consecutive d per 16-B store ----
    const int q = q0 + r;
    if (q < T) {
        const float inv = 1.0f / o[ND][0];
        float *op = out + ((size_t)b * T + q) * (H * DH) + head * DH;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = o[d][4 * g + e] * inv;
                *(f32x4 *)(op + 32 * d + 8 * g + 4 * h) = v;
            }
    }
}

// ------------------------------------------------------------------------------------------
// attn_fwd_lp, software-pipelined (the default for the bf16 / fp16x3 modes): attn_fwd_lp keeps every wave in
// the same phase (the per-tile barrier aligns them), so the matrix pipe idles during the softmax and the VALU
// during the products.  Here K runs one tile ahead of V: iteration t issues the MFMAs of S(t+1) = K(t+1) Q^T
// with the softmax of tile t (max3 chain, deferred-max check, exp2, hi/lo split) sliced between them, then the
// MFMAs of O += V(t)^T P(t)^T (+ row sums) with the staging of K(t+2) / V(t+1) (split, LDS writes) and the
// buffer loads of K(t+3) / V(t+2) between them.  One barrier per tile; K and V double-buffered in LDS, the
// key-mask bias triple-buffered (it is read two iterations after it is written).
// ------------------------------------------------------------------------------------------
// Round 3 (VALU diet; per 64-key tile and wave 105 -> ~70 vector instructions at bf16):
//  * the row constant c (the "running max", log2 units) is FOLDED INTO THE PRODUCT: S' = [K,1][Q,-c]^T, one extra MFMA
//    per 32-key block (k = 16 step whose only non-zero column is the ones / -c pair) replaces the accumulator
//    zero-init and the 32 subtractions in front of exp2.  c is kept exactly representable in the operand type (rounded
//    UP to a bf16 / f16 value), so the step adds exactly -c;
//  * NO max chain in the steady state: P = exp2(S') is computed straight away, and an OR over the packed P registers
//    tells whether any P >= 2 (bit 14 of a non-negative bf16 / f16 number): only then - a key beat the row constant by
//    more than 1 - the rare path recomputes the tile's maximum, raises c, rescales O and l and redoes the tile's P.
//    The constant is set CMARGIN = 6 above the maximum seen, so P <= 2^-6 after a raise and the next raise needs a
//    key 2^7 above that maximum (round 2's deferred maximum, THR 8, with the test moved from S to the packed P);
//  * a block of S' that was started with a row constant that has been raised since ("stale bias": the MFMAs of tile
//    t + 1 run while tile t's softmax may raise c) is shifted by the difference first (rare path as well).
// IO16 (PREC 1 only): q (already multiplied by scale * log2 e), k, v arrive as bf16 and the output is written as bf16
// (the QKV epilogue / the out-projection do the rounding this kernel / that kernel would do anyway: same bits)
// DIAG (diagnostic library only): per-wave stamps, diag[(block * NW + wave) * 16 + ..]: 0 prologue, 1 phase A (S(t+1) ||
// softmax(t)), 2 phase B (P.V + row sums || staging), 3 barrier, 7 epilogue, 8 total, 9 / 10 wall clock, 11 tiles
// ABL (diagnostic library only, timing ablations with WRONG results): 1 no staging inside the loop (no global loads,
// no LDS writes), 2 no block barriers inside the loop, 4 no fragment reads inside the loop (LDS offsets pinned to tile 0
// and hoistable), 8 no softmax work (P fragments constant)
template <int DH, int NW, int PREC, bool VARLEN = false, bool IO16 = false, int DIAG = 0, int ABL = 0>       // VARLEN: packed ragged batches, see attn_fwd_pipe
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_lp_pipe(
    const float *__restrict__ Q, const float *__restrict__ Kg, const float *__restrict__ Vg,
    const uint8_t *__restrict__ mask, float *__restrict__ out, int H, int T, float scale_log2e, int BH,
    const int *__restrict__ cu = nullptr, const int2 *__restrict__ work = nullptr, int Mtot = 0,
    unsigned long long *__restrict__ diag = nullptr) {
    unsigned long long dg[12] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}, tbeg = 0, tl = 0;
    if constexpr (DIAG != 0) { dg[9] = __builtin_amdgcn_s_memrealtime(); tbeg = vs_stamp(); tl = tbeg; }
    auto lap = [&](int slot) __attribute__((always_inline)) {
        if constexpr (DIAG != 0) { const unsigned long long t = vs_stamp(); dg[slot] += t - tl; tl = t; }
    };
    constexpr int KT = 64, NS = DH / 16, ND = DH / 32, NT = 64 * NW, QB = 32 * NW;
    constexpr int NP = PREC == 2 ? 2 : 1, NPROD = PREC == 2 ? 3 : 1;
    // DMA (bf16 storage, head dim 64, 8-wave blocks): K and V tiles are dense [64 keys][128 B] images filled by LDS-DMA
    // (global_load_lds_dwordx4: one 1-KiB piece = 8 key rows per wave instruction, wave w owns piece w of either tile -
    // no staging registers, no ds_write, no v_perm), XOR-swizzled through the per-lane SOURCE address so that the reads
    // are conflict-free: K fragments (ds_read_b128, lanes = 32 key rows) see 16-byte chunk c of row k at chunk
    // c ^ ((k >> 1) & 7); V^T fragments come from the ROW-MAJOR V tile by ds_read_b64_tr_b16 (4 keys x 16 d per 16
    // lanes, transposed by the LDS), with the two 64-byte halves of a row swapped where bit 1 of the key is set.
    constexpr bool DMA = PREC == 1 && IO16 && DH == 64 && NW == 8;
    constexpr int LDK = DMA ? DH : DH + 8, LDV = KT + 4, D4 = DH / 4;
    constexpr int KPT = KT * D4 / NT;
    static_assert(KPT == 2 || KPT == 4, "staging packs 2 or 4 keys per V^T store");
    constexpr float THR = 8.0f;
    typedef unsigned short h16;
    static_assert(!IO16 || PREC == 1, "bf16 storage belongs to the bf16 mode");
    constexpr int ES = IO16 ? 2 : 4;                  // bytes per stored q/k/v element
    // STAG (8-wave blocks: waves w and w + 4 share a SIMD): waves 4..7 run HALF A TILE BEHIND waves 0..3, so that on
    // every SIMD one wave is in phase A (10 MFMAs + the tile's softmax: VALU-bound) while its partner is in phase B
    // (the P.V MFMAs + staging: MFMA-bound) - the two pipes of the SIMD are asked for complementary work instead of
    // the same work twice (round 2: both waves in the same phase, 34 % of the wave cycles issue-stalled).  Two block
    // barriers per tile keep the alternation; K is staged three tiles ahead and V two (a ring of three LDS tiles:
    // the late half still reads a tile while the early half stages the next but one).
    constexpr bool FOLD = PREC == 1 && DH <= 64;   // the row constant folded into the product + OR test; fp16x3 / head dim 128: round 2's max chain
    constexpr bool STAG = NW == 8 && PREC == 1 && DH <= 64;   // (fp16x3 / head dim 128 fill the register file: they spill in this form)
    constexpr int NBUF = STAG ? 3 : 2, KAHEAD = STAG ? 3 : 2, VAHEAD = KAHEAD - 1, NMB = KAHEAD + 1;
    __shared__ __attribute__((aligned(16))) h16 Kb[NBUF][NP][KT * LDK];
    __shared__ __attribute__((aligned(16))) h16 Vt[NBUF][NP][DMA ? KT * DH : DH * LDV];
    __shared__ __attribute__((aligned(16))) float mb[NMB][KT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    int b, head, qt;
    size_t base, orow0;
    if constexpr (VARLEN) {
        const int2 wk = work[blockIdx.x];
        b = wk.x; qt = wk.y; head = blockIdx.y;
        const int c0 = cu[b];
        T = cu[b + 1] - c0;
        base = ((size_t)head * Mtot + c0) * DH;
        orow0 = (size_t)c0;
    } else {
        int bh;
        if (!attn_block_map((T + QB - 1) / QB, BH, bh, qt)) return;
        b = bh / H; head = bh - b * H;
        base = (size_t)bh * T * DH;
        orow0 = (size_t)b * T;
    }
    const int q0 = qt * QB + 32 * wave;
    const float NEG_INF = -__builtin_inff();

    auto pack = [&](float x0, float x1, unsigned (&pl)[NP]) __attribute__((always_inline)) {
        if constexpr (PREC == 2) split_f16(x0, x1, pl[0], pl[1]);
        else pl[0] = pack_bf16(x0, x1);
    };
    auto mma = [&](const u32x4 &a, const u32x4 &bq, const f32x16 &c) __attribute__((always_inline)) -> f32x16 {
        if constexpr (PREC == 2) return MFMA_F16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, bq), c);
        else return MFMA_BF16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bq), c);
    };
    // product `pr` of the plane expansion, small terms first: (lo,hi), (hi,lo), (hi,hi); bf16: (hi,hi) only
    auto mma_prod = [&](auto prc, const u32x4 (&a)[NP], const u32x4 (&bq)[NP], f32x16 &c) __attribute__((always_inline)) {
        constexpr int pr = decltype(prc)::value;
        if constexpr (PREC == 2 && pr == 0) c = mma(a[1], bq[0], c);
        else if constexpr (PREC == 2 && pr == 1) c = mma(a[0], bq[1], c);
        else c = mma(a[0], bq[0], c);
    };

    u32x4 qreg[NS][NP];
    {
        int qr = q0 + r; qr = qr < T ? qr : T - 1;
        const float *qp = Q + base + (size_t)qr * DH + 8 * h;
        if constexpr (IO16) {
            const h16 *qp16 = (const h16 *)Q + base + (size_t)qr * DH + 8 * h;
#pragma unroll
            for (int s = 0; s < NS; ++s) qreg[s][0] = *(const u32x4 *)(qp16 + 16 * s);
        } else
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const f32x4 v0 = *(const f32x4 *)(qp + 16 * s), v1 = *(const f32x4 *)(qp + 16 * s + 4);
            unsigned pl[4][NP];
            pack(v0[0] * scale_log2e, v0[1] * scale_log2e, pl[0]);
            pack(v0[2] * scale_log2e, v0[3] * scale_log2e, pl[1]);
            pack(v1[0] * scale_log2e, v1[1] * scale_log2e, pl[2]);
            pack(v1[2] * scale_log2e, v1[3] * scale_log2e, pl[3]);
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int e = 0; e < 4; ++e) qreg[s][p][e] = pl[e][p];
        }
    }
    f32x16 o[ND + 1];                               // O^T blocks + the row sums (block of ones)
#pragma unroll
    for (int d = 0; d <= ND; ++d)
#pragma unroll
        for (int t = 0; t < 16; ++t) o[d][t] = 0.f;
    // m_run: the row constant c subtracted from this query's scores (log2 units), exactly representable in the operand
    // type; -inf until the first live key.  nbias: element 0 of the bias step's B operand = (-c, 0) packed (lanes with
    // h == 1 hold k = 8..15 of the step: zero).
    float m_run = NEG_INF;
    unsigned nbias = 0u;
    constexpr unsigned ONE2 = PREC == 2 ? 0x3C003C00u : 0x3F803F80u;
    const u32x4 ones_u = {ONE2, ONE2, ONE2, ONE2};

    const int d4 = tid % D4, kq = tid / D4;
    const int ntiles = (T + KT - 1) / KT;
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((char *)const_cast<float *>(Kg) + base * ES, 0, T * DH * ES, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((char *)const_cast<float *>(Vg) + base * ES, 0, T * DH * ES, 0x00020000);
    int voff[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) voff[i] = ((KPT * kq + i) * DH + 4 * d4) * ES;
    f32x4 pk[KPT], pv[KPT];
    u32x2 pk16[KPT], pv16[KPT];                       // IO16: 4 bf16 of one key
    float pm = 0.f;
    // the pipeline runs up to three tiles past the end: those loads re-read the last tile (their products are
    // never consumed); rows beyond T inside the last tile read as zeros (buffer bounds check) and are masked
    auto gload_k = [&](int tile) __attribute__((always_inline)) {
        tile = tile < ntiles ? tile : ntiles - 1;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            if constexpr (IO16) pk16[i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(krs, voff[i], tile * (KT * DH * ES), 0));
            else pk[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, voff[i], tile * (KT * DH * ES), 0));
        }
    };
    auto gload_v = [&](int tile) __attribute__((always_inline)) {
        tile = tile < ntiles ? tile : ntiles - 1;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            if constexpr (IO16) pv16[i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(vrs, voff[i], tile * (KT * DH * ES), 0));
            else pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vrs, voff[i], tile * (KT * DH * ES), 0));
        }
    };
    auto gload_m = [&](int tile) __attribute__((always_inline)) {
        if (tid < KT) {
            const int key = tile * KT + tid;
            float pmv = key >= T ? NEG_INF : 0.f;
            if (mask != nullptr) pmv = mask[(size_t)b * T + (key < T ? key : T - 1)] != 0 ? NEG_INF : pmv;
            pm = pmv;
        }
    };
    auto stage_k1 = [&](int i, int buf) __attribute__((always_inline)) {
        if constexpr (IO16) {
            *(u32x2 *)&Kb[buf][0][(KPT * kq + i) * LDK + 4 * d4] = pk16[i];
            return;
        }
        unsigned p0[NP], p1[NP];
        pack(pk[i][0], pk[i][1], p0);
        pack(pk[i][2], pk[i][3], p1);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            u32x2 u; u[0] = p0[p]; u[1] = p1[p];
            *(u32x2 *)&Kb[buf][p][(KPT * kq + i) * LDK + 4 * d4] = u;
        }
    };
    auto stage_v1 = [&](int e, int buf) __attribute__((always_inline)) {
        if constexpr (IO16) {
            // element e of this lane's 4 d's, from two keys, into one dword: v_perm_b32 picks the (e & 1) halves
            const unsigned sel = (e & 1) ? 0x07060302u : 0x05040100u;
            if constexpr (KPT == 4) {
                u32x2 u;
                u[0] = __builtin_amdgcn_perm(pv16[1][e >> 1], pv16[0][e >> 1], sel);
                u[1] = __builtin_amdgcn_perm(pv16[3][e >> 1], pv16[2][e >> 1], sel);
                *(u32x2 *)&Vt[buf][0][(4 * d4 + e) * LDV + 4 * kq] = u;
            } else {
                *(unsigned *)&Vt[buf][0][(4 * d4 + e) * LDV + 2 * kq] = __builtin_amdgcn_perm(pv16[1][e >> 1], pv16[0][e >> 1], sel);
            }
            return;
        }
        if constexpr (KPT == 4) {
            unsigned p0[NP], p1[NP];
            pack(pv[0][e], pv[1][e], p0);
            pack(pv[2][e], pv[3][e], p1);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                u32x2 u; u[0] = p0[p]; u[1] = p1[p];
                *(u32x2 *)&Vt[buf][p][(4 * d4 + e) * LDV + 4 * kq] = u;
            }
        } else {
            unsigned p0[NP];
            pack(pv[0][e], pv[1][e], p0);
#pragma unroll
            for (int p = 0; p < NP; ++p) *(unsigned *)&Vt[buf][p][(4 * d4 + e) * LDV + 2 * kq] = p0[p];
        }
    };
    auto stage_m = [&](int mbuf) __attribute__((always_inline)) { if (tid < KT) mb[mbuf][tid] = pm; };
    // DMA images: per-lane byte offsets of the swizzled reads (row part + chunk part; tile / block / step offsets are
    // immediates) and of this lane's two LDS-DMA source pieces
    int koff[NS], voff_tr[ND], dma_row = 0, dma_krel = 0, dma_vrel = 0;
    if constexpr (DMA) {
        const int swz = (r >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) koff[ks] = r * 128 + (((2 * ks + h) ^ swz) << 4);
        const int i16 = lane & 15, y = (i16 >> 3) & 1, x = 2 * ((lane >> 4) & 1) + ((i16 >> 1) & 1);
#pragma unroll
        for (int d = 0; d < ND; ++d) voff_tr[d] = (4 * h + (i16 >> 2)) * 128 + ((4 * (d ^ y) + x) << 4) + 8 * (i16 & 1);
        dma_row = 8 * wave + (lane >> 3);
        dma_krel = ((lane & 7) ^ ((dma_row >> 1) & 7)) * 8;              // 16-bit elements inside the key's row
        dma_vrel = ((lane & 7) ^ (4 * ((dma_row >> 1) & 1))) * 8;
    }
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    auto k_frag = [&](int buf, int ks, int n, u32x4 (&ka)[NP]) __attribute__((always_inline)) {
        if constexpr (DMA) {
            ka[0] = *(const u32x4 *)((const char *)&Kb[buf][0][0] + koff[ks] + n * 4096);
            return;
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) ka[p] = *(const u32x4 *)&Kb[buf][p][(32 * n + r) * LDK + 16 * ks + 8 * h];
    };
    auto v_frag = [&](int buf, int n, int ks, int d, u32x4 (&va)[NP]) __attribute__((always_inline)) {
        if constexpr (DMA) {
            const char *vb = (const char *)&Vt[buf][0][0] + voff_tr[d] + (32 * n + 16 * ks) * 128;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)vb);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(vb + 8 * 128));
            const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
            va[0][0] = l2[0]; va[0][1] = l2[1]; va[0][2] = h2[0]; va[0][3] = h2[1];
            return;
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const h16 *vp = &Vt[buf][p][(32 * d + r) * LDV + 32 * n + 16 * ks + 4 * h];
            const u32x2 lo = *(const u32x2 *)vp, hi = *(const u32x2 *)(vp + 8);
            va[p][0] = lo[0]; va[p][1] = lo[1]; va[p][2] = hi[0]; va[p][3] = hi[1];
        }
    };

    // LDS-DMA of this wave's pieces of K(tk) and V(tv) (tiles beyond the end re-read the last one; rows beyond T re-read
    // row T - 1: finite values, masked by the key bias).  Asynchronous: counted by vmcnt, waited for at the end of phase A.
    const h16 *Kg16 = (const h16 *)Kg + base, *Vg16 = (const h16 *)Vg + base;
    auto dma_issue = [&](int tk, int tv, bool do_k, bool do_v) __attribute__((always_inline)) {
        if constexpr (DMA) {
            if (do_k) {
                tk = tk < ntiles ? tk : ntiles - 1;
                int key = tk * KT + dma_row; key = key < T ? key : T - 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Kg16 + (size_t)key * DH + dma_krel),
                                                 (__attribute__((address_space(3))) void *)&Kb[tk % NBUF][0][wave * 512], 16, 0, 0);
            }
            if (do_v) {
                tv = tv < ntiles ? tv : ntiles - 1;
                int key = tv * KT + dma_row; key = key < T ? key : T - 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Vg16 + (size_t)key * DH + dma_vrel),
                                                 (__attribute__((address_space(3))) void *)&Vt[tv % NBUF][0][wave * 512], 16, 0, 0);
            }
        }
    };
    if constexpr (DMA) {
        // every tile of the prologue is requested at once (one HBM round trip instead of KAHEAD sequential ones)
#pragma unroll
        for (int u = 0; u < KAHEAD; ++u) dma_issue(u, u, true, u < VAHEAD);
#pragma unroll
        for (int u = 0; u < KAHEAD; ++u) { gload_m(u); stage_m(u % NMB); }
        gload_m(KAHEAD);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // ---- prologue: K(0 .. KAHEAD-1), V(0 .. VAHEAD-1) in LDS; K(KAHEAD), V(VAHEAD) in registers; S(0) computed ----
#pragma unroll
        for (int u = 0; u < KAHEAD; ++u) {
            gload_k(u); gload_m(u);
            if (u < VAHEAD) gload_v(u);
#pragma unroll
            for (int i = 0; i < KPT; ++i) stage_k1(i, u % NBUF);
            if (u < VAHEAD) {
#pragma unroll
                for (int e = 0; e < 4; ++e) stage_v1(e, u % NBUF);
            }
            stage_m(u % NMB);
        }
        gload_k(KAHEAD); gload_m(KAHEAD); gload_v(VAHEAD);
    }
    __syncthreads();
    f32x16 sa[2], sb[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int t = 0; t < 16; ++t) { sa[n][t] = 0.f; sb[n][t] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < NS; ++ks)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            u32x4 ka[NP];
            k_frag(0, ks, n, ka);
            static_for<NPROD>([&](auto prc) { mma_prod(prc, ka, qreg[ks], sa[n]); });
        }

    // bf16: c = maximum + 6 and the OR test (P >= 2).  f16 (fp16x3): its 5-bit exponent would push the small P of a
    // "+6" constant into the subnormals (measured: 2.4e-4 on the output), so there c = maximum - 8 (P <= 2^8 after a
    // raise: the whole normal range below is precision) and the test is a packed-f16 MAX tree against 2^15.
    constexpr float CMARGIN = PREC == 2 ? -8.0f : 6.0f;
    (void)CMARGIN;
    // P fragments of the tile in flight
    u32x4 pf[2][2][NP];
    f16x2 pmaxh = {(_Float16)0.f, (_Float16)0.f};
    float cb_a = 0.f, cb_b = 0.f;                    // the row constant S(a) / S(b) were started with
    // Rare path: bring a tile's S' (started with the constant `cb`) onto the current row constant, raising it to the
    // tile's maximum if that is larger (or if there is none yet): S' shifted, O and l rescaled, the bias operand rebuilt.
    // `raise`: THIS row asked for a larger constant (one of its P reached 2).  Rows that only ride along (the branch is
    // wave-uniform) must come out bit for bit unchanged - a row's result may not depend on the rows it shares a wave
    // with (batch invariance) - so they are shifted by 0 and scaled by 1.
    auto rebase = [&](f32x16 (&sv)[2], float &cb, bool raise) __attribute__((always_inline)) {
        float mx = __builtin_fmaxf(__builtin_fmaxf(sv[0][0], sv[0][1]), sv[1][0]);
        mx = __builtin_fmaxf(mx, sv[1][1]);
#pragma unroll
        for (int t = 2; t < 16; t += 2) {
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, sv[0][t]), sv[0][t + 1]);
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, sv[1][t]), sv[1][t + 1]);
        }
        const float raw = pair_max(mx) + cb;         // the tile's maximum, absolute; -inf if every key is masked
        float c_new = m_run;
        if ((raise || m_run == NEG_INF) && raw + CMARGIN > m_run) {   // round UP to a value the operand type holds exactly
            // c = the tile's maximum + CMARGIN: P <= 2^-CMARGIN now, and the next raise only when a key beats this
            // maximum by 2^(CMARGIN + 1) (the deferred maximum of round 2, THR 8, in the OR test's terms)
            const float cl = __builtin_fminf(__builtin_fmaxf(raw + CMARGIN, -32768.f), 32768.f);
            const unsigned b = __builtin_bit_cast(unsigned, cl);
            constexpr unsigned LOW = PREC == 2 ? 0x1FFFu : 0xFFFFu;       // f16: 10 mantissa bits, bf16: 7
            c_new = __builtin_bit_cast(float, (cl >= 0.f ? b + LOW : b) & ~LOW);
            // From |c| = 2^13 on, one bf16 step is 64 and more: rounded UP, the row maximum's own P = 2^-(CMARGIN + distance)
            // fell below 2^-126 once the step reached 128 (logits beyond 2^14), was flushed to zero, and the row came out 0 / 0
            // (tools/fuzz_attn_w64.py, round 4: q = 4, k = 5000 -> NaN; q = 100, k = 200, the same logit on a bf16 grid point ->
            // fine).  There the constant is rounded to NEAREST: the maximum's P stays within 2^+-64 of 2^-CMARGIN.
            if (PREC == 1 && __builtin_fabsf(cl) >= 8192.f) c_new = __builtin_bit_cast(float, (b + 0x8000u) & ~LOW);
            if (PREC == 2 && __builtin_fabsf(c_new) < 0.0001220703125f) c_new = cl > 0.f ? 0.0001220703125f : 0.f;   // below 2^-13: not a normal f16
        }
        const float u_new = (c_new == NEG_INF) ? 0.f : c_new;
        const float shift = cb - u_new;              // finite
        const float alpha = __builtin_amdgcn_exp2f(m_run - u_new);      // 1 for rows that keep their constant; 0 from -inf
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int t = 0; t < 16; ++t) sv[n][t] += shift;
#pragma unroll
        for (int d = 0; d <= ND; ++d)
#pragma unroll
            for (int t = 0; t < 16; ++t) o[d][t] *= alpha;
        m_run = c_new;
        cb = u_new;
        unsigned pl[NP];
        pack(-u_new, 0.f, pl);
        nbias = h == 0 ? pl[0] : 0u;
    };
    auto p_pair = [&](auto kc, f32x16 (&sv)[2]) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value, n = k / 8, ks = (k / 4) % 2, j = k % 4;
        unsigned pl[NP];
        pack(__builtin_amdgcn_exp2f(sv[n][8 * ks + 2 * j]), __builtin_amdgcn_exp2f(sv[n][8 * ks + 2 * j + 1]), pl);
#pragma unroll
        for (int p = 0; p < NP; ++p) pf[n][ks][p][j] = pl[p];
        // f16: running packed maximum of the hi plane, pair by pair (v_pk_max_f16).  (A 15-max tree over the finished
        // fragments in unit 17 was folded by hipcc to 3 maxima over 4 of the 16 registers, with the other exponentials
        // sunk behind the test: an overflowing P went undetected - found by the fp64 kernel test.)
        if constexpr (PREC == 2) pmaxh = __builtin_elementwise_max(pmaxh, __builtin_bit_cast(f16x2, pl[0]));
    };
    // one unit of softmax(t) work on s_in (S' of tile t, started with the row constant cb): 0 stale-constant check,
    // 1..16 one (n, ks, j) pair each: exp2, pack / split into the P fragments, 17 did any P reach 2?
    // fp16x3 keeps round 2's form (the folded form needs ~25 registers more than the 256 this precision already fills:
    // measured 6.5 -> 7.9 ms at configs[4] with the spills): row maximum by a max3 chain, deferred raise (THR 8),
    // subtraction in front of exp2, accumulators started at zero.  Units 0..3 max chain, 4 check, 5..20 one pair each.
    float sm_mx = 0.f;
    f32x2 sm_mm = {0.f, 0.f};
    auto sm_unit_max = [&](auto uc, f32x16 (&sv)[2]) __attribute__((always_inline)) {
        constexpr int U = decltype(uc)::value;
        constexpr float THR = 8.0f;
        if constexpr (U == 0) {
            sm_mx = __builtin_fmaxf(__builtin_fmaxf(sv[0][0], sv[0][1]), sv[1][0]);
            sm_mx = __builtin_fmaxf(sm_mx, sv[1][1]);
        }
        if constexpr (U < 4) {
            constexpr int t0 = U == 0 ? 2 : 4 * U;
#pragma unroll
            for (int t = t0; t < 4 * U + 4; t += 2) {
                sm_mx = __builtin_fmaxf(__builtin_fmaxf(sm_mx, sv[0][t]), sv[0][t + 1]);
                sm_mx = __builtin_fmaxf(__builtin_fmaxf(sm_mx, sv[1][t]), sv[1][t + 1]);
            }
        }
        if constexpr (U == 4) {
            const float mx = pair_max(sm_mx);
            const bool raise = mx > m_run + THR || (m_run == NEG_INF && mx != NEG_INF);
            if (__builtin_expect(__any(raise), 0)) {
                const float m_new = raise ? mx : m_run;
                const float u_new = (m_new == NEG_INF) ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m_run - u_new);
#pragma unroll
                for (int d = 0; d <= ND; ++d)
#pragma unroll
                    for (int t = 0; t < 16; ++t) o[d][t] *= alpha;
                m_run = m_new;
            }
            const float m_use = (m_run == NEG_INF) ? 0.f : m_run;
            sm_mm[0] = m_use; sm_mm[1] = m_use;
        }
        if constexpr (U >= 5 && U < 21) {
            constexpr int k = U - 5, n = k / 8, ks = (k / 4) % 2, j = k % 4;
            const f32x2 sx = {sv[n][8 * ks + 2 * j], sv[n][8 * ks + 2 * j + 1]};
            const f32x2 dv = sx - sm_mm;
            unsigned pl[NP];
            pack(__builtin_amdgcn_exp2f(dv[0]), __builtin_amdgcn_exp2f(dv[1]), pl);
#pragma unroll
            for (int p = 0; p < NP; ++p) pf[n][ks][p][j] = pl[p];
        }
    };
    auto sm_unit = [&](auto uc, f32x16 (&sv)[2], float &cb) __attribute__((always_inline)) {
        constexpr int U = decltype(uc)::value;
        if constexpr (!FOLD) { sm_unit_max(uc, sv); return; }
        if constexpr (U == 0) {
            const float u_run = (m_run == NEG_INF) ? 0.f : m_run;
            if (__builtin_expect(__any(cb != u_run || m_run == NEG_INF), 0)) rebase(sv, cb, false);
        }
        if constexpr (U >= 1 && U < 17) p_pair(std::integral_constant<int, U - 1>{}, sv);
        if constexpr (U == 17) {
            unsigned acc = 0u;
            bool hit;
            if constexpr (PREC == 2) {
                // the packed f16 maximum of the hi plane: either half >= 2^15 (0x7800; also inf / nan)?
                acc = __builtin_bit_cast(unsigned, pmaxh);
                hit = (acc & 0xFFFFu) >= 0x7800u || (acc >> 16) >= 0x7800u;
                pmaxh = f16x2{(_Float16)0.f, (_Float16)0.f};
            } else {
                // non-negative bf16 numbers: bit 14 set <=> value >= 2 (also inf / nan)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) acc |= (pf[n][ks][0][0] | pf[n][ks][0][1]) | (pf[n][ks][0][2] | pf[n][ks][0][3]);
                hit = (acc & 0x40004000u) != 0u;
            }
            if (__builtin_expect(__any(hit), 0)) {
                const unsigned hbit = hit ? 1u : 0u;
                auto pr = __builtin_amdgcn_permlane32_swap(hbit, hbit, false, false);     // a row lives in lanes l and l ^ 32
                rebase(sv, cb, (((unsigned)pr[0] | (unsigned)pr[1]) & 1u) != 0u);
                static_for<16>([&](auto kc) { p_pair(kc, sv); });
                if constexpr (PREC == 2) pmaxh = f16x2{(_Float16)0.f, (_Float16)0.f};
            }
        }
    };
    constexpr int NU = FOLD ? 18 : 21;
    constexpr bool FULLPF = PREC == 1 && DH <= 64;       // all K / V fragments of a tile in registers (see phase A)
    constexpr int NKF = FULLPF ? NS * 2 : 2, NVF = FULLPF ? 4 * ND : 2;
    constexpr int NBIAS = FOLD ? 2 : 0;                  // bias steps in front of the products
    constexpr int NSLOT_A = 2 * NS * NPROD + NBIAS, RA = (NU + NSLOT_A - 1) / NSLOT_A;
    constexpr int NSLOT_B = 4 * (ND * NPROD + NP);
    constexpr int NITEM = KPT + 4 + 2, SB = NSLOT_B / NITEM > 0 ? NSLOT_B / NITEM : 1;    // staging items, slot stride

    u32x4 abl_frag = {0u, 0u, 0u, 0u};
    if constexpr ((ABL & 16) != 0) abl_frag = *(const u32x4 *)&Kb[0][0][r * LDK + 8 * h];
    // one iteration: s_in = S(t) (complete), s_out <- S(t+1)
    auto iteration = [&](int t, f32x16 (&s_in)[2], float &cb_in, f32x16 (&s_out)[2], float &cb_out) __attribute__((always_inline)) {
        const int cur = (ABL & 4) ? 0 : t % NBUF, nxt = (ABL & 4) ? 0 : (t + 1) % NBUF;                  // V(t) / K(t+1) live here
        if (mask != nullptr || (t + 1) * KT > T) {       // masked / ragged tile: key-mask bias first (rare)
            const float *mp = mb[t % NMB];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bv = *(const f32x4 *)&mp[32 * n + 8 * g + 4 * h];
#pragma unroll
                    for (int e = 0; e < 4; ++e) s_in[n][4 * g + e] += bv[e];
                }
        }
        // ---- phase A: S'(t+1) = [K,1][Q,-c]^T MFMAs, softmax(t) between them ----
        // both 32-key blocks of the tile take the row constant as it is NOW (softmax(t) below may raise it meanwhile)
        cb_out = (m_run == NEG_INF) ? 0.f : m_run;
        const u32x4 nb = {nbias, 0u, 0u, 0u};
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // FULLPF (bf16, head dim <= 64: the registers are there): every K fragment of tile t+1 is requested before
        // the first MFMA and every V fragment of tile t during phase A, so no MFMA waits for an LDS read issued one
        // gap earlier (round 2: `s_waitcnt lgkmcnt(0)` in front of every MFMA).  Otherwise: one fragment ahead.
        u32x4 ka[NKF][NP], va[NVF][NP];
        if constexpr ((ABL & 16) != 0) {
#pragma unroll
            for (int g = 0; g < NKF; ++g)
#pragma unroll
                for (int p = 0; p < NP; ++p) ka[g][p] = abl_frag;
#pragma unroll
            for (int g = 0; g < NVF; ++g)
#pragma unroll
                for (int p = 0; p < NP; ++p) va[g][p] = abl_frag;
        } else
        if constexpr (FULLPF) {
#pragma unroll
            for (int g = 0; g < NS * 2; ++g) k_frag(nxt, g % NS, g / NS, ka[g]);
        } else {
            k_frag(nxt, 0, 0, ka[0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<NSLOT_A>([&](auto ic) {
            // slots 0, 1: the two bias steps (they need no LDS operand: the K fragment reads are in flight meanwhile);
            // then the products, block-major
            constexpr int i = decltype(ic)::value, PERN = NS * NPROD, n = i < NBIAS ? i : (i - NBIAS) / PERN,
                          w = i < NBIAS ? 0 : 1 + (i - NBIAS) % PERN;
            if constexpr (w == 0) {
                s_out[n] = mma(ones_u, nb, zero16);                       // -c for every key row of the block
            } else {
                constexpr int ks = (w - 1) / NPROD, pr = (w - 1) % NPROD, g = n * NS + ks;       // fragment index: block-major
                if constexpr (!FOLD && w == 1) s_out[n] = zero16;
                if constexpr (!(ABL & 16) && !FULLPF && pr == 0 && g + 1 < NS * 2) k_frag(nxt, (g + 1) % NS, (g + 1) / NS, ka[(g + 1) & 1]);
                mma_prod(std::integral_constant<int, pr>{}, ka[FULLPF ? g : (g & 1)], qreg[ks], s_out[n]);
            }
            if constexpr (!(ABL & 16) && FULLPF && i >= NBIAS && i - NBIAS < NVF) {
                constexpr int f = i - NBIAS;
                v_frag(cur, f / ND / 2, (f / ND) % 2, f % ND, va[f]);
            }
            static_for<RA>([&](auto rc) {
                constexpr int U = i * RA + decltype(rc)::value;
                if constexpr (U < NU && !(ABL & 8)) sm_unit(std::integral_constant<int, U>{}, s_in, cb_in);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        lap(1);
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the pieces requested in phase B(t-1) have landed
        if constexpr (STAG && !(ABL & 2)) { __syncthreads(); lap(3); }  // the partner wave of this SIMD changes phase too
        // ---- phase B: O += V(t)^T P(t)^T and the row sums; staging of K(t+KAHEAD), V(t+VAHEAD), loads of the tiles after ----
        if constexpr (!FULLPF && !(ABL & 16)) v_frag(cur, 0, 0, 0, va[0]);
        __builtin_amdgcn_sched_barrier(0);
        static_for<NSLOT_B>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int PER = ND * NPROD + NP;                 // MFMAs per (n, ks): products per d block, then row sums
            constexpr int g = i / PER, w = i % PER, n = g / 2, ks = g % 2;
            if constexpr (w < ND * NPROD) {
                constexpr int d = w / NPROD, pr = w % NPROD, f = g * ND + d;       // fragment index
                if constexpr (!(ABL & 16) && !FULLPF && pr == 0 && f + 1 < 4 * ND)
                    v_frag(cur, (f + 1) / ND / 2, ((f + 1) / ND) % 2, (f + 1) % ND, va[(f + 1) & 1]);
                mma_prod(std::integral_constant<int, pr>{}, va[FULLPF ? f : (f & 1)], pf[n][ks], o[d]);
            } else {
                constexpr int p = NP - 1 - (w - ND * NPROD);     // lo plane first
                o[ND] = mma(ones_u, pf[n][ks][p], o[ND]);
            }
            if constexpr (DMA) {
                // K(t+KAHEAD) and V(t+VAHEAD) straight into their LDS tiles, as early in the phase as possible
                if constexpr (!(ABL & 1) && i == 0) dma_issue(t + KAHEAD, t + VAHEAD, true, true);
                if constexpr (!(ABL & 1) && i == 4) { stage_m((t + KAHEAD) % NMB); gload_m(t + KAHEAD + 1); }
            } else
            if constexpr (!(ABL & 1) && i % SB == SB - 1 && i / SB < NITEM) {
                constexpr int item = i / SB;
                if constexpr (item < KPT) stage_k1(item, (t + KAHEAD) % NBUF);
                else if constexpr (item < KPT + 4) stage_v1(item - KPT, (t + VAHEAD) % NBUF);
                else if constexpr (item == KPT + 4) { stage_m((t + KAHEAD) % NMB); }
                else { gload_k(t + KAHEAD + 1); gload_m(t + KAHEAD + 1); gload_v(t + VAHEAD + 1); }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        lap(2);
        if constexpr (!(ABL & 2)) __syncthreads();
        lap(3);
    };

    lap(0);
    // every wave passes the same number of barriers: the late half waits one out before its first tile, the early
    // half after its last (wave-uniform branches: the wave index is made a scalar first)
    const bool late = STAG && __builtin_amdgcn_readfirstlane(wave) >= NW / 2;
    if (late) __syncthreads();
    int t = 0;
    for (; t + 1 < ntiles; t += 2) {
        iteration(t, sa, cb_a, sb, cb_b);
        iteration(t + 1, sb, cb_b, sa, cb_a);
    }
    if (t < ntiles) iteration(t, sa, cb_a, sb, cb_b);
    if (STAG && !late) __syncthreads();

    const int q = q0 + r;
    if (q < T) {
        const float inv = 1.0f / o[ND][0];
        float *op = out + (orow0 + q) * (H * DH) + head * DH;
        h16 *op16 = (h16 *)out + (orow0 + q) * (H * DH) + head * DH;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = o[d][4 * g + e] * inv;
                if constexpr (IO16) {
                    u32x2 u; u[0] = pack_bf16(v[0], v[1]); u[1] = pack_bf16(v[2], v[3]);
                    *(u32x2 *)(op16 + 32 * d + 8 * g + 4 * h) = u;
                } else
                *(f32x4 *)(op + 32 * d + 8 * g + 4 * h) = v;
            }
    }
    if constexpr (DIAG != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lap(7);
        dg[8] = tl - tbeg; dg[10] = __builtin_amdgcn_s_memrealtime(); dg[11] = (unsigned long long)ntiles;
        if (diag != nullptr && lane == 0) {
            unsigned long long *o_ = diag + ((size_t)blockIdx.x * NW + wave) * 16;
#pragma unroll
            for (int i = 0; i < 12; ++i) o_[i] = dg[i];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Attention, software-pipelined (head dim 32 / 64): same math and operand trick as attn_fwd.
//
// Measured on gfx950 (profiles/): the fp32 MFMA shares the SIMD's FP32 lanes with
// ordinary VALU work — every VALU instruction costs ~4 of the 64 cycles an MFMA owns, "in its shadow"
// or not.  So this kernel is built to issue as few VALU instructions per MFMA as possible:
//  * K/V tiles arrive through buffer loads whose per-tile offset is a scalar (no address VALU, and the
//    hardware bounds check zero-fills the ragged tail);
//  * the running row max m is folded into the product: S' = [K,1]*[Q,-m]^T costs one extra MFMA per
//    32-key block and replaces the accumulator zero-init and the 16 subtracts before exp2;
//  * deferred max: m is only raised (and O, l rescaled) when a block's max exceeds it by more than
//    2^8 — exact in fp32 (p <= 256 instead of <= 1) and almost never taken after the first block;
//  * the key-mask bias (0/-inf) is added only on tiles that have masked keys or the ragged tail.
// Keys are consumed in 32-key blocks; block b+1's S' MFMAs issue while block b's softmax (max check,
// exp2, row sum) runs between them; then O^T += V_b^T * P_b^T with the V operands prefetched into
// registers.  K/V tiles of 64 keys are double-buffered in LDS; global loads of tile t+2 and LDS writes
// of tile t+1 ride in the MFMA stream of tile t.  NW waves per block (8: one block per CU, all
// blocks take the same time; 4: for short videos).
// ------------------------------------------------------------------------------------------
// VARLEN (packed ragged batches, vs_scorer_forward_packed): the videos' frames are concatenated ([Mtot, .] rows,
// video b = rows cu[b] .. cu[b+1]), q/k/v are head-major over the packed rows ([H][Mtot][DH]) and the grid is
// (work items, heads) with work[w] = (video, query tile): T, the operand bases and the output rows are per block.
// DIAG (diagnostic library only, tools/diag_attention.py): per-wave s_memtime stamps around the phases of a tile ->
// diag[(block * NW + wave) * 16 + ..]: 0 prologue, 1 barrier A, 2 half-step A (S' of block 1 || softmax of block 0),
// 3 P.V of block 0, 4 barrier B, 5 half-step B, 6 P.V of block 1, 7 epilogue, 8 total cycles, 9 / 10 wall clock
// (100 MHz) at wave start / end, 11 tiles, 12 Q fetch (inside 0)
template <int DH, bool HAS_MASK, int NW, bool VARLEN = false, int DIAG = 0>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_pipe(
    const float *__restrict__ Q, const float *__restrict__ Kg, const float *__restrict__ Vg,
    const uint8_t *__restrict__ mask, float *__restrict__ out, int H, int T, float scale_log2e, int BH,
    const int *__restrict__ cu = nullptr, const int2 *__restrict__ work = nullptr, int Mtot = 0,
    unsigned long long *__restrict__ diag = nullptr) {
    unsigned long long dg[13] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}, tbeg = 0, tl = 0;
    if constexpr (DIAG != 0) { dg[9] = __builtin_amdgcn_s_memrealtime(); tbeg = vs_stamp(); tl = tbeg; }
    auto lap = [&](int slot) __attribute__((always_inline)) {
        if constexpr (DIAG != 0) { const unsigned long long t = vs_stamp(); dg[slot] += t - tl; tl = t; }
    };
    constexpr int KT = 64, LD = DH + 4, NJ = DH / 8, ND = DH / 32;
    constexpr int NT = 64 * NW;                     // threads per block
    constexpr int F4 = KT * DH / 4 / NT;            // float4 per thread per operand tile
    constexpr int TILE = KT * LD;                   // floats per K (or V) tile in LDS
    constexpr float THR = 8.0f;                     // deferred-max threshold (log2 units)
    __shared__ __attribute__((aligned(16))) float smem[4 * TILE + 2 * KT];   // [buf]{K,V} + mask bias
    float *mbs = smem + 4 * TILE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    int b, head, qt;
    size_t base, orow0;                              // operand base (floats), first output row of this video
    if constexpr (VARLEN) {
        const int2 wk = work[blockIdx.x];
        b = wk.x; qt = wk.y; head = blockIdx.y;
        const int c0 = cu[b];
        T = cu[b + 1] - c0;
        base = ((size_t)head * Mtot + c0) * DH;
        orow0 = (size_t)c0;
    } else {
        int bh;
        if (!attn_block_map((T + 32 * NW - 1) / (32 * NW), BH, bh, qt)) return;
        b = bh / H; head = bh - b * H;
        base = (size_t)bh * T * DH;
        orow0 = (size_t)b * T;
    }
    const int q0 = qt * (32 * NW) + 32 * wave;
    const float NEG_INF = -__builtin_inff();
    const int ntiles = (T + KT - 1) / KT;

    // Q fragment (B operand), scaled by scale*log2(e).  A lane needs 16 B of ITS query row per 8 k - as a
    // direct load that is 32 B into 32 different lines per instruction (slow to issue) - so the wave's 32
    // rows are loaded coalesced (DH*128 contiguous bytes), parked in a wave-private LDS corner (the K/V
    // buffers are not in use yet) and read back as fragments.
    float qreg[4 * NJ];
    static_assert(4 * TILE >= NW * 32 * LD || NW == 8, "LDS corner per wave");
    float *wtp = smem + wave * (32 * LD);            // 32 x (DH+4) floats per wave (NW*32*LD <= 4*TILE for NW <= 8)
    {
#pragma unroll
        for (int i = 0; i < DH / 8; ++i) {           // 32 rows x DH floats = DH/8 wave loads of 1 KiB
            const int idx = lane + 64 * i;
            const int row = idx / (DH / 4), c4 = (idx % (DH / 4)) * 4;
            int qr = q0 + row; qr = qr < T ? qr : T - 1;
            *(f32x4 *)&wtp[row * LD + c4] = *(const f32x4 *)(Q + base + (size_t)qr * DH + c4);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const f32x4 v = *(const f32x4 *)&wtp[r * LD + 8 * j + 4 * h];
#pragma unroll
            for (int s = 0; s < 4; ++s) qreg[4 * j + s] = v[s] * scale_log2e;
        }
    }
    lap(12);
    __syncthreads();                                 // Q corners are read; K/V staging may overwrite them
    f32x16 o[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int t = 0; t < 16; ++t) o[d][t] = 0.f;
    // m_run: running max (log2 units) actually applied to O and l; -inf until the first live key.
    // m_use(m) = m, or 0 while m is still -inf (keeps exp2 arguments finite-or--inf, never inf-inf).
    float m_run = NEG_INF, l_run = 0.f;
    const float ones_a = h == 0 ? 1.0f : 0.0f;      // A operand of the bias step: adds B[0][q] to every key row

    // ---- staging: buffer loads (scalar per-tile offset, zero fill beyond T), then LDS writes ----
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(Kg + base), 0, T * DH * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(Vg + base), 0, T * DH * 4, 0x00020000);
    int voff[F4];
#pragma unroll
    for (int i = 0; i < F4; ++i) {
        const int idx = tid + NT * i;
        voff[i] = ((idx / (DH / 4)) * DH + (idx % (DH / 4)) * 4) * 4;
    }
    f32x4 pk[F4], pv[F4];
    float pm = 0.f;
    auto gload_k = [&](int i, int tile) __attribute__((always_inline)) { pk[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, voff[i], tile * (KT * DH * 4), 0)); };
    auto gload_v = [&](int i, int tile) __attribute__((always_inline)) { pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vrs, voff[i], tile * (KT * DH * 4), 0)); };
    auto gload_m = [&](int tile) __attribute__((always_inline)) {      // key-mask bias of key (tid & 63): 0 or -inf (also for keys >= T)
        const int key = tile * KT + (tid & (KT - 1));
        float pmv = key >= T ? NEG_INF : 0.f;
        if (HAS_MASK) pmv = mask[(size_t)b * T + (key < T ? key : T - 1)] != 0 ? NEG_INF : pmv;
        pm = pmv;
    };
    auto stage_k = [&](int i, int buf) __attribute__((always_inline)) {
        const int idx = tid + NT * i;
        *(f32x4 *)&smem[buf * 2 * TILE + (idx / (DH / 4)) * LD + (idx % (DH / 4)) * 4] = pk[i];
    };
    auto stage_v = [&](int i, int buf) __attribute__((always_inline)) {
        const int idx = tid + NT * i;
        *(f32x4 *)&smem[buf * 2 * TILE + TILE + (idx / (DH / 4)) * LD + (idx % (DH / 4)) * 4] = pv[i];
    };

    // ---- softmax state of the block in flight ----
    float sm_mx = 0.f, sm_psum = 0.f;
    float vreg[16 * ND];
    // Rare path (first live block, or a max jump > 2^THR, or a block whose bias is stale): bring the
    // block's S' onto the (possibly raised) running max and rescale O, l.  Wave-uniform branch.
    auto fixup = [&](f32x16 &sv, float m_bias, float raw_max) __attribute__((always_inline)) {
        const float m_new = (raw_max > m_run + THR || m_run == NEG_INF) ? fmaxf(m_run, raw_max) : m_run;
        const float u_new = (m_new == NEG_INF) ? 0.f : m_new;
        const float shift = m_bias - u_new;                     // finite
        const float alpha = __builtin_amdgcn_exp2f(m_run - u_new);   // m_run = -inf -> 0 (O = l = 0 then anyway)
#pragma unroll
        for (int t = 0; t < 16; ++t) sv[t] += shift;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int t = 0; t < 16; ++t) o[d][t] *= alpha;
        l_run *= alpha;
        m_run = m_new;
    };
    // One unit of block-b softmax work, issued between two MFMAs of block b+1's S'.  sv = S' of block b
    // (biased by -m_bias), MASKED: add the key-mask bias first.
    auto sm_unit = [&](auto uc, f32x16 &sv, float m_bias, const float *mb, auto masked_tag) __attribute__((always_inline)) {
        constexpr int U = decltype(uc)::value;
        constexpr bool MASKED = decltype(masked_tag)::value;
        if constexpr (MASKED && U < 4) {
            const f32x4 bv = *(const f32x4 *)(mb + 8 * U + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) sv[4 * U + e] += bv[e];
        }
        if constexpr (U == 4) {
            sm_mx = fmaxf(fmaxf(sv[0], sv[1]), sv[2]);
#pragma unroll
            for (int t = 3; t < 15; t += 2) sm_mx = fmaxf(fmaxf(sm_mx, sv[t]), sv[t + 1]);
            sm_mx = fmaxf(sm_mx, sv[15]);
        }
        if constexpr (U == 5) {
            const float raw_max = pair_max(sm_mx) + m_bias;      // -inf if every key so far is masked
            const float u_run = (m_run == NEG_INF) ? 0.f : m_run;
            const bool fix = (m_bias != u_run) || (raw_max > m_run + THR) || (m_run == NEG_INF && raw_max != NEG_INF);
            if (__builtin_expect(__any(fix), 0)) fixup(sv, m_bias, raw_max);
            sm_psum = 0.f;
        }
        if constexpr (U >= 8 && U < 16) {
            constexpr int k = U - 8;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float pe_ = __builtin_amdgcn_exp2f(sv[2 * k + e]);
                sv[2 * k + e] = pe_;
                sm_psum += pe_;
            }
        }
        if constexpr (U == 16) l_run += pair_sum(sm_psum);
    };
    // MFMA stream of one half-step:  s_out = [K_blk,1]*[Q,-m_bias]^T  (1 + 4*NJ MFMAs), with between
    // consecutive MFMAs: one V operand of block (Vsrc,vblk) into vreg, one softmax unit of s_in, and
    // (STAGE) the LDS writes of tile t+1 / buffer loads of tile t+2.
    auto half_step = [&](const float *Ksrc, int blk, f32x16 &s_out, float m_bias_out, f32x16 &s_in, float m_bias_in,
                         const float *mb, const float *Vsrc, int vblk, auto masked_tag, auto stage_tag, int nbuf, int ntile) __attribute__((always_inline)) {
        constexpr bool STAGE = decltype(stage_tag)::value;
        constexpr int NSLOT = 4 * NJ, R = 32 / NSLOT;      // softmax units per MFMA slot (DH=64: 1, DH=32: 2)
        static_assert(NSLOT == 16 * ND, "one V operand per MFMA slot");
        const float *kp = Ksrc + (32 * blk + r) * LD + 4 * h;
        const float *vp = Vsrc + (32 * vblk + 4 * h) * LD + r;
        f32x4 ka[2];
        ka[0] = *(const f32x4 *)kp;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        s_out = MFMA32(ones_a, -m_bias_out, zero);
        __builtin_amdgcn_sched_barrier(0);
        static_for<NSLOT>([&](auto ic) {
            constexpr int i = decltype(ic)::value, j = i / 4, st = i % 4;
            if constexpr (st == 0 && j + 1 < NJ) ka[(j + 1) & 1] = *(const f32x4 *)(kp + 8 * (j + 1));
            s_out = MFMA32(ka[j & 1][st], qreg[4 * j + st], s_out);
            {
                constexpr int t = i / ND, d = i % ND;
                vreg[i] = vp[((t & 3) + 8 * (t >> 2)) * LD + 32 * d];
            }
            static_for<R>([&](auto rc) {
                constexpr int U = i * R + decltype(rc)::value;
                sm_unit(std::integral_constant<int, U>{}, s_in, m_bias_in, mb, masked_tag);
                if constexpr (STAGE) {
                    if constexpr (U < 2 * F4) { if constexpr (U % 2 == 0) stage_k(U / 2, nbuf); else stage_v(U / 2, nbuf); }
                    if constexpr (U == 2 * F4) mbs[nbuf * KT + (tid & (KT - 1))] = pm;
                    if constexpr (U >= 17 && U < 17 + 2 * F4) { if constexpr ((U - 17) % 2 == 0) gload_k((U - 17) / 2, ntile); else gload_v((U - 17) / 2, ntile); }
                    if constexpr (U == 17 + 2 * F4) gload_m(ntile);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // softmax alone (last block of the video: nothing left to overlap with)
    auto softmax_only = [&](f32x16 &sv, float m_bias, const float *mb, const float *Vsrc, int vblk, auto masked_tag) __attribute__((always_inline)) {
        const float *vp = Vsrc + (32 * vblk + 4 * h) * LD + r;
#pragma unroll
        for (int i = 0; i < 16 * ND; ++i) vreg[i] = vp[(((i / ND) & 3) + 8 * ((i / ND) >> 2)) * LD + 32 * (i % ND)];
        static_for<32>([&](auto uc) { sm_unit(uc, sv, m_bias, mb, masked_tag); });
    };
    // O^T += V[32 keys]^T * P^T : 16*ND MFMAs, operands already in registers
    auto pv_acc = [&](const f32x16 &p) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int d = 0; d < ND; ++d) o[d] = MFMA32(vreg[ND * t + d], p[t], o[d]);
    };

    // ---- prologue: tile 0 into LDS, tile 1 into registers, S' of block (0,0) with bias 0 ----
#pragma unroll
    for (int i = 0; i < F4; ++i) { gload_k(i, 0); gload_v(i, 0); }
    gload_m(0);
#pragma unroll
    for (int i = 0; i < F4; ++i) { stage_k(i, 0); stage_v(i, 0); }
    mbs[tid & (KT - 1)] = pm;
    {
        const int t1 = ntiles > 1 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < F4; ++i) { gload_k(i, t1); gload_v(i, t1); }
        gload_m(t1);
    }
    __syncthreads();
    f32x16 s_cur, s_nxt;
    float mb_cur = 0.f, mb_nxt = 0.f;            // bias each in-flight block was started with
    {
        const float *kp = smem + r * LD + 4 * h;
#pragma unroll
        for (int t = 0; t < 16; ++t) s_cur[t] = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const f32x4 ka = *(const f32x4 *)(kp + 8 * j);
#pragma unroll
            for (int st = 0; st < 4; ++st) s_cur = MFMA32(ka[st], qreg[4 * j + st], s_cur);
        }
    }

    // One 64-key tile.  LAST (the video's final tile) is peeled out of the loop so that the loop body has
    // no branch besides the rare fix-up: a branch there costs 64 accumulator-register copies per tile.
    auto tile_step = [&](int t, auto masked_tag, auto last_tag) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int buf = t & 1;
        const float *Ks = smem + buf * 2 * TILE, *Vs = Ks + TILE;
        const float *Kn = smem + (buf ^ 1) * 2 * TILE;
        __syncthreads();                       // everyone is done with buffer buf^1 (tile t-1)
        lap(1);
        // S' of block (t,1)  ||  softmax of block (t,0), LDS writes of tile t+1, loads of tile t+2
        mb_nxt = (m_run == NEG_INF) ? 0.f : m_run;
        half_step(Ks, 1, s_nxt, mb_nxt, s_cur, mb_cur, mbs + buf * KT, Vs, 0, masked_tag, std::integral_constant<bool, !LAST>{},
                  buf ^ 1, t + 2 < ntiles ? t + 2 : ntiles - 1);
        lap(2);
        pv_acc(s_cur);
        lap(3);
        __syncthreads();                       // tile t+1 is visible in buffer buf^1
        lap(4);
        if constexpr (!LAST) {
            // S' of block (t+1,0)  ||  softmax of block (t,1)
            mb_cur = (m_run == NEG_INF) ? 0.f : m_run;
            half_step(Kn, 0, s_cur, mb_cur, s_nxt, mb_nxt, mbs + buf * KT + 32, Vs, 1, masked_tag, std::false_type{}, 0, 0);
        } else {
            softmax_only(s_nxt, mb_nxt, mbs + buf * KT + 32, Vs, 1, masked_tag);
        }
        lap(5);
        pv_acc(s_nxt);
        lap(6);
    };
    lap(0);
    // without a mask only the ragged last tile carries dead keys
    for (int t = 0; t + 1 < ntiles; ++t) tile_step(t, std::integral_constant<bool, HAS_MASK>{}, std::false_type{});
    if (HAS_MASK || (T % KT) != 0) tile_step(ntiles - 1, std::true_type{}, std::true_type{});
    else                           tile_step(ntiles - 1, std::false_type{}, std::true_type{});

    // ---- epilogue: O^T[d][q] / l -> out[b, q, head*DH + d], through the wave-private LDS corner so that
    // every store instruction writes whole 128-byte lines (DH/4 lanes per row) ----
    __syncthreads();                                 // every wave is done reading K/V tiles
    {
        const float inv = 1.0f / l_run;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = o[d][4 * g + e] * inv;
                *(f32x4 *)&wtp[r * LD + 32 * d + 8 * g + 4 * h] = v;
            }
        constexpr int LPR = DH / 4, RPI = 64 / LPR;       // lanes per row, rows per store instruction
        const int orow = lane / LPR, oc4 = (lane % LPR) * 4;
#pragma unroll
        for (int p = 0; p < 32 / RPI; ++p) {
            const int rr = orow + RPI * p, q = q0 + rr;
            const f32x4 v = *(const f32x4 *)&wtp[rr * LD + oc4];
            if (q < T) *(f32x4 *)(out + (orow0 + q) * (H * DH) + head * DH + oc4) = v;
        }
    }
    if constexpr (DIAG != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the stores have left the wave
        lap(7);
        dg[8] = tl - tbeg; dg[10] = __builtin_amdgcn_s_memrealtime(); dg[11] = (unsigned long long)ntiles;
        dg[0] -= dg[12];
        if (diag != nullptr && lane == 0) {
            unsigned long long *o = diag + ((size_t)blockIdx.x * NW + wave) * 16;
#pragma unroll
            for (int i = 0; i < 13; ++i) o[i] = dg[i];
        }
    }
}

}  // namespace

#ifdef VS_WITH_DIAG
// tools/diag_attention.py: the stamped instantiation of the exact head-dim-64 kernel (8-wave blocks, no mask)
int vsk_diag_attention(const float *q, const float *k, const float *v, float *out, int B, int H, int T, float scale,
                       unsigned long long *diag, hipStream_t st) {
    const int BH = B * H, nq = (T + 255) / 256;
    dim3 g(8 * ((BH + 7) / 8) * nq);
    hipLaunchKernelGGL((attn_fwd_pipe<64, false, 8, false, 1>), g, dim3(512), 0, st, q, k, v, nullptr, out, H, T,
                       scale * 1.4426950408889634f, BH, nullptr, nullptr, 0, diag);
    VSK_CHECK_LAUNCH();
    return (int)g.x;     // > 0: blocks launched (the caller sizes / reads diag[blocks * 8 * 16])
}
// prec 1: bf16 in / out (q pre-scaled; the bf16 mode's form), prec 2: fp16x3 on fp32 q / k / v
int vsk_diag_attention_lp(const float *q, const float *k, const float *v, float *out, int B, int H, int T, float scale,
                          int prec, unsigned long long *diag, hipStream_t st) {
    const int BH = B * H, nq = (T + 255) / 256;
    dim3 g(8 * ((BH + 7) / 8) * nq);
    const float sl2 = scale * 1.4426950408889634f;
    if (prec >= 100) {      // timing ablations of the bf16 kernel (no stamps): prec = 100 + ABL
#define VSK_ABL(A_) case A_: hipLaunchKernelGGL((attn_fwd_lp_pipe<64, 8, 1, false, true, 0, A_>), g, dim3(512), 0, st, q, k, v, nullptr, out, H, T, sl2, BH, nullptr, nullptr, 0, nullptr); break;
        switch (prec - 100) {
            VSK_ABL(0) VSK_ABL(1) VSK_ABL(2) VSK_ABL(3) VSK_ABL(4) VSK_ABL(5) VSK_ABL(7) VSK_ABL(16) VSK_ABL(17) VSK_ABL(19)
            default: return -1;
        }
#undef VSK_ABL
        VSK_CHECK_LAUNCH();
        return (int)g.x;
    }
    if (prec == 1)
        hipLaunchKernelGGL((attn_fwd_lp_pipe<64, 8, 1, false, true, 1>), g, dim3(512), 0, st, q, k, v, nullptr, out, H, T, sl2, BH,
                           nullptr, nullptr, 0, diag);
    else
        hipLaunchKernelGGL((attn_fwd_lp_pipe<64, 8, 2, false, false, 1>), g, dim3(512), 0, st, q, k, v, nullptr, out, H, T, sl2, BH,
                           nullptr, nullptr, 0, diag);
    VSK_CHECK_LAUNCH();
    return (int)g.x;
}
#endif

float vsk_attention_qscale(float scale) { return scale * 1.4426950408889634f; }

int vsk_attention(const float *q, const float *k, const float *v, const uint8_t *mask, float *out,
                  int B, int H, int T, int dh, float scale, hipStream_t st) {
    const float sl2 = vsk_attention_qscale(scale);
    const int BH = B * H;
    dim3 grid(8 * ((BH + 7) / 8) * ((T + 127) / 128));
#ifdef VS_WITH_DIAG     // A/B switch for tools/ (diagnostic library only): the non-pipelined kernel at head dim 32 / 64
    const bool legacy = vsk_options().attn_legacy != 0;
    if (dh == 32 && legacy)
        hipLaunchKernelGGL((attn_fwd<32, 2>), grid, dim3(256), 0, st, q, k, v, mask, out, H, T, sl2, BH);
    else if (dh == 64 && legacy)
        hipLaunchKernelGGL((attn_fwd<64, 2>), grid, dim3(256), 0, st, q, k, v, mask, out, H, T, sl2, BH);
    else
#endif
    if (dh == 32 || dh == 64) {
        // 8-wave blocks (one per CU, 256 query rows) unless the ragged tail would waste more rows than
        // 4-wave blocks (two per CU, 128 query rows) do
        const int r8 = (T + 255) / 256 * 256, r4 = (T + 127) / 128 * 128;
        const bool wide = !vsk_options().attn_nw4 && r8 * 100 <= r4 * 105;
        const int nq = wide ? r8 / 256 : r4 / 128;
        dim3 g(8 * ((BH + 7) / 8) * nq), blk(wide ? 512 : 256);
#define VSK_ATTN(DH_, MASK_, NW_) \
    hipLaunchKernelGGL((attn_fwd_pipe<DH_, MASK_, NW_, false>), g, blk, 0, st, q, k, v, mask, out, H, T, sl2, BH, nullptr, nullptr, 0, nullptr)
        if (dh == 32) {
            if (mask) { if (wide) VSK_ATTN(32, true, 8); else VSK_ATTN(32, true, 4); }
            else      { if (wide) VSK_ATTN(32, false, 8); else VSK_ATTN(32, false, 4); }
        } else {
            if (mask) { if (wide) VSK_ATTN(64, true, 8); else VSK_ATTN(64, true, 4); }
            else      { if (wide) VSK_ATTN(64, false, 8); else VSK_ATTN(64, false, 4); }
        }
#undef VSK_ATTN
    }
    else if (dh == 128)
        hipLaunchKernelGGL((attn_fwd<128, 1>), grid, dim3(256), 0, st, q, k, v, mask, out, H, T, sl2, BH);
    else
        return -1;
    VSK_CHECK_LAUNCH();
    return 0;
}

// latency mode: keys split over the 8 waves of a block (exact fp32, head dim 32 / 64); -1 if the shape has no such kernel
int vsk_attention_splitkv(const float *q, const float *k, const float *v, const uint8_t *mask, float *out,
                          int B, int H, int T, int dh, float scale, hipStream_t st) {
    const float sl2 = vsk_attention_qscale(scale);
    const dim3 grid(((T + 31) / 32) * B * H);
    if (dh == 64) hipLaunchKernelGGL((attn_fwd_splitkv<64>), grid, dim3(512), 0, st, q, k, v, mask, out, H, T, sl2, B * H);
    else if (dh == 128) hipLaunchKernelGGL((attn_fwd_splitkv<128>), grid, dim3(512), 0, st, q, k, v, mask, out, H, T, sl2, B * H);
    else if (dh == 32) hipLaunchKernelGGL((attn_fwd_splitkv<32>), grid, dim3(512), 0, st, q, k, v, mask, out, H, T, sl2, B * H);
    else return -1;
    VSK_CHECK_LAUNCH();
    return 0;
}

// packed ragged batch: q/k/v [H][Mtot][dh] head-major over the packed rows, cu [B+1] row offsets, work[nwork] =
// (video, query tile of 32*nw rows); exact fp32, head dim 32 / 64
int vsk_attention_packed(const float *q, const float *k, const float *v, float *out, int H, int Mtot, int dh,
                         float scale, const int *cu, const int *work, int nwork, int nw, int prec, hipStream_t st) {
    const float sl2 = vsk_attention_qscale(scale);
    if (nwork <= 0) return 0;
    if (prec == (1 | VSK_STORE16) && dh == 64 && nw == 8 && vsk_options().attn_w64)      // one wave per SIMD: the same 256-row work items
        return vsk_attention_bf16_w64_packed(q, k, v, out, H, Mtot, cu, work, nwork, st);
    dim3 grid(nwork, H);
    const int2 *wk = (const int2 *)work;
#define VSK_ATTN_PX(DH_, NW_) \
    hipLaunchKernelGGL((attn_fwd_pipe<DH_, false, NW_, true>), grid, dim3(64 * NW_), 0, st, q, k, v, nullptr, out, H, 0, sl2, 0, cu, wk, Mtot, nullptr)
#define VSK_ATTN_PE(DH_, NW_, P_) \
    hipLaunchKernelGGL((attn_fwd_lp_pipe<DH_, NW_, P_, true>), grid, dim3(64 * NW_), 0, st, q, k, v, nullptr, out, H, 0, sl2, 0, cu, wk, Mtot)
#define VSK_ATTN_P16(DH_, NW_) \
    hipLaunchKernelGGL((attn_fwd_lp_pipe<DH_, NW_, 1, true, true>), grid, dim3(64 * NW_), 0, st, q, k, v, nullptr, out, H, 0, sl2, 0, cu, wk, Mtot)
    if (prec == (1 | VSK_STORE16)) {      // bf16 q (pre-scaled) / k / v in, bf16 out
        if (dh == 64 && nw == 8) VSK_ATTN_P16(64, 8);
        else if (dh == 64 && nw == 4) VSK_ATTN_P16(64, 4);
        else if (dh == 32 && nw == 4) VSK_ATTN_P16(32, 4);
        else return -1;
    } else if (prec == 2) {
        if (dh == 64 && nw == 8) VSK_ATTN_PE(64, 8, 2);
        else if (dh == 64 && nw == 4) VSK_ATTN_PE(64, 4, 2);
        else if (dh == 32 && nw == 4) VSK_ATTN_PE(32, 4, 2);
        else return -1;
    } else if (prec == 1) {
        if (dh == 128 && nw == 8) VSK_ATTN_PE(128, 8, 1);       // head dim 128 (round 4), fp32 q / k / v in HBM
        else if (dh == 64 && nw == 8) VSK_ATTN_PE(64, 8, 1);
        else if (dh == 64 && nw == 4) VSK_ATTN_PE(64, 4, 1);
        else if (dh == 32 && nw == 4) VSK_ATTN_PE(32, 4, 1);
        else return -1;
    } else {
        if (dh == 128 && nw == 4)       // exact fp32, head dim 128 (M-B): 128-row work items through the non-pipelined kernel
            hipLaunchKernelGGL((attn_fwd<128, 1, true>), grid, dim3(256), 0, st, q, k, v, nullptr, out, H, 0, sl2, 0, cu, wk, Mtot);
        else if (dh == 64 && nw == 8) VSK_ATTN_PX(64, 8);
        else if (dh == 64 && nw == 4) VSK_ATTN_PX(64, 4);
        else if (dh == 32 && nw == 8) VSK_ATTN_PX(32, 8);
        else if (dh == 32 && nw == 4) VSK_ATTN_PX(32, 4);
        else return -1;
    }
#undef VSK_ATTN_PX
#undef VSK_ATTN_PE
#undef VSK_ATTN_P16
    VSK_CHECK_LAUNCH();
    return 0;
}

int vsk_attention_bf16(const float *q, const float *k, const float *v, const uint8_t *mask, float *out,
                       int B, int H, int T, int dh, float scale, int prec, hipStream_t st) {
    const float sl2 = vsk_attention_qscale(scale);
    const int BH = B * H;
    // 8-wave blocks (256 query rows share one staged K/V tile) unless the ragged tail would waste too many rows
    const int r8 = (T + 255) / 256 * 256, r4 = (T + 127) / 128 * 128;
    const bool wide = dh == 64 && !vsk_options().attn_nw4 && r8 * 100 <= r4 * 105;
    dim3 grid(8 * ((BH + 7) / 8) * (wide ? r8 / 256 : r4 / 128));
    const bool simple = vsk_options().attn_lp_simple != 0;      // A/B switch for tools/, not a fallback
    if (dh == 128) {     // head dim 128 (M-B): bf16 only, 8-wave blocks only (fp16x3 does not fit the register file: 48 spills)
        if (prec != 1 && prec != (1 | VSK_STORE16)) return -1;
        dim3 g128(8 * ((BH + 7) / 8) * (r8 / 256));
        if (prec == (1 | VSK_STORE16))      // bf16 q (pre-scaled) / k / v in, bf16 out (the wide models' bf16-operand path)
            hipLaunchKernelGGL((attn_fwd_lp_pipe<128, 8, 1, false, true>), g128, dim3(512), 0, st, q, k, v, mask, out, H, T, sl2, BH);
        else
        hipLaunchKernelGGL((attn_fwd_lp_pipe<128, 8, 1>), g128, dim3(512), 0, st, q, k, v, mask, out, H, T, sl2, BH);
        VSK_CHECK_LAUNCH();
        return 0;
    }
#define VSK_ATTN_LP(KERN_, P_)                                                                                           \
    if (dh == 64 && wide)                                                                                                \
        hipLaunchKernelGGL((KERN_<64, 8, P_>), grid, dim3(512), 0, st, q, k, v, mask, out, H, T, sl2, BH);               \
    else if (dh == 64)                                                                                                   \
        hipLaunchKernelGGL((KERN_<64, 4, P_>), grid, dim3(256), 0, st, q, k, v, mask, out, H, T, sl2, BH);               \
    else if (dh == 32)                                                                                                   \
        hipLaunchKernelGGL((KERN_<32, 4, P_>), grid, dim3(256), 0, st, q, k, v, mask, out, H, T, sl2, BH);               \
    else                                                                                                                 \
        return -1;
    if (prec == (1 | VSK_STORE16)) {      // bf16 q (pre-scaled) / k / v in, bf16 out: the pipelined kernel only
        if (dh == 64 && wide && vsk_options().attn_w64) {      // one wave per SIMD, 4 waves x 64 query rows (vs_attention_w64.hip)
            const int rc = vsk_attention_bf16_w64(q, k, v, mask, out, B, H, T, st);
            if (rc != -1) return rc;
        }
        if (dh == 64 && wide)
            hipLaunchKernelGGL((attn_fwd_lp_pipe<64, 8, 1, false, true>), grid, dim3(512), 0, st, q, k, v, mask, out, H, T, sl2, BH);
        else if (dh == 64)
            hipLaunchKernelGGL((attn_fwd_lp_pipe<64, 4, 1, false, true>), grid, dim3(256), 0, st, q, k, v, mask, out, H, T, sl2, BH);
        else if (dh == 32)
            hipLaunchKernelGGL((attn_fwd_lp_pipe<32, 4, 1, false, true>), grid, dim3(256), 0, st, q, k, v, mask, out, H, T, sl2, BH);
        else
            return -1;
    } else
    if (simple) { if (prec == 2) { VSK_ATTN_LP(attn_fwd_lp, 2) } else { VSK_ATTN_LP(attn_fwd_lp, 1) } }
    else        { if (prec == 2) { VSK_ATTN_LP(attn_fwd_lp_pipe, 2) } else { VSK_ATTN_LP(attn_fwd_lp_pipe, 1) } }
#undef VSK_ATTN_LP
    VSK_CHECK_LAUNCH();
    return 0;
}

