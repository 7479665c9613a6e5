// vs_train_gemm_rows.hip — the two widest GEMMs of a bf16 training step at small K (reference simnet.py:180-181 and its
// backward): C16[M, N] = epilogue(A[M, K] W16[N, K]^T + bias), K = d_model in {256, 512}, N = 4 d_model, C written as bf16:
//   EPI_RELU_DROP  mlp.fc1 + ReLU + mlp.dropout (forward)            -> the bf16-stored hidden tensor
//   EPI_GATE       d(hidden) = (dY W2) gated by that tensor's sign, times 1 / (1 - p) (backward) -> the bf16-stored gradient
//   EPI_QKV        the q / k / v projections (N = 3 d_model; simnet.py:148-153): bf16 [3][B][H][T][dh] planes, q times scale * log2 e
//
// Why not gemm_nt_128.  With K = 256 a 256 x 256 output tile is four k-tiles of work: its A panel is re-read by each of the
// N / 256 column tiles and W by each of the M / 256 row tiles (all through L2, ~10 TB/s chip-wide), and its epilogue -
// dropout hash or gate, bf16 packing, 128 KB of stores at ~23 GB/s per CU - runs with the matrix pipe idle.  Measured at
// M = 65 536: 115 us (fc1) and 148 us (gate) against 33 / 56 us of HBM bytes (profiles/r03z_pmc_summary_train_bf16.txt:
// 0.09-0.12 of the matrix pipe).
//
// A-stationary, in REGISTERS.  A wave owns 32 rows of A for the whole kernel: their K values, rounded to bf16 once, are the
// B operands of every MFMA it issues - K / 16 fragments of 4 registers (64 registers at K = 256).  A block is 8 waves = 256
// rows; the grid is M / 256 blocks (one per CU at the bench shape).  W streams past in stages of 64 output columns x K
// (bf16 image, 32 KB at K = 256: 17 B per clock and CU from L2 - the rate this chip sustains), double-buffered in LDS and
// shared by the 8 waves; every stage is 2 x K / 16 MFMAs per wave and ends in its own epilogue, so the stores of stage s
// drain under the MFMAs of stage s + 1 instead of forming a store phase.  A is read from HBM exactly once, C written once.
// Accumulators start at the bias and the k order is ascending, like gemm_nt_128's: the results are bit-identical to it.
#include "vs_train_device.h"
#include "vs_train_kernels.h"

namespace {

typedef unsigned short h16;
enum { GR_RELU_DROP = 0, GR_GATE = 1, GR_RELU = 2, GR_QKV = 3 };

// W image row (K bf16) in LDS: 16-byte chunk c at position c ^ (row & 15) (rows are 512 / 1024 B apart: without the XOR the
// 16 lanes of a ds_read_b128 group - 16 different rows, one chunk each - would all hit the same four banks)
template <int KT>
__device__ __forceinline__ int wimg_off(int row, int chunk) { return row * (2 * KT) + ((chunk ^ (row & 15)) << 4); }

// KT = 256: 64-column stages (two accumulator chains per wave); KT = 512 (d_model 512: the reference's default architecture):
// 128 registers of A per lane and 32-column stages (one chain), so that two W stages still fit beside the scratch.
template <int KT, int EPI, bool A16>      // A16: A is stored as bf16 (not used by the call sites today; kept for the rows-copy form)
__global__ __launch_bounds__(512, 2) void gemm_rows16(
    const float *__restrict__ A, const h16 *__restrict__ W, const float *__restrict__ bias, h16 *__restrict__ C,
    const h16 *__restrict__ gate, int M, int N, float scale, unsigned long long seed, unsigned site, float p, int T, int H, int dh) {
    constexpr int NS = KT / 16;                                    // k-steps
    constexpr int NB = KT == 256 ? 2 : 1, GR_NT = 32 * NB;         // 32-column blocks / output columns per stage
    constexpr int STAGE = GR_NT * 2 * KT;                          // bytes (32 KB)
    constexpr int NLD = GR_NT * KT * 2 / 16 / 512;                 // 16-byte chunks per thread and stage (4)
    constexpr int LPR = 4 * NB, RPP = 64 / LPR, NPASS = 32 / RPP;  // read-back of the transposed stage: lanes per row, rows per pass, passes
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[2][STAGE];
    __shared__ __attribute__((aligned(16))) unsigned char tbuf[8][32 * 144];      // per-wave transposition scratch: 32 rows x (64 bf16 + pad)
    __shared__ __attribute__((aligned(16))) float bias_s[4096];                   // the whole bias vector (N <= 4096): an accumulator's
                                                                                  // initial value must not wait for an L2 round trip per stage

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 256 + 32 * wave;
    const int mrow = m0 + r < M ? m0 + r : M - 1;

    // ---- this wave's 32 rows of A as B-operand fragments: lane (r, h) <- A[mrow][16 ks + 8 h .. + 7]
    bf16x8 af[NS];
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
        if constexpr (A16) {
            af[ks] = __builtin_bit_cast(bf16x8, *(const u32x4 *)((const h16 *)A + (size_t)mrow * KT + 16 * ks + 8 * h));
        } else {
            const float *ap = A + (size_t)mrow * KT + 16 * ks + 8 * h;
            const f32x4 v0 = *(const f32x4 *)ap, v1 = *(const f32x4 *)(ap + 4);
            const u32x4 u = {pack_bf16(v0[0], v0[1]), pack_bf16(v0[2], v0[3]), pack_bf16(v1[0], v1[1]), pack_bf16(v1[2], v1[3])};
            af[ks] = __builtin_bit_cast(bf16x8, u);
        }
    }

    // ---- W stages: 64 rows x KT bf16, 16-byte chunks dealt to the 512 threads
    u32x4 wreg[NLD];
    auto wload = [&](int st) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + 512 * i, row = idx / (KT / 8), chunk = idx % (KT / 8);
            int n = st * GR_NT + row; n = n < N ? n : N - 1;
            wreg[i] = *(const u32x4 *)(W + (size_t)n * KT + 8 * chunk);
        }
    };
    auto wstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + 512 * i, row = idx / (KT / 8), chunk = idx % (KT / 8);
            *(u32x4 *)(wbuf[buf] + wimg_off<KT>(row, chunk)) = wreg[i];
        }
    };
    const int nst = (N + GR_NT - 1) / GR_NT;
    const DropSite ds = drop_site(seed, site, p);
    const unsigned rkm = drop_rowkey(ds, (unsigned)(m0 + r));                // the owner lane's row
    for (int i = tid; i < N / 4; i += 512) *(f32x4 *)&bias_s[4 * i] = *(const f32x4 *)(bias + 4 * i);
    wload(0);
    wstore(0);
    __syncthreads();
    unsigned char *tb = tbuf[wave];
    const int trow = lane / LPR, tc8 = (lane % LPR) * 8;
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1, n0 = st * GR_NT;
        if (st + 1 < nst) wload(st + 1);
        u32x4 gpre[NPASS];                          // this stage's gate values, requested before the MFMAs that they will mask
        if (EPI == GR_GATE) {
#pragma unroll
            for (int pp = 0; pp < NPASS; ++pp) {
                int row = m0 + trow + RPP * pp; row = row < M ? row : M - 1;
                int col = n0 + tc8; col = col < N ? col : N - 8;
                gpre[pp] = *(const u32x4 *)(gate + (size_t)row * N + col);
            }
        }
        // acc[nb][t] = C[m = r][n = n0 + 32 nb + 8 (t >> 2) + 4 h + (t & 3)]: starts at the bias
        f32x16 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int tg = 0; tg < 4; ++tg) {
                const int c = n0 + 32 * nb + 8 * tg + 4 * h;
                const f32x4 bv = c < N ? *(const f32x4 *)&bias_s[c] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[nb][4 * tg + e] = bv[e];
            }
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const bf16x8 wf = __builtin_bit_cast(bf16x8, *(const u32x4 *)(wbuf[buf] + wimg_off<KT>(32 * nb + r, 2 * ks + h)));
                acc[nb] = MFMA_BF16(wf, af[ks], acc[nb]);
            }
        }
        // ---- epilogue of this stage: the owner packs its 64 columns (ReLU here; dropout and gate need the transposed,
        // row-contiguous view), the wave transposes them through its scratch, 16-byte stores of whole 128-byte lines
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int tg = 0; tg < 4; ++tg) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[nb][4 * tg + e];
                    if (EPI == GR_RELU_DROP || EPI == GR_RELU) v[e] = relu1(v[e]);
                    if (EPI == GR_GATE) v[e] *= scale;
                    if (EPI == GR_QKV && n0 < H * dh) v[e] *= scale;        // q (a 64-column stage never straddles q | k | v: d % 64 == 0)
                }
                if (EPI == GR_RELU_DROP) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = drop_keep(ds, rkm, (unsigned)(n0 + 32 * nb + 8 * tg + 4 * h + e)) ? v[e] * ds.scale : 0.f;
                }
                *(u32x2 *)(tb + r * 144 + 64 * nb + 16 * tg + 8 * h) = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
            }
#pragma unroll
        for (int pp = 0; pp < NPASS; ++pp) {
            const int row = m0 + trow + RPP * pp, col = n0 + tc8;
            u32x4 pk = *(const u32x4 *)(tb + (trow + RPP * pp) * 144 + 2 * tc8);
            if (row < M && col < N) {
                if (EPI == GR_GATE) {          // the bf16-stored activation: > 0 <=> its 16 bits are a positive integer
                    const u32x4 g = gpre[pp];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned gw = g[e], pw = pk[e];
                        const unsigned lo = (short)(gw & 0xffffu) > 0 ? (pw & 0xffffu) : 0u;
                        const unsigned hi = (short)(gw >> 16) > 0 ? (pw & 0xffff0000u) : 0u;
                        pk[e] = lo | hi;
                    }
                }
                if (EPI == GR_QKV) {       // head-major scatter: 8 consecutive columns stay inside one head (dh % 8 == 0)
                    const int d = H * dh, which = col / d, c = col - which * d, head = c / dh, e0 = c - head * dh;
                    const int b = row / T, t = row - b * T;
                    *(u32x4 *)(C + (size_t)which * M * d + (((size_t)b * H + head) * T + t) * dh + e0) = pk;
                } else
                *(u32x4 *)(C + (size_t)row * N + col) = pk;
            }
        }
        if (st + 1 < nst) wstore(buf ^ 1);
        __syncthreads();
    }
}

}  // namespace

// From 3/4 of a 256-row block per CU up (the grid is M / 256 blocks: at 16 384 rows it would fill a quarter of the chip and
// measured 7 % slower per step than the tiled kernels' 256 tiles); MI355X: 256 CUs.
// (any_rows: the kernel itself handles every M - tests pin it on small batches)
bool vst_gemm_rows16_supported(int M, int N, int K, bool any_rows) {
    return (K == 256 || K == 512) && N % 8 == 0 && N >= 64 && N <= 4096 && M > 0 && (any_rows || M >= 192 * 256);
}

// epi 0: dropout(relu(.)) (seed, site, p); 1: gate (gate16, scale); 2: relu; 3: q / k / v planes (T, H, dh; scale = q's factor).
// A fp32 [M, K]; W16 bf16 [N, K]; C16 bf16 [M, N] (epi 3: [3][M / T][H][T][dh]).
int vst_gemm_rows16(const float *A, const void *W16, const float *bias, void *C16, const void *gate16, int M, int N, int K, int epi,
                    float scale, unsigned long long seed, unsigned site, float p, hipStream_t st, int T, int H, int dh) {
    if (!vst_gemm_rows16_supported(M, N, K, true)) return -1;
    if (epi == GR_QKV && (T <= 0 || H <= 0 || dh % 8 || N != 3 * H * dh || (H * dh) % 64 || M % T)) return -1;
    const dim3 grid((M + 255) / 256);
    const h16 *w = (const h16 *)W16, *g = (const h16 *)gate16;
    h16 *c = (h16 *)C16;
#define VST_GR(KT_, EPI_) hipLaunchKernelGGL((gemm_rows16<KT_, EPI_, false>), grid, dim3(512), 0, st, A, w, bias, c, g, M, N, scale, seed, site, p, T, H, dh)
    if (K == 256) {
        if (epi == GR_RELU_DROP) VST_GR(256, GR_RELU_DROP); else if (epi == GR_GATE) VST_GR(256, GR_GATE); else if (epi == GR_RELU) VST_GR(256, GR_RELU); else if (epi == GR_QKV) VST_GR(256, GR_QKV); else return -1;
    } else {
        if (epi == GR_RELU_DROP) VST_GR(512, GR_RELU_DROP); else if (epi == GR_GATE) VST_GR(512, GR_GATE); else if (epi == GR_RELU) VST_GR(512, GR_RELU); else if (epi == GR_QKV) VST_GR(512, GR_QKV); else return -1;
    }
#undef VST_GR
    VSK_CHECK_LAUNCH();
    return 0;
}
