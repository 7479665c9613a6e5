// vs_gemm_ring.hip — the bf16 mode's projection GEMM for wide models (d_model > 256; reference simnet.py:150-183 under
// VS_FLAG_BF16_LINEAR):  C[M,N] = A16[M,K] * W16[N,K]^T + bias (+ ReLU | + the q/k/v head-major scatter), BOTH operands
// bf16 in HBM, fp32 accumulation, C fp32 or bf16.
//
// Why a second GEMM kernel.  gemm_nt_128's bf16 instantiation reads fp32 operands and rounds them on their way into LDS.
// Measured on M-B (profiles/r03c_gemm_bf16_diag.txt): its k-loop alone runs at 600 TF = 0.24 of the bf16 peak, and neither
// a twice-as-long distance between a global load and its use nor half the barriers moved it (+3 %): a 256x256 tile moves
// 2 KB of fp32 operands from L2 per k for 131 kflop, 64 flop/B, and the CUs pull ~11 TB/s out of the L2s at that rate.
// The remedy is bytes, not latency: operands that are ONLY ever matrix operands live in HBM as bf16 (weights as a bf16
// image, activations written as bf16 by their producers - q/k/v, the attention output, the MLP hidden tensor, and a bf16
// COPY of each LayerNorm output beside the fp32 residual stream), which doubles the flops per byte, and nothing is left
// to convert, so the tiles go from L2 into LDS by LDS-DMA (global_load_lds_dwordx4) with no register staging at all.
//
// Shape (K % 64 == 0).  256 x 256 output tile per 8-wave block (4 x 2 waves of 64 x 128: 2 x 4 MFMA tiles of 32x32, 128
// accumulator registers), one block per CU, k-tiles of 64 (one 128-byte row = one cache line per operand row) in a ring
// of 2 LDS stages of 64 KB; per k-tile a wave issues 8 DMA instructions (1 KB each: 8 rows), waits for the pieces it
// issued one k-tile ago, meets the block at ONE barrier, and runs 32 MFMAs from 24 conflict-free ds_read_b128.
// Otherwise (K % 32 == 0): 128 x 256 tiles on 4-wave blocks, two per CU, k-tiles of 32 (64-byte rows), 3 stages of 24 KB.
// Measured at 65 536 rows (tools/bench_gemm16.py, profiles/r03c_gemm16.txt): 680-1 060 TF against 390-670 TF for
// gemm_nt_128's bf16 form; k-loop alone ~1 100-1 200 TF on every shape (0.45 of the dense peak); the epilogue - stores at
// ~23 GB/s per CU whatever their width - is 15-35 % of a tile at K = 512 and is not overlapped (one block per CU).
// Configurations measured slower: 128 x 256 x 32 in 3 stages, two blocks per CU (570-910 TF: 64-byte rows, 85 flop/B);
// 256 x 256 x 32 in 4 stages (590-945); 256 x 128 x 64 in 3 stages (510-860); 128 x 256 x 64 in 2 stages, two blocks
// per CU (455-870).
// LDS image: chunk c (16 B = 8 k) of row R is stored at position c ^ swz(R), swz = (R >> 1) & 7 for 128-byte rows,
// (R >> 2) & 3 for 64-byte rows - the DMA writes lane-linear, so the swizzle is applied to the SOURCE address; a fragment
// read (lane = row, one chunk) is then conflict-free in each 16-lane group of ds_read_b128 (MI355X_MICROARCH.md, LDS table).
// Output tiles are numbered so that the blocks of one XCD (blockIdx % 8) own a contiguous chunk of the tile list, N
// fastest: the N-tiles of one A row-panel run together and read it through one L2.
#include "vs_kernels.h"
#include "vs_device.h"

namespace {

typedef unsigned short h16;
enum { RG_BIAS = 0, RG_RELU = 1, RG_QKV = 3 };

constexpr int RTLD = 36;                           // epilogue transposition scratch: floats per row

template <int N_> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// NWM: waves along M (block = NWM x 2 waves, tile = 64 NWM x 256);  BK: k-tile (32: 64-byte rows, 64: 128-byte rows);
// NST: ring stages
// PERSISTENT: the grid is one block per CU (two for the 4-wave forms) and every block walks its list of output tiles
// (gemm_nt_128's: the blocks of one XCD own one contiguous chunk of the tile list, N fastest) with ONE continuous ring:
// the first k-tile(s) of the next output tile are requested before the epilogue of the current one, into the stage(s)
// the epilogue's transposition scratch does not use.
// (Tried and measured without effect, so not kept: waiting at the next tile's first k-tile with `s_waitcnt vmcnt(#stores)`
// so that the stores drain under the new MFMAs; starting every other block half a tile late; spreading a k-tile's DMA
// requests over its k-steps; rotating the k-loop's start by column tile; a 3-stage ring of 256 x 128 tiles.)
template <int EPI, int C16, int NWM, int BK, int NST, int NJ = 4>
__global__ __launch_bounds__(128 * NWM, 2) void gemm16_ring(
    const h16 *__restrict__ A, const h16 *__restrict__ W, const float *__restrict__ bias, float *__restrict__ C,
    int M, int N, int K, int T, int H, int dh, float qscale) {
    constexpr int BM = 64 * NWM, NW = 2 * NWM, RBN = 64 * NJ;
    constexpr int ROWB = 2 * BK, CPR = ROWB / 16, RPP = 1024 / ROWB;       // row bytes, 16-byte chunks per row, rows per 1-KB piece
    constexpr int APW = BM / RPP / NW, WPW = RBN / RPP / NW, PPS = APW + WPW;   // DMA pieces per wave and stage
    constexpr int STAGE_BYTES = (BM + RBN) * ROWB;
    static_assert(STAGE_BYTES >= NW * 32 * RTLD * 4, "one stage doubles as the epilogue's transposition scratch");
    static_assert((NST - 1) * PPS <= 63, "vmcnt immediate");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE_BYTES];
    // chunk c of row R lives at chunk position c ^ swz(R): conflict-free ds_read_b128 with lane = row
    auto swz = [](int R) __attribute__((always_inline)) { return CPR == 4 ? (R >> 2) & 3 : (R >> 1) & 7; };
    auto ring_off = [&](int R, int c) __attribute__((always_inline)) { return R * ROWB + ((c ^ swz(R)) << 4); };

    const int tiles_n = (N + RBN - 1) / RBN;
    const int ntiles = ((M + BM - 1) / BM) * tiles_n;
    // this block's tile list: start + j, start + j + G, ...   (chunk [start, start + len) per XCD label)
    const int xl = blockIdx.x & 7, j = blockIdx.x >> 3, G = gridDim.x >> 3;
    const int cq = ntiles >> 3, cr = ntiles & 7;
    const int start = xl * cq + (xl < cr ? xl : cr), len = cq + (xl < cr ? 1 : 0);
    const int my_tiles = len > j ? (len - j + G - 1) / G : 0;
    if (my_tiles == 0) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int nk = K / BK;

    // ---- this lane's DMA sources: pieces wave + NW p of A (APW) and of W (WPW); a piece = RPP rows
    const int prow = lane / CPR, pcp = lane % CPR;
    const h16 *asrc[APW], *wsrc[WPW];
    int nx_m0 = 0, nx_n0 = 0;
    auto set_tile = [&](int it) __attribute__((always_inline)) {
        const int tile = start + j + it * G;
        nx_m0 = (tile / tiles_n) * BM;
        nx_n0 = (tile % tiles_n) * RBN;
#pragma unroll
        for (int p = 0; p < APW; ++p) {
            const int R = RPP * (wave + NW * p) + prow;
            int gr = nx_m0 + R; gr = gr < M ? gr : M - 1;
            asrc[p] = A + (size_t)gr * K + 8 * (pcp ^ swz(R));
        }
#pragma unroll
        for (int p = 0; p < WPW; ++p) {
            const int R = RPP * (wave + NW * p) + prow;
            int gr = nx_n0 + R; gr = gr < N ? gr : N - 1;
            wsrc[p] = W + (size_t)gr * K + 8 * (pcp ^ swz(R));
        }
    };
    // k-tile kt of the tile set_tile() named last -> ring stage `slot`
    auto issue = [&](int kt, int slot) __attribute__((always_inline)) {
        unsigned char *sb = smem + slot * STAGE_BYTES;
        const int koff = kt * BK;
#pragma unroll
        for (int p = 0; p < APW; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(asrc[p] + koff),
                                             (__attribute__((address_space(3))) void *)(sb + 1024 * (wave + NW * p)), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < WPW; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc[p] + koff),
                                             (__attribute__((address_space(3))) void *)(sb + BM * ROWB + 1024 * (wave + NW * p)), 16, 0, 0);
    };

    int g = 0;                              // k-tiles consumed so far: k-tile kt of the current tile lives in stage (g + kt) % NST
    set_tile(0);
#pragma unroll
    for (int u = 0; u < NST - 1; ++u)
        if (u < nk) issue(u, u % NST);
    for (int it = 0; it < my_tiles; ++it) {
        const int m0 = nx_m0, n0 = nx_n0;
        // acc[i][jj][t] = C[m = 64 wr + 32 i + r][n = 32 NJ wc + 32 jj + 8 (t >> 2) + 4 h + (t & 3)]   (lane = output ROW: the
        // W fragment is the MFMA's A operand, the activation fragment its B operand - as in gemm_nt_128)
        f32x16 acc[2][NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                for (int t = 0; t < 16; ++t) acc[i][jj][t] = 0.f;

        for (int kt = 0; kt < nk; ++kt) {
            // the pieces this wave requested for k-tile kt have landed (those of the k-tiles after it may still fly) ...
            const int ahead = nk - 1 - kt;            // k-tiles requested after kt, capped at NST - 2
            if (NST >= 4 && ahead >= 2 && (kt >= NST - 1 || it == 0)) wait_vm<2 * PPS>();
            else if (NST >= 3 && ahead >= 1 && (kt >= NST - 1 || it == 0)) wait_vm<PPS>();
            else wait_vm<0>();
            __syncthreads();      // ... and so have everybody's; every wave has finished the k-tile before: its stage is free
            if (kt + NST - 1 < nk) issue(kt + NST - 1, (g + kt + NST - 1) % NST);
            const unsigned char *sa = smem + ((g + kt) % NST) * STAGE_BYTES, *sw = sa + BM * ROWB;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 fa[2], fw[NJ];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(sa + ring_off(64 * wr + 32 * i + r, 2 * ks + h)));
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) fw[jj] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(sw + ring_off(32 * NJ * wc + 32 * jj + r, 2 * ks + h)));
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    acc[0][jj] = MFMA_BF16(fw[jj], fa[0], acc[0][jj]);
                    acc[1][jj] = MFMA_BF16(fw[jj], fa[1], acc[1][jj]);
                }
            }
        }
        g += nk;
        __syncthreads();          // every wave has read its last fragments: that stage becomes the transposition scratch,
                                  // the others take the next tile's first k-tiles
        if (it + 1 < my_tiles) {
            set_tile(it + 1);
#pragma unroll
            for (int u = 0; u < NST - 1; ++u)
                if (u < nk) issue(u, (g + u) % NST);
        }

        // ---- epilogue (gemm_nt_128's): a lane owns output ROWS; each 32x32 sub-tile is transposed through a wave-private
        // corner of LDS so that every store instruction writes whole 128-byte lines
        float *tp = reinterpret_cast<float *>(smem + ((g + NST - 1) % NST) * STAGE_BYTES) + wave * (32 * RTLD);
        const int trow = lane >> 3, tc4 = (lane & 7) * 4;
        int b0 = 0, t0 = 0;
        if (EPI == RG_QKV) { b0 = m0 / T; t0 = m0 - b0 * T; }
        // bf16 output: TWO sub-tiles (64 columns) per round, transposed as packed bf16 (bias / ReLU / q scale applied in the
        // owner's registers first): half the LDS traffic of the fp32 transposition and 16-byte stores.  (The store phase
        // itself is bound by bytes, not instructions - ~23 GB/s per CU with 8-byte and with 16-byte stores alike.)
        if (C16 != 0 && NJ % 2 == 0 && (EPI != RG_QKV || dh % 64 == 0)) {
            unsigned char *tb = reinterpret_cast<unsigned char *>(tp);          // 32 rows x (128 B + 16 B pad)
            const int tc8 = (lane & 7) * 8;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int jp = 0; jp < NJ / 2; ++jp) {
                    const int c64 = n0 + 32 * NJ * wc + 64 * jp;
                    int which = 0, head = 0, e0 = 0;
                    if (EPI == RG_QKV) { const int d = H * dh; which = c64 / d; const int c = c64 - which * d; head = c / dh; e0 = c - head * dh; }
                    const float mul = (EPI == RG_QKV && which == 0) ? qscale : 1.0f;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int jj = 2 * jp + u;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int cb = c64 + 32 * u + 8 * q + 4 * h;
                            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                            if (cb < N) bv = *(const f32x4 *)(bias + cb);
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[e] = acc[i][jj][4 * q + e] + bv[e];
                                if (EPI == RG_RELU) v[e] = relu1(v[e]);
                                if (EPI == RG_QKV) v[e] *= mul;
                            }
                            u32x2 pk; pk[0] = pack_bf16(v[0], v[1]); pk[1] = pack_bf16(v[2], v[3]);
                            *(u32x2 *)(tb + r * 144 + 64 * u + 16 * q + 8 * h) = pk;
                        }
                    }
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int ro = 64 * wr + 32 * i + trow + 8 * p;
                        const u32x4 pk = *(const u32x4 *)(tb + (trow + 8 * p) * 144 + 2 * tc8);
                        const int row = m0 + ro;
                        int bb = b0, tt = t0 + ro;
                        if (EPI == RG_QKV) { while (tt >= T) { tt -= T; ++bb; } }
                        if (row < M && c64 + tc8 < N) {
                            h16 *C2 = (h16 *)C;
                            if (EPI == RG_QKV)
                                *(u32x4 *)(C2 + (size_t)which * M * (H * dh) + (((size_t)bb * H + head) * T + tt) * dh + e0 + tc8) = pk;
                            else
                                *(u32x4 *)(C2 + (size_t)row * N + c64 + tc8) = pk;
                        }
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][jj][4 * q + e];
                    *(f32x4 *)&tp[r * RTLD + 8 * q + 4 * h] = v;
                }
                const int c32 = n0 + 32 * NJ * wc + 32 * jj;              // a 32-column block never straddles a head
                int which = 0, head = 0, e0 = 0;
                if (EPI == RG_QKV) { const int d = H * dh; which = c32 / d; const int c = c32 - which * d; head = c / dh; e0 = c - head * dh; }
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (c32 < N) bv = *(const f32x4 *)(bias + c32 + tc4);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int ro = 64 * wr + 32 * i + trow + 8 * p;
                    f32x4 v = *(const f32x4 *)&tp[(trow + 8 * p) * RTLD + tc4] + bv;
                    if (EPI == RG_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = relu1(v[e]);
                    }
                    const int row = m0 + ro;
                    int bb = b0, tt = t0 + ro;
                    if (EPI == RG_QKV) { while (tt >= T) { tt -= T; ++bb; } }
                    if (row < M && c32 < N) {
                        if constexpr (C16 != 0) {
                            if (EPI == RG_QKV && which == 0) v *= qscale;
                            u32x2 u; u[0] = pack_bf16(v[0], v[1]); u[1] = pack_bf16(v[2], v[3]);
                            h16 *C2 = (h16 *)C;
                            if (EPI == RG_QKV)
                                *(u32x2 *)(C2 + (size_t)which * M * (H * dh) + (((size_t)bb * H + head) * T + tt) * dh + e0 + tc4) = u;
                            else
                                *(u32x2 *)(C2 + (size_t)row * N + c32 + tc4) = u;
                        } else if (EPI == RG_QKV)
                            *(f32x4 *)(C + (size_t)which * M * (H * dh) + (((size_t)bb * H + head) * T + tt) * dh + e0 + tc4) = v;
                        else
                            *(f32x4 *)(C + (size_t)row * N + c32 + tc4) = v;
                    }
                }
            }
        }
    }
}

// dst[i] = bf16(src[i])  (round to nearest even), 8 elements per thread
__global__ __launch_bounds__(256) void to_bf16(const float *__restrict__ src, h16 *__restrict__ dst, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const f32x4 a = *(const f32x4 *)(src + 8 * i), b = *(const f32x4 *)(src + 8 * i + 4);
        const u32x4 u = {pack_bf16(a[0], a[1]), pack_bf16(a[2], a[3]), pack_bf16(b[0], b[1]), pack_bf16(b[2], b[3])};
        *(u32x4 *)(dst + 8 * i) = u;
    }
}

}  // namespace

bool vsk_gemm16_supported(int M, int N, int K) { return M > 0 && N > 0 && N % 32 == 0 && K >= 32 && K % 32 == 0; }

// epi: 0 bias, 1 bias + ReLU, 3 bias + q/k/v head-major scatter (T, H, dh; q additionally times qscale when c16).
// A16 [M, K], W16 [N, K] bf16 row-major, 16-byte aligned; C fp32 [M, N] (c16 = 0) or bf16 (c16 = 1).
int vsk_gemm16(const void *A16, const void *W16, const float *bias, void *C, int M, int N, int K, int epi, int c16,
               int T, int H, int dh, float qscale, hipStream_t st) {
    if (!vsk_gemm16_supported(M, N, K)) return -1;
    if (epi == RG_QKV && (T <= 0 || H <= 0 || dh <= 0 || dh % 32 || N != 3 * H * dh || M % T)) return -1;
    // 256 x 256 tiles with 128-byte rows (whole cache lines per row and DMA) measured fastest wherever K allows them
    const int cfg = K % 64 == 0 ? 2 : 0;
    const h16 *a = (const h16 *)A16, *w = (const h16 *)W16;
    float *c = (float *)C;
    const int cus = vsk_device_cus();
    if (cus <= 0) return (int)hipErrorInvalidDevice;
#define VSK_RING2(EPI_, C16_, NWM_, BK_, NST_, NJ_)                                                                               \
    do {                                                                                                                         \
        const int nt_ = ((M + 64 * NWM_ - 1) / (64 * NWM_)) * ((N + 64 * NJ_ - 1) / (64 * NJ_));                                \
        int g_ = (NWM_ == 2 ? 2 : 1) * cus; g_ -= g_ % 8; if (g_ < 8) g_ = 8;                                                    \
        const int need_ = (nt_ + 7) / 8 * 8;                                                                                     \
        hipLaunchKernelGGL((gemm16_ring<EPI_, C16_, NWM_, BK_, NST_, NJ_>), dim3(need_ < g_ ? need_ : g_), dim3(128 * NWM_), 0, st, \
                           a, w, bias, c, M, N, K, T, H, dh, qscale);                                                            \
    } while (0)
#define VSK_RING(EPI_, C16_)                                    \
    do {                                                        \
        if (cfg == 2) VSK_RING2(EPI_, C16_, 4, 64, 2, 4);       \
        else VSK_RING2(EPI_, C16_, 2, 32, 3, 4);                \
    } while (0)
    if (epi == RG_BIAS) { if (c16) VSK_RING(RG_BIAS, 1); else VSK_RING(RG_BIAS, 0); }
    else if (epi == RG_RELU) { if (c16) VSK_RING(RG_RELU, 1); else VSK_RING(RG_RELU, 0); }
    else if (epi == RG_QKV) { if (c16) VSK_RING(RG_QKV, 1); else VSK_RING(RG_QKV, 0); }
    else return -1;
#undef VSK_RING
#undef VSK_RING2
    VSK_CHECK_LAUNCH();
    return 0;
}

// dst <- bf16(src), n elements (n % 8 == 0, both 16-byte aligned)
int vsk_to_bf16(const float *src, void *dst, size_t n, hipStream_t st) {
    if (n % 8) return -1;
    const size_t n8 = n / 8;
    if (n8 == 0) return 0;
    const size_t want = (n8 + 255) / 256;
    const int blocks = (int)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(to_bf16, dim3(blocks), dim3(256), 0, st, src, (h16 *)dst, n8);
    VSK_CHECK_LAUNCH();
    return 0;
}
