// vs_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the frame-importance scorer.
//
// Everything on this path is a dense fp32 contraction; the 1e-4 parity bar
// rules out bf16 operands, so every product runs on the exact-fp32 matrix instruction
// v_mfma_f32_32x32x2_f32 (64 cycles/SIMD, bit-equal to an fmaf chain, 157 TFLOP/s chip peak).
// At that rate one MFMA covers ~14 VALU issue slots and 64 LDS cycles, so the design goal is
// simply: keep one dependent MFMA chain per wave issuing back-to-back, two or three waves per
// SIMD to cover barriers, and put all elementwise work (bias, positional table, ReLU, residual,
// LayerNorm, score head, softmax) into the shadow of the MFMAs of the same kernel.
//
// Operand convention used by all kernels (lane l, r = l & 31, h = l >> 5):
//   A operand of 32x32x2: A[i = r][k = h]     B operand: B[k = h][j = r]
//   accumulator reg t (0..15): C[row = (t&3) + 8*(t>>2) + 4*h][col = r]
// A lane fetches FOUR consecutive k of its row with one 16-byte LDS read (k = 8*g + 4*h + s,
// s = 0..3) and feeds them to four MFMA steps; step s therefore contracts k in {8g+s, 8g+4+s}.
// The k order inside a sum is irrelevant as long as A and B use the same one.
#include <atomic>

#include "vs_train_device.h"     // dropout hash (vs_device.h comes with it)
#include "vs_kernels.h"

namespace {

// ------------------------------------------------------------------------------------------
// Generic projection GEMM:  C[M,N] = A[M,K] * W[N,K]^T + bias  (+ epilogue)
//   128x128 block tile, BK = 32, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles.
//   LDS rows are padded to 36 floats: 16 distinct rows then cover all 64 banks with their
//   16-byte reads (36*r mod 64 = 4*(9r mod 16)), so ds_read_b128 is conflict-free.
//   PERSISTENT: the grid is sized to the chip (2 blocks per CU) and every block walks a list of
//   output tiles with ONE continuous global->register->LDS pipeline: the first k-tile of the next
//   output tile is prefetched under the last k-tile of the current one, and the epilogue's stores
//   drain under the next tile's MFMAs.  Without this every block of the chip loads, computes and
//   stores in the same phase and the load latency / store bursts are exposed once per round.
//   Blocks that share an XCD (blockIdx % 8) own one contiguous chunk of the tile list, so the
//   N-tiles of one A row-panel are consumed through one L2.
// ------------------------------------------------------------------------------------------
enum { EPI_BIAS = 0, EPI_RELU = 1, EPI_PE = 2, EPI_QKV = 3,
       // training path only (vs_train.cpp):
       EPI_GATE = 4,          // C = gate > 0 ? C * scale : 0, gate = `pe` [M,N]: ReLU + dropout backward riding in the dgrad GEMM
       EPI_RELU_DROP = 5 };   // C = dropout(relu(C)): mlp.dropout riding in the fc1 GEMM (reference simnet.py:181)
struct EpiArgs {              // extra epilogue operands of EPI_GATE / EPI_RELU_DROP (zero for every other epilogue)
    unsigned long long seed;
    unsigned site;
    float p;
    float scale;
};

// Diagnostic stamps: vs_stamp() of vs_device.h; compiled only into the DIAG instantiation, which is reachable only
// through vs_diag_gemm(); no product launch executes a stamp.
__device__ __forceinline__ unsigned long long stamp() { return vs_stamp(); }

// NWM = waves along M (block = NWM x 2 waves, tile = 64*NWM x 128).  NWM = 4: one 8-wave block per CU -
// the two waves of every SIMD then belong to the same block and are coupled by its barriers, so neither
// can starve the other and all blocks finish together (two independent 4-wave blocks per CU share the
// SIMD unfairly: the older one finishes ~25 % earlier and the younger runs a lonely tail).
// NJ = 32-column MFMA tiles per wave along N (2: block tile 64*NWM x 128; 4: 64*NWM x 256 - a third fewer
// staging instructions and a quarter fewer fragment reads per MFMA, 128 accumulator registers).
// PREC 1 (opt-in, VS_FLAG_BF16_LINEAR): the product runs as v_mfma_f32_32x32x16_bf16.  Operands stay fp32 in HBM and
// are rounded to bf16 once, on their way into LDS (rows of 32 k = 64 B, padded to 80 B: conflict-free b128
// fragment reads); bias, accumulation and the epilogue are the fp32 ones.
// PREC 2 (opt-in, VS_FLAG_F16X3_LINEAR): fp32 emulated on the f16 pipe (split_f16): an LDS row holds the 32 hi
// halves then the 32 lo halves (128 B + 16 B pad = the fp32 row stride), three MFMAs per fragment pair.
// C16 (PREC 1 only; fc1 and QKV): C is stored as bf16 - its only consumers (attention, fc2) round it to bf16 on entry
// anyway, so the results are bit-identical and the tensor costs half the HBM bytes.  EPI_QKV then also pre-multiplies
// q by ea.scale (the attention's scale * log2 e), which the bf16-input attention kernel no longer does.
// KW (PREC 1 only): k-tile width.  A bf16 k-tile of 32 is 16 MFMAs of 32 cycles per wave - 1 024 cycles for the SIMD's two
// waves, half the latency of the global loads that were issued at its start (measured: 3 255 cycles per k-tile).  KW = 64
// (K % 64 == 0) doubles the distance between a load and its use and halves the barriers; the LDS rows (64 bf16 + 16 B pad)
// then have the fp32 layout's 144-byte stride.
// A16 (PREC 1 only; training path's bf16 storage of the MLP hidden tensor and of its gradient): A is bf16 in memory - its
// 8-byte pieces go into the LDS image as they are.  With C16, EPI_GATE's gate tensor (`pe`) is bf16 as well.
template <int EPI, int NWM = 2, int DIAG = 0, int NJ = 2, int PREC = 0, int C16 = 0, int KW = 32, int A16 = 0>     // DIAG (tools/diag_gemm.py only) 2: epilogue skipped (wrong output) + per-wave cycles/wall clock; 1, 3: the same with the epilogue
__global__ __launch_bounds__(128 * NWM, 2) void gemm_nt_128(
    const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
    float *__restrict__ C, int M, int N, int K, const float *__restrict__ pe, int T, int H, int dh,
    unsigned long long *__restrict__ diag = nullptr, EpiArgs ea = EpiArgs{0ull, 0u, 0.f, 0.f}) {
    static_assert(KW == 32 || (KW == 64 && (PREC == 1 || PREC == 3)), "64-wide k-tiles exist for the 16-bit operands only");
    static_assert(A16 == 0 || PREC == 1 || PREC == 3, "16-bit-stored A belongs to the 16-bit operand modes");
    constexpr bool LP1 = PREC == 1 || PREC == 3;      // ONE 16-bit plane per operand: bf16 (1) or f16 (3: the training path's fp16 mode)
    constexpr int F16 = PREC == 3 ? 1 : 0;
    (void)LP1; (void)F16;
    typedef unsigned short a16_t;
    constexpr int BM = 64 * NWM, BN = 64 * NJ, BK = KW, LD = 36;
    constexpr int NT = 128 * NWM;                       // threads
    constexpr int F4R = BK / 4;                         // float4 per k-tile row
    constexpr int LA = BM * F4R / NT, LW = BN * F4R / NT;   // float4 of A / of W per thread per k-tile (4, 4 | 4, 2; KW 64: 8, 8)
    constexpr int RS = NT / F4R;                        // row stride of the staging map
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LD];

    const int tiles_n = (N + BN - 1) / BN;
    const int ntiles = ((M + BM - 1) / BM) * tiles_n;
    // this block's tile list: start + j, start + j + G, ...   (chunk [start, start+len) per XCD label)
    const int xl = blockIdx.x & 7, j = blockIdx.x >> 3, G = gridDim.x >> 3;
    const int cq = ntiles >> 3, cr = ntiles & 7;
    const int start = xl * cq + (xl < cr ? xl : cr), len = cq + (xl < cr ? 1 : 0);
    const int my_tiles = len > j ? (len - j + G - 1) / G : 0;
    if (my_tiles == 0) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int lrow = tid / F4R, lc4 = (tid % F4R) * 4;   // staging map: rows lrow + RS*i
    const int nk = K / BK;

    // Per-output-tile state, set up ONCE per tile (integer division, 64-bit row pointers, this lane's
    // bias values).  `nx_*` belongs to the tile being prefetched, `cu_*` to the tile being accumulated.
    const float *aptr[LA], *wptr[LW];
    int nx_m0 = 0, nx_n0 = 0, cu_m0 = 0, cu_n0 = 0;
    float nx_bias[NJ], cu_bias[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) { nx_bias[jj] = 0.f; cu_bias[jj] = 0.f; }
    auto set_tile = [&](int it) __attribute__((always_inline)) {
        const int tile = start + j + it * G;
        nx_m0 = (tile / tiles_n) * BM;
        nx_n0 = (tile % tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            int ar = nx_m0 + lrow + RS * i; ar = ar < M ? ar : M - 1;
            aptr[i] = A16 ? (const float *)((const a16_t *)A + (size_t)ar * K + lc4) : A + (size_t)ar * K + lc4;
        }
#pragma unroll
        for (int i = 0; i < LW; ++i) {
            int wrow = nx_n0 + lrow + RS * i; wrow = wrow < N ? wrow : N - 1;
            wptr[i] = W + (size_t)wrow * K + lc4;
        }
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int col = nx_n0 + 32 * NJ * wc + 32 * jj + r;
            const float bv = bias[col < N ? col : N - 1];
            nx_bias[jj] = h == 0 ? bv : 0.f;      // A operand of the bias step: A[n = r][k = h]
        }
    };
    f32x4 pa[LA], pw[LW];
    // (A16: a piece is 4 bf16 = the first two dwords of pa[i])
    auto load_a = [&](int i, int koff) __attribute__((always_inline)) {
        if constexpr (A16 != 0) {
            const f32x2 v = *(const f32x2 *)((const a16_t *)aptr[i] + koff);      // 8 bytes, moved as they are
            pa[i][0] = v[0]; pa[i][1] = v[1];
        } else
            pa[i] = *(const f32x4 *)(aptr[i] + koff);
    };
    constexpr int LDB = BK == 64 ? 36 : 20;             // BF: LDS row stride in floats (32 bf16 + pad = 80 B | 64 bf16 + pad = 144 B)
    auto stage = [&](int buf) __attribute__((always_inline)) {
        float *As = smem + buf * (BM + BN) * LD, *Ws = As + BM * LD;
        if constexpr (PREC == 2) {
#pragma unroll
            for (int i = 0; i < LA; ++i) {
                u32x2 hi, lo;
                split_f16x4((const float *)&pa[i], hi, lo);
                *(u32x2 *)&As[(lrow + RS * i) * LD + lc4 / 2] = hi;
                *(u32x2 *)&As[(lrow + RS * i) * LD + 16 + lc4 / 2] = lo;
            }
#pragma unroll
            for (int i = 0; i < LW; ++i) {
                u32x2 hi, lo;
                const f32x4 ws = pw[i] * F16X3_WS;
                split_f16x4((const float *)&ws, hi, lo);
                *(u32x2 *)&Ws[(lrow + RS * i) * LD + lc4 / 2] = hi;
                *(u32x2 *)&Ws[(lrow + RS * i) * LD + 16 + lc4 / 2] = lo;
            }
        } else if constexpr (LP1) {
#pragma unroll
            for (int i = 0; i < LA; ++i) {
                u32x2 u;
                if constexpr (A16 != 0) { u[0] = f32_bits(pa[i][0]); u[1] = f32_bits(pa[i][1]); }
                else { u[0] = pack_lp<F16>(pa[i][0], pa[i][1]); u[1] = pack_lp<F16>(pa[i][2], pa[i][3]); }
                *(u32x2 *)&As[(lrow + RS * i) * LDB + lc4 / 2] = u;
            }
#pragma unroll
            for (int i = 0; i < LW; ++i) {
                u32x2 u; u[0] = pack_lp<F16>(pw[i][0], pw[i][1]); u[1] = pack_lp<F16>(pw[i][2], pw[i][3]);
                *(u32x2 *)&Ws[(lrow + RS * i) * LDB + lc4 / 2] = u;
            }
        } else {
#pragma unroll
            for (int i = 0; i < LA; ++i) *(f32x4 *)&As[(lrow + RS * i) * LD + lc4] = pa[i];
#pragma unroll
            for (int i = 0; i < LW; ++i) *(f32x4 *)&Ws[(lrow + RS * i) * LD + lc4] = pw[i];
        }
    };

    // acc[i][jj][t] = C[m = 64wr + 32i + r][n = 64wc + 32jj + acc_row(t,h)]  (lane = output ROW:
    // the W fragment is the MFMA A operand, the activation fragment the B operand)
    f32x16 acc[2][NJ];
    unsigned long long dsum[5] = {0, 0, 0, 0, 0}, ts0 = 0;
    int fpar = 0;                                   // LDS buffer holding the k-tile about to be consumed

    // One k-tile: 64 MFMAs from LDS buffer `fpar`, with the global loads of the NEXT k-tile (at
    // aptr/wptr + koff) in the first half of the MFMA stream and their LDS writes in the last quarter.
    auto ktile = [&](int koff) __attribute__((always_inline)) {
        const float *As = smem + fpar * (BM + BN) * LD, *Ws = As + BM * LD;
        if constexpr (PREC == 2) {
            // 2 k-steps of 16: lane (r,h) supplies k = 16ks + 8h .. +7 of the hi half-row and of the lo half-row
            const float *ap = As + (64 * wr + r) * LD + 4 * h;
            const float *wp = Ws + (32 * NJ * wc + r) * LD + 4 * h;
#pragma unroll
            for (int i = 0; i < LA; ++i) pa[i] = *(const f32x4 *)(aptr[i] + koff);
#pragma unroll
            for (int i = 0; i < LW; ++i) pw[i] = *(const f32x4 *)(wptr[i] + koff);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 ah[2], al[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[i] = __builtin_bit_cast(f16x8, *(const u32x4 *)(ap + 32 * i * LD + 8 * ks));
                    al[i] = __builtin_bit_cast(f16x8, *(const u32x4 *)(ap + 32 * i * LD + 16 + 8 * ks));
                }
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    const f16x8 wh = __builtin_bit_cast(f16x8, *(const u32x4 *)(wp + 32 * jj * LD + 8 * ks));
                    const f16x8 wl = __builtin_bit_cast(f16x8, *(const u32x4 *)(wp + 32 * jj * LD + 16 + 8 * ks));
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        acc[i][jj] = MFMA_F16(wl, ah[i], acc[i][jj]);
                        acc[i][jj] = MFMA_F16(wh, al[i], acc[i][jj]);
                        acc[i][jj] = MFMA_F16(wh, ah[i], acc[i][jj]);
                    }
                }
            }
            stage(fpar ^ 1);
            __syncthreads();
            fpar ^= 1;
            return;
        }
        if constexpr (LP1) {
            // BK / 16 k-steps of 16: lane (r,h) supplies k = 16ks + 8h .. +7 (one b128 of the bf16 row)
            const float *ap = As + (64 * wr + r) * LDB + 4 * h;
            const float *wp = Ws + (32 * NJ * wc + r) * LDB + 4 * h;
#pragma unroll
            for (int i = 0; i < LA; ++i) load_a(i, koff);
#pragma unroll
            for (int i = 0; i < LW; ++i) pw[i] = *(const f32x4 *)(wptr[i] + koff);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                u32x4 fa[2], fw[NJ];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = *(const u32x4 *)(ap + 32 * i * LDB + 8 * ks);
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) fw[jj] = *(const u32x4 *)(wp + 32 * jj * LDB + 8 * ks);
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    acc[0][jj] = mfma_lp<F16>(fw[jj], fa[0], acc[0][jj]);
                    acc[1][jj] = mfma_lp<F16>(fw[jj], fa[1], acc[1][jj]);
                }
            }
            stage(fpar ^ 1);
            __syncthreads();
            fpar ^= 1;
            return;
        }
        const float *ap = As + (64 * wr + r) * LD + 4 * h;
        const float *wp = Ws + (32 * NJ * wc + r) * LD + 4 * h;
        f32x4 fa[2][2], fw[2][NJ];
        fa[0][0] = *(const f32x4 *)(ap);           fa[0][1] = *(const f32x4 *)(ap + 32 * LD);
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) fw[0][jj] = *(const f32x4 *)(wp + 32 * jj * LD);
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            const int c = g & 1, n = c ^ 1;
            if (g + 1 < BK / 8) {
                fa[n][0] = *(const f32x4 *)(ap + 8 * (g + 1)); fa[n][1] = *(const f32x4 *)(ap + 32 * LD + 8 * (g + 1));
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) fw[n][jj] = *(const f32x4 *)(wp + 32 * jj * LD + 8 * (g + 1));
            }
            if (g < 2) {
#pragma unroll
                for (int i = 0; i < LA / 2; ++i) pa[g * (LA / 2) + i] = *(const f32x4 *)(aptr[g * (LA / 2) + i] + koff);
#pragma unroll
                for (int i = 0; i < LW / 2; ++i) pw[g * (LW / 2) + i] = *(const f32x4 *)(wptr[g * (LW / 2) + i] + koff);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    acc[0][jj] = MFMA32(fw[c][jj][s], fa[c][0][s], acc[0][jj]);
                    acc[1][jj] = MFMA32(fw[c][jj][s], fa[c][1][s], acc[1][jj]);
                }
            }
            if (g == BK / 8 - 1) stage(fpar ^ 1);
            if (g < 2) {
#pragma unroll
                for (int q = 0; q < (LA + LW) / 2; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 8 * NJ / ((LA + LW) / 2), 0);   // MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                           // 1 VMEM read
                }
            } else if (g == BK / 8 - 1) {
#pragma unroll
                for (int q = 0; q < LA + LW; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 8 * NJ / (LA + LW), 0);          // MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                           // 1 DS write
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        fpar ^= 1;
    };

    set_tile(0);
#pragma unroll
    for (int i = 0; i < LA; ++i) load_a(i, 0);
#pragma unroll
    for (int i = 0; i < LW; ++i) pw[i] = *(const f32x4 *)wptr[i];
    stage(0);
    __syncthreads();
    if (DIAG != 0) ts0 = stamp();
    const unsigned long long tbegin = ts0;
    if (DIAG >= 2) dsum[3] = __builtin_amdgcn_s_memrealtime();      // 100 MHz wall clock: wave start

    for (int it = 0; it < my_tiles; ++it) {
        cu_m0 = nx_m0; cu_n0 = nx_n0;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) cu_bias[jj] = nx_bias[jj];
        // accumulators start at the bias: one "bias x ones" MFMA per 32x32 tile with C = 0 replaces the
        // zero-init and 64 adds, and keeps every load out of the epilogue
        {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = MFMA32(cu_bias[jj], PREC == 2 ? F16X3_WS : 1.0f, zero);
        }
        for (int kt = 0; kt + 1 < nk; ++kt) ktile((kt + 1) * BK);
        // last k-tile of this output tile: prefetch the first k-tile of the next one (or a harmless
        // duplicate after the final tile - no branch in the MFMA stream)
        if (it + 1 < my_tiles) set_tile(it + 1);
        ktile(0);

        if constexpr (DIAG == 2) {
            // keep the accumulators live without an epilogue.  (No inline asm here: an asm operand of
            // dependent array type makes hipcc silently drop the kernel's HOST stub - undefined symbol at dlopen.)
            float keep = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                    for (int t = 0; t < 16; ++t) keep += acc[i][jj][t];
            if (keep == 1.2345e-30f && diag != nullptr) diag[0] = 1;
            continue;
        }
        // ---- epilogue.  A lane owns output ROWS (4 consecutive columns per register quad), so a direct
        // store instruction would write 32 B into 32 different cache lines (measured: ~460 cycles per
        // store, 7.4 K cycles per tile).  Instead each 32x32 sub-tile is transposed through a wave-private
        // corner of the LDS staging buffer that is idle right now (its k-tile was just consumed; the other
        // buffer already holds the next tile's first k-tile): write as owned (ds_write_b128), read back with
        // 8 lanes per row, and every store instruction writes 8 full 128-byte lines.  No block barrier is
        // needed for the transposition itself (one wave, in-order LDS), only one afterwards, before any
        // wave restages that buffer.
        const int m0 = cu_m0, n0 = cu_n0;
        float *tp = smem + (fpar ^ 1) * (BM + BN) * LD + wave * (32 * LD);     // 32 x 36 floats per wave
        const int trow = lane >> 3, tc4 = (lane & 7) * 4;                       // read-back map: rows trow + 8p
        int b0 = 0, t0 = 0;                       // (video, frame) of row m0, for EPI_PE / EPI_QKV
        if (EPI == EPI_PE || EPI == EPI_QKV) { b0 = m0 / T; t0 = m0 - b0 * T; }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[i][jj][4 * q + e];
                        if (EPI == EPI_RELU || EPI == EPI_RELU_DROP) v[e] = relu1(v[e]);
                    }
                    *(f32x4 *)&tp[r * LD + 8 * q + 4 * h] = v;
                }
                const int c32 = n0 + 32 * NJ * wc + 32 * jj;          // a 32-column block never straddles a head
                int which = 0, head = 0, e0 = 0;
                if (EPI == EPI_QKV) { const int d = H * dh; which = c32 / d; const int c = c32 - which * d; head = c / dh; e0 = c - head * dh; }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int ro = 64 * wr + 32 * i + trow + 8 * p;
                    f32x4 v = *(const f32x4 *)&tp[(trow + 8 * p) * LD + tc4];
                    if constexpr (PREC == 2) v *= 1.0f / F16X3_WS;
                    const int row = m0 + ro;
                    int bb = b0, tt = t0 + ro;
                    if (EPI == EPI_PE || EPI == EPI_QKV) { while (tt >= T) { tt -= T; ++bb; } }
                    if (row < M && c32 < N) {
                        if (EPI == EPI_PE) {
                            const f32x4 pv = *(const f32x4 *)(pe + (size_t)tt * N + c32 + tc4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += pv[e];
                        }
                        if (EPI == EPI_GATE) {
                            if constexpr (C16 != 0) {       // bf16-stored activation: > 0 <=> its 16 bits are a positive integer
                                const u32x2 g2 = *(const u32x2 *)((const a16_t *)pe + (size_t)row * N + c32 + tc4);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = (short)(g2[e >> 1] >> (16 * (e & 1))) > 0 ? v[e] * ea.scale : 0.f;
                            } else {
                            const f32x4 gv = *(const f32x4 *)(pe + (size_t)row * N + c32 + tc4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = gv[e] > 0.f ? v[e] * ea.scale : 0.f;
                            }
                        }
                        if (EPI == EPI_RELU_DROP) {
                            const DropSite dsite = drop_site(ea.seed, ea.site, ea.p);
                            const unsigned rk = drop_rowkey(dsite, (unsigned)row);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = drop_keep(dsite, rk, (unsigned)(c32 + tc4 + e)) ? v[e] * dsite.scale : 0.f;
                        }
                        if constexpr (C16 != 0) {
                            if (EPI == EPI_QKV && which == 0) v *= ea.scale;
                            u32x2 u; u[0] = pack_lp<F16>(v[0], v[1]); u[1] = pack_lp<F16>(v[2], v[3]);
                            unsigned short *C2 = (unsigned short *)C;
                            if (EPI == EPI_QKV)
                                *(u32x2 *)(C2 + (size_t)which * M * (H * dh) + (((size_t)bb * H + head) * T + tt) * dh + e0 + tc4) = u;
                            else
                                *(u32x2 *)(C2 + (size_t)row * N + c32 + tc4) = u;
                        } else if (EPI == EPI_QKV)
                            *(f32x4 *)(C + (size_t)which * M * (H * dh) + (((size_t)bb * H + head) * T + tt) * dh + e0 + tc4) = v;
                        else
                            *(f32x4 *)(C + (size_t)row * N + c32 + tc4) = v;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (DIAG >= 2) { ts0 = stamp(); dsum[4] = __builtin_amdgcn_s_memrealtime(); }
    if (DIAG != 0 && diag != nullptr && lane == 0) {
        unsigned long long *o = diag + ((size_t)blockIdx.x * (2 * NWM) + wave) * 8;
#pragma unroll
        for (int i = 0; i < 5; ++i) o[i] = dsum[i];
        o[5] = ts0 - tbegin; o[6] = (unsigned long long)(my_tiles * nk); o[7] = tbegin;
    }
}

// ------------------------------------------------------------------------------------------
// Projection + residual + LayerNorm (+ score head):  one block owns 64 full rows of d = 64*NB
// columns, so mean/variance and the final_layer dot product are reductions inside the block.
//   4 waves as 2 (rows) x 2 (column halves); each wave 32 rows x 32*NB columns.
//   BK = 16, LDS row pad 20 floats (20*r mod 64 = 4*(5r mod 16): conflict-free b128 reads).
// ------------------------------------------------------------------------------------------
template <int NB, int PREC = 0>      // PREC 2: fp32 emulated on the f16 pipe (rows hold 16 hi then 16 lo halves: the same 80 B)
__global__ __launch_bounds__(256) void gemm_res_ln(
    const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
    const float *__restrict__ res, const float *__restrict__ gamma, const float *__restrict__ beta,
    float *__restrict__ out, int M, int K,
    const float *__restrict__ score_w, const float *__restrict__ score_b, int num_classes,
    int sigmoid, float *__restrict__ scores) {
    constexpr int BM = 64, N = 64 * NB, BK = 16, LD = BK + 4;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + N) * LD];

    const int m0 = blockIdx.x * BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;

    const int lrow = tid >> 2, lc4 = (tid & 3) * 4;      // 64 rows x 4 float4 per pass
    int arow = m0 + lrow; arow = arow < M ? arow : M - 1;
    const float *ag = A + (size_t)arow * K + lc4;
    const float *wg = W + (size_t)lrow * K + lc4;         // + 64*i rows
    f32x4 pa, pw[NB];
    pa = *(const f32x4 *)ag;
#pragma unroll
    for (int i = 0; i < NB; ++i) pw[i] = *(const f32x4 *)(wg + (size_t)64 * i * K);

    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[j][t] = 0.f;

    auto put = [&](float *rowp, f32x4 v, float scale) __attribute__((always_inline)) {
        if constexpr (PREC == 2) {
            v *= scale;
            u32x2 hi, lo;
            split_f16x4((const float *)&v, hi, lo);
            *(u32x2 *)&rowp[lc4 / 2] = hi;
            *(u32x2 *)&rowp[8 + lc4 / 2] = lo;
        } else {
            *(f32x4 *)&rowp[lc4] = v;
        }
    };
    {
        float *As = smem, *Ws = smem + BM * LD;
        put(&As[lrow * LD], pa, 1.0f);
#pragma unroll
        for (int i = 0; i < NB; ++i) put(&Ws[(lrow + 64 * i) * LD], pw[i], F16X3_WS);
    }
    __syncthreads();

    const int nk = K / BK;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const float *As = smem + cur * (BM + N) * LD, *Ws = As + BM * LD;
        const bool more = kt + 1 < nk;
        if (more) {
            pa = *(const f32x4 *)(ag + (kt + 1) * BK);
#pragma unroll
            for (int i = 0; i < NB; ++i) pw[i] = *(const f32x4 *)(wg + (size_t)64 * i * K + (kt + 1) * BK);
        }
        const float *ap = As + (32 * wr + r) * LD + 4 * h;
        const float *wp = Ws + (32 * NB * wc + r) * LD + 4 * h;
        if constexpr (PREC == 2) {
            // one k-step of 16: lane (r,h) supplies k = 8h .. 8h+7 of the hi half-row and of the lo half-row
            const f16x8 ah = __builtin_bit_cast(f16x8, *(const u32x4 *)ap), al = __builtin_bit_cast(f16x8, *(const u32x4 *)(ap + 8));
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const f16x8 bh = __builtin_bit_cast(f16x8, *(const u32x4 *)(wp + 32 * j * LD));
                const f16x8 bl = __builtin_bit_cast(f16x8, *(const u32x4 *)(wp + 32 * j * LD + 8));
                acc[j] = MFMA_F16(ah, bl, acc[j]);
                acc[j] = MFMA_F16(al, bh, acc[j]);
                acc[j] = MFMA_F16(ah, bh, acc[j]);
            }
        } else {
#pragma unroll
            for (int g = 0; g < BK / 8; ++g) {
                const f32x4 a = *(const f32x4 *)(ap + 8 * g);
                f32x4 b[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) b[j] = *(const f32x4 *)(wp + 32 * j * LD + 8 * g);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc[j] = MFMA32(a[s], b[j][s], acc[j]);
            }
        }
        if (more) {
            float *An = smem + (cur ^ 1) * (BM + N) * LD, *Wn = An + BM * LD;
            put(&An[lrow * LD], pa, 1.0f);
#pragma unroll
            for (int i = 0; i < NB; ++i) put(&Wn[(lrow + 64 * i) * LD], pw[i], F16X3_WS);
        }
        __syncthreads();
    }

    // ---- epilogue: v = acc + bias + residual; two-pass LayerNorm over the row ----
    float *red = smem;                       // [2][64] exchange between the two column halves
    const int cbase = 32 * NB * wc + r;      // + 32*j
    float bj[NB], gj[NB], bej[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) { bj[j] = bias[cbase + 32 * j]; gj[j] = gamma[cbase + 32 * j]; bej[j] = beta[cbase + 32 * j]; }

    float part[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        int row = m0 + 32 * wr + acc_row(t, h); row = row < M ? row : M - 1;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float v = (PREC == 2 ? acc[j][t] * (1.0f / F16X3_WS) : acc[j][t]) + bj[j] + res[(size_t)row * N + cbase + 32 * j];
            acc[j][t] = v;
            s += v;
        }
        part[t] = half_sum(s);
    }
    if (r == 0) {
#pragma unroll
        for (int t = 0; t < 16; ++t) red[wc * 64 + 32 * wr + acc_row(t, h)] = part[t];
    }
    __syncthreads();
    float mean[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int rl = 32 * wr + acc_row(t, h);
        mean[t] = (red[rl] + red[64 + rl]) * (1.0f / N);
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j) { const float c = acc[j][t] - mean[t]; acc[j][t] = c; s += c * c; }
        part[t] = half_sum(s);
    }
    if (r == 0) {
#pragma unroll
        for (int t = 0; t < 16; ++t) red[wc * 64 + 32 * wr + acc_row(t, h)] = part[t];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int rl = 32 * wr + acc_row(t, h);
        const float rstd = 1.0f / sqrtf((red[rl] + red[64 + rl]) * (1.0f / N) + 1e-5f);
        const int row = m0 + rl;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float y = acc[j][t] * rstd * gj[j] + bej[j];
            acc[j][t] = y;
            if (row < M) out[(size_t)row * N + cbase + 32 * j] = y;
        }
    }
    // ---- optional score head: scores[row, c] = y . score_w[c,:] + score_b[c] ----
    if (score_w != nullptr) {
        for (int c = 0; c < num_classes; ++c) {
            __syncthreads();
            float wj[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) wj[j] = score_w[(size_t)c * N + cbase + 32 * j];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < NB; ++j) s += acc[j][t] * wj[j];
                part[t] = half_sum(s);
            }
            if (r == 0) {
#pragma unroll
                for (int t = 0; t < 16; ++t) red[wc * 64 + 32 * wr + acc_row(t, h)] = part[t];
            }
            __syncthreads();
            if (tid < 64) {
                const int row = m0 + tid;
                if (row < M) {
                    float s = red[tid] + red[64 + tid] + score_b[c];
                    if (sigmoid) s = 1.0f / (1.0f + expf(-s));
                    scores[(size_t)row * num_classes + c] = s;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Projection + residual + LayerNorm (+ score head), d_model <= 256: each WAVE owns 32 full rows.
//   out = LN(A*W^T + bias + residual)*gamma + beta;  scores = out . score_w + score_b
//   Swapped operands (W fragment = MFMA A operand) put one output ROW on each lane pair (l, l^32):
//   lane (r,h) holds row 32w+r, columns 32j + 8q + 4h + e of all N = 32*NT columns in acc[NT].
//   So mean, variance and the score dot product are in-lane sums plus ONE exchange with lane^32 -
//   no LDS reduction and no block barrier in the epilogue - and the stores are 16-byte row pieces.
//   residual + bias are loaded straight into the accumulators before the first MFMA (C-in), so the
//   epilogue issues no loads from HBM at all; gamma/beta/score_w sit in LDS.
//   Block = 4 waves = 128 rows, BK = 16 (LDS rows padded to 20 floats), 2 blocks per CU.
// ------------------------------------------------------------------------------------------
// A16 (PREC 1 only): A lives in HBM as bf16 (written so by the bf16 attention / the C16 fc1 epilogue)
template <int NT, int PREC = 0, int A16 = 0>     // PREC 1: bf16 MFMA operands (see gemm_nt_128), LDS rows of 16 bf16 padded to 48 B; 2: f16 hi|lo rows (80 B)
__global__ __launch_bounds__(256, 2) void gemm_ln_rows(
    const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
    const float *__restrict__ res, const float *__restrict__ gamma, const float *__restrict__ beta,
    float *__restrict__ out, int M, int K,
    const float *__restrict__ score_w, const float *__restrict__ score_b, int num_classes,
    int sigmoid, float *__restrict__ scores) {
    constexpr int BM = 128, N = 32 * NT, BK = 16, LD = BK + 4;
    constexpr int WL = (N * BK / 4 + 255) / 256;       // float4 of W per thread per k-tile (N=256: 4)
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + N) * LD + 4 * N];
    float *gam_s = smem + 2 * (BM + N) * LD, *bet_s = gam_s + N, *sw_s = bet_s + N, *bias_s = sw_s + N;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int ntiles = (M + BM - 1) / BM;
    const int nk = K / BK;

    constexpr float SC = PREC == 2 ? F16X3_WS : 1.0f;      // scale of the accumulators (see F16X3_WS)
    for (int i = tid; i < N; i += 256) { gam_s[i] = gamma[i]; bet_s[i] = beta[i]; bias_s[i] = bias[i] * SC; }

    // staging map: A 128 rows x 4 float4 (2 per thread), W N rows x 4 float4 (WL per thread)
    const int lrow = tid >> 2, lc4 = (tid & 3) * 4;
    f32x4 pa[2], pw[WL];
    u32x2 pa16[2];                                     // A16: 4 bf16 per thread and k-tile
    const float *aptr[2], *wptr[WL];
    const unsigned short *aptr16[2];
#pragma unroll
    for (int i = 0; i < WL; ++i) {
        int wrow = lrow + 64 * i; wrow = wrow < N ? wrow : N - 1;
        wptr[i] = W + (size_t)wrow * K + lc4;
    }
    constexpr int LDB = 12;                            // BF: LDS row stride in floats (48 B: conflict-free b128)
    auto stage = [&](int buf) __attribute__((always_inline)) {
        float *As = smem + buf * (BM + N) * LD, *Ws = As + BM * LD;
        if constexpr (PREC == 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                u32x2 hi, lo;
                split_f16x4((const float *)&pa[i], hi, lo);
                *(u32x2 *)&As[(lrow + 64 * i) * LD + lc4 / 2] = hi;
                *(u32x2 *)&As[(lrow + 64 * i) * LD + 8 + lc4 / 2] = lo;
            }
#pragma unroll
            for (int i = 0; i < WL; ++i)
                if (lrow + 64 * i < N) {
                    u32x2 hi, lo;
                    const f32x4 ws = pw[i] * F16X3_WS;
                    split_f16x4((const float *)&ws, hi, lo);
                    *(u32x2 *)&Ws[(lrow + 64 * i) * LD + lc4 / 2] = hi;
                    *(u32x2 *)&Ws[(lrow + 64 * i) * LD + 8 + lc4 / 2] = lo;
                }
        } else if constexpr (PREC == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                u32x2 u;
                if constexpr (A16 != 0) u = pa16[i];
                else { u[0] = pack_bf16(pa[i][0], pa[i][1]); u[1] = pack_bf16(pa[i][2], pa[i][3]); }
                *(u32x2 *)&As[(lrow + 64 * i) * LDB + lc4 / 2] = u;
            }
#pragma unroll
            for (int i = 0; i < WL; ++i)
                if (lrow + 64 * i < N) {
                    u32x2 u; u[0] = pack_bf16(pw[i][0], pw[i][1]); u[1] = pack_bf16(pw[i][2], pw[i][3]);
                    *(u32x2 *)&Ws[(lrow + 64 * i) * LDB + lc4 / 2] = u;
                }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) *(f32x4 *)&As[(lrow + 64 * i) * LD + lc4] = pa[i];
#pragma unroll
            for (int i = 0; i < WL; ++i)
                if (lrow + 64 * i < N) *(f32x4 *)&Ws[(lrow + 64 * i) * LD + lc4] = pw[i];
        }
    };

    f32x16 acc[NT];
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * BM;
        int row = m0 + 32 * wave + r;
        const bool row_ok = row < M;
        row = row_ok ? row : M - 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int ar = m0 + lrow + 64 * i; ar = ar < M ? ar : M - 1;
            if constexpr (A16 != 0) {
                aptr16[i] = (const unsigned short *)A + (size_t)ar * K + lc4;
                pa16[i] = *(const u32x2 *)aptr16[i];
            } else {
                aptr[i] = A + (size_t)ar * K + lc4;
                pa[i] = *(const f32x4 *)aptr[i];
            }
        }
#pragma unroll
        for (int i = 0; i < WL; ++i) pw[i] = *(const f32x4 *)wptr[i];
        // accumulators start at the residual (C-in of the first MFMA); bias joins in the epilogue from LDS
        {
            const float *rp = res + (size_t)row * N + 4 * h;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 rv = *(const f32x4 *)(rp + 32 * j + 8 * q);
                    if constexpr (PREC == 2) rv *= SC;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][4 * q + e] = rv[e];
                }
        }
        __syncthreads();                 // previous tile's readers are done with both LDS buffers
        stage(0);
        __syncthreads();

        for (int kt = 0; kt < nk; ++kt) {
            const int kn = kt + 1 < nk ? kt + 1 : kt;         // last step reloads a duplicate: branch-free stream
            const float *As = smem + (kt & 1) * (BM + N) * LD, *Ws = As + BM * LD;
            if constexpr (PREC == 2) {
                const float *arow = As + (32 * wave + r) * LD + 4 * h;
                const f16x8 ah = __builtin_bit_cast(f16x8, *(const u32x4 *)arow);
                const f16x8 al = __builtin_bit_cast(f16x8, *(const u32x4 *)(arow + 8));
#pragma unroll
                for (int i = 0; i < 2; ++i) pa[i] = *(const f32x4 *)(aptr[i] + kn * BK);
#pragma unroll
                for (int i = 0; i < WL; ++i) pw[i] = *(const f32x4 *)(wptr[i] + kn * BK);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float *wrow = Ws + (32 * j + r) * LD + 4 * h;
                    const f16x8 wh = __builtin_bit_cast(f16x8, *(const u32x4 *)wrow);
                    const f16x8 wl = __builtin_bit_cast(f16x8, *(const u32x4 *)(wrow + 8));
                    acc[j] = MFMA_F16(wl, ah, acc[j]);
                    acc[j] = MFMA_F16(wh, al, acc[j]);
                    acc[j] = MFMA_F16(wh, ah, acc[j]);
                }
                stage((kt + 1) & 1);
                __syncthreads();
                continue;
            }
            if constexpr (PREC == 1) {
                // one k-step of 16: lane (r,h) supplies k = 8h .. 8h+7
                const bf16x8 fa = __builtin_bit_cast(bf16x8, *(const u32x4 *)(As + (32 * wave + r) * LDB + 4 * h));
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if constexpr (A16 != 0) pa16[i] = *(const u32x2 *)(aptr16[i] + kn * BK);
                    else pa[i] = *(const f32x4 *)(aptr[i] + kn * BK);
                }
#pragma unroll
                for (int i = 0; i < WL; ++i) pw[i] = *(const f32x4 *)(wptr[i] + kn * BK);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const bf16x8 fw = __builtin_bit_cast(bf16x8, *(const u32x4 *)(Ws + (32 * j + r) * LDB + 4 * h));
                    acc[j] = MFMA_BF16(fw, fa, acc[j]);
                }
                stage((kt + 1) & 1);
                __syncthreads();
                continue;
            }
            const float *ap = As + (32 * wave + r) * LD + 4 * h;
            const float *wp = Ws + r * LD + 4 * h;
#pragma unroll
            for (int g = 0; g < BK / 8; ++g) {
                const f32x4 fa = *(const f32x4 *)(ap + 8 * g);
                if (g == 0) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) pa[i] = *(const f32x4 *)(aptr[i] + kn * BK);
#pragma unroll
                    for (int i = 0; i < WL; ++i) pw[i] = *(const f32x4 *)(wptr[i] + kn * BK);
                }
                // one 32-column tile at a time: a 4-step dependent chain on acc[j] issues back-to-back
                // (latency == issue interval for 32x32x2), and only one weight fragment is live
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const f32x4 fw = *(const f32x4 *)(wp + 32 * j * LD + 8 * g);
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[j] = MFMA32(fw[s], fa[s], acc[j]);
                }
                if (g == BK / 8 - 1) stage((kt + 1) & 1);
                if (g == 0) {
#pragma unroll
                    for (int q = 0; q < 2 + WL; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, (4 * NT) / (2 + WL), 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 2 + WL; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, (4 * NT) / (2 + WL), 0);
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }

        // ---- epilogue: LayerNorm over the row (lane-local + one lane^32 exchange), 16-byte stores ----
        // (reduction order shared with skinny_ln: per 32-column block, blocks ascending, partner last)
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float pj = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *(const f32x4 *)&bias_s[32 * j + 8 * q + 4 * h];
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[j][4 * q + e] += bv[e]; pj += acc[j][4 * q + e]; }
            }
            sum += pj;
        }
        sum = pair_sum(sum);
        const float mean = sum * (1.0f / N);
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float pj = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) { const float c = acc[j][t] - mean; acc[j][t] = c; pj += c * c; }
            sq += pj;
        }
        sq = pair_sum(sq);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / N) + 1e-5f * (SC * SC));
        // stores: each 32x32 block is transposed through a wave-private corner of the (now idle) staging
        // buffers so that a store instruction writes 8 full 128-byte lines instead of 32 B into 32 lines
        float *tp = smem + wave * (32 * 36);
        const int trow = lane >> 3, tc4 = (lane & 7) * 4;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 gv = *(const f32x4 *)&gam_s[32 * j + 8 * q + 4 * h];
                const f32x4 bv = *(const f32x4 *)&bet_s[32 * j + 8 * q + 4 * h];
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) { y[e] = acc[j][4 * q + e] * rstd * gv[e] + bv[e]; acc[j][4 * q + e] = y[e]; }
                *(f32x4 *)&tp[r * 36 + 8 * q + 4 * h] = y;
            }
#pragma unroll
            for (int pq = 0; pq < 4; ++pq) {
                const f32x4 v = *(const f32x4 *)&tp[(trow + 8 * pq) * 36 + tc4];
                const int orow = m0 + 32 * wave + trow + 8 * pq;
                if (orow < M) *(f32x4 *)(out + (size_t)orow * N + 32 * j + tc4) = v;
            }
        }
        if (score_w != nullptr) {
            for (int c = 0; c < num_classes; ++c) {
                __syncthreads();
                for (int i = tid; i < N; i += 256) sw_s[i] = score_w[(size_t)c * N + i];
                __syncthreads();
                float dot = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    float pj = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 wv = *(const f32x4 *)&sw_s[32 * j + 8 * q + 4 * h];
#pragma unroll
                        for (int e = 0; e < 4; ++e) pj += acc[j][4 * q + e] * wv[e];
                    }
                    dot += pj;
                }
                dot = pair_sum(dot);
                if (h == 0 && row_ok) {
                    float sc = dot + score_b[c];
                    if (sigmoid) sc = 1.0f / (1.0f + expf(-sc));
                    scores[(size_t)row * num_classes + c] = sc;
                }
            }
        }
    }
}

#ifdef VS_WITH_DIAG      // negative result kept for tools/ only: not in the product library
// ------------------------------------------------------------------------------------------
// Fused MLP block (d_model = 256, hidden = 1024):
//     out = LN2( relu(h1 * W1^T + b1) * W2^T + b2 + h1 ) * gamma + beta      (+ score head)
// (reference simnet.py:109-110,180-183,42) in ONE kernel, activations never leaving registers:
//   * a wave owns 32 rows; X = its h1 rows as MFMA B-operand fragments (128 registers: lane (r,h) holds
//     row r, columns 32j + 8q + 4h + e in X[j][4q+e]);
//   * per 128-column chunk of the hidden layer: U = W1[chunk] * X^T  (X registers are the B operands),
//     ReLU in place, then Y += W2[:, chunk] * U^T  (the U accumulators are the B operands, the same
//     k permutation);  Y starts at X (the residual) and ends in the LayerNorm epilogue of gemm_ln_rows.
//   Only the WEIGHT tiles stream through LDS (double buffer, one barrier per 32-k tile); LDS reads per
//   MFMA are half those of the tiled GEMMs, the [M,1024] hidden tensor (256 MiB write + read at M=65536)
//   never exists, and three launches become two.  One block (4 waves, 128 rows, ~380 registers per
//   lane) per CU.  Summation orders equal those of gemm_nt_128 / gemm_ln_rows / the skinny kernels, so the
//   result is bit-identical to the unfused path.
// ------------------------------------------------------------------------------------------
template <int ABL>      // diagnostic ablation (timing only): 1 no weight loads/LDS writes, 2 no barriers in the step loop
__global__ __launch_bounds__(256, 1) void mlp_fused_256(
    const float *__restrict__ H1, const float *__restrict__ W1, const float *__restrict__ b1,
    const float *__restrict__ W2, const float *__restrict__ b2, const float *__restrict__ gamma,
    const float *__restrict__ beta, float *__restrict__ out, int M,
    const float *__restrict__ score_w, const float *__restrict__ score_b, int num_classes,
    int sigmoid, float *__restrict__ scores) {
    constexpr int D = 256, HID = 1024, CH = 128, BK = 32, LD = BK + 4, NT = 8;
    constexpr int BUF = 256 * LD;                                   // floats per staging buffer
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF + 4 * D];
    float *gam_s = smem + 2 * BUF, *bet_s = gam_s + D, *sw_s = bet_s + D, *bias_s = sw_s + D;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    for (int i = tid; i < D; i += 256) { gam_s[i] = gamma[i]; bet_s[i] = beta[i]; bias_s[i] = b2[i]; }

    // Weight tiles, 128 MFMAs per wave each, 8 float4 per thread:
    //   fc1 tile (chunk c, k-tile kt of 64): W1 rows 128c + .., columns 64kt + ..  -> LDS [128][68]
    //   fc2 tile (chunk c, k-tile kt of 32): W2 rows 0..255, columns 128c + 32kt + .. -> LDS [256][36]
    constexpr int LD1 = 68, LD2 = 36;
    const int r1 = tid >> 4, c1 = (tid & 15) * 4;          // fc1 tile: rows r1 + 16*i
    const int r2 = tid >> 3, c2 = (tid & 7) * 4;           // fc2 tile: rows r2 + 32*i
    f32x4 pa[8];
    auto load_w1 = [&](int c, int kt) __attribute__((always_inline)) {
        const float *p = W1 + (size_t)(CH * c + r1) * D + 64 * kt + c1;
#pragma unroll
        for (int i = 0; i < 8; ++i) pa[i] = *(const f32x4 *)(p + (size_t)16 * i * D);
    };
    auto load_w2 = [&](int c, int kt) __attribute__((always_inline)) {
        const float *p = W2 + (size_t)r2 * HID + CH * c + BK * kt + c2;
#pragma unroll
        for (int i = 0; i < 8; ++i) pa[i] = *(const f32x4 *)(p + (size_t)32 * i * HID);
    };
    auto stage_w1 = [&](int buf) __attribute__((always_inline)) {
        float *Ws = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < 8; ++i) *(f32x4 *)&Ws[(r1 + 16 * i) * LD1 + c1] = pa[i];
    };
    auto stage_w2 = [&](int buf) __attribute__((always_inline)) {
        float *Ws = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < 8; ++i) *(f32x4 *)&Ws[(r2 + 32 * i) * LD2 + c2] = pa[i];
    };
    // one step's schedule: 8 global loads among the first 32 MFMAs, 8 LDS writes among the last 32
    auto step_schedule = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 64, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
    };

    f32x16 X[NT], Y[NT], U[4];
    const int ntiles = (M + 127) / 128;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * 128 + 32 * wave;
        // ---- X: this wave's 32 rows of h1, loaded coalesced and transposed through a wave-private LDS corner
        // (two column halves of 128; 4 waves x 32 x 132 floats = 66 KiB of the idle staging area) ----
        __syncthreads();
        {
            float *tp = smem + wave * (32 * 132);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {                                  // 32 rows x 128 floats = 16 wave loads
                    const int idx = lane + 64 * i, row = idx >> 5, c4 = (idx & 31) * 4;
                    int gr = m0 + row; gr = gr < M ? gr : M - 1;
                    *(f32x4 *)&tp[row * 132 + c4] = *(const f32x4 *)(H1 + (size_t)gr * D + 128 * half + c4);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = *(const f32x4 *)&tp[r * 132 + 32 * j + 8 * q + 4 * h];
#pragma unroll
                        for (int e = 0; e < 4; ++e) X[4 * half + j][4 * q + e] = v[e];
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) Y[j] = X[j];                               // residual (simnet.py:110)
        __syncthreads();
        load_w1(0, 0);
        stage_w1(0);
        __syncthreads();

        int fpar = 0;
        for (int c = 0; c < HID / CH; ++c) {
            // U = b1[chunk] (bias x ones MFMA, C = 0)
            {
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) U[jj] = MFMA32(h == 0 ? b1[CH * c + 32 * jj + r] : 0.f, 1.0f, zero);
            }
            // ---- fc1: 4 k-tiles of 64, B operand = X[2kt], X[2kt+1] ----
            static_for<4>([&](auto ktc) {
                constexpr int kt = decltype(ktc)::value;
                const float *Ws = smem + fpar * BUF + r * LD1 + 4 * h;
                if constexpr (!(ABL & 1)) { if constexpr (kt < 3) load_w1(c, kt + 1); else load_w2(c, 0); }
                {   // weight fragment double-buffered by hand: the next ds_read is in flight under 4 MFMAs
                    f32x4 wn = *(const f32x4 *)Ws;
#pragma unroll
                    for (int i = 0; i < 32; ++i) {
                        const int g = i >> 2, jj = i & 3;                       // g: 8 groups of 8 k
                        const f32x4 w = wn;
                        if (i + 1 < 32) wn = *(const f32x4 *)(Ws + 32 * ((i + 1) & 3) * LD1 + 8 * ((i + 1) >> 2));
#pragma unroll
                        for (int st = 0; st < 4; ++st) U[jj] = MFMA32(w[st], X[2 * kt + (g >> 2)][4 * (g & 3) + st], U[jj]);
                    }
                }
                if constexpr (!(ABL & 1)) { if constexpr (kt < 3) stage_w1(fpar ^ 1); else stage_w2(fpar ^ 1); }
                step_schedule();
                if constexpr (!(ABL & 2)) __syncthreads();
                fpar ^= 1;
            });
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int t = 0; t < 16; ++t) U[jj][t] = relu1(U[jj][t]);
            // ---- fc2 partial: 4 k-tiles of 32 of this chunk, B operand = relu(U)[kt] ----
            static_for<4>([&](auto ktc) {
                constexpr int kt = decltype(ktc)::value;
                const float *Ws = smem + fpar * BUF + r * LD2 + 4 * h;
                const int cn = c + 1 < HID / CH ? c + 1 : c;                    // after the last chunk: harmless reload
                if constexpr (!(ABL & 1)) { if constexpr (kt < 3) load_w2(c, kt + 1); else load_w1(cn, 0); }
                {
                    f32x4 wn = *(const f32x4 *)Ws;
#pragma unroll
                    for (int i = 0; i < 32; ++i) {
                        const int g = i >> 3, j = i & 7;
                        const f32x4 w = wn;
                        if (i + 1 < 32) wn = *(const f32x4 *)(Ws + 32 * ((i + 1) & 7) * LD2 + 8 * ((i + 1) >> 3));
#pragma unroll
                        for (int st = 0; st < 4; ++st) Y[j] = MFMA32(w[st], U[kt][4 * g + st], Y[j]);
                    }
                }
                if constexpr (!(ABL & 1)) { if constexpr (kt < 3) stage_w2(fpar ^ 1); else stage_w1(fpar ^ 1); }
                step_schedule();
                if constexpr (!(ABL & 2)) __syncthreads();
                fpar ^= 1;
            });
        }

        // ---- epilogue: + b2, LayerNorm (same reduction trees as gemm_ln_rows), coalesced stores, score head ----
        const int row = m0 + r;
        const bool row_ok = row < M;
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float pj = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *(const f32x4 *)&bias_s[32 * j + 8 * q + 4 * h];
#pragma unroll
                for (int e = 0; e < 4; ++e) { Y[j][4 * q + e] += bv[e]; pj += Y[j][4 * q + e]; }
            }
            sum += pj;
        }
        sum = pair_sum(sum);
        const float mean = sum * (1.0f / D);
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float pj = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) { const float cv = Y[j][t] - mean; Y[j][t] = cv; pj += cv * cv; }
            sq += pj;
        }
        sq = pair_sum(sq);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / D) + 1e-5f);
        float *tp = smem + wave * (32 * 36);
        const int trow = lane >> 3, tc4 = (lane & 7) * 4;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 gv = *(const f32x4 *)&gam_s[32 * j + 8 * q + 4 * h];
                const f32x4 bv = *(const f32x4 *)&bet_s[32 * j + 8 * q + 4 * h];
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) { y[e] = Y[j][4 * q + e] * rstd * gv[e] + bv[e]; Y[j][4 * q + e] = y[e]; }
                *(f32x4 *)&tp[r * 36 + 8 * q + 4 * h] = y;
            }
#pragma unroll
            for (int pq = 0; pq < 4; ++pq) {
                const f32x4 v = *(const f32x4 *)&tp[(trow + 8 * pq) * 36 + tc4];
                const int orow = m0 + trow + 8 * pq;
                if (orow < M) *(f32x4 *)(out + (size_t)orow * D + 32 * j + tc4) = v;
            }
        }
        if (score_w != nullptr) {
            for (int c = 0; c < num_classes; ++c) {
                __syncthreads();
                for (int i = tid; i < D; i += 256) sw_s[i] = score_w[(size_t)c * D + i];
                __syncthreads();
                float dot = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    float pj = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 wv = *(const f32x4 *)&sw_s[32 * j + 8 * q + 4 * h];
#pragma unroll
                        for (int e = 0; e < 4; ++e) pj += Y[j][4 * q + e] * wv[e];
                    }
                    dot += pj;
                }
                dot = pair_sum(dot);
                if (h == 0 && row_ok) {
                    float sc = dot + score_b[c];
                    if (sigmoid) sc = 1.0f / (1.0f + expf(-sc));
                    scores[(size_t)row * num_classes + c] = sc;
                }
            }
        }
    }
}
#endif  // VS_WITH_DIAG

// ------------------------------------------------------------------------------------------
// Skinny-M kernels (latency path, M <= a few thousand rows: single videos, small batches).
// The tiled kernels above put a whole K loop on each of a handful of blocks when M is small (fc2 + LN
// of one 320-frame video: 152 us on 3 blocks).  Here every WAVE owns one 32x32 output tile and streams
// both operands straight from L2 into registers (16 bytes per lane per 8 k; 4 groups in flight): no LDS
// staging and no barrier in the K loop, so the chip is filled with (M/32)*(N/32) independent waves and a
// stage costs K/2 MFMAs = 3.4 us (K=256) .. 13.7 us (K=1024).  At large M this form is L1-bandwidth-bound
// (2 KiB per 4 MFMAs per wave) and the LDS-tiled kernels take over.
//   skinny_gemm<EPI>: block = 4 waves = 32 rows x 128 columns.
//   skinny_ln<NW>:    block = NW waves = 32 rows x 32*NW = d_model columns; LayerNorm statistics and the
//                     score dot product are exchanged between the waves through LDS.
// Same operand convention as everywhere: weight fragment = MFMA A operand, so lane (r,h) ends up with
// row m0+r and columns n0 + 8q + 4h + e in acc[4q+e]; bias enters as a "bias x ones" MFMA.
// ------------------------------------------------------------------------------------------
// K loop of one wave: chunks of 32 k (4 groups of 8 = 16 MFMAs), FOUR register sets so that the loads of
// chunks c+1..c+3 are in flight while chunk c computes (~3000 MFMA cycles of cover for an L2 round trip);
// unrolled by the four sets so no register rotation is needed.  Requires K % 128 == 0.  The MFMA order
// (groups ascending, steps 0..3) is the same as in the LDS-tiled kernels, so both families produce
// bit-identical sums.
__device__ __forceinline__ void skinny_mainloop(f32x16 &acc, const float *__restrict__ ap,
                                                const float *__restrict__ wp, int K) {
    f32x4 a[4][4], w[4][4];
    auto load = [&](int set, int off) __attribute__((always_inline)) {      // off in floats from ap / wp
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            a[set][g] = *(const f32x4 *)(ap + off + 8 * g);
            w[set][g] = *(const f32x4 *)(wp + off + 8 * g);
        }
    };
    auto mma = [&](int set) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int st = 0; st < 4; ++st) acc = MFMA32(w[set][g][st], a[set][g][st], acc);
    };
    const int nc = K / 32;
    load(0, 0); load(1, 32); load(2, 64);
    for (int c = 0; c < nc - 4; c += 4) {
        load(3, 96);  mma(0);
        load(0, 128); mma(1);
        load(1, 160); mma(2);
        load(2, 192); mma(3);
        ap += 128; wp += 128;
    }
    load(3, 96);
    mma(0); mma(1); mma(2); mma(3);
}

template <int EPI>
__global__ __launch_bounds__(256) void skinny_gemm(
    const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
    float *__restrict__ C, int M, int N, int K, const float *__restrict__ pe, int T, int H, int dh, EpiArgs ea) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 128 + 32 * wave;
    if (n0 >= N) return;
    const int row = m0 + r;
    const int arow = row < M ? row : M - 1;
    const float *ap = A + (size_t)arow * K + 4 * h, *wp = W + (size_t)(n0 + r) * K + 4 * h;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 acc = MFMA32(h == 0 ? bias[n0 + r] : 0.f, 1.0f, zero);
    skinny_mainloop(acc, ap, wp, K);
    if (row >= M) return;
    int bb = 0, tt = 0;
    if (EPI == EPI_PE || EPI == EPI_QKV) { bb = row / T; tt = row - bb * T; }
    int which = 0, head = 0, e0 = 0;
    if (EPI == EPI_QKV) { const int d = H * dh; which = n0 / d; const int c = n0 - which * d; head = c / dh; e0 = c - head * dh; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int co = 8 * q + 4 * h;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[4 * q + e];
        if (EPI == EPI_RELU || EPI == EPI_RELU_DROP) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = relu1(v[e]);
        }
        if (EPI == EPI_PE) {
            const f32x4 pv = *(const f32x4 *)(pe + (size_t)tt * N + n0 + co);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += pv[e];
        }
        if (EPI == EPI_GATE) {
            const f32x4 gv = *(const f32x4 *)(pe + (size_t)row * N + n0 + co);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gv[e] > 0.f ? v[e] * ea.scale : 0.f;
        }
        if (EPI == EPI_RELU_DROP) {
            const DropSite dsite = drop_site(ea.seed, ea.site, ea.p);
            const unsigned rk = drop_rowkey(dsite, (unsigned)row);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = drop_keep(dsite, rk, (unsigned)(n0 + co + e)) ? v[e] * dsite.scale : 0.f;
        }
        if (EPI == EPI_QKV)
            *(f32x4 *)(C + (size_t)which * M * (H * dh) + (((size_t)bb * H + head) * T + tt) * dh + e0 + co) = v;
        else
            *(f32x4 *)(C + (size_t)row * N + n0 + co) = v;
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void skinny_ln(
    const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
    const float *__restrict__ res, const float *__restrict__ gamma, const float *__restrict__ beta,
    float *__restrict__ out, int M, int K,
    const float *__restrict__ score_w, const float *__restrict__ score_b, int num_classes,
    int sigmoid, float *__restrict__ scores) {
    constexpr int N = 32 * NW;
    __shared__ float red[NW * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = 32 * wave;
    const int row = m0 + r;
    const bool row_ok = row < M;
    const int arow = row_ok ? row : M - 1;
    const float *ap = A + (size_t)arow * K + 4 * h, *wp = W + (size_t)(n0 + r) * K + 4 * h;
    // Same arithmetic order as gemm_ln_rows, so a video scores bit-identically through either family:
    // accumulator starts at the residual, the K loop, then + bias; row statistics are summed per 32-column
    // block in-lane, the blocks in ascending order, and the lane^32 partner last.
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 rv = *(const f32x4 *)(res + (size_t)arow * N + n0 + 8 * q + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * q + e] = rv[e];
    }
    skinny_mainloop(acc, ap, wp, K);

    auto row_total = [&](float v) __attribute__((always_inline)) {      // v: this lane's sum over its 16 columns
        __syncthreads();                       // previous use of red[] is finished
        red[wave * 64 + lane] = v;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[w * 64 + lane];            // blocks ascending, own half
        return pair_sum(t);                                              // + lane^32 partner
    };
    float s1 = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *(const f32x4 *)(bias + n0 + 8 * q + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[4 * q + e] += bv[e]; s1 += acc[4 * q + e]; }
    }
    const float mean = row_total(s1) * (1.0f / N);
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) { const float c = acc[t] - mean; acc[t] = c; s2 += c * c; }
    const float rstd = 1.0f / sqrtf(row_total(s2) * (1.0f / N) + 1e-5f);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int co = n0 + 8 * q + 4 * h;
        const f32x4 gv = *(const f32x4 *)(gamma + co), bv = *(const f32x4 *)(beta + co);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) { y[e] = acc[4 * q + e] * rstd * gv[e] + bv[e]; acc[4 * q + e] = y[e]; }
        if (row_ok) *(f32x4 *)(out + (size_t)row * N + co) = y;
    }
    if (score_w != nullptr) {
        for (int c = 0; c < num_classes; ++c) {
            float dot = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 wv = *(const f32x4 *)(score_w + (size_t)c * N + n0 + 8 * q + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) dot += acc[4 * q + e] * wv[e];
            }
            const float tot = row_total(dot);
            if (wave == 0 && h == 0 && row_ok) {
                float sc = tot + score_b[c];
                if (sigmoid) sc = 1.0f / (1.0f + expf(-sc));
                scores[(size_t)row * num_classes + c] = sc;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Skinny-M kernels, packed form (used by vs_scorer_forward).  The gather above costs ~260 ns per wave
// load (32 rows x 16 B).  Here (a) the weights are pre-packed once, at vs_weights_pack time, in
// FRAGMENT-MAJOR order  Wf[n/32][k/8][lane][4] = W[32*(n/32) + (lane&31)][8*(k/8) + 4*(lane>>5) + e],
// so every wave load is one contiguous 1 KiB; (b) the block's 32 activation rows are staged once per
// 1024-wide K phase through LDS with coalesced loads and read back as fragments (ds_read_b128).
// MFMA order and reductions are unchanged -> results stay bit-identical to the other kernels.
// ------------------------------------------------------------------------------------------
__global__ void pack_fragments(const float *__restrict__ W, float *__restrict__ Wf, int N, int K) {
    const size_t total = (size_t)N * K / 4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const size_t gi = idx >> 6;
        const int g = (int)(gi % (K / 8)), nb = (int)(gi / (K / 8));
        const int row = 32 * nb + (lane & 31), col = 8 * g + 4 * (lane >> 5);
        *(f32x4 *)(Wf + idx * 4) = *(const f32x4 *)(W + (size_t)row * K + col);
    }
}

// K loop of one wave over one K phase [k0, k0 + kp): activation fragments from LDS (As, row stride lda),
// weight fragments from the packed array; 4 register sets of one 32-k chunk each, as skinny_mainloop.
__device__ __forceinline__ void skinny2_phase(f32x16 &acc, const float *__restrict__ As_row,
                                              const float *__restrict__ wf, int kp) {
    // wf points at this lane's 4 floats of the phase's first group; consecutive groups are 256 floats apart
    f32x4 w[4][4];
    auto load = [&](int set, int chunk) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) w[set][g] = *(const f32x4 *)(wf + (size_t)(4 * chunk + g) * 256);
    };
    auto mma = [&](int set, int chunk) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 a = *(const f32x4 *)(As_row + 32 * chunk + 8 * g);
#pragma unroll
            for (int st = 0; st < 4; ++st) acc = MFMA32(w[set][g][st], a[st], acc);
        }
    };
    const int nc = kp / 32;                         // multiple of 4
    load(0, 0); load(1, 1); load(2, 2);
    int c = 0;
    for (; c < nc - 4; c += 4) {
        load(3, c + 3); mma(0, c);
        load(0, c + 4); mma(1, c + 1);
        load(1, c + 5); mma(2, c + 2);
        load(2, c + 6); mma(3, c + 3);
    }
    load(3, c + 3);
    mma(0, c); mma(1, c + 1); mma(2, c + 2); mma(3, c + 3);
}

// stage rows [m0, m0+32) x [k0, k0+kp) of A into LDS (row stride kp + 4), coalesced
template <int NT>
__device__ __forceinline__ void skinny2_stage(float *As, const float *__restrict__ A, int M, int K, int m0, int k0, int kp) {
    const int f4row = kp / 4, total = 32 * f4row;
    for (int idx = threadIdx.x; idx < total; idx += NT) {
        const int row = idx / f4row, c4 = idx - row * f4row;
        int ar = m0 + row; ar = ar < M ? ar : M - 1;
        *(f32x4 *)&As[row * (kp + 4) + 4 * c4] = *(const f32x4 *)(A + (size_t)ar * K + k0 + 4 * c4);
    }
}

// ---- the same latency kernels with the fp32 product EMULATED on the f16 pipe (PREC 2, "fp16x3") ----
// Packed weights: Wh[n/32][k/16][lane][hi 8 x f16 | lo 8 x f16] = split_f16(2^10 * W[32*(n/32) + (lane&31)]
// [16*(k/16) + 8*(lane>>5) + j]): 32 B per lane and k-group, the same bytes as the fp32 fragment copy.
// Activations are split once while they are staged into LDS: a row holds its kp hi halves, then its kp lo halves
// (4*kp + 16 B: the fp32 row stride, same conflict-free reads).  Product order per 16 k: lo*hi, hi*lo, hi*hi -
// the order of the tiled PREC 2 kernels, so a video's scores do not depend on the batch it is scored in.
__global__ void pack_fragments_f16x3(const float *__restrict__ W, unsigned *__restrict__ Wh, int N, int K) {
    const size_t total = (size_t)N * K / 8;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const size_t gi = idx >> 6;
        const int g = (int)(gi % (K / 16)), nb = (int)(gi / (K / 16));
        const int row = 32 * nb + (lane & 31), col = 16 * g + 8 * (lane >> 5);
        const f32x4 v0 = *(const f32x4 *)(W + (size_t)row * K + col) * F16X3_WS;
        const f32x4 v1 = *(const f32x4 *)(W + (size_t)row * K + col + 4) * F16X3_WS;
        u32x2 h0, l0, h1, l1;
        split_f16x4((const float *)&v0, h0, l0);
        split_f16x4((const float *)&v1, h1, l1);
        u32x4 hi, lo;
        hi[0] = h0[0]; hi[1] = h0[1]; hi[2] = h1[0]; hi[3] = h1[1];
        lo[0] = l0[0]; lo[1] = l0[1]; lo[2] = l1[0]; lo[3] = l1[1];
        *(u32x4 *)(Wh + idx * 8) = hi;
        *(u32x4 *)(Wh + idx * 8 + 4) = lo;
    }
}

// K loop of one wave over one K phase of kp: as_row = this lane's LDS row + 16*h bytes (hi half; lo half 2*kp
// bytes further), wh = this lane's 8 dwords of the phase's first 16-k group (groups are 512 dwords apart).
// 4 register sets of one 32-k chunk (2 groups) each, as skinny2_phase.
__device__ __forceinline__ void skinny3_phase(f32x16 &acc, const unsigned char *__restrict__ as_row,
                                              const unsigned *__restrict__ wh, int kp) {
    u32x4 w[4][2][2];
    auto load = [&](int set, int chunk) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            w[set][g][0] = *(const u32x4 *)(wh + (size_t)(2 * chunk + g) * 512);
            w[set][g][1] = *(const u32x4 *)(wh + (size_t)(2 * chunk + g) * 512 + 4);
        }
    };
    auto mma = [&](int set, int chunk) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const f16x8 ah = __builtin_bit_cast(f16x8, *(const u32x4 *)(as_row + 64 * chunk + 32 * g));
            const f16x8 al = __builtin_bit_cast(f16x8, *(const u32x4 *)(as_row + 2 * kp + 64 * chunk + 32 * g));
            const f16x8 wh8 = __builtin_bit_cast(f16x8, w[set][g][0]), wl8 = __builtin_bit_cast(f16x8, w[set][g][1]);
            acc = MFMA_F16(wl8, ah, acc);
            acc = MFMA_F16(wh8, al, acc);
            acc = MFMA_F16(wh8, ah, acc);
        }
    };
    const int nc = kp / 32;                         // multiple of 4
    load(0, 0); load(1, 1); load(2, 2);
    int c = 0;
    for (; c < nc - 4; c += 4) {
        load(3, c + 3); mma(0, c);
        load(0, c + 4); mma(1, c + 1);
        load(1, c + 5); mma(2, c + 2);
        load(2, c + 6); mma(3, c + 3);
    }
    load(3, c + 3);
    mma(0, c); mma(1, c + 1); mma(2, c + 2); mma(3, c + 3);
}

// stage rows [m0, m0+32) x [k0, k0+kp) of A into LDS as f16 hi | lo half-rows (row stride 4*kp + 16 bytes)
template <int NT>
__device__ __forceinline__ void skinny3_stage(unsigned char *As, const float *__restrict__ A, int M, int K, int m0, int k0, int kp) {
    const int f4row = kp / 4, total = 32 * f4row;
    for (int idx = threadIdx.x; idx < total; idx += NT) {
        const int row = idx / f4row, c4 = idx - row * f4row;
        int ar = m0 + row; ar = ar < M ? ar : M - 1;
        const f32x4 v = *(const f32x4 *)(A + (size_t)ar * K + k0 + 4 * c4);
        u32x2 hi, lo;
        split_f16x4((const float *)&v, hi, lo);
        unsigned char *rowp = As + (size_t)row * (4 * kp + 16);
        *(u32x2 *)(rowp + 8 * c4) = hi;
        *(u32x2 *)(rowp + 2 * kp + 8 * c4) = lo;
    }
}

template <int EPI, int PREC = 0>       // PREC 2: Wf is the pack_fragments_f16x3 copy
__global__ __launch_bounds__(256) void skinny2_gemm(
    const float *__restrict__ A, const float *__restrict__ Wf, const float *__restrict__ bias,
    float *__restrict__ C, int M, int N, int K, const float *__restrict__ pe, int T, int H, int dh, EpiArgs ea) {
    extern __shared__ __attribute__((aligned(16))) float As[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 128 + 32 * wave;
    const bool live = n0 < N;
    const int row = m0 + r;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 acc = MFMA32((h == 0 && live) ? bias[n0 + r] : 0.f, PREC == 2 ? F16X3_WS : 1.0f, zero);
    const int kpmax = K < 1024 ? K : 1024;
    for (int k0 = 0; k0 < K; k0 += kpmax) {
        const int kp = K - k0 < kpmax ? K - k0 : kpmax;
        if (k0) __syncthreads();
        if constexpr (PREC == 2) skinny3_stage<256>((unsigned char *)As, A, M, K, m0, k0, kp);
        else skinny2_stage<256>(As, A, M, K, m0, k0, kp);
        __syncthreads();
        if (live) {
            if constexpr (PREC == 2)
                skinny3_phase(acc, (const unsigned char *)As + (size_t)r * (4 * kp + 16) + 16 * h,
                              (const unsigned *)Wf + ((size_t)(n0 / 32) * (K / 16) + k0 / 16) * 512 + lane * 8, kp);
            else
                skinny2_phase(acc, As + r * (kp + 4) + 4 * h, Wf + ((size_t)(n0 / 32) * (K / 8) + k0 / 8) * 256 + lane * 4, kp);
        }
    }
    if (!live || row >= M) return;
    int bb = 0, tt = 0;
    if (EPI == EPI_PE || EPI == EPI_QKV) { bb = row / T; tt = row - bb * T; }
    int which = 0, head = 0, e0 = 0;
    if (EPI == EPI_QKV) { const int d = H * dh; which = n0 / d; const int c = n0 - which * d; head = c / dh; e0 = c - head * dh; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int co = 8 * q + 4 * h;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[4 * q + e];
        if constexpr (PREC == 2) v *= 1.0f / F16X3_WS;
        if (EPI == EPI_RELU || EPI == EPI_RELU_DROP) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = relu1(v[e]);
        }
        if (EPI == EPI_PE) {
            const f32x4 pv = *(const f32x4 *)(pe + (size_t)tt * N + n0 + co);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += pv[e];
        }
        if (EPI == EPI_GATE) {
            const f32x4 gv = *(const f32x4 *)(pe + (size_t)row * N + n0 + co);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gv[e] > 0.f ? v[e] * ea.scale : 0.f;
        }
        if (EPI == EPI_RELU_DROP) {
            const DropSite dsite = drop_site(ea.seed, ea.site, ea.p);
            const unsigned rk = drop_rowkey(dsite, (unsigned)row);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = drop_keep(dsite, rk, (unsigned)(n0 + co + e)) ? v[e] * dsite.scale : 0.f;
        }
        if (EPI == EPI_QKV)
            *(f32x4 *)(C + (size_t)which * M * (H * dh) + (((size_t)bb * H + head) * T + tt) * dh + e0 + co) = v;
        else
            *(f32x4 *)(C + (size_t)row * N + n0 + co) = v;
    }
}

// ---- latency mode (VS_FLAG_SPLITK, round 4): one reference-sized video (T = 320 rows) leaves the chip idle - the K = 1024
// products (embedding, fc2) run on 10-20 blocks, each walking the whole K on one CU.  Split K over blockIdx.z: every block
// multiplies a fixed K-slice and writes its partial [M, N] tile; the consumer (row LayerNorm pass / sum_parts_pe) adds the
// partials in slice order, then bias, then the residual - a FIXED order, so results are deterministic and do not depend on
// the batch, but they are not the bits of the unsplit kernels (different summation tree: within ~1e-6, tests pin 1e-4 of the
// goldens).  Opt-in; bit-identity between batched and single-video scoring stays the default.
template <int PREC = 0>       // PREC 2: the fp32 product emulated on the f16 pipe (Wf = the pack_fragments_f16x3 copy), as skinny2_gemm<EPI, 2>
__global__ __launch_bounds__(256) void skinny2_gemm_parts(
    const float *__restrict__ A, const float *__restrict__ Wf, float *__restrict__ parts, int M, int N, int K, int kslice) {
    extern __shared__ __attribute__((aligned(16))) float As[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 128 + 32 * wave;
    const int k0 = blockIdx.z * kslice;
    const bool live = n0 < N;
    const int row = m0 + r;
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // the slice in phases of at most 256 k (33 KiB of LDS whatever the slice: fc2 of d_model 512 runs four slices of 512)
    const int kp = kslice < 256 ? kslice : 256;
    for (int kk = 0; kk < kslice; kk += kp) {
        if (kk) __syncthreads();
        if constexpr (PREC == 2) skinny3_stage<256>((unsigned char *)As, A, M, K, m0, k0 + kk, kp);
        else skinny2_stage<256>(As, A, M, K, m0, k0 + kk, kp);
        __syncthreads();
        if (live) {
            if constexpr (PREC == 2)
                skinny3_phase(acc, (const unsigned char *)As + (size_t)r * (4 * kp + 16) + 16 * h,
                              (const unsigned *)Wf + ((size_t)(n0 / 32) * (K / 16) + (k0 + kk) / 16) * 512 + lane * 8, kp);
            else
                skinny2_phase(acc, As + r * (kp + 4) + 4 * h, Wf + ((size_t)(n0 / 32) * (K / 8) + (k0 + kk) / 8) * 256 + lane * 4, kp);
        }
    }
    if (!live || row >= M) return;
    float *C = parts + (size_t)blockIdx.z * M * N;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[4 * q + e];
        if constexpr (PREC == 2) v *= 1.0f / F16X3_WS;      // (the packed weights carry 2^10: see F16X3_WS)
        *(f32x4 *)(C + (size_t)row * N + n0 + 8 * q + 4 * h) = v;
    }
}

// out[row, :] = ((parts[0] + parts[1]) + ... ) + bias + pe[row % T]   (the embedding's epilogue in latency mode)
__global__ __launch_bounds__(256) void sum_parts_pe(const float *__restrict__ parts, int nsplit, const float *__restrict__ bias,
                                                    const float *__restrict__ pe, int T, float *__restrict__ out, int M, int N) {
    const int n4 = N / 4;
    const size_t total = (size_t)M * n4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int row = (int)(i / n4), c = (int)(i % n4) * 4;
        f32x4 v = *(const f32x4 *)(parts + (size_t)row * N + c);
        for (int s = 1; s < nsplit; ++s) v += *(const f32x4 *)(parts + ((size_t)s * M + row) * N + c);
        v += *(const f32x4 *)(bias + c);
        if (pe != nullptr) v += *(const f32x4 *)(pe + (size_t)(row % T) * N + c);
        *(f32x4 *)(out + (size_t)row * N + c) = v;
    }
}

template <int NW, int PREC = 0>
__global__ __launch_bounds__(64 * NW) void skinny2_ln(
    const float *__restrict__ A, const float *__restrict__ Wf, const float *__restrict__ bias,
    const float *__restrict__ res, const float *__restrict__ gamma, const float *__restrict__ beta,
    float *__restrict__ out, int M, int K,
    const float *__restrict__ score_w, const float *__restrict__ score_b, int num_classes,
    int sigmoid, float *__restrict__ scores) {
    constexpr int N = 32 * NW;
    extern __shared__ __attribute__((aligned(16))) float As[];
    __shared__ float red[NW * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = 32 * wave;
    const int row = m0 + r;
    const bool row_ok = row < M;
    const int arow = row_ok ? row : M - 1;
    constexpr float SC = PREC == 2 ? F16X3_WS : 1.0f;      // scale of the accumulators (see F16X3_WS, gemm_ln_rows)
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 rv = *(const f32x4 *)(res + (size_t)arow * N + n0 + 8 * q + 4 * h);
        if constexpr (PREC == 2) rv *= SC;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * q + e] = rv[e];
    }
    const int kpmax = K < 1024 ? K : 1024;
    for (int k0 = 0; k0 < K; k0 += kpmax) {
        const int kp = K - k0 < kpmax ? K - k0 : kpmax;
        if (k0) __syncthreads();
        if constexpr (PREC == 2) skinny3_stage<64 * NW>((unsigned char *)As, A, M, K, m0, k0, kp);
        else skinny2_stage<64 * NW>(As, A, M, K, m0, k0, kp);
        __syncthreads();
        if constexpr (PREC == 2)
            skinny3_phase(acc, (const unsigned char *)As + (size_t)r * (4 * kp + 16) + 16 * h,
                          (const unsigned *)Wf + ((size_t)wave * (K / 16) + k0 / 16) * 512 + lane * 8, kp);
        else
            skinny2_phase(acc, As + r * (kp + 4) + 4 * h, Wf + ((size_t)wave * (K / 8) + k0 / 8) * 256 + lane * 4, kp);
    }
    auto row_total = [&](float v) __attribute__((always_inline)) {      // same tree as skinny_ln / gemm_ln_rows
        __syncthreads();
        red[wave * 64 + lane] = v;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[w * 64 + lane];
        return pair_sum(t);
    };
    float s1 = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 bv = *(const f32x4 *)(bias + n0 + 8 * q + 4 * h);
        if constexpr (PREC == 2) bv *= SC;
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[4 * q + e] += bv[e]; s1 += acc[4 * q + e]; }
    }
    const float mean = row_total(s1) * (1.0f / N);
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) { const float c = acc[t] - mean; acc[t] = c; s2 += c * c; }
    const float rstd = 1.0f / sqrtf(row_total(s2) * (1.0f / N) + 1e-5f * (SC * SC));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int co = n0 + 8 * q + 4 * h;
        const f32x4 gv = *(const f32x4 *)(gamma + co), bv = *(const f32x4 *)(beta + co);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) { y[e] = acc[4 * q + e] * rstd * gv[e] + bv[e]; acc[4 * q + e] = y[e]; }
        if (row_ok) *(f32x4 *)(out + (size_t)row * N + co) = y;
    }
    if (score_w != nullptr) {
        for (int c = 0; c < num_classes; ++c) {
            float dot = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 wv = *(const f32x4 *)(score_w + (size_t)c * N + n0 + 8 * q + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) dot += acc[4 * q + e] * wv[e];
            }
            const float tot = row_total(dot);
            if (wave == 0 && h == 0 && row_ok) {
                float sc = tot + score_b[c];
                if (sigmoid) sc = 1.0f / (1.0f + expf(-sc));
                scores[(size_t)row * num_classes + c] = sc;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// out = LayerNorm(a + residual) * gamma + beta (+ score head, + sigmoid), one wave per row  (d_model > 256).
// For wide models the fused projection + LayerNorm kernel (gemm_res_ln: 4 waves share 64 rows, statistics through
// LDS) runs at 0.58-0.67 of the matrix peak while the plain persistent GEMM reaches 0.88 on the same product
// (M-B, d 512: fc2 1.31 ms fused vs 1.00 ms + this 0.09 ms pass).  So d_model > 256 takes the plain GEMM (bias in
// its accumulator init) and this HBM-bound row pass: two-pass variance, lane-local float4s, wave shuffles only.
// Per-row arithmetic does not depend on M: a video's scores stay bit-identical whatever batch it is scored in.
// ------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void rows_res_ln(const float *__restrict__ a, const float *__restrict__ res,
                                                   const float *__restrict__ gamma, const float *__restrict__ beta,
                                                   float *__restrict__ out, int M, int d,
                                                   const float *__restrict__ score_w, const float *__restrict__ score_b,
                                                   int num_classes, int sigmoid, float *__restrict__ scores,
                                                   unsigned short *__restrict__ out16, int dn,
                                                   int nsplit, const float *__restrict__ pbias) {
    // dn (== d for every natively shaped model): the LayerNorm width.  A model EMBEDDED in a wider supported shape (round 4:
    // zero-padded weights, vs_weights_set_norm_width) has its true d_model = dn < d; columns dn .. d-1 of the residual stream
    // are identically zero (zero weight rows / bias / gamma / beta) and stay out of the statistics.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto wsum = [](float v) __attribute__((always_inline)) { v += __shfl_xor(v, 32); return half_sum(v); };
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        f32x4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = 4 * lane + 256 * u;
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < d) {
                f32x4 av = *(const f32x4 *)(a + (size_t)row * d + c);
                if (nsplit > 0) {       // latency mode: `a` = nsplit K-slice partials [nsplit][M][d], added in slice order, then the bias
                    for (int sp = 1; sp < nsplit; ++sp) av += *(const f32x4 *)(a + ((size_t)sp * M + row) * d + c);
                    av += *(const f32x4 *)(pbias + c);
                }
                v[u] = av + *(const f32x4 *)(res + (size_t)row * d + c);
                s += v[u][0] + v[u][1] + v[u][2] + v[u][3];
            }
        }
        const float mean = wsum(s) / (float)dn;
        float s2 = 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            if (4 * lane + 256 * u < dn) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float t = v[u][e] - mean; s2 += t * t; }
            }
        }
        const float rstd = 1.0f / sqrtf(wsum(s2) / (float)dn + 1e-5f);
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = 4 * lane + 256 * u;
            if (c < d) {
                const f32x4 g = *(const f32x4 *)(gamma + c), b = *(const f32x4 *)(beta + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[u][e] = (v[u][e] - mean) * rstd * g[e] + b[e];
                *(f32x4 *)(out + (size_t)row * d + c) = v[u];
                if (out16 != nullptr) {       // the bf16 copy the next bf16-operand GEMM reads (vs_gemm_ring.hip)
                    u32x2 u2; u2[0] = pack_bf16(v[u][0], v[u][1]); u2[1] = pack_bf16(v[u][2], v[u][3]);
                    *(u32x2 *)(out16 + (size_t)row * d + c) = u2;
                }
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
                dot = wsum(dot);
                if (lane == 0) {
                    float sc = dot + score_b[cls];
                    if (sigmoid) sc = 1.0f / (1.0f + expf(-sc));
                    scores[(size_t)row * num_classes + cls] = sc;
                }
            }
        }
    }
}

// ---- packed ragged batches (vs_scorer_forward_packed) ----
// one thread: B is a few hundred at most, and the launch is stream-ordered before its consumers
// work_cap: pairs `work` can hold; the host sized the launch from ITS copy of the lengths, so entries beyond the
// capacity (device lengths that disagree with the host's) are dropped instead of written past the workspace
__global__ void plan_packed(const int *__restrict__ lengths, int B, int qb, int *__restrict__ cu, int *__restrict__ work,
                            int work_cap) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int row = 0, n = 0;
    for (int b = 0; b < B; ++b) {
        cu[b] = row;
        const int t = lengths[b];
        for (int q = 0; q < (t + qb - 1) / qb; ++q) {
            if (n < work_cap) { work[2 * n] = b; work[2 * n + 1] = q; }
            ++n;
        }
        row += t;
    }
    cu[B] = row;
}

// rows[cu[b] + t, :] = pe[t, :]  (grid: 64-frame chunks x videos)
__global__ __launch_bounds__(256) void gather_rows(const float *__restrict__ pe, const int *__restrict__ cu, int d,
                                                    float *__restrict__ rows) {
    const int b = blockIdx.y, c0 = cu[b], len = cu[b + 1] - c0, f4 = d / 4;
    for (int idx = threadIdx.x; idx < 64 * f4; idx += 256) {
        const int t = blockIdx.x * 64 + idx / f4, c = (idx % f4) * 4;
        if (t < len) *(f32x4 *)(rows + (size_t)(c0 + t) * d + c) = *(const f32x4 *)(pe + (size_t)t * d + c);
    }
}

// segment blockIdx.y of a batch of plain float copies; 16-byte accesses where both ends and the length allow
__global__ __launch_bounds__(256) void copy_segments(const VskCopySegs s) {
    const int g = blockIdx.y;
    const float *src = s.src[g];
    float *dst = s.dst[g];
    const unsigned n = s.n[g];
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0 && (n & 3) == 0) {
        for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n / 4; i += gridDim.x * 256)
            ((f32x4 *)dst)[i] = ((const f32x4 *)src)[i];
    } else {
        for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dst[i] = src[i];
    }
}

// use_cls=True (reference simnet.py:214-216, 47-51): h[b, 0, :] = class token, h[b, 1 + t, :] = e[b, t, :], and the key
// mask gets a leading "not padding" entry per video.  grid = (ceil((T + 1) / 64), B)
__global__ __launch_bounds__(256) void insert_cls(const float *__restrict__ e, const float *__restrict__ cls,
                                                   const uint8_t *__restrict__ mask, float *__restrict__ h,
                                                   uint8_t *__restrict__ mask1, int T, int d) {
    const int b = blockIdx.y, f4 = d / 4, T1 = T + 1;
    for (int idx = threadIdx.x; idx < 64 * f4; idx += 256) {
        const int t = blockIdx.x * 64 + idx / f4, c = (idx % f4) * 4;
        if (t < T1)
            *(f32x4 *)(h + ((size_t)b * T1 + t) * d + c) =
                t == 0 ? *(const f32x4 *)(cls + c) : *(const f32x4 *)(e + ((size_t)b * T + t - 1) * d + c);
    }
    if (mask1 != nullptr && threadIdx.x < 64) {
        const int t = blockIdx.x * 64 + threadIdx.x;
        if (t < T1) mask1[(size_t)b * T1 + t] = t == 0 ? 0 : mask[(size_t)b * T + t - 1];
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host-side launchers (plain C++ interface used by vs_scorer.cpp)
// ------------------------------------------------------------------------------------------

// persistent grid: `per_cu` blocks per CU, a multiple of 8 so the XCD chunking is exact
static int persistent_blocks(int ntiles, int per_cu = 2) {
    const int cus = vsk_device_cus();
    if (cus <= 0) return -1;
    int g = per_cu * cus;
    g -= g % 8;
    if (g < 8) g = 8;
    const int need = (ntiles + 7) / 8 * 8;
    return need < g ? need : g;
}

// 256x128 tiles on 8-wave blocks when there is enough work to give every CU >= 2 such tiles and the
// ragged M edge does not waste more than 128-row tiles would; else 128x128 tiles on 4-wave blocks
static bool use_wide_tiles(int M, int N) {
    if (vsk_options().gemm_nwm2) return false;
    const long r256 = (M + 255) / 256 * 256, r128 = (M + 127) / 128 * 128;
    const long tiles = (r256 / 256) * ((N + 127) / 128);
    return r256 * 100 <= r128 * 105 && tiles >= 512;
}

// latency path: below this many rows the LDS-tiled kernels cannot fill the chip
// measured hand-over (tools/sweep_skinny.py, T=1024, M-A): skinny wins through M = 16384, tiled from 32768
int vsk_plan_packed(const int *lengths_dev, int B, int qb, int *cu, int *work, int work_cap, hipStream_t st) {
    hipLaunchKernelGGL(plan_packed, dim3(1), dim3(64), 0, st, lengths_dev, B, qb, cu, work, work_cap);
    VSK_CHECK_LAUNCH();
    return 0;
}

int vsk_gather_rows(const float *pe, const int *cu, int B, int tmax, int d, float *rows, hipStream_t st) {
    hipLaunchKernelGGL(gather_rows, dim3((tmax + 63) / 64, B), dim3(256), 0, st, pe, cu, d, rows);
    VSK_CHECK_LAUNCH();
    return 0;
}

int vsk_copy_segments(const VskCopySegs &segs, hipStream_t st) {
    if (segs.count <= 0 || segs.count > VSK_COPY_MAX_SEGS) return -1;
    hipLaunchKernelGGL(copy_segments, dim3(64, segs.count), dim3(256), 0, st, segs);
    VSK_CHECK_LAUNCH();
    return 0;
}

int vsk_insert_cls(const float *e, const float *cls, const uint8_t *mask, float *h, uint8_t *mask1, int B, int T, int d,
                   hipStream_t st) {
    if (d % 4) return -1;
    hipLaunchKernelGGL(insert_cls, dim3((T + 1 + 63) / 64, B), dim3(256), 0, st, e, cls, mask, h, mask1, T, d);
    VSK_CHECK_LAUNCH();
    return 0;
}

int vsk_rows_res_ln(const float *a, const float *res, const float *gamma, const float *beta, float *out, int M, int d,
                    const float *score_w, const float *score_b, int num_classes, int sigmoid, float *scores,
                    hipStream_t st, void *out16, int dn, int nsplit, const float *pbias) {
    if (d % 4 || d > 1024) return -1;
    if (nsplit > 0 && pbias == nullptr) return -1;
    if (dn <= 0) dn = d;
    if (dn % 4 || dn > d) return -1;
    const int rows4 = (M + 3) / 4;
    const dim3 grid(rows4 < 8192 ? (rows4 < 1 ? 1 : rows4) : 8192);
    switch ((d + 255) / 256) {          // float4 per lane (256 columns each)
        case 1: hipLaunchKernelGGL(rows_res_ln<1>, grid, dim3(256), 0, st, a, res, gamma, beta, out, M, d, score_w, score_b, num_classes, sigmoid, scores, (unsigned short *)out16, dn, nsplit, pbias); break;
        case 2: hipLaunchKernelGGL(rows_res_ln<2>, grid, dim3(256), 0, st, a, res, gamma, beta, out, M, d, score_w, score_b, num_classes, sigmoid, scores, (unsigned short *)out16, dn, nsplit, pbias); break;
        case 3: hipLaunchKernelGGL(rows_res_ln<3>, grid, dim3(256), 0, st, a, res, gamma, beta, out, M, d, score_w, score_b, num_classes, sigmoid, scores, (unsigned short *)out16, dn, nsplit, pbias); break;
        default: hipLaunchKernelGGL(rows_res_ln<4>, grid, dim3(256), 0, st, a, res, gamma, beta, out, M, d, score_w, score_b, num_classes, sigmoid, scores, (unsigned short *)out16, dn, nsplit, pbias); break;
    }
    VSK_CHECK_LAUNCH();
    return 0;
}

int vsk_skinny_max_rows() { return vsk_options().skinny_rows; }
static int skinny_max_rows() { return vsk_skinny_max_rows(); }

// dynamic LDS of the packed skinny kernels: 32 activation rows x (min(K,1024) + 4) floats (up to 128.5 KiB)
static size_t skinny2_lds(int K) { return (size_t)32 * ((K < 1024 ? K : 1024) + 4) * sizeof(float); }

// latency mode: K-slice partial products of A [M, K] x W^T (Wf: fragment-major copy) -> parts [nsplit][M][N], no bias
int vsk_linear_parts(const float *A, const float *Wf, float *parts, int M, int N, int K, int nsplit, hipStream_t st, int f16x3) {
    if (Wf == nullptr || nsplit < 1 || K % nsplit) return -1;
    const int kslice = K / nsplit;
    if (kslice % 128 || (kslice > 256 && kslice % 256) || N % 32) return -1;      // phases of <= 256 k: 33 KiB of LDS, no opt-in attribute
    dim3 grid((M + 31) / 32, (N + 127) / 128, nsplit);
    if (f16x3) hipLaunchKernelGGL(skinny2_gemm_parts<2>, grid, dim3(256), skinny2_lds(kslice < 256 ? kslice : 256), st, A, Wf, parts, M, N, K, kslice);
    else hipLaunchKernelGGL(skinny2_gemm_parts<0>, grid, dim3(256), skinny2_lds(kslice < 256 ? kslice : 256), st, A, Wf, parts, M, N, K, kslice);
    VSK_CHECK_LAUNCH();
    return 0;
}

int vsk_sum_parts_pe(const float *parts, int nsplit, const float *bias, const float *pe, int T, float *out, int M, int N, hipStream_t st) {
    if (N % 4 || nsplit < 1) return -1;
    const size_t total = (size_t)M * (N / 4);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(sum_parts_pe, dim3(blocks), dim3(256), 0, st, parts, nsplit, bias, pe, T, out, M, N);
    VSK_CHECK_LAUNCH();
    return 0;
}
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: set it once per (kernel, device).  `done` is
// the call site's own per-device flag array (one per kernel instantiation).
enum { VSK_MAX_DEVICES = 64 };
static int allow_big_lds_on_device(const void *kernel, std::atomic<unsigned char> *done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipErrorInvalidDevice;
    if (dev >= 0 && dev < VSK_MAX_DEVICES && done[dev].load(std::memory_order_acquire)) return 0;
    const int rc = (int)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 1028 * 4);
    if (rc == 0 && dev >= 0 && dev < VSK_MAX_DEVICES) done[dev].store(1, std::memory_order_release);
    return rc;
}
#define VSK_ALLOW_BIG_LDS(kernel_)                                          \
    ([]() -> int {                                                          \
        static std::atomic<unsigned char> done_[VSK_MAX_DEVICES];           \
        return allow_big_lds_on_device((const void *)(kernel_), done_);     \
    }())

// LPP: the PREC of the 16-bit-operand branch - 1 bf16, 3 f16 (the training path's fp16 mode; bf16 flag | VSK_F16)
template <int EPI, int LPP = 1>
static int launch_gemm(const float *A, const float *W, const float *Wf, const float *bias, float *C, int M, int N, int K,
                       const float *pe, int T, int H, int dh, int bf16, hipStream_t st,
                       EpiArgs ea = EpiArgs{0ull, 0u, 0.f, 0.f}) {
    // bf16 form: 64-wide k-tiles from K = 512 up (measured: +6 % on M-B's K = 512 / 2048 products; at K = 256 - four
    // k-tiles per output tile - the 35 registers it spills cost more than the halved barriers give: -4 %)
    const bool kw64 = K % 64 == 0 && K >= 512;
    if (bf16 == 2 && Wf != nullptr && M <= skinny_max_rows() && N % 32 == 0 && K % 128 == 0) {
        // fp16x3 latency kernels (Wf is then the pack_fragments_f16x3 copy)
        if (const int attr_rc = VSK_ALLOW_BIG_LDS((skinny2_gemm<EPI, 2>))) return attr_rc;
        dim3 grid((M + 31) / 32, (N + 127) / 128);
        hipLaunchKernelGGL((skinny2_gemm<EPI, 2>), grid, dim3(256), skinny2_lds(K), st, A, Wf, bias, C, M, N, K, pe, T, H, dh, ea);
        VSK_CHECK_LAUNCH();
        return 0;
    }
    if (bf16 == 2) {     // fp32 emulated on the f16 pipe (opt-in): the LDS-tiled kernels
        if (N % 256 == 0 && M > 128) {
            const int blocks = persistent_blocks(((M + 255) / 256) * (N / 256), 1);
            if (blocks < 0) return (int)hipErrorInvalidDevice;
            hipLaunchKernelGGL((gemm_nt_128<EPI, 4, 0, 4, 2>), dim3(blocks), dim3(512), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr, ea);
        } else {
            const int blocks = persistent_blocks(((M + 127) / 128) * ((N + 127) / 128), 2);
            if (blocks < 0) return (int)hipErrorInvalidDevice;
            hipLaunchKernelGGL((gemm_nt_128<EPI, 2, 0, 2, 2>), dim3(blocks), dim3(256), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr, ea);
        }
        VSK_CHECK_LAUNCH();
        return 0;
    }
    if constexpr (EPI == EPI_RELU_DROP || EPI == EPI_GATE || EPI == EPI_BIAS) {
        if (bf16 == (1 | VSK_STORE16)) {      // training path, bf16 storage of the MLP hidden tensor / of its gradient / of dO
            const bool big = N % 256 == 0 && M > 128;
            const int blocks = big ? persistent_blocks(((M + 255) / 256) * (N / 256), 1) : persistent_blocks(((M + 127) / 128) * ((N + 127) / 128), 2);
            if (blocks < 0) return (int)hipErrorInvalidDevice;
            if (big && kw64) hipLaunchKernelGGL((gemm_nt_128<EPI, 4, 0, 4, LPP, 1, 64>), dim3(blocks), dim3(512), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr, ea);
            else if (big) hipLaunchKernelGGL((gemm_nt_128<EPI, 4, 0, 4, LPP, 1>), dim3(blocks), dim3(512), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr, ea);
            else hipLaunchKernelGGL((gemm_nt_128<EPI, 2, 0, 2, LPP, 1>), dim3(blocks), dim3(256), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr, ea);
            VSK_CHECK_LAUNCH();
            return 0;
        }
    }
    if constexpr (EPI == EPI_BIAS || EPI == EPI_PE) {
        if (bf16 == (1 | VSK_A16)) {          // training path: A (the bf16-stored hidden tensor / its gradient) read as bf16
            const bool big = N % 256 == 0 && M > 128;
            const int blocks = big ? persistent_blocks(((M + 255) / 256) * (N / 256), 1) : persistent_blocks(((M + 127) / 128) * ((N + 127) / 128), 2);
            if (blocks < 0) return (int)hipErrorInvalidDevice;
            if (big && kw64) hipLaunchKernelGGL((gemm_nt_128<EPI, 4, 0, 4, LPP, 0, 64, 1>), dim3(blocks), dim3(512), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr, ea);
            else if (big) hipLaunchKernelGGL((gemm_nt_128<EPI, 4, 0, 4, LPP, 0, 32, 1>), dim3(blocks), dim3(512), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr, ea);
            else hipLaunchKernelGGL((gemm_nt_128<EPI, 2, 0, 2, LPP, 0, 32, 1>), dim3(blocks), dim3(256), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr, ea);
            VSK_CHECK_LAUNCH();
            return 0;
        }
    }
    if (bf16 & VSK_A16) return -1;           // only the two consumers above read a bf16-stored A
    if constexpr (EPI == EPI_RELU || EPI == EPI_QKV) {
        if (bf16 == (1 | VSK_STORE16)) {      // bf16 matrix pipe, C stored as bf16 (fc1, QKV)
            if (N % 256 == 0 && M > 128) {
                const int blocks = persistent_blocks(((M + 255) / 256) * (N / 256), 1);
                if (blocks < 0) return (int)hipErrorInvalidDevice;
                if (kw64) hipLaunchKernelGGL((gemm_nt_128<EPI, 4, 0, 4, LPP, 1, 64>), dim3(blocks), dim3(512), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr, ea);
                else hipLaunchKernelGGL((gemm_nt_128<EPI, 4, 0, 4, LPP, 1>), dim3(blocks), dim3(512), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr, ea);
            } else {
                const int blocks = persistent_blocks(((M + 127) / 128) * ((N + 127) / 128), 2);
                if (blocks < 0) return (int)hipErrorInvalidDevice;
                if (kw64) hipLaunchKernelGGL((gemm_nt_128<EPI, 2, 0, 2, LPP, 1, 64>), dim3(blocks), dim3(256), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr, ea);
                else hipLaunchKernelGGL((gemm_nt_128<EPI, 2, 0, 2, LPP, 1>), dim3(blocks), dim3(256), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr, ea);
            }
            VSK_CHECK_LAUNCH();
            return 0;
        }
    }
    if (bf16 & VSK_STORE16) return -1;       // only the two producers above have a bf16-output form
    if (bf16) {          // bf16 matrix pipe (opt-in): always the LDS-tiled kernels
        if (N % 256 == 0 && M > 128) {
            const int blocks = persistent_blocks(((M + 255) / 256) * (N / 256), 1);
            if (blocks < 0) return (int)hipErrorInvalidDevice;
            if (kw64) hipLaunchKernelGGL((gemm_nt_128<EPI, 4, 0, 4, LPP, 0, 64>), dim3(blocks), dim3(512), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr, ea);
            else hipLaunchKernelGGL((gemm_nt_128<EPI, 4, 0, 4, LPP>), dim3(blocks), dim3(512), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr, ea);
        } else {
            const int blocks = persistent_blocks(((M + 127) / 128) * ((N + 127) / 128), 2);
            if (blocks < 0) return (int)hipErrorInvalidDevice;
            if (kw64) hipLaunchKernelGGL((gemm_nt_128<EPI, 2, 0, 2, LPP, 0, 64>), dim3(blocks), dim3(256), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr, ea);
            else hipLaunchKernelGGL((gemm_nt_128<EPI, 2, 0, 2, LPP>), dim3(blocks), dim3(256), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr, ea);
        }
        VSK_CHECK_LAUNCH();
        return 0;
    }
    if (Wf != nullptr && M <= skinny_max_rows() && N % 32 == 0 && K % 128 == 0) {
        if (const int attr_rc = VSK_ALLOW_BIG_LDS((skinny2_gemm<EPI, 0>))) return attr_rc;
        dim3 grid((M + 31) / 32, (N + 127) / 128);
        hipLaunchKernelGGL((skinny2_gemm<EPI, 0>), grid, dim3(256), skinny2_lds(K), st, A, Wf, bias, C, M, N, K, pe, T, H, dh, ea);
        VSK_CHECK_LAUNCH();
        return 0;
    }
    if (M <= skinny_max_rows() && N % 32 == 0 && K % 128 == 0) {
        dim3 grid((M + 31) / 32, (N + 127) / 128);
        hipLaunchKernelGGL((skinny_gemm<EPI>), grid, dim3(256), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, ea);
        VSK_CHECK_LAUNCH();
        return 0;
    }
    if (use_wide_tiles(M, N) && N % 256 == 0 && !vsk_options().gemm_nj2) {
        const int blocks = persistent_blocks(((M + 255) / 256) * (N / 256), 1);
        if (blocks < 0) return (int)hipErrorInvalidDevice;
        hipLaunchKernelGGL((gemm_nt_128<EPI, 4, 0, 4, 0>), dim3(blocks), dim3(512), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr, ea);
    } else if (use_wide_tiles(M, N)) {
        const int blocks = persistent_blocks(((M + 255) / 256) * ((N + 127) / 128), 1);
        if (blocks < 0) return (int)hipErrorInvalidDevice;
        hipLaunchKernelGGL((gemm_nt_128<EPI, 4, 0, 2, 0>), dim3(blocks), dim3(512), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr, ea);
    } else {
        const int blocks = persistent_blocks(((M + 127) / 128) * ((N + 127) / 128), 2);
        if (blocks < 0) return (int)hipErrorInvalidDevice;
        hipLaunchKernelGGL((gemm_nt_128<EPI, 2, 0, 2, 0>), dim3(blocks), dim3(256), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr, ea);
    }
    VSK_CHECK_LAUNCH();
    return 0;
}

int vsk_linear(const float *A, const float *W, const float *Wf, const float *bias, float *C, int M, int N, int K,
               int relu, const float *pe, int T, int bf16, hipStream_t st) {
    if (bf16 & VSK_F16) {          // f16 operands (training path's fp16 mode): the same kernels, PREC 3
        bf16 &= ~VSK_F16;
        if (pe != nullptr) return launch_gemm<EPI_PE, 3>(A, W, Wf, bias, C, M, N, K, pe, T, 0, 0, bf16, st);
        if (relu) return launch_gemm<EPI_RELU, 3>(A, W, Wf, bias, C, M, N, K, nullptr, 1, 0, 0, bf16, st);
        return launch_gemm<EPI_BIAS, 3>(A, W, Wf, bias, C, M, N, K, nullptr, 1, 0, 0, bf16, st);
    }
    if (pe != nullptr) return launch_gemm<EPI_PE>(A, W, Wf, bias, C, M, N, K, pe, T, 0, 0, bf16, st);
    if (relu) return launch_gemm<EPI_RELU>(A, W, Wf, bias, C, M, N, K, nullptr, 1, 0, 0, bf16, st);
    return launch_gemm<EPI_BIAS>(A, W, Wf, bias, C, M, N, K, nullptr, 1, 0, 0, bf16, st);
}

// training path: C = (gate > 0 ? (A W^T + bias) * scale : 0), gate [M,N] - the ReLU / mlp.dropout backward in the
// epilogue of the fc2 dgrad GEMM (exact fp32 kernels only)
int vsk_linear_gate(const float *A, const float *W, const float *Wf, const float *bias, const float *gate, float scale,
                    float *C, int M, int N, int K, hipStream_t st, int bf16) {
    if (bf16 & VSK_F16) return launch_gemm<EPI_GATE, 3>(A, W, Wf, bias, C, M, N, K, gate, 1, 0, 0, bf16 & ~VSK_F16, st, EpiArgs{0ull, 0u, 0.f, scale});
    return launch_gemm<EPI_GATE>(A, W, Wf, bias, C, M, N, K, gate, 1, 0, 0, bf16, st, EpiArgs{0ull, 0u, 0.f, scale});
}

// training path: C = dropout_{seed,site,p}(relu(A W^T + bias)) - mlp.fc1 + ReLU + mlp.dropout in one GEMM
int vsk_linear_relu_dropout(const float *A, const float *W, const float *Wf, const float *bias, float *C, int M, int N,
                            int K, unsigned long long seed, unsigned site, float p, hipStream_t st, int bf16) {
    if (bf16 & VSK_F16) return launch_gemm<EPI_RELU_DROP, 3>(A, W, Wf, bias, C, M, N, K, nullptr, 1, 0, 0, bf16 & ~VSK_F16, st, EpiArgs{seed, site, p, 0.f});
    return launch_gemm<EPI_RELU_DROP>(A, W, Wf, bias, C, M, N, K, nullptr, 1, 0, 0, bf16, st, EpiArgs{seed, site, p, 0.f});
}

#ifdef VS_WITH_DIAG
int vsk_mlp_fused(const float *H1, const float *W1, const float *b1, const float *W2, const float *b2,
                  const float *gamma, const float *beta, float *out, int M, int d,
                  const float *score_w, const float *score_b, int num_classes, int sigmoid, float *scores,
                  hipStream_t st) {
    if (d != 256) return -1;
    int blocks = persistent_blocks((M + 127) / 128, 1);
    if (blocks < 0) return (int)hipErrorInvalidDevice;
    if (blocks > (M + 127) / 128) blocks = (M + 127) / 128;
    const int abl = vsk_options().mlp_abl;
#define VSK_MLP(A_) hipLaunchKernelGGL(mlp_fused_256<A_>, dim3(blocks), dim3(256), 0, st, H1, W1, b1, W2, b2, gamma, beta, out, M, score_w, score_b, num_classes, sigmoid, scores)
    if (abl == 1) VSK_MLP(1); else if (abl == 2) VSK_MLP(2); else if (abl == 3) VSK_MLP(3); else VSK_MLP(0);
#undef VSK_MLP
    VSK_CHECK_LAUNCH();
    return 0;
}
#endif  // VS_WITH_DIAG

int vsk_pack_fragments(const float *W, float *Wf, int N, int K, hipStream_t st) {
    if (N % 32 || K % 8) return -1;
    hipLaunchKernelGGL(pack_fragments, dim3(256), dim3(256), 0, st, W, Wf, N, K);
    VSK_CHECK_LAUNCH();
    return 0;
}

// blockIdx.y = job, blockIdx.x strides over the job's float4 groups (same element map as pack_fragments)
__global__ void pack_fragments_batch(VskMatJobs jobs) {
    const int job = blockIdx.y;
    const float *__restrict__ W = jobs.in[job];
    float *__restrict__ Wf = jobs.out[job];
    const int N = jobs.rows[job], K = jobs.cols[job];
    const size_t total = (size_t)N * K / 4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const size_t gi = idx >> 6;
        const int g = (int)(gi % (K / 8)), nb = (int)(gi / (K / 8));
        const int row = 32 * nb + (lane & 31), col = 8 * g + 4 * (lane >> 5);
        *(f32x4 *)(Wf + idx * 4) = *(const f32x4 *)(W + (size_t)row * K + col);
    }
}

int vsk_pack_fragments_batch(const VskMatJobs &jobs, hipStream_t st) {
    if (jobs.n < 1 || jobs.n > VskMatJobs::MAX) return -1;
    for (int i = 0; i < jobs.n; ++i) if (jobs.rows[i] % 32 || jobs.cols[i] % 8) return -1;
    hipLaunchKernelGGL(pack_fragments_batch, dim3(64, jobs.n), dim3(256), 0, st, jobs);
    VSK_CHECK_LAUNCH();
    return 0;
}

int vsk_pack_fragments_f16x3(const float *W, float *Wh, int N, int K, hipStream_t st) {
    if (N % 32 || K % 16) return -1;
    hipLaunchKernelGGL(pack_fragments_f16x3, dim3(256), dim3(256), 0, st, W, (unsigned *)Wh, N, K);
    VSK_CHECK_LAUNCH();
    return 0;
}

#ifdef VS_WITH_DIAG
// diagnostic: the fc1-shaped GEMM with per-wave stamps (VS_DIAG_MODE 1..3, VS_DIAG_NWM 2|4); `grid` <= 0
// selects the product grid.  diag == nullptr runs the product kernel.
int vsk_diag_gemm(const float *A, const float *W, const float *bias, float *C, int M, int N, int K,
                  int grid, unsigned long long *diag, hipStream_t st) {
    const char *mode = getenv("VS_DIAG_MODE"), *nw = getenv("VS_DIAG_NWM");
    const int m = mode ? atoi(mode) : 1, nwm = nw ? atoi(nw) : 2;
    int blocks = nwm == 4 ? persistent_blocks(((M + 255) / 256) * ((N + 127) / 128), 1)
                          : persistent_blocks(((M + 127) / 128) * ((N + 127) / 128), 2);
    if (blocks < 0) return (int)hipErrorInvalidDevice;
    if (grid > 0) blocks = grid;
    const char *dprec = getenv("VS_DIAG_PREC");
    if (dprec && atoi(dprec) == 1) {      // the bf16 instantiation (256x256 tiles): mode 2 = without, else with the epilogue
        blocks = grid > 0 ? grid : persistent_blocks(((M + 255) / 256) * (N / 256), 1);
        if (m == 2)
            hipLaunchKernelGGL((gemm_nt_128<EPI_RELU, 4, 2, 4, 1, 0, 64>), dim3(blocks), dim3(512), 0, st, A, W, bias, C, M, N, K, nullptr, 1, 0, 0, diag);
        else
            hipLaunchKernelGGL((gemm_nt_128<EPI_RELU, 4, 3, 4, 1, 0, 64>), dim3(blocks), dim3(512), 0, st, A, W, bias, C, M, N, K, nullptr, 1, 0, 0, diag);
        VSK_CHECK_LAUNCH();
        return 0;
    }
    if (dprec) {                          // the fp16x3 instantiation (256x256 tiles): 2 = without, else with the epilogue
        blocks = grid > 0 ? grid : persistent_blocks(((M + 255) / 256) * (N / 256), 1);
        if (m == 2)
            hipLaunchKernelGGL((gemm_nt_128<EPI_RELU, 4, 2, 4, 2>), dim3(blocks), dim3(512), 0, st, A, W, bias, C, M, N, K, nullptr, 1, 0, 0, diag);
        else
            hipLaunchKernelGGL((gemm_nt_128<EPI_RELU, 4, 3, 4, 2>), dim3(blocks), dim3(512), 0, st, A, W, bias, C, M, N, K, nullptr, 1, 0, 0, diag);
        VSK_CHECK_LAUNCH();
        return 0;
    }
#define VSK_DG(NWM_, D_) hipLaunchKernelGGL((gemm_nt_128<EPI_RELU, NWM_, D_, 2>), dim3(blocks), dim3(128 * NWM_), 0, st, A, W, bias, C, M, N, K, nullptr, 1, 0, 0, diag)
    if (diag == nullptr) { if (nwm == 4) VSK_DG(4, 0); else VSK_DG(2, 0); }
    else if (m == 1) { if (nwm == 4) VSK_DG(4, 1); else VSK_DG(2, 1); }
    else if (m == 2) { if (nwm == 4) VSK_DG(4, 2); else VSK_DG(2, 2); }
    else { if (nwm == 4) VSK_DG(4, 3); else VSK_DG(2, 3); }
#undef VSK_DG
    VSK_CHECK_LAUNCH();
    return 0;
}
#endif  // VS_WITH_DIAG

// bf16 | VSK_STORE16: q (times qscale), k, v are written as bf16, three [B,H,T,dh] planes of M*d 2-byte elements
int vsk_qkv(const float *h, const float *Wqkv, const float *Wf, const float *bqkv, float *qkv, int B, int T, int d,
            int H, int bf16, hipStream_t st, float qscale) {
    if (bf16 & VSK_F16)
        return launch_gemm<EPI_QKV, 3>(h, Wqkv, Wf, bqkv, qkv, B * T, 3 * d, d, nullptr, T, H, d / H, bf16 & ~VSK_F16, st,
                                       EpiArgs{0ull, 0u, 0.f, qscale});
    return launch_gemm<EPI_QKV>(h, Wqkv, Wf, bqkv, qkv, B * T, 3 * d, d, nullptr, T, H, d / H, bf16, st,
                                EpiArgs{0ull, 0u, 0.f, qscale});
}

int vsk_linear_res_ln(const float *A, const float *W, const float *Wf, const float *bias, const float *res,
                      const float *gamma, const float *beta, float *out, int M, int N, int K,
                      const float *score_w, const float *score_b, int num_classes, int sigmoid,
                      float *scores, int bf16, hipStream_t st) {
    if (bf16 == 2 && Wf != nullptr && M <= skinny_max_rows() && N <= 256 && N % 32 == 0 && K % 128 == 0) {
        const int blocks = (M + 31) / 32;       // fp16x3 latency kernel (Wf: pack_fragments_f16x3 copy)
#define VSK_SLN3_CASE(NW_)                                                                                 \
    case NW_: {                                                                                            \
        if (const int attr_rc = VSK_ALLOW_BIG_LDS((skinny2_ln<NW_, 2>))) return attr_rc;                   \
        hipLaunchKernelGGL((skinny2_ln<NW_, 2>), dim3(blocks), dim3(64 * NW_), skinny2_lds(K), st, A, Wf, bias, res, \
                           gamma, beta, out, M, K, score_w, score_b, num_classes, sigmoid, scores);        \
        break;                                                                                             \
    }
        switch (N / 32) {
            VSK_SLN3_CASE(2) VSK_SLN3_CASE(4) VSK_SLN3_CASE(6) VSK_SLN3_CASE(8)
            default: return -1;
        }
#undef VSK_SLN3_CASE
        VSK_CHECK_LAUNCH();
        return 0;
    }
    const bool a16 = bf16 == (1 | VSK_STORE16);           // A stored as bf16 (bf16 mode only)
    if (a16) bf16 = 1;
    if (bf16 == 1 && (N > 256 || N % 32)) return -1;      // bf16: d_model <= 256 only
    if (bf16 && N <= 256) {          // low-precision lane-owns-a-row kernels (wider fp16x3: gemm_res_ln<NB, 2> below)
        if (N % 32) return -1;
        int blocks = persistent_blocks((M + 127) / 128);
        if (blocks < 0) return (int)hipErrorInvalidDevice;
        if (blocks > (M + 127) / 128) blocks = (M + 127) / 128;
#define VSK_LNB_CASE(NT_)                                                                                   \
    case NT_:                                                                                               \
        if (bf16 == 2)                                                                                      \
            hipLaunchKernelGGL((gemm_ln_rows<NT_, 2>), dim3(blocks), dim3(256), 0, st, A, W, bias, res, gamma, \
                               beta, out, M, K, score_w, score_b, num_classes, sigmoid, scores);            \
        else if (a16)                                                                                       \
            hipLaunchKernelGGL((gemm_ln_rows<NT_, 1, 1>), dim3(blocks), dim3(256), 0, st, A, W, bias, res, gamma, \
                               beta, out, M, K, score_w, score_b, num_classes, sigmoid, scores);            \
        else                                                                                                \
            hipLaunchKernelGGL((gemm_ln_rows<NT_, 1>), dim3(blocks), dim3(256), 0, st, A, W, bias, res, gamma, \
                               beta, out, M, K, score_w, score_b, num_classes, sigmoid, scores);            \
        break;
        switch (N / 32) {
            VSK_LNB_CASE(2) VSK_LNB_CASE(4) VSK_LNB_CASE(6) VSK_LNB_CASE(8)
            default: return -1;
        }
#undef VSK_LNB_CASE
        VSK_CHECK_LAUNCH();
        return 0;
    }
    if (Wf != nullptr && M <= skinny_max_rows() && N <= 256 && N % 32 == 0 && K % 128 == 0) {
        const int blocks = (M + 31) / 32;
#define VSK_SLN2_CASE(NW_)                                                                                 \
    case NW_: {                                                                                            \
        if (const int attr_rc = VSK_ALLOW_BIG_LDS((skinny2_ln<NW_, 0>))) return attr_rc;                   \
        hipLaunchKernelGGL((skinny2_ln<NW_, 0>), dim3(blocks), dim3(64 * NW_), skinny2_lds(K), st, A, Wf, bias, res, \
                           gamma, beta, out, M, K, score_w, score_b, num_classes, sigmoid, scores);        \
        break;                                                                                             \
    }
        switch (N / 32) {
            VSK_SLN2_CASE(2) VSK_SLN2_CASE(4) VSK_SLN2_CASE(6) VSK_SLN2_CASE(8)
            default: return -1;
        }
#undef VSK_SLN2_CASE
        VSK_CHECK_LAUNCH();
        return 0;
    }
    if (M <= skinny_max_rows() && N <= 256 && N % 32 == 0 && K % 128 == 0) {
        const int blocks = (M + 31) / 32;
#define VSK_SLN_CASE(NW_)                                                                          \
    case NW_:                                                                                      \
        hipLaunchKernelGGL(skinny_ln<NW_>, dim3(blocks), dim3(64 * NW_), 0, st, A, W, bias, res, gamma, \
                           beta, out, M, K, score_w, score_b, num_classes, sigmoid, scores);       \
        break;
        switch (N / 32) {
            VSK_SLN_CASE(2) VSK_SLN_CASE(4) VSK_SLN_CASE(6) VSK_SLN_CASE(8)
            default: return -1;
        }
#undef VSK_SLN_CASE
        VSK_CHECK_LAUNCH();
        return 0;
    }
    if (N <= 256 && N % 32 == 0) {
        int blocks = persistent_blocks((M + 127) / 128);
        if (blocks < 0) return (int)hipErrorInvalidDevice;
        if (blocks > (M + 127) / 128) blocks = (M + 127) / 128;
#define VSK_LNR_CASE(NT_)                                                                             \
    case NT_:                                                                                         \
        hipLaunchKernelGGL((gemm_ln_rows<NT_, 0>), dim3(blocks), dim3(256), 0, st, A, W, bias, res, gamma, \
                           beta, out, M, K, score_w, score_b, num_classes, sigmoid, scores);          \
        break;
        switch (N / 32) {
            VSK_LNR_CASE(2) VSK_LNR_CASE(4) VSK_LNR_CASE(6) VSK_LNR_CASE(8)
            default: return -1;
        }
#undef VSK_LNR_CASE
        VSK_CHECK_LAUNCH();
        return 0;
    }
    const int blocks = (M + 63) / 64;
#define VSK_LN_CASE(NB_)                                                                                  \
    case NB_:                                                                                             \
        if (bf16 == 2)                                                                                    \
            hipLaunchKernelGGL((gemm_res_ln<NB_, 2>), dim3(blocks), dim3(256), 0, st, A, W, bias, res, gamma, \
                               beta, out, M, K, score_w, score_b, num_classes, sigmoid, scores);          \
        else                                                                                              \
            hipLaunchKernelGGL((gemm_res_ln<NB_, 0>), dim3(blocks), dim3(256), 0, st, A, W, bias, res, gamma, \
                               beta, out, M, K, score_w, score_b, num_classes, sigmoid, scores);          \
        break;
    switch (N / 64) {
        VSK_LN_CASE(1) VSK_LN_CASE(2) VSK_LN_CASE(3) VSK_LN_CASE(4)
        VSK_LN_CASE(5) VSK_LN_CASE(6) VSK_LN_CASE(7) VSK_LN_CASE(8)
        default: return -1;
    }
#undef VSK_LN_CASE
    VSK_CHECK_LAUNCH();
    return 0;
}
