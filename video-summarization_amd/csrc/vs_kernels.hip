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
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "vs_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

namespace {

__device__ __forceinline__ int acc_row(int t, int h) { return (t & 3) + 8 * (t >> 2) + 4 * h; }

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{})
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Attention block -> (video*head, query tile).  Blocks b and b+8 share an XCD and are dispatched in
// order, so the nq query tiles of one (video, head) are given to nq CONSECUTIVE blocks of one XCD:
// they run at the same time and that head's K/V is fetched from HBM/MALL into one L2 once instead
// of once per query tile (measured: 5.7x the algorithmic bytes without this).  Speed only.
__device__ __forceinline__ bool attn_block_map(int nq, int BH, int &bh, int &qt) {
    const int L = blockIdx.x, x = L & 7, s = L >> 3;
    qt = s % nq;
    bh = x + 8 * (s / nq);
    return bh < BH;
}

// v_permlane32_swap of a register with itself yields {x_lo | x_lo} and {x_hi | x_hi}: every lane then
// sees both its own and its lane^32 partner's value, so a symmetric combine needs no select.
__device__ __forceinline__ float pair_max(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)pr[0]), __builtin_bit_cast(float, (unsigned)pr[1]));
}
__device__ __forceinline__ float pair_sum(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)pr[0]) + __builtin_bit_cast(float, (unsigned)pr[1]);
}

// ---- bf16 matrix pipe (opt-in paths) ----
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// two floats -> packed bf16 pair (round to nearest even): lowers to one v_cvt_pk_bf16_f32.  NOT inline asm:
// the hazard recogniser must see this instruction - it needs a wait state after a v_exp_f32 (trans unit)
// producing its input, and an asm statement does not get one (measured: wrong products).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const f32x2 f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}

// ---- f16 matrix pipe used to EMULATE fp32 (opt-in "fp16x3"): x ~= hi + lo with hi = f16(x), lo = f16(x - hi)
// (22 significant bits), and a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi accumulated in fp32 (the dropped
// a_lo*b_lo is ~2^-22 of the product).  Three 32x32x16 f16 MFMAs (96 cycles) replace eight 32x32x2 fp32 MFMAs
// (512 cycles).  Operand magnitudes must stay below the f16 range (65504).
// Weights are multiplied by F16X3_WS = 2^10 on their way into the split (exact), so that the lo half of a typical
// weight (|w| ~ 0.03, lo ~ 2^-12 |w|) is a normal f16 number instead of a subnormal with 2^-24 absolute steps;
// the accumulators then hold 2^10 times the result and the epilogue multiplies by 2^-10 (exact) - or, in the
// LayerNorm kernel, scales residual, bias and eps instead (LayerNorm is scale-invariant).  |w| must be < 63.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float F16X3_WS = 1024.0f;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ void split_f16(float x0, float x1, unsigned &hi, unsigned &lo) {
    const f32x2 x = {x0, x1};
    const f16x2 hv = __builtin_convertvector(x, f16x2);
    const f32x2 rem = x - __builtin_convertvector(hv, f32x2);
    hi = __builtin_bit_cast(unsigned, hv);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(rem, f16x2));
}
__device__ __forceinline__ void split_f16x4(const float *v, u32x2 &hi, u32x2 &lo) {   // 4 floats -> 2+2 dwords
    unsigned h0, l0, h1, l1;
    split_f16(v[0], v[1], h0, l0);
    split_f16(v[2], v[3], h1, l1);
    hi[0] = h0; hi[1] = h1; lo[0] = l0; lo[1] = l1;
}

// max(x, 0) as ONE instruction: integer max on the bit pattern (negative floats are negative integers).
// fmaxf would first canonicalise its MFMA-produced input with a second v_max; and NOT inline asm: an asm
// statement reading an MFMA result gets no hazard wait states from the compiler (measured: wrong values when
// the scheduler placed it right behind the MFMA).
__device__ __forceinline__ float relu1(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

__device__ __forceinline__ float half_sum(float v) {   // sum over the 32 lanes of a half-wave
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

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
enum { EPI_BIAS = 0, EPI_RELU = 1, EPI_PE = 2, EPI_QKV = 3 };

// Diagnostic stamps (cdna guide §7 "In-kernel stamps"): compiled only into the DIAG instantiation,
// which is reachable only through vs_diag_gemm(); no product launch executes a stamp.
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

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
template <int EPI, int NWM = 2, int DIAG = 0, int NJ = 2, int PREC = 0>     // DIAG (tools/diag_gemm.py only) 2: epilogue skipped (wrong output) + per-wave cycles/wall clock; 1, 3: the same with the epilogue
__global__ __launch_bounds__(128 * NWM, 2) void gemm_nt_128(
    const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
    float *__restrict__ C, int M, int N, int K, const float *__restrict__ pe, int T, int H, int dh,
    unsigned long long *__restrict__ diag = nullptr) {
    constexpr int BM = 64 * NWM, BN = 64 * NJ, BK = 32, LD = BK + 4;
    constexpr int NT = 128 * NWM;                       // threads
    constexpr int LA = BM * 8 / NT, LW = BN * 8 / NT;   // float4 of A / of W per thread per k-tile (4, 4 | 4, 2)
    constexpr int RS = NT / 8;                          // row stride of the staging map
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
    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;      // staging map: rows lrow + RS*i
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
            aptr[i] = A + (size_t)ar * K + lc4;
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
    constexpr int LDB = 20;                             // BF: LDS row stride in floats (80 B = 32 bf16 + pad)
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
        } else if constexpr (PREC == 1) {
#pragma unroll
            for (int i = 0; i < LA; ++i) {
                u32x2 u; u[0] = pack_bf16(pa[i][0], pa[i][1]); u[1] = pack_bf16(pa[i][2], pa[i][3]);
                *(u32x2 *)&As[(lrow + RS * i) * LDB + lc4 / 2] = u;
            }
#pragma unroll
            for (int i = 0; i < LW; ++i) {
                u32x2 u; u[0] = pack_bf16(pw[i][0], pw[i][1]); u[1] = pack_bf16(pw[i][2], pw[i][3]);
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
        if constexpr (PREC == 1) {
            // 2 k-steps of 16: lane (r,h) supplies k = 16ks + 8h .. +7 (one b128 of the bf16 row)
            const float *ap = As + (64 * wr + r) * LDB + 4 * h;
            const float *wp = Ws + (32 * NJ * wc + r) * LDB + 4 * h;
#pragma unroll
            for (int i = 0; i < LA; ++i) pa[i] = *(const f32x4 *)(aptr[i] + koff);
#pragma unroll
            for (int i = 0; i < LW; ++i) pw[i] = *(const f32x4 *)(wptr[i] + koff);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fa[2], fw[NJ];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(ap + 32 * i * LDB + 8 * ks));
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) fw[jj] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(wp + 32 * jj * LDB + 8 * ks));
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    acc[0][jj] = MFMA_BF16(fw[jj], fa[0], acc[0][jj]);
                    acc[1][jj] = MFMA_BF16(fw[jj], fa[1], acc[1][jj]);
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
    for (int i = 0; i < LA; ++i) pa[i] = *(const f32x4 *)aptr[i];
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
                        if (EPI == EPI_RELU) v[e] = relu1(v[e]);
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
                        if (EPI == EPI_QKV)
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
template <int NB>
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

    {
        float *As = smem, *Ws = smem + BM * LD;
        *(f32x4 *)&As[lrow * LD + lc4] = pa;
#pragma unroll
        for (int i = 0; i < NB; ++i) *(f32x4 *)&Ws[(lrow + 64 * i) * LD + lc4] = pw[i];
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
        if (more) {
            float *An = smem + (cur ^ 1) * (BM + N) * LD, *Wn = An + BM * LD;
            *(f32x4 *)&An[lrow * LD + lc4] = pa;
#pragma unroll
            for (int i = 0; i < NB; ++i) *(f32x4 *)&Wn[(lrow + 64 * i) * LD + lc4] = pw[i];
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
            const float v = acc[j][t] + bj[j] + res[(size_t)row * N + cbase + 32 * j];
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
template <int NT, int PREC = 0>     // PREC 1: bf16 MFMA operands (see gemm_nt_128), LDS rows of 16 bf16 padded to 48 B; 2: f16 hi|lo rows (80 B)
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
    const float *aptr[2], *wptr[WL];
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
                u32x2 u; u[0] = pack_bf16(pa[i][0], pa[i][1]); u[1] = pack_bf16(pa[i][2], pa[i][3]);
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
            aptr[i] = A + (size_t)ar * K + lc4;
            pa[i] = *(const f32x4 *)aptr[i];
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
                for (int i = 0; i < 2; ++i) pa[i] = *(const f32x4 *)(aptr[i] + kn * BK);
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
    float *__restrict__ C, int M, int N, int K, const float *__restrict__ pe, int T, int H, int dh) {
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
        if (EPI == EPI_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = relu1(v[e]);
        }
        if (EPI == EPI_PE) {
            const f32x4 pv = *(const f32x4 *)(pe + (size_t)tt * N + n0 + co);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += pv[e];
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

template <int EPI>
__global__ __launch_bounds__(256) void skinny2_gemm(
    const float *__restrict__ A, const float *__restrict__ Wf, const float *__restrict__ bias,
    float *__restrict__ C, int M, int N, int K, const float *__restrict__ pe, int T, int H, int dh) {
    extern __shared__ __attribute__((aligned(16))) float As[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 128 + 32 * wave;
    const bool live = n0 < N;
    const int row = m0 + r;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 acc = MFMA32((h == 0 && live) ? bias[n0 + r] : 0.f, 1.0f, zero);
    const int kpmax = K < 1024 ? K : 1024;
    for (int k0 = 0; k0 < K; k0 += kpmax) {
        const int kp = K - k0 < kpmax ? K - k0 : kpmax;
        if (k0) __syncthreads();
        skinny2_stage<256>(As, A, M, K, m0, k0, kp);
        __syncthreads();
        if (live)
            skinny2_phase(acc, As + r * (kp + 4) + 4 * h, Wf + ((size_t)(n0 / 32) * (K / 8) + k0 / 8) * 256 + lane * 4, kp);
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
        if (EPI == EPI_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = relu1(v[e]);
        }
        if (EPI == EPI_PE) {
            const f32x4 pv = *(const f32x4 *)(pe + (size_t)tt * N + n0 + co);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += pv[e];
        }
        if (EPI == EPI_QKV)
            *(f32x4 *)(C + (size_t)which * M * (H * dh) + (((size_t)bb * H + head) * T + tt) * dh + e0 + co) = v;
        else
            *(f32x4 *)(C + (size_t)row * N + n0 + co) = v;
    }
}

template <int NW>
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
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 rv = *(const f32x4 *)(res + (size_t)arow * N + n0 + 8 * q + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * q + e] = rv[e];
    }
    const int kpmax = K < 1024 ? K : 1024;
    for (int k0 = 0; k0 < K; k0 += kpmax) {
        const int kp = K - k0 < kpmax ? K - k0 : kpmax;
        if (k0) __syncthreads();
        skinny2_stage<64 * NW>(As, A, M, K, m0, k0, kp);
        __syncthreads();
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
// Attention: softmax(q k^T * scale + keymask) v without materialising [T,T].
//   grid = (ceil(T/128), B*H); 4 waves, each owns 32 query rows and walks all key tiles.
//   Both products keep the QUERY on the lane: S^T = K * Q^T  (A = K tile from LDS, B = Q in
//   registers) leaves, for query r, 16 keys per register set; O^T = V^T * P^T then takes that
//   accumulator register t directly as its B operand (keys (t&3)+8(t>>2)+4h — exactly the
//   k-pair of MFMA step t) with A = V[key][d-column] read from LDS.  So the softmax row
//   statistics (max, sum, rescale) are lane-local plus one exchange with lane^32, and P never
//   leaves registers.
// ------------------------------------------------------------------------------------------
template <int DH, int NKB>   // NKB 32-key blocks per tile
__global__ __launch_bounds__(256, 2) void attn_fwd(
    const float *__restrict__ Q, const float *__restrict__ Kg, const float *__restrict__ Vg,
    const uint8_t *__restrict__ mask, float *__restrict__ out, int H, int T, float scale_log2e, int BH) {
    constexpr int KT = 32 * NKB, LD = DH + 4, NJ = DH / 8, ND = DH / 32;
    constexpr int F4 = KT * DH / 4 / 256;          // float4 per thread per operand tile
    __shared__ __attribute__((aligned(16))) float Ks[KT * LD];
    __shared__ __attribute__((aligned(16))) float Vs[KT * LD];
    __shared__ float mb[KT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    int bh, qt;
    if (!attn_block_map((T + 127) / 128, BH, bh, qt)) return;
    const int b = bh / H, head = bh - b * H;
    const size_t base = (size_t)bh * T * DH;
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
template <int DH, int NW, int PREC>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_lp_pipe(
    const float *__restrict__ Q, const float *__restrict__ Kg, const float *__restrict__ Vg,
    const uint8_t *__restrict__ mask, float *__restrict__ out, int H, int T, float scale_log2e, int BH) {
    constexpr int KT = 64, NS = DH / 16, ND = DH / 32, NT = 64 * NW, QB = 32 * NW;
    constexpr int NP = PREC == 2 ? 2 : 1, NPROD = PREC == 2 ? 3 : 1;
    constexpr int LDK = DH + 8, LDV = KT + 4, D4 = DH / 4;
    constexpr int KPT = KT * D4 / NT;
    static_assert(KPT == 2 || KPT == 4, "staging packs 2 or 4 keys per V^T store");
    constexpr float THR = 8.0f;
    typedef unsigned short h16;
    __shared__ __attribute__((aligned(16))) h16 Kb[2][NP][KT * LDK];
    __shared__ __attribute__((aligned(16))) h16 Vt[2][NP][DH * LDV];
    __shared__ __attribute__((aligned(16))) float mb[3][KT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    int bh, qt;
    if (!attn_block_map((T + QB - 1) / QB, BH, bh, qt)) return;
    const int b = bh / H, head = bh - b * H;
    const size_t base = (size_t)bh * T * DH;
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
    float m_run = NEG_INF;
    constexpr unsigned ONE2 = PREC == 2 ? 0x3C003C00u : 0x3F803F80u;
    const u32x4 ones_u = {ONE2, ONE2, ONE2, ONE2};

    const int d4 = tid % D4, kq = tid / D4;
    const int ntiles = (T + KT - 1) / KT;
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Kg + base), 0, T * DH * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Vg + base), 0, T * DH * 4, 0x00020000);
    int voff[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) voff[i] = ((KPT * kq + i) * DH + 4 * d4) * 4;
    f32x4 pk[KPT], pv[KPT];
    float pm = 0.f;
    // the pipeline runs up to three tiles past the end: those loads re-read the last tile (their products are
    // never consumed); rows beyond T inside the last tile read as zeros (buffer bounds check) and are masked
    auto gload_k = [&](int tile) __attribute__((always_inline)) {
        tile = tile < ntiles ? tile : ntiles - 1;
#pragma unroll
        for (int i = 0; i < KPT; ++i)
            pk[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, voff[i], tile * (KT * DH * 4), 0));
    };
    auto gload_v = [&](int tile) __attribute__((always_inline)) {
        tile = tile < ntiles ? tile : ntiles - 1;
#pragma unroll
        for (int i = 0; i < KPT; ++i)
            pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vrs, voff[i], tile * (KT * DH * 4), 0));
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
    auto k_frag = [&](int buf, int ks, int n, u32x4 (&ka)[NP]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NP; ++p) ka[p] = *(const u32x4 *)&Kb[buf][p][(32 * n + r) * LDK + 16 * ks + 8 * h];
    };
    auto v_frag = [&](int buf, int n, int ks, int d, u32x4 (&va)[NP]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const h16 *vp = &Vt[buf][p][(32 * d + r) * LDV + 32 * n + 16 * ks + 4 * h];
            const u32x2 lo = *(const u32x2 *)vp, hi = *(const u32x2 *)(vp + 8);
            va[p][0] = lo[0]; va[p][1] = lo[1]; va[p][2] = hi[0]; va[p][3] = hi[1];
        }
    };

    // ---- prologue: K(0), V(0), K(1) in LDS; K(2), V(1) in registers; S(0) computed ----
    gload_k(0); gload_v(0); gload_m(0);
#pragma unroll
    for (int i = 0; i < KPT; ++i) stage_k1(i, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) stage_v1(e, 0);
    stage_m(0);
    gload_k(1); gload_m(1);
#pragma unroll
    for (int i = 0; i < KPT; ++i) stage_k1(i, 1);
    stage_m(1);
    gload_k(2); gload_m(2); gload_v(1);
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

    // softmax state of the tile in flight
    float sm_mx = 0.f;
    f32x2 sm_mm = {0.f, 0.f};
    u32x4 pf[2][2][NP];
    // one unit of softmax(t) work on s_in: 0..3 max3 chain, 4 deferred-max check (+ rare rescale), 5..20 one
    // (n, ks, j) pair each: subtract, exp2, pack / split into the P fragments
    auto sm_unit = [&](auto uc, f32x16 (&sv)[2]) __attribute__((always_inline)) {
        constexpr int U = decltype(uc)::value;
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
    constexpr int NU = 21;
    constexpr int NSLOT_A = NS * 2 * NPROD, RA = (NU + NSLOT_A - 1) / NSLOT_A;
    constexpr int NSLOT_B = 4 * (ND * NPROD + NP);
    constexpr int NITEM = KPT + 4 + 2, SB = NSLOT_B / NITEM > 0 ? NSLOT_B / NITEM : 1;    // staging items, slot stride

    // one iteration: s_in = S(t) (complete), s_out <- S(t+1)
    auto iteration = [&](int t, f32x16 (&s_in)[2], f32x16 (&s_out)[2]) __attribute__((always_inline)) {
        const int cur = t & 1, nxt = cur ^ 1;
        if (mask != nullptr || (t + 1) * KT > T) {       // masked / ragged tile: key-mask bias first (rare)
            const float *mp = mb[t % 3];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bv = *(const f32x4 *)&mp[32 * n + 8 * g + 4 * h];
#pragma unroll
                    for (int e = 0; e < 4; ++e) s_in[n][4 * g + e] += bv[e];
                }
        }
        // ---- phase A: S(t+1) MFMAs, softmax(t) between them ----
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int x = 0; x < 16; ++x) s_out[n][x] = 0.f;
        u32x4 ka[2][NP];
        k_frag(nxt, 0, 0, ka[0]);
        __builtin_amdgcn_sched_barrier(0);
        static_for<NSLOT_A>([&](auto ic) {
            constexpr int i = decltype(ic)::value, g = i / NPROD, pr = i % NPROD, ks = g / 2, n = g % 2;
            if constexpr (pr == 0 && g + 1 < NS * 2) k_frag(nxt, (g + 1) / 2, (g + 1) % 2, ka[(g + 1) & 1]);
            mma_prod(std::integral_constant<int, pr>{}, ka[g & 1], qreg[ks], s_out[n]);
            static_for<RA>([&](auto rc) {
                constexpr int U = i * RA + decltype(rc)::value;
                if constexpr (U < NU) sm_unit(std::integral_constant<int, U>{}, s_in);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        // ---- phase B: O += V(t)^T P(t)^T and the row sums; staging of K(t+2), V(t+1), loads of K(t+3), V(t+2) ----
        u32x4 va[2][NP];
        v_frag(cur, 0, 0, 0, va[0]);
        __builtin_amdgcn_sched_barrier(0);
        static_for<NSLOT_B>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int PER = ND * NPROD + NP;                 // MFMAs per (n, ks): products per d block, then row sums
            constexpr int g = i / PER, w = i % PER, n = g / 2, ks = g % 2;
            if constexpr (w < ND * NPROD) {
                constexpr int d = w / NPROD, pr = w % NPROD, f = g * ND + d;       // fragment index
                if constexpr (pr == 0 && f + 1 < 4 * ND)
                    v_frag(cur, (f + 1) / ND / 2, ((f + 1) / ND) % 2, (f + 1) % ND, va[(f + 1) & 1]);
                mma_prod(std::integral_constant<int, pr>{}, va[f & 1], pf[n][ks], o[d]);
            } else {
                constexpr int p = NP - 1 - (w - ND * NPROD);     // lo plane first
                o[ND] = mma(ones_u, pf[n][ks][p], o[ND]);
            }
            if constexpr (i % SB == SB - 1 && i / SB < NITEM) {
                constexpr int item = i / SB;
                if constexpr (item < KPT) stage_k1(item, cur);
                else if constexpr (item < KPT + 4) stage_v1(item - KPT, nxt);
                else if constexpr (item == KPT + 4) { stage_m((t + 2) % 3); }
                else { gload_k(t + 3); gload_m(t + 3); gload_v(t + 2); }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        __syncthreads();
    };

    int t = 0;
    for (; t + 1 < ntiles; t += 2) {
        iteration(t, sa, sb);
        iteration(t + 1, sb, sa);
    }
    if (t < ntiles) iteration(t, sa, sb);

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
template <int DH, bool HAS_MASK, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_pipe(
    const float *__restrict__ Q, const float *__restrict__ Kg, const float *__restrict__ Vg,
    const uint8_t *__restrict__ mask, float *__restrict__ out, int H, int T, float scale_log2e, int BH) {
    constexpr int KT = 64, LD = DH + 4, NJ = DH / 8, ND = DH / 32;
    constexpr int NT = 64 * NW;                     // threads per block
    constexpr int F4 = KT * DH / 4 / NT;            // float4 per thread per operand tile
    constexpr int TILE = KT * LD;                   // floats per K (or V) tile in LDS
    constexpr float THR = 8.0f;                     // deferred-max threshold (log2 units)
    __shared__ __attribute__((aligned(16))) float smem[4 * TILE + 2 * KT];   // [buf]{K,V} + mask bias
    float *mbs = smem + 4 * TILE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    int bh, qt;
    if (!attn_block_map((T + 32 * NW - 1) / (32 * NW), BH, bh, qt)) return;
    const int b = bh / H, head = bh - b * H;
    const size_t base = (size_t)bh * T * DH;
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
        // S' of block (t,1)  ||  softmax of block (t,0), LDS writes of tile t+1, loads of tile t+2
        mb_nxt = (m_run == NEG_INF) ? 0.f : m_run;
        half_step(Ks, 1, s_nxt, mb_nxt, s_cur, mb_cur, mbs + buf * KT, Vs, 0, masked_tag, std::integral_constant<bool, !LAST>{},
                  buf ^ 1, t + 2 < ntiles ? t + 2 : ntiles - 1);
        pv_acc(s_cur);
        __syncthreads();                       // tile t+1 is visible in buffer buf^1
        if constexpr (!LAST) {
            // S' of block (t+1,0)  ||  softmax of block (t,1)
            mb_cur = (m_run == NEG_INF) ? 0.f : m_run;
            half_step(Kn, 0, s_cur, mb_cur, s_nxt, mb_nxt, mbs + buf * KT + 32, Vs, 1, masked_tag, std::false_type{}, 0, 0);
        } else {
            softmax_only(s_nxt, mb_nxt, mbs + buf * KT + 32, Vs, 1, masked_tag);
        }
        pv_acc(s_nxt);
    };
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
            if (q < T) *(f32x4 *)(out + ((size_t)b * T + q) * (H * DH) + head * DH + oc4) = v;
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host-side launchers (plain C++ interface used by vs_scorer.cpp)
// ------------------------------------------------------------------------------------------
#define VSK_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

// persistent grid: `per_cu` blocks per CU, a multiple of 8 so the XCD chunking is exact
static int persistent_blocks(int ntiles, int per_cu = 2) {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        cus = prop.multiProcessorCount;
    }
    int g = per_cu * cus;
    g -= g % 8;
    if (g < 8) g = 8;
    const int need = (ntiles + 7) / 8 * 8;
    return need < g ? need : g;
}

// 256x128 tiles on 8-wave blocks when there is enough work to give every CU >= 2 such tiles and the
// ragged M edge does not waste more than 128-row tiles would; else 128x128 tiles on 4-wave blocks
static bool use_wide_tiles(int M, int N) {
    if (getenv("VS_GEMM_NWM2")) return false;
    const long r256 = (M + 255) / 256 * 256, r128 = (M + 127) / 128 * 128;
    const long tiles = (r256 / 256) * ((N + 127) / 128);
    return r256 * 100 <= r128 * 105 && tiles >= 512;
}

// latency path: below this many rows the LDS-tiled kernels cannot fill the chip
// measured hand-over (tools/sweep_skinny.py, T=1024, M-A): skinny wins through M = 16384, tiled from 32768
int vsk_skinny_max_rows() {
    const char *e = getenv("VS_SKINNY_ROWS");      // read per call so tests can pin either path
    return e ? atoi(e) : 16384;
}
static int skinny_max_rows() { return vsk_skinny_max_rows(); }

// dynamic LDS of the packed skinny kernels: 32 activation rows x (min(K,1024) + 4) floats (up to 128.5 KiB)
static size_t skinny2_lds(int K) { return (size_t)32 * ((K < 1024 ? K : 1024) + 4) * sizeof(float); }
template <class F>
static int allow_big_lds(F *kernel) {
    return (int)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 1028 * 4);
}

template <int EPI>
static int launch_gemm(const float *A, const float *W, const float *Wf, const float *bias, float *C, int M, int N, int K,
                       const float *pe, int T, int H, int dh, int bf16, hipStream_t st) {
    if (bf16 == 2) {     // fp32 emulated on the f16 pipe (opt-in): always the LDS-tiled kernels
        if (N % 256 == 0 && M > 128) {
            const int blocks = persistent_blocks(((M + 255) / 256) * (N / 256), 1);
            if (blocks < 0) return (int)hipErrorInvalidDevice;
            hipLaunchKernelGGL((gemm_nt_128<EPI, 4, 0, 4, 2>), dim3(blocks), dim3(512), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr);
        } else {
            const int blocks = persistent_blocks(((M + 127) / 128) * ((N + 127) / 128), 2);
            if (blocks < 0) return (int)hipErrorInvalidDevice;
            hipLaunchKernelGGL((gemm_nt_128<EPI, 2, 0, 2, 2>), dim3(blocks), dim3(256), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr);
        }
        VSK_CHECK_LAUNCH();
        return 0;
    }
    if (bf16) {          // bf16 matrix pipe (opt-in): always the LDS-tiled kernels
        if (N % 256 == 0 && M > 128) {
            const int blocks = persistent_blocks(((M + 255) / 256) * (N / 256), 1);
            if (blocks < 0) return (int)hipErrorInvalidDevice;
            hipLaunchKernelGGL((gemm_nt_128<EPI, 4, 0, 4, 1>), dim3(blocks), dim3(512), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr);
        } else {
            const int blocks = persistent_blocks(((M + 127) / 128) * ((N + 127) / 128), 2);
            if (blocks < 0) return (int)hipErrorInvalidDevice;
            hipLaunchKernelGGL((gemm_nt_128<EPI, 2, 0, 2, 1>), dim3(blocks), dim3(256), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr);
        }
        VSK_CHECK_LAUNCH();
        return 0;
    }
    if (Wf != nullptr && M <= skinny_max_rows() && N % 32 == 0 && K % 128 == 0) {
        static const int attr_rc = allow_big_lds(skinny2_gemm<EPI>);
        if (attr_rc) return attr_rc;
        dim3 grid((M + 31) / 32, (N + 127) / 128);
        hipLaunchKernelGGL((skinny2_gemm<EPI>), grid, dim3(256), skinny2_lds(K), st, A, Wf, bias, C, M, N, K, pe, T, H, dh);
        VSK_CHECK_LAUNCH();
        return 0;
    }
    if (M <= skinny_max_rows() && N % 32 == 0 && K % 128 == 0) {
        dim3 grid((M + 31) / 32, (N + 127) / 128);
        hipLaunchKernelGGL((skinny_gemm<EPI>), grid, dim3(256), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh);
        VSK_CHECK_LAUNCH();
        return 0;
    }
    if (use_wide_tiles(M, N) && N % 256 == 0 && !getenv("VS_GEMM_NJ2")) {
        const int blocks = persistent_blocks(((M + 255) / 256) * (N / 256), 1);
        if (blocks < 0) return (int)hipErrorInvalidDevice;
        hipLaunchKernelGGL((gemm_nt_128<EPI, 4, 0, 4, 0>), dim3(blocks), dim3(512), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr);
    } else if (use_wide_tiles(M, N)) {
        const int blocks = persistent_blocks(((M + 255) / 256) * ((N + 127) / 128), 1);
        if (blocks < 0) return (int)hipErrorInvalidDevice;
        hipLaunchKernelGGL((gemm_nt_128<EPI, 4, 0, 2, 0>), dim3(blocks), dim3(512), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr);
    } else {
        const int blocks = persistent_blocks(((M + 127) / 128) * ((N + 127) / 128), 2);
        if (blocks < 0) return (int)hipErrorInvalidDevice;
        hipLaunchKernelGGL((gemm_nt_128<EPI, 2, 0, 2, 0>), dim3(blocks), dim3(256), 0, st, A, W, bias, C, M, N, K, pe, T, H, dh, nullptr);
    }
    VSK_CHECK_LAUNCH();
    return 0;
}

int vsk_linear(const float *A, const float *W, const float *Wf, const float *bias, float *C, int M, int N, int K,
               int relu, const float *pe, int T, int bf16, hipStream_t st) {
    if (pe != nullptr) return launch_gemm<EPI_PE>(A, W, Wf, bias, C, M, N, K, pe, T, 0, 0, bf16, st);
    if (relu) return launch_gemm<EPI_RELU>(A, W, Wf, bias, C, M, N, K, nullptr, 1, 0, 0, bf16, st);
    return launch_gemm<EPI_BIAS>(A, W, Wf, bias, C, M, N, K, nullptr, 1, 0, 0, bf16, st);
}

int vsk_mlp_fused(const float *H1, const float *W1, const float *b1, const float *W2, const float *b2,
                  const float *gamma, const float *beta, float *out, int M, int d,
                  const float *score_w, const float *score_b, int num_classes, int sigmoid, float *scores,
                  hipStream_t st) {
    if (d != 256) return -1;
    int blocks = persistent_blocks((M + 127) / 128, 1);
    if (blocks < 0) return (int)hipErrorInvalidDevice;
    if (blocks > (M + 127) / 128) blocks = (M + 127) / 128;
    const char *ab = getenv("VS_MLP_ABL");
    const int abl = ab ? atoi(ab) : 0;
#define VSK_MLP(A_) hipLaunchKernelGGL(mlp_fused_256<A_>, dim3(blocks), dim3(256), 0, st, H1, W1, b1, W2, b2, gamma, beta, out, M, score_w, score_b, num_classes, sigmoid, scores)
    if (abl == 1) VSK_MLP(1); else if (abl == 2) VSK_MLP(2); else if (abl == 3) VSK_MLP(3); else VSK_MLP(0);
#undef VSK_MLP
    VSK_CHECK_LAUNCH();
    return 0;
}

int vsk_pack_fragments(const float *W, float *Wf, int N, int K, hipStream_t st) {
    if (N % 32 || K % 8) return -1;
    hipLaunchKernelGGL(pack_fragments, dim3(256), dim3(256), 0, st, W, Wf, N, K);
    VSK_CHECK_LAUNCH();
    return 0;
}

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
    if (getenv("VS_DIAG_PREC")) {         // the fp16x3 instantiation (256x256 tiles): 2 = without, else with the epilogue
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

int vsk_qkv(const float *h, const float *Wqkv, const float *Wf, const float *bqkv, float *qkv, int B, int T, int d,
            int H, int bf16, hipStream_t st) {
    return launch_gemm<EPI_QKV>(h, Wqkv, Wf, bqkv, qkv, B * T, 3 * d, d, nullptr, T, H, d / H, bf16, st);
}

int vsk_attention(const float *q, const float *k, const float *v, const uint8_t *mask, float *out,
                  int B, int H, int T, int dh, float scale, hipStream_t st) {
    const float sl2 = scale * 1.4426950408889634f;
    const int BH = B * H;
    dim3 grid(8 * ((BH + 7) / 8) * ((T + 127) / 128));
    static const bool legacy = getenv("VS_ATTN_LEGACY") != nullptr;      // A/B switch for tools/, not a fallback
    if (dh == 32 && legacy)
        hipLaunchKernelGGL((attn_fwd<32, 2>), grid, dim3(256), 0, st, q, k, v, mask, out, H, T, sl2, BH);
    else if (dh == 64 && legacy)
        hipLaunchKernelGGL((attn_fwd<64, 2>), grid, dim3(256), 0, st, q, k, v, mask, out, H, T, sl2, BH);
    else if (dh == 32 || dh == 64) {
        // 8-wave blocks (one per CU, 256 query rows) unless the ragged tail would waste more rows than
        // 4-wave blocks (two per CU, 128 query rows) do
        const int r8 = (T + 255) / 256 * 256, r4 = (T + 127) / 128 * 128;
        const bool wide = !getenv("VS_ATTN_NW4") && r8 * 100 <= r4 * 105;
        const int nq = wide ? r8 / 256 : r4 / 128;
        dim3 g(8 * ((BH + 7) / 8) * nq), blk(wide ? 512 : 256);
#define VSK_ATTN(DH_, MASK_, NW_) \
    hipLaunchKernelGGL((attn_fwd_pipe<DH_, MASK_, NW_>), g, blk, 0, st, q, k, v, mask, out, H, T, sl2, BH)
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

int vsk_attention_bf16(const float *q, const float *k, const float *v, const uint8_t *mask, float *out,
                       int B, int H, int T, int dh, float scale, int prec, hipStream_t st) {
    const float sl2 = scale * 1.4426950408889634f;
    const int BH = B * H;
    // 8-wave blocks (256 query rows share one staged K/V tile) unless the ragged tail would waste too many rows
    const int r8 = (T + 255) / 256 * 256, r4 = (T + 127) / 128 * 128;
    const bool wide = dh == 64 && !getenv("VS_ATTN_NW4") && r8 * 100 <= r4 * 105;
    dim3 grid(8 * ((BH + 7) / 8) * (wide ? r8 / 256 : r4 / 128));
    static const bool simple = getenv("VS_ATTN_LP_SIMPLE") != nullptr;      // A/B switch for tools/, not a fallback
#define VSK_ATTN_LP(KERN_, P_)                                                                                           \
    if (dh == 64 && wide)                                                                                                \
        hipLaunchKernelGGL((KERN_<64, 8, P_>), grid, dim3(512), 0, st, q, k, v, mask, out, H, T, sl2, BH);               \
    else if (dh == 64)                                                                                                   \
        hipLaunchKernelGGL((KERN_<64, 4, P_>), grid, dim3(256), 0, st, q, k, v, mask, out, H, T, sl2, BH);               \
    else if (dh == 32)                                                                                                   \
        hipLaunchKernelGGL((KERN_<32, 4, P_>), grid, dim3(256), 0, st, q, k, v, mask, out, H, T, sl2, BH);               \
    else                                                                                                                 \
        return -1;
    if (simple) { if (prec == 2) { VSK_ATTN_LP(attn_fwd_lp, 2) } else { VSK_ATTN_LP(attn_fwd_lp, 1) } }
    else        { if (prec == 2) { VSK_ATTN_LP(attn_fwd_lp_pipe, 2) } else { VSK_ATTN_LP(attn_fwd_lp_pipe, 1) } }
#undef VSK_ATTN_LP
    VSK_CHECK_LAUNCH();
    return 0;
}

int vsk_linear_res_ln(const float *A, const float *W, const float *Wf, const float *bias, const float *res,
                      const float *gamma, const float *beta, float *out, int M, int N, int K,
                      const float *score_w, const float *score_b, int num_classes, int sigmoid,
                      float *scores, int bf16, hipStream_t st) {
    if (bf16) {          // bf16 matrix pipe (opt-in): d_model <= 256 only
        if (N > 256 || N % 32) return -1;
        int blocks = persistent_blocks((M + 127) / 128);
        if (blocks < 0) return (int)hipErrorInvalidDevice;
        if (blocks > (M + 127) / 128) blocks = (M + 127) / 128;
#define VSK_LNB_CASE(NT_)                                                                                   \
    case NT_:                                                                                               \
        if (bf16 == 2)                                                                                      \
            hipLaunchKernelGGL((gemm_ln_rows<NT_, 2>), dim3(blocks), dim3(256), 0, st, A, W, bias, res, gamma, \
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
        static const int attr_rc = allow_big_lds(skinny2_ln<NW_>);                                         \
        if (attr_rc) return attr_rc;                                                                       \
        hipLaunchKernelGGL(skinny2_ln<NW_>, dim3(blocks), dim3(64 * NW_), skinny2_lds(K), st, A, Wf, bias, res, \
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
#define VSK_LN_CASE(NB_)                                                                             \
    case NB_:                                                                                        \
        hipLaunchKernelGGL(gemm_res_ln<NB_>, dim3(blocks), dim3(256), 0, st, A, W, bias, res, gamma, \
                           beta, out, M, K, score_w, score_b, num_classes, sigmoid, scores);         \
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
