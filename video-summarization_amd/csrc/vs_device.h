// vs_device.h — device-side helpers shared by the kernel translation units (vs_kernels.hip: Linear kernels,
// vs_attention.hip: attention kernels).  Everything here lives in an anonymous namespace: each TU gets its own copy.
//
// Operand convention used by all kernels (lane l, r = l & 31, h = l >> 5):
//   A operand of 32x32x2: A[i = r][k = h]     B operand: B[k = h][j = r]
//   accumulator reg t (0..15): C[row = (t&3) + 8*(t>>2) + 4*h][col = r]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

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
// Bit pattern of a float / float of a bit pattern, through a by-value parameter ON PURPOSE: this hipcc (ROCm 7.2) compiles
// `__builtin_bit_cast(unsigned, vec[k])` on an ext-vector ELEMENT as if k were 0 (seen as `load <1 x i32>` + splat in the
// IR: the second dword of an 8-byte load was never fetched).  A scalar copy first is compiled correctly.
__device__ __forceinline__ unsigned f32_bits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bits_f32(unsigned x) { return __builtin_bit_cast(float, x); }
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
// ---- ONE 16-bit operand type per kernel instantiation: bf16 (F16 = 0) or IEEE f16 (F16 = 1; round 4: the training path's
// "fp16" mode - the reference's own autocast type, train.py:120 - 11 significant bits against bf16's 8, range 65 504:
// an overflowing operand becomes inf and travels to the gradients, which is what the reference's GradScaler looks for) ----
__device__ __forceinline__ unsigned pack_f16(float lo, float hi) {
    const f32x2 f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, f16x2));
}
template <int F16>
__device__ __forceinline__ unsigned pack_lp(float lo, float hi) {
    if constexpr (F16) return pack_f16(lo, hi);
    else return pack_bf16(lo, hi);
}
template <int F16>
__device__ __forceinline__ f32x16 mfma_lp(const u32x4 &a, const u32x4 &b, const f32x16 &c) {
    if constexpr (F16) return MFMA_F16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c);
    else return MFMA_BF16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c);
}
// the two values of a packed pair as floats
template <int F16>
__device__ __forceinline__ f32x2 unpack_lp(unsigned u) {
    if constexpr (F16) return __builtin_convertvector(__builtin_bit_cast(f16x2, u), f32x2);
    else return f32x2{bits_f32(u << 16), bits_f32(u & 0xffff0000u)};
}

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

}  // namespace

// Diagnostic stamps (cdna guide section 7 "In-kernel stamps"): used only by the DIAG instantiations of the diagnostic
// library (-DVS_WITH_DIAG, tools/); no product launch executes a stamp.  The wait also drains the wave's LDS reads.
namespace {
__device__ __forceinline__ unsigned long long vs_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
}  // namespace

#define VSK_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
