// valu_probe.hip — issue cost of the softmax ingredients beside the bf16 matrix pipe (round 3: what bounds the
// low-precision attention at head dim 64, where every 16-key x 32-query MFMA comes with 1.33 v_exp_f32?).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_probe tools/valu_probe.hip ; run: ./valu_probe
// Every variant runs ITER iterations of a fixed instruction mix per wave, on 1 or 2 waves per SIMD (one 256- or
// 512-thread block per CU); reported: shader cycles (s_memtime, median over waves) per iteration and per SIMD.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const f32x2 f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}

// VAR: 0 = 16 exp            1 = 16 exp + 8 cvt_pk (dependent)   2 = 8 MFMA alone
//      3 = 8 MFMA, each followed by 2 exp + 1 cvt     4 = 8 MFMA, each followed by 4 exp + 2 cvt
//      5 = 32 v_add_f32      6 = 8 MFMA, each followed by 4 v_add     7 = 8 MFMA each followed by 8 v_add
//      8 = 4 exp + 2 cvt per MFMA, but the cvt of pair k is issued one gap later (no trans -> VALU dependency)
template <int VAR>
__global__ void probe(const float *__restrict__ src, float *__restrict__ dst, unsigned long long *__restrict__ cyc, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = src[tid + 64 * i] * 1e-3f;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[i][t] = 0.f;
    u32x4 a = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u}, b = a;
    unsigned pk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float e[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) e[i] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (VAR == 0 || VAR == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) e[i] = __builtin_amdgcn_exp2f(x[i]);
            if constexpr (VAR == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) pk[i] ^= pack_bf16(e[2 * i], e[2 * i + 1]);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = e[i] * 1e-9f;
            }
        } else if constexpr (VAR == 5) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] += 1.0f;
        } else {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                acc[m & 1] = MFMA_BF16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[m & 1]);
                if constexpr (VAR == 3) {
                    e[2 * m] = __builtin_amdgcn_exp2f(x[2 * m]); e[2 * m + 1] = __builtin_amdgcn_exp2f(x[2 * m + 1]);
                    pk[m] ^= pack_bf16(e[2 * m], e[2 * m + 1]);
                }
                if constexpr (VAR == 4) {
                    const int i0 = (4 * m) & 15;
                    e[i0] = __builtin_amdgcn_exp2f(x[i0]); e[i0 + 1] = __builtin_amdgcn_exp2f(x[i0 + 1]);
                    e[i0 + 2] = __builtin_amdgcn_exp2f(x[i0 + 2]); e[i0 + 3] = __builtin_amdgcn_exp2f(x[i0 + 3]);
                    pk[m] ^= pack_bf16(e[i0], e[i0 + 1]);
                    pk[(m + 4) & 7] ^= pack_bf16(e[i0 + 2], e[i0 + 3]);
                }
                if constexpr (VAR == 8) {
                    const int i0 = (4 * m) & 15, j0 = (4 * (m + 7)) & 15;      // the pair exponentiated one gap earlier
                    pk[m] ^= pack_bf16(e[j0], e[j0 + 1]);
                    pk[(m + 4) & 7] ^= pack_bf16(e[j0 + 2], e[j0 + 3]);
                    e[i0] = __builtin_amdgcn_exp2f(x[i0]); e[i0 + 1] = __builtin_amdgcn_exp2f(x[i0 + 1]);
                    e[i0 + 2] = __builtin_amdgcn_exp2f(x[i0 + 2]); e[i0 + 3] = __builtin_amdgcn_exp2f(x[i0 + 3]);
                }
                if constexpr (VAR == 6 || VAR == 7) {
#pragma unroll
                    for (int i = 0; i < (VAR == 6 ? 4 : 8); ++i) x[(8 * m + i) & 15] += 1.0f;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += x[i] + e[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += (float)pk[i];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 16; ++t) sum += acc[i][t];
    dst[(size_t)blockIdx.x * blockDim.x + tid] = sum;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int VAR>
void run(const char *name, const float *src, float *dst, unsigned long long *cyc, int iters, int mfma, int exps, int others) {
    for (int wps = 1; wps <= 2; ++wps) {
        const int threads = 256 * wps, grid = 256;
        hipMemset(cyc, 0, grid * 8 * sizeof(unsigned long long));
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe<VAR>, dim3(grid), dim3(threads), 0, 0, src, dst, cyc, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid * 8);
        hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<double> v;
        for (int bI = 0; bI < grid; ++bI)
            for (int w = 0; w < 4 * wps; ++w) v.push_back((double)h[bI * 8 + w] / iters);
        std::sort(v.begin(), v.end());
        const double med = v[v.size() / 2];
        printf("%-58s %d wave(s)/SIMD: %7.1f cycles per iteration and wave  -> per SIMD %7.1f cycles for %2d MFMA + %2d exp + %2d other VALU",
               name, wps, med, med, mfma * wps, exps * wps, others * wps);
        if (mfma) printf("  (MFMA pipe busy %.2f)", 32.0 * mfma * wps / med);
        printf("\n");
    }
}

// The instruction mix of one 64-key tile of the bf16 attention (head dim 64, 32 queries per wave), registers only:
//   phase A: 10 MFMAs (2 x (bias + 4 products)) into the NEXT tile's two S accumulators, with the 32 exp2 + 16 cvt_pk
//            of the CURRENT tile's S between them (4 exp + 2 cvt per gap in the first 8 gaps) and an 8-instruction OR tree;
//   phase B: 14 MFMAs (8 P.V + 4 row sums + 2 spare) that read the packed P.
// MODE 0: both phases back to back in every wave; MODE 1: phase A only; MODE 2: phase B only;
// MODE 3: waves 4..7 start with phase B (the staggered schedule, free-running: no barriers).
template <int MODE>
__global__ void tile_mix(const float *__restrict__ src, float *__restrict__ dst, unsigned long long *__restrict__ cyc, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x16 sA[2], sB[2], o[3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 16; ++t) { sA[i][t] = src[tid + t] * 1e-3f; sB[i][t] = src[tid + 16 + t] * 1e-3f; }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int t = 0; t < 16; ++t) o[i][t] = 0.f;
    u32x4 kf = {0x3A803A80u, 0x3A803A80u, 0x3A803A80u, 0x3A803A80u}, qf = kf;        // small operands: S stays bounded
    u32x4 pf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pf[i] = kf;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto phaseA = [&](f32x16 (&sin)[2], f32x16 (&sout)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 10; ++m) {
            const int n = m / 5;
            if (m % 5 == 0) sout[n] = MFMA_BF16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf), zero);
            else sout[n] = MFMA_BF16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf), sout[n]);
            if (m < 8) {
                const int nn = m / 4, j0 = 4 * (m % 4);
                const float e0 = __builtin_amdgcn_exp2f(sin[nn][j0]), e1 = __builtin_amdgcn_exp2f(sin[nn][j0 + 1]);
                const float e2 = __builtin_amdgcn_exp2f(sin[nn][j0 + 2]), e3 = __builtin_amdgcn_exp2f(sin[nn][j0 + 3]);
                pf[m / 2][2 * (m & 1)] = pack_bf16(e0, e1);
                pf[m / 2][2 * (m & 1) + 1] = pack_bf16(e2, e3);
            }
            if (m == 9) {
                unsigned acc = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) acc |= (pf[i][0] | pf[i][1]) | (pf[i][2] | pf[i][3]);
                if (__builtin_expect(__any((acc & 0x40004000u) != 0u), 0)) sin[0][0] += 1.0f;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto phaseB = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 14; ++m) {
            o[m % 3] = MFMA_BF16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, pf[m & 3]), o[m % 3]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const bool late = MODE == 3 && __builtin_amdgcn_readfirstlane(wave) >= 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (late) phaseB();
    for (int it = 0; it < iters; it += 2) {
        if (MODE != 2) phaseA(sA, sB);
        if (MODE != 1) phaseB();
        if (MODE != 2) phaseA(sB, sA);
        if (MODE != 1) phaseB();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int t = 0; t < 16; ++t) sum += o[i][t];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 16; ++t) sum += sA[i][t] + sB[i][t];
    dst[(size_t)blockIdx.x * blockDim.x + tid] = sum;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
void run_tile(const char *name, const float *src, float *dst, unsigned long long *cyc, int iters) {
    for (int wps = 1; wps <= 2; ++wps) {
        if (MODE == 3 && wps == 1) continue;
        const int threads = 256 * wps, grid = 256;
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(tile_mix<MODE>, dim3(grid), dim3(threads), 0, 0, src, dst, cyc, iters);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid * 8);
        (void)hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<double> v;
        for (int bI = 0; bI < grid; ++bI)
            for (int w = 0; w < 4 * wps; ++w) v.push_back((double)h[bI * 8 + w] / iters);
        std::sort(v.begin(), v.end());
        const double med = v[v.size() / 2];
        const int nm = MODE == 1 ? 10 : MODE == 2 ? 14 : 24;
        printf("%-64s %d wave(s)/SIMD: %7.1f cycles per tile and wave; MFMA pipe busy %.2f (useful 20 of 24: %.2f)\n", name, wps, med,
               32.0 * nm * wps / med, MODE == 0 || MODE == 3 ? 32.0 * 20 * wps / med : 0.0);
    }
}

int main() {
    float *src, *dst;
    unsigned long long *cyc;
    hipMalloc(&src, 1 << 20); hipMalloc(&dst, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    std::vector<float> h(1 << 18);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(rand() % 1000) / 1000.f;
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int it = 2000;
    run<0>("16 v_exp_f32 (+16 v_mul)", src, dst, cyc, it, 0, 16, 16);
    run<1>("16 v_exp_f32 + 8 v_cvt_pk_bf16_f32 (+8 xor)", src, dst, cyc, it, 0, 16, 16);
    run<5>("32 v_add_f32", src, dst, cyc, it, 0, 0, 32);
    run<2>("8 MFMA 32x32x16 bf16", src, dst, cyc, it, 8, 0, 0);
    run<3>("8 x (MFMA, 2 exp, 1 cvt_pk + xor)", src, dst, cyc, it, 8, 16, 16);
    run<4>("8 x (MFMA, 4 exp, 2 cvt_pk + 2 xor)", src, dst, cyc, it, 8, 32, 32);
    run<8>("8 x (MFMA, 2 cvt_pk of the previous gap's exps, 4 exp)", src, dst, cyc, it, 8, 32, 32);
    run<6>("8 x (MFMA, 4 v_add)", src, dst, cyc, it, 8, 0, 32);
    run<7>("8 x (MFMA, 8 v_add)", src, dst, cyc, it, 8, 0, 64);
    run_tile<1>("tile mix, phase A only (10 MFMA + 32 exp + 16 cvt + OR tree)", src, dst, cyc, it);
    run_tile<2>("tile mix, phase B only (14 MFMA)", src, dst, cyc, it);
    run_tile<0>("tile mix, A then B in every wave", src, dst, cyc, it);
    run_tile<3>("tile mix, waves 4..7 half a tile behind (free-running)", src, dst, cyc, it);
    return 0;
}
