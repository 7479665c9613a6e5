// mfma_probe.hip — what does each ingredient of the GEMM/attention loops cost the fp32 MFMA pipe?
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe tools/mfma_probe.hip ; run: ./mfma_probe
// Each variant issues the same number of v_mfma_f32_32x32x2_f32 per wave; utilisation =
// 64 cycles * MFMAs per SIMD / elapsed shader cycles (s_memtime, median over waves).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

enum { V_PURE4 = 0, V_PURE1 = 1, V_LDSREAD = 2, V_BARRIER = 3, V_STAGE = 4, V_STAGE_NOBAR = 5 };

template <int VAR>
__global__ __launch_bounds__(256, 2) void probe(const float *__restrict__ src, float *__restrict__ dst,
                                                 unsigned long long *__restrict__ cyc, int iters) {
    constexpr int LD = 36;
    __shared__ __attribute__((aligned(16))) float smem[2 * 256 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < 2 * 256 * LD; i += 256) smem[i] = src[i & 4095];
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int t = 0; t < 16; ++t) acc[i][t] = 0.f;
    float a = src[tid], b = src[tid + 256];
    const float *gp = src + (size_t)(blockIdx.x % 64) * 8192 + tid * 4;
    f32x4 pa[4], pw[4];
    for (int i = 0; i < 4; ++i) { pa[i] = *(const f32x4 *)(gp + 1024 * i); pw[i] = *(const f32x4 *)(gp + 4096 + 1024 * i); }
    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;

    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {        // 64 MFMAs per iteration
        if (VAR == V_PURE4) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                acc[0] = MFMA32(a, b, acc[0]); acc[1] = MFMA32(a, b, acc[1]);
                acc[2] = MFMA32(a, b, acc[2]); acc[3] = MFMA32(a, b, acc[3]);
            }
        } else if (VAR == V_PURE1) {
#pragma unroll
            for (int s = 0; s < 64; ++s) acc[0] = MFMA32(a, b, acc[0]);
        } else {
            const float *As = smem + (it & 1) * 256 * LD, *Ws = As + 128 * LD;
            const float *ap = As + (64 * (wave >> 1) + r) * LD + 4 * h, *wp = Ws + (64 * (wave & 1) + r) * LD + 4 * h;
            if (VAR == V_STAGE || VAR == V_STAGE_NOBAR) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { pa[i] = *(const f32x4 *)(gp + 1024 * i + (it & 7) * 4); pw[i] = *(const f32x4 *)(gp + 4096 + 1024 * i + (it & 7) * 4); }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 a0 = *(const f32x4 *)(ap + 8 * g), a1 = *(const f32x4 *)(ap + 32 * LD + 8 * g);
                f32x4 b0 = *(const f32x4 *)(wp + 8 * g), b1 = *(const f32x4 *)(wp + 32 * LD + 8 * g);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc[0] = MFMA32(b0[s], a0[s], acc[0]); acc[1] = MFMA32(b1[s], a0[s], acc[1]);
                    acc[2] = MFMA32(b0[s], a1[s], acc[2]); acc[3] = MFMA32(b1[s], a1[s], acc[3]);
                }
            }
            if (VAR == V_STAGE || VAR == V_STAGE_NOBAR) {
                float *An = smem + ((it + 1) & 1) * 256 * LD, *Wn = An + 128 * LD;
#pragma unroll
                for (int i = 0; i < 4; ++i) { *(f32x4 *)&An[(lrow + 32 * i) * LD + lc4] = pa[i]; *(f32x4 *)&Wn[(lrow + 32 * i) * LD + lc4] = pw[i]; }
            }
            if (VAR == V_BARRIER || VAR == V_STAGE) __syncthreads();
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int t = 0; t < 16; ++t) sum += acc[i][t];
    dst[(size_t)blockIdx.x * 256 + tid] = sum;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int VAR>
void run(const char *name, int grid, const float *src, float *dst, unsigned long long *cyc, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = getenv("PROBE_REPS") ? atoi(getenv("PROBE_REPS")) : 1;   // sustained-load mode: many launches
    for (int i = 0; i < reps; ++i) probe<VAR><<<grid, 256>>>(src, dst, cyc, iters);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) probe<VAR><<<grid, 256>>>(src, dst, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::vector<unsigned long long> c(grid * 4);
    hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double med = (double)c[c.size() / 2];
    const int waves_per_simd = grid / 256;                       // 256 CUs, 4 waves per block = 1 per SIMD
    const double util = 64.0 * 64.0 * iters * (waves_per_simd > 0 ? waves_per_simd : 1) / med;
    const double tf = 4096.0 * 64.0 * iters * grid * 4 / (ms * 1e-3) / 1e12;
    printf("%-14s grid %4d: %.3f ms  %6.1f TFLOP/s  med cycles/wave %.0f  (%.1f cyc per MFMA issued on the SIMD)  pipe util %.3f  clock %.2f GHz\n",
           name, grid, ms, tf, med, med / (64.0 * iters * std::max(1, waves_per_simd)), util, med / (ms * 1e6));
}

int main() {
    float *src, *dst; unsigned long long *cyc;
    hipMalloc(&src, 64 * 8192 * 4 + 65536); hipMalloc(&dst, 1024 * 256 * 4); hipMalloc(&cyc, 4096 * 8);
    std::vector<float> hsrc(64 * 8192 + 16384);
    for (size_t i = 0; i < hsrc.size(); ++i) hsrc[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(src, hsrc.data(), hsrc.size() * 4, hipMemcpyHostToDevice);
    const int iters = 512;
    for (int grid : {256, 512}) {
        run<V_PURE4>("pure4", grid, src, dst, cyc, iters);
        run<V_PURE1>("pure1", grid, src, dst, cyc, iters);
        run<V_LDSREAD>("ldsread", grid, src, dst, cyc, iters);
        run<V_BARRIER>("ldsread+bar", grid, src, dst, cyc, iters);
        run<V_STAGE_NOBAR>("stage-nobar", grid, src, dst, cyc, iters);
        run<V_STAGE>("stage+bar", grid, src, dst, cyc, iters);
    }
    return 0;
}
