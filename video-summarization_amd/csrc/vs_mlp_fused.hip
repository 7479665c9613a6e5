// vs_mlp_fused.hip — the tail of one encoder layer as ONE kernel on the bf16 matrix pipe (opt-in bf16 mode,
// d_model = 256, hidden = 1024).  The core is the MLP block
//     out = LayerNorm( relu(h1 W1^T + b1) W2^T + b2 + h1 ) * gamma + beta      (+ score head)
// (reference simnet.py:109-110 EncoderBlock, 180-183 MLP, 42 final_layer); in front of it, when the attention output is
// stored as bf16, the out-projection + residual + norm1 that produces h1 (simnet.py:107, 163; "TAIL"), and behind it the
// NEXT layer's q / k / v projection of the rows just produced (simnet.py:148-153; "QKV epilogue").  One launch per layer
// besides the attention; h1, the hidden activations and the re-read of the layer output never touch HBM.
//
// Why: as separate kernels every stage is bound by its HBM round trip, store bursts and load latency, not by the matrix
// pipe (stamped build: fc1 spends 55 % of its time in the store epilogue and 3 200 cycles per 32-k tile waiting for loads;
// MFMA busy 0.06-0.14): fc1 + fc2 took 209 us per layer at M = 65 536 against an MFMA floor of 27 us.
//
// Layout.  Block = 8 waves = 256 rows, one block per CU (or 4 waves = 128 rows when 256-row tiles would occupy at most
// half the CUs: template parameter NWV); wave w owns rows 32w .. 32w+31 (lane (r, h) = row r):
//   Y[8]   fp32 accumulators of the output row block, started at the residual h1 (C-in), 128 registers;
//   X[16]  the same h1 values rounded to bf16, as the B operands of fc1 (64 registers) - they come out of the
//          residual load for free: lane (r, h) holds columns 32j + 8q + 4h + e, and taking registers 8qq .. 8qq+7 of a
//          32-column block as ONE 16-k step only permutes k inside the step (position 8h + i <-> column
//          8(i>>2) + 4h + (i&3): bits 2 and 3 swapped);
//   per 32 hidden units:  U = W1[32 rows] X^T (16 MFMAs, one dependent chain: gfx950 issues it back to back),
//          ReLU + round to bf16 in place (the U accumulator registers 8qq .. 8qq+7 ARE the next B operand, the same
//          permutation), Y += W2[:, 32 columns] U^T (16 MFMAs).
//   The weights are pre-packed once per vs_weights_pack/update (vsk_pack_mlp_bf16 / vsk_pack_qkv_bf16) into the exact LDS
//   IMAGE of every chunk: bf16, the permutation applied where the B operand is permuted, rows padded (528 / 144 / 80 B:
//   conflict-free ds_read_b128), 40 KiB per chunk.  Staging is therefore LDS-DMA (global_load_lds_dwordx4: a wave
//   instruction copies 1 KiB verbatim, no staging registers, no LDS-write instructions): every wave issues 5 pieces per
//   chunk, TWO chunks ahead, into a ring of three LDS buffers; a counted s_waitcnt vmcnt(5) + one raw s_barrier per chunk
//   retire the chunk needed next and leave the one after it in flight.  That count is only right if nothing else sits on
//   the vector-memory counter inside a chunk: tests/test_host.py checks the ISA of every chunk at build time (and lane-
//   derived addresses are recomputed per tile from an opaque copy of the thread index - left to the compiler they are
//   hoisted to kernel entry, ~100 registers of them, and spilled around the loops).  Per MFMA: one 1-KiB fragment read per
//   wave and nothing else.
// Numerics.  Rounding points equal those of the separate kernels (activations, weights and relu(fc1) to bf16; fp32
// everything else).  The out-projection and the QKV epilogue multiply in the k order of the kernels they replace:
// bit-identical.  The MLP part permutes k inside an MFMA step: equal to fp32 rounding, not bitwise.
// Measured (MI355X, M = 65 536): MLP block alone 79 us (two kernels: 209), of which 25 us are the exposed prologue +
// epilogue of the single tile each CU owns; + out-projection/norm1 93 us (was 44 + 79); + next QKV 116-120 us (was + 66).
// Tried and dropped: 4 waves x 64 rows with the whole 512-register file per wave (one fragment read feeds two MFMAs):
// correct, but as hipcc compiles it (accumulators in AGPRs, 96 v_accvgpr moves per chunk) 291 us against 82, MFMA busy
// 0.10, with no waiting on any counter - cause not found.  It is not the occupancy: this file's own code at one wave per
// SIMD (NWV = 4, two 128-row tiles per CU) takes 106 us against 79.
#include "vs_device.h"
#include "vs_kernels.h"

#include <atomic>

namespace {

typedef unsigned short h16;

constexpr int MLP_D = 256, MLP_HID = 1024, MLP_CH = 32, MLP_NCH = MLP_HID / MLP_CH;
constexpr int MLP_LD1 = 528;                    // bytes per W1 row in the image (256 bf16 + 16: stride = 4 mod 64 dwords)
constexpr int MLP_LD2 = 80;                     // bytes per W2 row (32 bf16 + 16: 20 dwords, conflict-free b128)
constexpr int MLP_W2OFF = 17408;                // W1 part: 32 * 528 = 16896, rounded up to a 1-KiB piece boundary
constexpr int MLP_IMG = 40960;                  // W2 part: 256 * 80 = 20480 -> 37 pieces, padded to 40 (5 per wave)
// the out-projection in front (TAIL kernels): Wo [256][256] as 4 chunks of 64 k, rows of 64 bf16 + 16 B (144 B = 36 dwords:
// conflict-free b128), 256 * 144 = 36 864 B = 36 pieces, padded to the same 40-piece chunk
constexpr int MLP_LDO = 144, MLP_NCHO = 4;
constexpr int MLP_CHUNKS = MLP_NCHO + MLP_NCH;  // image = [4 Wo chunks][32 MLP chunks]
// the NEXT layer's QKV projection behind (QKV epilogue): Wqkv [768][256] as 12 chunks of 64 output columns, rows of 256 bf16
// + 16 B like W1 (64 * 528 = 33 792 B = 33 pieces, padded to 40), natural k order; its own image
constexpr int MLP_NCHQ = 12;

__device__ __forceinline__ int swap23(int p) { return (p & 3) | ((p & 4) << 1) | ((p & 8) >> 1); }

// Wo [256][256], W1 [1024][256], W2 [256][1024] fp32 -> img [36 chunks][MLP_IMG bytes] (pad bytes stay as zeroed)
__global__ void pack_mlp_bf16(const float *__restrict__ Wo, const float *__restrict__ W1, const float *__restrict__ W2,
                              unsigned char *__restrict__ img) {
    const int n = MLP_D * MLP_HID / 2;          // bf16 pairs per MLP matrix
    unsigned char *mlp = img + (size_t)MLP_NCHO * MLP_IMG;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        {   // W1: pair at row = hidden unit, position p (even) of the permuted k order
            const int row = (2 * i) / MLP_D, p = (2 * i) % MLP_D, b16 = p & ~15;
            const float *src = W1 + (size_t)row * MLP_D + b16;
            unsigned *dst = (unsigned *)(mlp + (size_t)(row / MLP_CH) * MLP_IMG + (row % MLP_CH) * MLP_LD1 + 2 * p);
            *dst = pack_bf16(src[swap23(p & 15)], src[swap23((p & 15) + 1)]);
        }
        {   // W2: pair at row = output column, hidden unit 32c + position p32 (even) of the permuted order
            const int row = (2 * i) / MLP_HID, hcol = (2 * i) % MLP_HID, c = hcol / MLP_CH, p32 = hcol % MLP_CH;
            const float *src = W2 + (size_t)row * MLP_HID + MLP_CH * c + (p32 & 16);
            unsigned *dst = (unsigned *)(mlp + (size_t)c * MLP_IMG + MLP_W2OFF + row * MLP_LD2 + 2 * p32);
            *dst = pack_bf16(src[swap23(p32 & 15)], src[swap23((p32 & 15) + 1)]);
        }
        if (i < MLP_D * MLP_D / 2) {   // Wo: natural k order (its B operand is read from HBM in that order), 64 k per chunk
            const int row = (2 * i) / MLP_D, k = (2 * i) % MLP_D;
            unsigned *dst = (unsigned *)(img + (size_t)(k / 64) * MLP_IMG + row * MLP_LDO + 2 * (k % 64));
            *dst = pack_bf16(Wo[(size_t)row * MLP_D + k], Wo[(size_t)row * MLP_D + k + 1]);
        }
    }
}

// Wqkv [768][256] fp32 -> img [12 chunks][MLP_IMG bytes]
__global__ void pack_qkv_bf16(const float *__restrict__ Wqkv, unsigned char *__restrict__ img) {
    const int n = 3 * MLP_D * MLP_D / 2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int row = (2 * i) / MLP_D, k = (2 * i) % MLP_D;
        unsigned *dst = (unsigned *)(img + (size_t)(row / 64) * MLP_IMG + (row % 64) * MLP_LD1 + 2 * k);
        *dst = pack_bf16(Wqkv[(size_t)row * MLP_D + k], Wqkv[(size_t)row * MLP_D + k + 1]);
    }
}

// one wave instruction: lane l copies 16 bytes from its `g` to lds_wave_base + 16 l (asynchronous, counted by vmcnt)
__device__ __forceinline__ void glds16(const void *g, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

struct TailArgs {            // TAIL kernels: the out-projection + norm1 in front of the MLP block
    const h16 *att;          // [M, 256] bf16 attention output
    const float *res;        // [M, 256] fp32 residual (the layer input)
    const float *bo, *gamma1, *beta1;
};

struct QkvArgs {             // QKV epilogue: the next layer's q / k / v projection of the rows this block just produced
    const unsigned char *img;    // vsk_pack_qkv_bf16 image of the NEXT layer's Wqkv, or nullptr (no epilogue)
    const float *bqkv;           // [768]
    h16 *qkv;                    // out: three [B, H, T, dh] bf16 planes of M * 256 elements (q pre-multiplied by qscale)
    int T, H, dh;
    float qscale;
};

// LDS ring + the LDS-DMA copy of one 40-KiB chunk image into it (40 pieces of 1 KiB, 5 per wave): source = wave-uniform
// base (scalar registers) + one per-lane byte offset, no per-piece address registers
template <int NWV>              // waves per block: 8 (5 pieces per wave and chunk) or 4 (10)
struct RingDma {
    static constexpr int PIECES = 40 / NWV;
    unsigned char *ring;
    int wave_u;
    unsigned lane16;
    __device__ __forceinline__ void dma(const unsigned char *image, int chunk, int bufoff) const {
        const unsigned char *src = image + (size_t)chunk * MLP_IMG + wave_u * 1024;
        unsigned l16 = lane16;
        asm volatile("" : "+v"(l16));           // (opaque: else per-piece per-lane 64-bit pointers are precomputed and spilled)
#pragma unroll
        for (int i = 0; i < PIECES; ++i) glds16(src + 1024 * NWV * i + l16, ring + bufoff + wave_u * 1024 + 1024 * NWV * i);
    }
};
// the counted wait that leaves exactly the newest N vector-memory operations in flight (N a compile-time constant)
template <int N>
__device__ __forceinline__ void wait_vmcnt_le() {
    static_assert(N == 0 || N == 5 || N == 10 || N == 13 || N == 18, "counts this file uses");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
}

// QKV epilogue shared by the layer-tail and the embedding kernels (see the header of mlp_fused_bf16): Y = this wave's 32
// finished rows (fp32, lane (er, eh) = row er, columns 32j + 8q + 4eh + e), X = scratch for their bf16 B operands.
template <int FR, int NWV>
__device__ __forceinline__ void qkv_epilogue(f32x16 (&Y)[8], u32x4 (&X)[16], const QkvArgs &qa, const RingDma<NWV> &rd,
                                             const float *bqkv_s, int m0, int er, int eh, int M) {
    constexpr int NT = 8, D = MLP_D;
    // X[ks] = this lane's 8 natural-order k of the normalised row: k-step ks = 2j + qq covers columns 32j + 16qq ..
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const unsigned e0 = pack_bf16(Y[j][8 * qq + 0], Y[j][8 * qq + 1]), e1 = pack_bf16(Y[j][8 * qq + 2], Y[j][8 * qq + 3]);
            const unsigned o0 = pack_bf16(Y[j][8 * qq + 4], Y[j][8 * qq + 5]), o1 = pack_bf16(Y[j][8 * qq + 6], Y[j][8 * qq + 7]);
            auto s0 = __builtin_amdgcn_permlane32_swap(e0, o0, false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(e1, o1, false, false);
            X[2 * j + qq][0] = s0[0]; X[2 * j + qq][1] = s1[0]; X[2 * j + qq][2] = s0[1]; X[2 * j + qq][3] = s1[1];
        }
    __syncthreads();                    // every wave is done with its transposition corner of the ring
    rd.dma(qa.img, 0, 0);
    rd.dma(qa.img, 1, MLP_IMG);
    const int row = m0 + er;
    const int vb = row / qa.T, vt = row - vb * qa.T;      // (video, frame) of this lane's row
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int qcur = 0, qn1 = MLP_IMG, qn2 = 2 * MLP_IMG;
    u32x4 pend[4];                      // the previous chunk's packed results, stored after its barrier
    for (int c = 0; c <= MLP_NCHQ; ++c) {
        if (c > 0) {                    // stores of chunk c - 1: 16 bytes per lane = 8 consecutive columns of its row
            const int n0 = 64 * (c - 1), which = n0 >> 8;
            if (row < M) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        const int cd = (n0 & 255) + 32 * t + 16 * pp + 8 * eh, head = cd / qa.dh, e = cd - head * qa.dh;
                        h16 *dst = qa.qkv + (size_t)which * M * D + (((size_t)vb * qa.H + head) * qa.T + vt) * qa.dh + e;
                        *(u32x4 *)dst = pend[2 * t + pp];
                    }
            }
            if (c == MLP_NCHQ) break;
        }
        rd.dma(qa.img, c + 2 < MLP_NCHQ ? c + 2 : MLP_NCHQ - 1, qn2);      // past the end: a harmless re-copy
        const unsigned char *wbase = rd.ring + qcur + er * MLP_LD1 + 16 * eh;
        u32x4 fw[FR];
        auto frag = [&](auto fc) __attribute__((always_inline)) {       // fragment f: k-step f / 2, column block f % 2
            constexpr int f = decltype(fc)::value;
            fw[f % FR] = *(const u32x4 *)(wbase + 32 * (f % 2) * MLP_LD1 + 32 * (f / 2));
        };
        f32x16 U[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float *bp = bqkv_s + 64 * c + 32 * t + 4 * eh;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *(const f32x4 *)(bp + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) U[t][4 * q + e] = bv[e];
            }
        }
        static_for<FR - 1>([&](auto fc) { frag(fc); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<32>([&](auto fc) {
            constexpr int f = decltype(fc)::value;
            if constexpr (f + FR - 1 < 32) frag(std::integral_constant<int, f + FR - 1>{});
            U[f % 2] = MFMA_BF16(__builtin_bit_cast(bf16x8, fw[f % FR]), __builtin_bit_cast(bf16x8, X[f / 2]), U[f % 2]);
            __builtin_amdgcn_sched_barrier(0);
        });
        const float osc = c < 4 ? qa.qscale : 1.0f;       // chunks 0..3 are q
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                // registers 8pp .. 8pp+7 of a tile: q = 2pp (even group), 2pp + 1 (odd group)
                const unsigned e0 = pack_bf16(U[t][8 * pp + 0] * osc, U[t][8 * pp + 1] * osc), e1 = pack_bf16(U[t][8 * pp + 2] * osc, U[t][8 * pp + 3] * osc);
                const unsigned o0 = pack_bf16(U[t][8 * pp + 4] * osc, U[t][8 * pp + 5] * osc), o1 = pack_bf16(U[t][8 * pp + 6] * osc, U[t][8 * pp + 7] * osc);
                auto s0 = __builtin_amdgcn_permlane32_swap(e0, o0, false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(e1, o1, false, false);
                pend[2 * t + pp][0] = s0[0]; pend[2 * t + pp][1] = s1[0]; pend[2 * t + pp][2] = s0[1]; pend[2 * t + pp][3] = s1[1];
            }
        // <= 5 operations outstanding: loads retire in order among themselves, so chunk c+1's pieces (older than the 5
        // of chunk c+2) have landed whatever the stores issued at the top of this iteration are doing
        wait_vmcnt_le<40 / NWV>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int t = qcur; qcur = qn1; qn1 = qn2; qn2 = t;
    }
}

// TAIL: the kernel starts one step earlier in the encoder layer: H1 is not read but computed,
//     h1 = LayerNorm(att Wo^T + bo + res) * gamma1 + beta1          (reference simnet.py:107 norm1(x + sa(x)))
// by four more chunks through the same ring (Wo in 64-k slabs; the bf16 attention rows are the B operands straight from HBM,
// the residual is the C-in of Y, bias and LayerNorm in registers exactly as gemm_ln_rows does them), and the
// result is at once the X / Y input of the MLP part: h1 never exists in HBM, one launch and one prologue fewer per layer.
// QKV epilogue (qa.img != nullptr; every layer but the last): the rows this block has just normalised are the next
// layer's input, so its q / k / v projection runs here, on registers: X = bf16(out rows) brought into natural k order by
// one v_permlane32_swap per dword (lanes l and l^32 hold the two halves of every 8-column group), 12 more chunks of
// 64 output columns through the ring, two 16-MFMA chains per chunk in the k order of the stand-alone QKV kernel
// (gemm_nt_128<EPI_QKV, .., C16>: bit-identical q * qscale, k, v), and the results leave as 16-byte pieces per lane
// (again one permlane swap per dword) straight into the head-major bf16 planes - no LDS round trip.  A chunk's stores are
// issued after its barrier, at the top of the next iteration: the counted vmcnt(5) at that iteration's end still proves
// that the chunk needed next has landed (loads retire in order among loads, so "<= 5 outstanding" cannot hold while a
// piece older than the newest five is pending, whatever the stores do); at worst it also waits for those stores, which
// are a whole chunk old by then.
// ABL (diagnostic library only, timing runs with wrong results): 1 no weight staging, 2 no barrier / wait in the chunk
// loop, 4 fragment reads replaced by register moves, 8 no chunk loop at all (prologue + epilogue only)
// NWV = waves per block: 8 (256-row tiles, two waves per SIMD: the throughput form) or 4 (128-row tiles, one wave per SIMD,
// the same per-wave code and register budget - launch_bounds stays 512: 74 % of the two-wave throughput per CU, but twice
// as many blocks, for row counts that would leave half the chip idle with 256-row tiles)
template <bool TAIL, int NWV, int ABL = 0>
__global__ __launch_bounds__(512, 1) void mlp_fused_bf16(
    const float *H1, TailArgs ta, QkvArgs qa, const unsigned char *__restrict__ Wimg, const float *__restrict__ b1,
    const float *__restrict__ b2, const float *__restrict__ gamma,
    const float *__restrict__ beta, float *out /* may be the residual buffer: every block rewrites only rows it has read */,
    int M, const float *__restrict__ score_w, const float *__restrict__ score_b, int num_classes,
    int sigmoid, float *__restrict__ scores) {
    constexpr int D = MLP_D, HID = MLP_HID, NT = 8, IMG = MLP_IMG, NTHR = 64 * NWV, ROWS = 32 * NWV;
    constexpr int C0 = TAIL ? 0 : MLP_NCHO;                      // first chunk of the image this kernel consumes
    extern __shared__ __attribute__((aligned(1024))) unsigned char dyn_smem[];      // ONE LDS object (see the header)
    unsigned char *ring = dyn_smem;                              // [3][IMG]
    float *b1s = (float *)(dyn_smem + 3 * IMG);                  // [HID]
    float *gam_s = b1s + HID, *bet_s = gam_s + D, *sw_s = bet_s + D, *bias_s = sw_s + D;
    float *gam1_s = bias_s + D, *bet1_s = gam1_s + D, *bo_s = bet1_s + D;     // TAIL only
    float *bqkv_s = bo_s + D;                                    // [3 D], QKV epilogue only

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    for (int i = tid; i < HID; i += NTHR) b1s[i] = b1[i];
    for (int i = tid; i < D; i += NTHR) {
        gam_s[i] = gamma[i]; bet_s[i] = beta[i]; bias_s[i] = b2[i];
        if constexpr (TAIL) { gam1_s[i] = ta.gamma1[i]; bet1_s[i] = ta.beta1[i]; bo_s[i] = ta.bo[i]; }
    }
    if (qa.img != nullptr)
        for (int i = tid; i < 3 * D; i += NTHR) bqkv_s[i] = qa.bqkv[i];

    const RingDma<NWV> rd{ring, __builtin_amdgcn_readfirstlane(wave), (unsigned)lane * 16u};
    auto dma_chunk = [&](int chunk, int bufoff) __attribute__((always_inline)) { rd.dma(Wimg, chunk, bufoff); };
    // end of a chunk: the next chunk (5 pieces, issued one iteration ago) has landed, the one after it stays in flight;
    // its data is read only after the barrier every wave passes behind its own wait
    auto chunk_done = [&]() __attribute__((always_inline)) {
        if constexpr (!(ABL & 2)) {
            if constexpr (!(ABL & 1)) wait_vmcnt_le<40 / NWV>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    };
    constexpr int FR = 4;                                        // fragment ring: FR - 1 LDS reads in flight

    f32x16 Y[NT];
    u32x4 X[2 * NT];
    const int ntiles = (M + ROWS - 1) / ROWS;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * ROWS + 32 * wave;                   // this wave's rows: m0 + r
        __syncthreads();                        // the previous tile's epilogue is done with the ring
        if constexpr (!(ABL & 1)) { dma_chunk(C0, 0); dma_chunk(C0 + 1, IMG); }
        // (lane-derived addresses are recomputed per tile from an opaque copy of the thread index: left to the compiler
        // they are hoisted to kernel entry - ~100 registers of loop-invariant addresses - and spilled around the loop)
        int lp = tid;
        asm volatile("" : "+v"(lp));
        const int rp_r = lp & 31, rp_h = (lp >> 5) & 1;
        {
            int row = m0 + rp_r;
            row = row < M ? row : M - 1;
            const float *rp = (TAIL ? ta.res : H1) + (size_t)row * D + 4 * rp_h;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 rv = *(const f32x4 *)(rp + 32 * j + 8 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) Y[j][4 * q + e] = rv[e];
                    if constexpr (!TAIL) {
                        X[2 * j + (q >> 1)][2 * (q & 1)] = pack_bf16(rv[0], rv[1]);
                        X[2 * j + (q >> 1)][2 * (q & 1) + 1] = pack_bf16(rv[2], rv[3]);
                    }
                }
            if constexpr (TAIL) {               // B operands of the out-projection: k-step ks = att[row][16 ks + 8 h .. + 7]
                const h16 *ap = ta.att + (size_t)row * D + 8 * rp_h;
#pragma unroll
                for (int ks = 0; ks < 2 * NT; ++ks) X[ks] = *(const u32x4 *)(ap + 16 * ks);
            }
        }
        __syncthreads();                        // (waits for every outstanding load, the two DMA'd chunks included)

        int cur = 0, nx1 = IMG, nx2 = 2 * IMG;  // ring offsets of chunk c, c+1, c+2
        if constexpr (TAIL && !(ABL & 8)) {
            // ---- out-projection: 4 chunks of 64 k; fragment f: k-step f / 8 of the chunk, output block f % 8 ----
            static_for<MLP_NCHO>([&](auto cc) {                    // (unrolled: the B operands X[4c + ..] are registers)
                constexpr int c = decltype(cc)::value;
                if constexpr (!(ABL & 1)) dma_chunk(c + 2, nx2);
                const unsigned char *wbase = ring + cur + r * MLP_LDO + 16 * h;
                u32x4 fw[FR];
                auto frag = [&](auto fc) __attribute__((always_inline)) {
                    constexpr int f = decltype(fc)::value;
                    if constexpr ((ABL & 4) != 0) { fw[f % FR] = X[f % 16]; return; }
                    fw[f % FR] = *(const u32x4 *)(wbase + 32 * (f % 8) * MLP_LDO + 32 * (f / 8));
                };
                static_for<FR - 1>([&](auto fc) { frag(fc); });
                __builtin_amdgcn_sched_barrier(0);
                static_for<32>([&](auto fc) {
                    constexpr int f = decltype(fc)::value;
                    if constexpr (f + FR - 1 < 32) frag(std::integral_constant<int, f + FR - 1>{});
                    Y[f % 8] = MFMA_BF16(__builtin_bit_cast(bf16x8, fw[f % FR]), __builtin_bit_cast(bf16x8, X[4 * c + f / 8]), Y[f % 8]);
                    __builtin_amdgcn_sched_barrier(0);
                });
                chunk_done();
                const int t = cur; cur = nx1; nx1 = nx2; nx2 = t;
            });
            // ---- + bo, norm1 (the arithmetic of gemm_ln_rows' epilogue), then h1 is both X (bf16) and Y (residual) ----
            int l1 = tid;
            asm volatile("" : "+v"(l1));
            const int h = (l1 >> 5) & 1;        // (opaque copy: see the prologue)
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float pj = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bv = *(const f32x4 *)&bo_s[32 * j + 8 * q + 4 * h];
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
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 gv = *(const f32x4 *)&gam1_s[32 * j + 8 * q + 4 * h];
                    const f32x4 bv = *(const f32x4 *)&bet1_s[32 * j + 8 * q + 4 * h];
                    f32x4 y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { y[e] = Y[j][4 * q + e] * rstd * gv[e] + bv[e]; Y[j][4 * q + e] = y[e]; }
                    X[2 * j + (q >> 1)][2 * (q & 1)] = pack_bf16(y[0], y[1]);
                    X[2 * j + (q >> 1)][2 * (q & 1) + 1] = pack_bf16(y[2], y[3]);
                }
        }
        for (int c = 0; c < ((ABL & 8) ? 0 : MLP_NCH); ++c) {
            const int g = MLP_NCHO + c;                          // chunk index in the image
            if constexpr (!(ABL & 1)) dma_chunk(g + 2 < MLP_CHUNKS ? g + 2 : MLP_CHUNKS - 1, nx2);      // past the end: a harmless re-copy
            const unsigned char *w1base = ring + cur + r * MLP_LD1 + 16 * h;
            const unsigned char *w2base = ring + cur + MLP_W2OFF + r * MLP_LD2 + 16 * h;
            // fragment f: f < 16: fc1 k-step f (W1 row r); f >= 16: fc2 k-step qq = (f-16)/8 of output block
            // j = (f-16)%8 (W2 row 32j + r)
            u32x4 fw[FR];
            auto frag = [&](auto fc) __attribute__((always_inline)) {
                constexpr int f = decltype(fc)::value;
                if constexpr ((ABL & 4) != 0) { fw[f % FR] = X[f % 16]; return; }
                if constexpr (f < 16) fw[f % FR] = *(const u32x4 *)(w1base + 32 * f);
                else fw[f % FR] = *(const u32x4 *)(w2base + 32 * ((f - 16) % 8) * MLP_LD2 + 32 * ((f - 16) / 8));
            };
            f32x16 U;
            {
                const float *bp = b1s + MLP_CH * c + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bv = *(const f32x4 *)(bp + 8 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) U[4 * q + e] = bv[e];
                }
            }
            static_for<FR - 1>([&](auto fc) { frag(fc); });
            __builtin_amdgcn_sched_barrier(0);
            // ---- fc1: U = b1 + W1[32 rows] X^T (one dependent chain: issues back to back on gfx950) ----
            static_for<16>([&](auto fc) {
                constexpr int f = decltype(fc)::value;
                frag(std::integral_constant<int, f + FR - 1>{});
                U = MFMA_BF16(__builtin_bit_cast(bf16x8, fw[f % FR]), __builtin_bit_cast(bf16x8, X[f]), U);
                __builtin_amdgcn_sched_barrier(0);
            });
            // ---- ReLU, round to bf16: registers 8qq .. 8qq+7 are the B operand of k-step qq ----
            u32x4 P[2];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
                    P[qq][ii] = pack_bf16(relu1(U[8 * qq + 2 * ii]), relu1(U[8 * qq + 2 * ii + 1]));
            __builtin_amdgcn_sched_barrier(0);
            // ---- fc2: Y += W2[:, these 32 hidden units] P^T ----
            static_for<16>([&](auto gc) {
                constexpr int k = decltype(gc)::value, f = 16 + k;
                if constexpr (f + FR - 1 < 32) frag(std::integral_constant<int, f + FR - 1>{});
                Y[k % 8] = MFMA_BF16(__builtin_bit_cast(bf16x8, fw[f % FR]), __builtin_bit_cast(bf16x8, P[k / 8]), Y[k % 8]);
                __builtin_amdgcn_sched_barrier(0);
            });
            chunk_done();
            const int t = cur; cur = nx1; nx1 = nx2; nx2 = t;
        }
        __syncthreads();                        // drains the DMAs still in flight before the ring is reused below

        // ---- epilogue (that of gemm_ln_rows): + b2, LayerNorm over the row (lane-local + one lane^32 exchange), stores
        // transposed through a wave-private corner of the now idle ring, score head ----
        int le = tid;
        asm volatile("" : "+v"(le));
        const int er = le & 31, eh = (le >> 5) & 1;              // (opaque copies of r, h: see the prologue)
        float *tp = (float *)dyn_smem + wave * (32 * 36);
        const int trow = (le & 63) >> 3, tc4 = (le & 7) * 4;
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float pj = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *(const f32x4 *)&bias_s[32 * j + 8 * q + 4 * eh];
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
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 gv = *(const f32x4 *)&gam_s[32 * j + 8 * q + 4 * eh];
                const f32x4 bv = *(const f32x4 *)&bet_s[32 * j + 8 * q + 4 * eh];
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) { y[e] = Y[j][4 * q + e] * rstd * gv[e] + bv[e]; Y[j][4 * q + e] = y[e]; }
                *(f32x4 *)&tp[er * 36 + 8 * q + 4 * eh] = y;
            }
#pragma unroll
            for (int pq = 0; pq < 4; ++pq) {
                const f32x4 v = *(const f32x4 *)&tp[(trow + 8 * pq) * 36 + tc4];
                const int orow = m0 + trow + 8 * pq;
                if (orow < M) *(f32x4 *)(out + (size_t)orow * D + 32 * j + tc4) = v;
            }
        }
        if (score_w != nullptr) {
            for (int cc = 0; cc < num_classes; ++cc) {
                __syncthreads();
                for (int i = tid; i < D; i += NTHR) sw_s[i] = score_w[(size_t)cc * D + i];
                __syncthreads();
                float dot = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    float pj = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 wv = *(const f32x4 *)&sw_s[32 * j + 8 * q + 4 * eh];
#pragma unroll
                        for (int e = 0; e < 4; ++e) pj += Y[j][4 * q + e] * wv[e];
                    }
                    dot += pj;
                }
                dot = pair_sum(dot);
                const int row = m0 + er;
                if (eh == 0 && row < M) {
                    float sc = dot + score_b[cc];
                    if (sigmoid) sc = 1.0f / (1.0f + expf(-sc));
                    scores[(size_t)row * num_classes + cc] = sc;
                }
            }
        }
        if (qa.img != nullptr) qkv_epilogue<FR, NWV>(Y, X, qa, rd, bqkv_s, m0, er, eh, M);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Embedding + positional table + the FIRST layer's QKV (reference simnet.py:211, 237-238, 148-153), same design: a wave
// keeps its 32 output rows in Y (started at the bias), W_embed streams through the ring in 64-k chunks (the
// out-projection's chunk format, natural k order), and the B operands are the wave's own rows of x: every lane reads
// its row's 64 fp32 of a chunk straight from HBM TWO chunks ahead (2 x 8 x 16 bytes in registers - x is read exactly
// once, an LDS stage would only add a barrier dependency) and rounds them to bf16 one chunk ahead.  A chunk issues its 5
// DMA pieces and 8 x loads together, so the counted wait is vmcnt(13): loads retire in order, "<= 13 outstanding" =
// everything issued before this iteration has landed.  Same k order and epilogue arithmetic as
// gemm_nt_128<EPI_PE, .., bf16>: bit-identical rows, which then take the QKV epilogue above.
// Measured (MI355X, M = 65 536): K = 1024: 139 us with the first QKV included (generic GEMM 139 + QKV kernel 67); the
// embedding part itself (~100 us for 268 MB of x) is no faster than the generic kernel, and neither a deeper x prefetch
// (one chunk ahead: the same) nor blocks starting their k loop at different chunks (132 us) moves it.
struct EmbedArgs {
    const float *x;          // [M, K] fp32 features
    int K;                   // in_features, a multiple of 64
    const float *bias;       // [256]
    const float *pe;         // [T, 256] positional rows (row m takes pe[m % T]) or nullptr
    int T;
};

template <int NWV>      // waves per block, see mlp_fused_bf16
__global__ __launch_bounds__(512, 1) void embed_qkv_bf16(EmbedArgs ea, QkvArgs qa, const unsigned char *__restrict__ Wimg,
                                                         float *__restrict__ out, int M) {
    constexpr int D = MLP_D, NT = 8, IMG = MLP_IMG, FR = 4, NTHR = 64 * NWV, ROWS = 32 * NWV;
    extern __shared__ __attribute__((aligned(1024))) unsigned char dyn_smem[];      // ONE LDS object
    unsigned char *ring = dyn_smem;                              // [3][IMG]
    float *bias_s = (float *)(dyn_smem + 3 * IMG);               // [D]
    float *bqkv_s = bias_s + D;                                  // [3 D]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    for (int i = tid; i < D; i += NTHR) bias_s[i] = ea.bias[i];
    if (qa.img != nullptr)
        for (int i = tid; i < 3 * D; i += NTHR) bqkv_s[i] = qa.bqkv[i];
    const RingDma<NWV> rd{ring, __builtin_amdgcn_readfirstlane(wave), (unsigned)lane * 16u};
    const int nch = ea.K / 64;
    auto gch = [&](int c) __attribute__((always_inline)) { return c < nch ? c : nch - 1; };      // past the end: a harmless re-copy / reload

    f32x16 Y[NT];
    u32x4 X[2 * NT];
    f32x4 xr[2][8];                                              // this lane's 64 fp32 of the next TWO chunks: k-step ks = [2ks], [2ks+1]
    const int ntiles = (M + ROWS - 1) / ROWS;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * ROWS + 32 * wave;
        __syncthreads();                        // consts visible / the previous tile is done with the ring
        rd.dma(Wimg, gch(0), 0);
        rd.dma(Wimg, gch(1), IMG);
        int lp = tid;
        asm volatile("" : "+v"(lp));            // (lane-derived addresses recomputed per tile: see mlp_fused_bf16)
        const int pr = lp & 31, ph = (lp >> 5) & 1;
        int row = m0 + pr;
        row = row < M ? row : M - 1;
        const float *xp = ea.x + (size_t)row * ea.K + 8 * ph;
        auto xload = [&](int set, int chunk) __attribute__((always_inline)) {
            const float *xn = xp + 64 * gch(chunk);
#pragma unroll
            for (int i = 0; i < 8; ++i) xr[set][i] = *(const f32x4 *)(xn + 16 * (i >> 1) + 4 * (i & 1));
        };
        xload(0, 0);
        xload(1, 1);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *(const f32x4 *)&bias_s[32 * j + 8 * q + 4 * ph];
#pragma unroll
                for (int e = 0; e < 4; ++e) Y[j][4 * q + e] = bv[e];
            }
        __syncthreads();                        // (waits for every outstanding load, the two DMA'd chunks included)

        int cur = 0, nx1 = IMG, nx2 = 2 * IMG;
        u32x4 xb[4];                            // a chunk's four B operands: natural k order, 8 k per lane and step
        auto convert = [&](int set) __attribute__((always_inline)) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                xb[ks][0] = pack_bf16(xr[set][2 * ks][0], xr[set][2 * ks][1]);         xb[ks][1] = pack_bf16(xr[set][2 * ks][2], xr[set][2 * ks][3]);
                xb[ks][2] = pack_bf16(xr[set][2 * ks + 1][0], xr[set][2 * ks + 1][1]); xb[ks][3] = pack_bf16(xr[set][2 * ks + 1][2], xr[set][2 * ks + 1][3]);
            }
        };
        convert(0);
        // one chunk; `set` = the register set chunk c came from (free again: it takes chunk c + 2)
        auto chunk = [&](int c, auto setc) __attribute__((always_inline)) {
            constexpr int set = decltype(setc)::value;
            rd.dma(Wimg, gch(c + 2), nx2);
            xload(set, c + 2);
            const unsigned char *wbase = ring + cur + r * MLP_LDO + 16 * h;
            u32x4 fw[FR];
            auto frag = [&](auto fc) __attribute__((always_inline)) {    // fragment f: k-step f / 8, output block f % 8
                constexpr int f = decltype(fc)::value;
                fw[f % FR] = *(const u32x4 *)(wbase + 32 * (f % 8) * MLP_LDO + 32 * (f / 8));
            };
            static_for<FR - 1>([&](auto fc) { frag(fc); });
            __builtin_amdgcn_sched_barrier(0);
            static_for<32>([&](auto fc) {
                constexpr int f = decltype(fc)::value;
                if constexpr (f + FR - 1 < 32) frag(std::integral_constant<int, f + FR - 1>{});
                Y[f % 8] = MFMA_BF16(__builtin_bit_cast(bf16x8, fw[f % FR]), __builtin_bit_cast(bf16x8, xb[f / 8]), Y[f % 8]);
                __builtin_amdgcn_sched_barrier(0);
            });
            // <= 13 loads outstanding = this iteration's 5 DMA pieces + 8 x loads: everything older has landed - the
            // chunk the ring serves next AND the x registers of the next chunk, which are converted right here
            wait_vmcnt_le<40 / NWV + 8>();
            convert(set ^ 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int t = cur; cur = nx1; nx1 = nx2; nx2 = t;
        };
        int c = 0;
        for (; c + 1 < nch; c += 2) {
            chunk(c, std::integral_constant<int, 0>{});
            chunk(c + 1, std::integral_constant<int, 1>{});
        }
        if (c < nch) chunk(c, std::integral_constant<int, 0>{});
        __syncthreads();                        // drains what is still in flight before the ring is reused below

        // ---- + positional row, stores transposed through a wave-private corner of the idle ring (full 128-byte lines) ----
        int le = tid;
        asm volatile("" : "+v"(le));
        const int er = le & 31, eh = (le >> 5) & 1;
        float *tp = (float *)dyn_smem + wave * (32 * 36);
        const int trow = (le & 63) >> 3, tc4 = (le & 7) * 4;
        int prow = m0 + er;
        prow = prow < M ? prow : M - 1;
        const float *pp = ea.pe != nullptr ? ea.pe + (size_t)(prow % ea.T) * D + 4 * eh : nullptr;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = Y[j][4 * q + e];
                if (pp != nullptr) {
                    const f32x4 pv = *(const f32x4 *)(pp + 32 * j + 8 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] += pv[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) Y[j][4 * q + e] = y[e];
                *(f32x4 *)&tp[er * 36 + 8 * q + 4 * eh] = y;
            }
#pragma unroll
            for (int pq = 0; pq < 4; ++pq) {
                const f32x4 v = *(const f32x4 *)&tp[(trow + 8 * pq) * 36 + tc4];
                const int orow = m0 + trow + 8 * pq;
                if (orow < M) *(f32x4 *)(out + (size_t)orow * D + 32 * j + tc4) = v;
            }
        }
        if (qa.img != nullptr) qkv_epilogue<FR, NWV>(Y, X, qa, rd, bqkv_s, m0, er, eh, M);
    }
}

// W_embed [256][K] fp32 -> img [K / 64 chunks][MLP_IMG bytes]: rows of 64 bf16 + 16 B, natural k order
__global__ void pack_embed_bf16(const float *__restrict__ W, unsigned char *__restrict__ img, int K) {
    const int n = MLP_D * K / 2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int row = (2 * i) / K, k = (2 * i) % K;
        unsigned *dst = (unsigned *)(img + (size_t)(k / 64) * MLP_IMG + row * MLP_LDO + 2 * (k % 64));
        *dst = pack_bf16(W[(size_t)row * K + k], W[(size_t)row * K + k + 1]);
    }
}

constexpr size_t EMBED_LDS = (size_t)3 * MLP_IMG + 4 * MLP_D * sizeof(float);

constexpr size_t MLP_LDS = (size_t)3 * MLP_IMG + (MLP_HID + 10 * MLP_D) * sizeof(float);      // 134 KiB

// hipFuncAttributeMaxDynamicSharedMemorySize is per DEVICE: set once per (kernel instantiation, device)
template <bool TAIL, int NWV, int ABL>
int allow_lds() {
    static std::atomic<unsigned char> done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipErrorInvalidDevice;
    if (dev >= 0 && dev < 64 && done[dev].load(std::memory_order_acquire)) return 0;
    const int rc = (int)hipFuncSetAttribute((const void *)mlp_fused_bf16<TAIL, NWV, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)MLP_LDS);
    if (rc == 0 && dev >= 0 && dev < 64) done[dev].store(1, std::memory_order_release);
    return rc;
}

}  // namespace

size_t vsk_mlp_bf16_image_bytes(int d) { return d == MLP_D ? (size_t)MLP_CHUNKS * MLP_IMG : 0; }

int vsk_pack_mlp_bf16(const float *Wo, const float *W1, const float *W2, void *img, int d, hipStream_t st) {
    if (d != MLP_D) return -1;
    if (hipMemsetAsync(img, 0, vsk_mlp_bf16_image_bytes(d), st) != hipSuccess) return (int)hipGetLastError();
    hipLaunchKernelGGL(pack_mlp_bf16, dim3(256), dim3(256), 0, st, Wo, W1, W2, (unsigned char *)img);
    VSK_CHECK_LAUNCH();
    return 0;
}

// 128-row tiles on 4-wave blocks (one wave per SIMD) when 256-row tiles would leave at least half the CUs without a block
static bool vsk_fused_half_tiles(int M, int cus) { return !vsk_options().lp_tile256 && (M + 255) / 256 <= cus / 2; }

size_t vsk_embed_bf16_image_bytes(int d, int K) { return d == MLP_D && K > 0 && K % 64 == 0 ? (size_t)(K / 64) * MLP_IMG : 0; }

int vsk_pack_embed_bf16(const float *W, void *img, int d, int K, hipStream_t st) {
    const size_t bytes = vsk_embed_bf16_image_bytes(d, K);
    if (!bytes) return -1;
    if (hipMemsetAsync(img, 0, bytes, st) != hipSuccess) return (int)hipGetLastError();
    hipLaunchKernelGGL(pack_embed_bf16, dim3(256), dim3(256), 0, st, W, (unsigned char *)img, K);
    VSK_CHECK_LAUNCH();
    return 0;
}

// h0 = x W^T + bias + pe[row % T]  (fp32 out, bf16 matrix pipe), and - next != nullptr - the first layer's q / k / v
int vsk_embed_bf16(const float *x, const void *img, const float *bias, const float *pe, int T, float *out, int M, int d, int K,
                   const VskNextQkv *next, hipStream_t st) {
    if (!vsk_embed_bf16_image_bytes(d, K) || M <= 0 || T <= 0) return -1;
    if (next && (next->T <= 0 || M % next->T || next->H <= 0 || d % next->H || (d / next->H) % 16)) return -1;
    const int cus = vsk_device_cus();
    if (cus <= 0) return (int)hipErrorInvalidDevice;
    const EmbedArgs ea{x, K, bias, pe, T};
    const QkvArgs qa = next ? QkvArgs{(const unsigned char *)next->img, next->bqkv, (h16 *)next->qkv16, next->T, next->H, d / next->H, next->qscale}
                            : QkvArgs{nullptr, nullptr, nullptr, 1, 1, d, 1.0f};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipErrorInvalidDevice;
#define VSK_EMBED_LAUNCH(W_)                                                                                            \
    do {                                                                                                                \
        static std::atomic<unsigned char> done[64];      /* hipFuncAttributeMaxDynamicSharedMemorySize is per DEVICE */  \
        if (!(dev >= 0 && dev < 64 && done[dev].load(std::memory_order_acquire))) {                                     \
            const int rc = (int)hipFuncSetAttribute((const void *)embed_qkv_bf16<W_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)EMBED_LDS); \
            if (rc) return rc;                                                                                          \
            if (dev >= 0 && dev < 64) done[dev].store(1, std::memory_order_release);                                    \
        }                                                                                                               \
        const int ntiles = (M + 32 * W_ - 1) / (32 * W_);                                                               \
        hipLaunchKernelGGL(embed_qkv_bf16<W_>, dim3(ntiles < cus ? ntiles : cus), dim3(64 * W_), EMBED_LDS, st, ea, qa,  \
                           (const unsigned char *)img, out, M);                                                         \
    } while (0)
    if (vsk_fused_half_tiles(M, cus)) VSK_EMBED_LAUNCH(4); else VSK_EMBED_LAUNCH(8);
#undef VSK_EMBED_LAUNCH
    VSK_CHECK_LAUNCH();
    return 0;
}

bool vsk_mlp_bf16_supported(int d) { return d == MLP_D; }

size_t vsk_qkv_bf16_image_bytes(int d) { return d == MLP_D ? (size_t)MLP_NCHQ * MLP_IMG : 0; }

int vsk_pack_qkv_bf16(const float *Wqkv, void *img, int d, hipStream_t st) {
    if (d != MLP_D) return -1;
    if (hipMemsetAsync(img, 0, vsk_qkv_bf16_image_bytes(d), st) != hipSuccess) return (int)hipGetLastError();
    hipLaunchKernelGGL(pack_qkv_bf16, dim3(256), dim3(256), 0, st, Wqkv, (unsigned char *)img);
    VSK_CHECK_LAUNCH();
    return 0;
}

// att16 != nullptr: the layer tail (out-projection + norm1 + MLP block): h = the residual (layer input), att16 the bf16
// attention output;  att16 == nullptr: the MLP block alone on h = h1
int vsk_mlp_bf16(const float *h, const void *att16, const float *bo, const float *gamma1, const float *beta1,
                 const void *img, const float *b1, const float *b2,
                 const float *gamma, const float *beta, float *out, int M, int d,
                 const float *score_w, const float *score_b, int num_classes, int sigmoid, float *scores,
                 const VskNextQkv *next, hipStream_t st) {
    if (!vsk_mlp_bf16_supported(d) || M <= 0) return -1;
    if (next && (next->T <= 0 || M % next->T || next->H <= 0 || d % next->H || (d / next->H) % 16)) return -1;
    const int cus = vsk_device_cus();
    if (cus <= 0) return (int)hipErrorInvalidDevice;
    const TailArgs ta{(const h16 *)att16, h, bo, gamma1, beta1};
    const QkvArgs qa = next ? QkvArgs{(const unsigned char *)next->img, next->bqkv, (h16 *)next->qkv16, next->T, next->H, d / next->H, next->qscale}
                            : QkvArgs{nullptr, nullptr, nullptr, 1, 1, d, 1.0f};
#define VSK_MLP_LAUNCH(T_, W_, A_)                                                                                     \
    do {                                                                                                               \
        if (const int rc = allow_lds<T_, W_, A_>()) return rc;                                                         \
        const int ntiles = (M + 32 * W_ - 1) / (32 * W_);                                                              \
        hipLaunchKernelGGL((mlp_fused_bf16<T_, W_, A_>), dim3(ntiles < cus ? ntiles : cus), dim3(64 * W_), MLP_LDS, st, h, ta, qa, \
                           (const unsigned char *)img, b1, b2, gamma, beta, out, M, score_w, score_b, num_classes, sigmoid, scores); \
    } while (0)
#ifdef VS_WITH_DIAG
    switch (vsk_options().mlp_abl) {       // timing-only ablations of the MLP-block kernel (tools/bench_mlp_fused.py)
        case 0: break;
        case 1: VSK_MLP_LAUNCH(false, 8, 1); VSK_CHECK_LAUNCH(); return 0;
        case 2: VSK_MLP_LAUNCH(false, 8, 2); VSK_CHECK_LAUNCH(); return 0;
        case 3: VSK_MLP_LAUNCH(false, 8, 3); VSK_CHECK_LAUNCH(); return 0;
        case 4: VSK_MLP_LAUNCH(false, 8, 4); VSK_CHECK_LAUNCH(); return 0;
        case 5: VSK_MLP_LAUNCH(false, 8, 5); VSK_CHECK_LAUNCH(); return 0;
        case 7: VSK_MLP_LAUNCH(false, 8, 7); VSK_CHECK_LAUNCH(); return 0;
        default: VSK_MLP_LAUNCH(false, 8, 8); VSK_CHECK_LAUNCH(); return 0;
    }
#endif
    if (vsk_fused_half_tiles(M, cus)) {
        if (att16 != nullptr) VSK_MLP_LAUNCH(true, 4, 0); else VSK_MLP_LAUNCH(false, 4, 0);
    } else {
        if (att16 != nullptr) VSK_MLP_LAUNCH(true, 8, 0); else VSK_MLP_LAUNCH(false, 8, 0);
    }
#undef VSK_MLP_LAUNCH
    VSK_CHECK_LAUNCH();
    return 0;
}
