// vs_attention_w64.hip — bf16 attention, head dim 64, ONE WAVE PER SIMD (reference simnet.py:155-161; the bf16 mode's
// attention for bf16-stored q * scale * log2 e / k / v planes and a bf16 output).
//
// A block is 4 waves = 256 query rows of one (video, head); a wave owns 64 query rows = two 32-row blocks A and B and
// the whole 512-register file of its SIMD.  Why this shape: in the 8-wave
// kernel (attn_fwd_lp_pipe, 32 rows per wave, two waves per SIMD) every K / V^T fragment read from LDS feeds ONE
// MFMA and the two waves of a SIMD compete for its issue port; here a fragment feeds TWO MFMAs (one per row block),
// the row constant rides in as the accumulator's initial value (no bias MFMA), and the softmax of one row block is
// placed in the MFMA gaps of the other block's products:
//
//   iteration t:   MFMA stream                         vector / LDS stream beside it
//     step 1       S'(t+1, A) = K(t+1) Q_A^T - c_A     softmax(t, B) second part, OR test;  V(t) fragment reads
//     step 2       O_A += V(t)^T P(t, A)^T, l_A        softmax(t+1, A) first part
//     step 3       S'(t+1, B) = K(t+1) Q_B^T - c_B     softmax(t+1, A) second part, OR test
//     wait for this wave's LDS-DMA pieces of K(t+2) / V(t+1), block barrier
//     step 4       O_B += V(t)^T P(t, B)^T, l_B        softmax(t+1, B) first part;  K(t+2) fragment reads; DMA K(t+4), V(t+3)
//
// 40 MFMAs per 64-key tile and wave (32 products + 8 row-sum steps on a block of ones), 64 exp2 + 32 packs between them.
// Same operand layouts as attn_fwd_lp_pipe<64, 8, 1, ., IO16 = true> (vs_attention.hip): K / V tiles are dense
// [64 keys][128 B] LDS images filled by LDS-DMA and XOR-swizzled through the source address, K fragments by
// ds_read_b128, V^T fragments by ds_read_b64_tr_b16, the S' accumulator registers packed pairwise ARE the B operand of
// the second product.  Softmax: P = exp2(S') against a row constant c kept >= max - 1 (raised to max + 6 when a P
// reaches 2: OR test on the packed P; rare path), l from the rounded P on the matrix pipe.
#include <atomic>

#include "vs_device.h"
#include "vs_kernels.h"

// The instruction stream itself is generated (tools/gen_attn_w64.py -> vs_attention_w64_asm.inc): hipcc, given the
// 512-register budget, moves the S' accumulators through v_accvgpr_read / write around every exp2.
// This file is the shell: block -> (video, head, query tile), the key-bias table and tile flags, the per-lane offsets.

namespace {

typedef unsigned short h16;

// ABL (diagnostic library only): timing ablations of the stream with WRONG results (tools/gen_attn_w64.py --abl)
template <bool VARLEN, bool HASMASK, int ABL = 0>
__global__ __launch_bounds__(256, 1) void attn_fwd_bf16_w64(
    const h16 *__restrict__ Q, const h16 *__restrict__ Kg, const h16 *__restrict__ Vg, const uint8_t *__restrict__ mask,
    h16 *__restrict__ out, int H, int T, int BH, const int *__restrict__ cu, const int2 *__restrict__ work, int Mtot, int mode0) {
    constexpr int DH = 64, KT = 64, NBUF = 3;
    __shared__ __attribute__((aligned(1024))) h16 Kb[NBUF][KT * DH];      // 8 KiB per tile, ring of three
    __shared__ __attribute__((aligned(1024))) h16 Vb[NBUF][KT * DH];
    // key bias (0 / -inf): HASMASK the whole row [ntiles * 64] followed by one flag byte per tile; else the last tile only
    extern __shared__ __attribute__((aligned(16))) float mbias[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
        if (!attn_block_map((T + 255) / 256, BH, bh, qt)) return;
        b = bh / H; head = bh - b * H;
        base = (size_t)bh * T * DH;
        orow0 = (size_t)b * T;
    }
    const int ntiles = (T + KT - 1) / KT;
    const float NEG_INF = -__builtin_inff();

    // A ragged last tile: its rows beyond the video are outside the buffer bounds of the LDS-DMA.  Whether such a lane
    // writes zeros or nothing, the ring then never holds uninitialised bits (0 * garbage could be a NaN in P.V).
    if ((T & (KT - 1)) != 0) {
        u32x4 *kz = (u32x4 *)&Kb[0][0], *vz = (u32x4 *)&Vb[0][0];
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int i = tid; i < NBUF * KT * DH * 2 / 16; i += 256) { kz[i] = z; vz[i] = z; }
    }
    // tile flags: bit (t & 31) of lane (t >> 5) set <=> tile t has a masked key or runs past the end of the video
    unsigned flags = 0u;
    int lastf = -1;                                  // no-mask stream: the one tile that needs the key bias (-1: none)
    // optimistic pass: minus the number of keys beyond the end of the video in a ragged last tile that is not tile 0
    const float npadn = (ntiles > 1 && (T & (KT - 1)) != 0) ? -(float)(KT - (T & (KT - 1))) : 0.f;
    unsigned mb_addr = (unsigned)(size_t)(__attribute__((address_space(3))) float *)mbias;
    if constexpr (HASMASK) {
        const int tpad = ntiles * KT;
        unsigned char *tflag = (unsigned char *)(mbias + tpad);
        const uint8_t *mrow = mask + (size_t)b * T;
        for (int k = tid; k < tpad; k += 256) mbias[k] = (k >= T || mrow[k] != 0) ? NEG_INF : 0.f;
        __syncthreads();
        for (int t = tid; t < ntiles; t += 256) {
            bool any = false;
            for (int k = 0; k < KT; ++k) any |= mbias[t * KT + k] != 0.f;
            tflag[t] = any ? 1 : 0;
        }
        __syncthreads();
        for (int j = 0; j < 32; ++j) {
            const int t = 32 * lane + j;
            if (t < ntiles && tflag[t] != 0) flags |= 1u << j;
        }
    } else {
        const int tl = ntiles - 1;
        if (tid < KT) mbias[tid] = (tl * KT + tid >= T) ? NEG_INF : 0.f;
        if ((T & (KT - 1)) != 0) lastf = tl;
        mb_addr -= (unsigned)tl * 256u;          // the stream addresses the table as mb + tile * 256
        __syncthreads();
    }

    // per-lane byte offsets (see attn_fwd_lp_pipe's DMA form, vs_attention.hip, for the two LDS images)
    const int swz = (r >> 1) & 7;
    const int koff = r * 128 + ((h ^ swz) << 4);                           // K fragment (key r, k step 0); step ks: ^ (ks << 5)
    const int i16 = lane & 15, y = (i16 >> 3) & 1, xx = 2 * ((lane >> 4) & 1) + ((i16 >> 1) & 1);
    const int voff = (4 * h + (i16 >> 2)) * 128 + ((4 * y + xx) << 4) + 8 * (i16 & 1);     // V^T fragment, d block 0; block 1: ^ 64
    const int row0 = 16 * wave + (lane >> 3);                              // key row of this lane in DMA piece 2w; piece 2w+1: + 8
    const int dk0 = row0 * 128 + (((lane & 7) ^ ((row0 >> 1) & 7)) << 4);
    const int dv0 = row0 * 128 + (((lane & 7) ^ (4 * ((row0 >> 1) & 1))) << 4);
    const int qrow = qt * 256 + 64 * wave + r;
    const int qoff = qrow * 128 + 16 * h;
    const int ooff = qrow * (H * 128) + 8 * h;
    const unsigned long long qp = (unsigned long long)(Q + base), kp = (unsigned long long)(Kg + base), vp = (unsigned long long)(Vg + base),
                             op = (unsigned long long)(out + orow0 * (size_t)(H * DH) + (size_t)head * DH);
    const unsigned nrec = (unsigned)T * 128u, nreco = ((unsigned)(T - 1) * (unsigned)H * 64u + 64u) * 2u;
    const unsigned kb = (unsigned)(size_t)(__attribute__((address_space(3))) h16 *)&Kb[0][0];
    const unsigned vb = (unsigned)(size_t)(__attribute__((address_space(3))) h16 *)&Vb[0][0];
    // every "s" operand must be provably wave-uniform
    auto sc = [](unsigned v) __attribute__((always_inline)) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
#define VS_W64_OPERANDS                                                                                                              \
        : [qlo] "s"(sc((unsigned)qp)), [qhi] "s"(sc((unsigned)(qp >> 32))), [klo] "s"(sc((unsigned)kp)), [khi] "s"(sc((unsigned)(kp >> 32))), \
          [vlo] "s"(sc((unsigned)vp)), [vhi] "s"(sc((unsigned)(vp >> 32))), [olo] "s"(sc((unsigned)op)), [ohi] "s"(sc((unsigned)(op >> 32))), \
          [nrec] "s"(sc(nrec)), [nreco] "s"(sc(nreco)), [ntiles] "s"(sc((unsigned)ntiles)), [mb] "s"(sc(mb_addr)), [kb] "s"(sc(kb)),     \
          [vb] "s"(sc(vb)), [orowb] "s"(sc((unsigned)(32 * H * 128))), [wave] "s"(wave), [lastf] "s"(sc((unsigned)lastf)), [mode0] "s"(sc((unsigned)mode0)), [npadn] "s"(sc(__builtin_bit_cast(unsigned, npadn))),               \
          [koff] "v"(koff), [voff] "v"(voff), [dk0] "v"(dk0), [dv0] "v"(dv0), [qoff] "v"(qoff), [ooff] "v"(ooff), [flags] "v"(flags)
    if constexpr (ABL == 0 && HASMASK) {      // per-tile flags, "row without a constant yet" checked on every tile
        asm volatile(
            "s_nop 4\n"
#include "vs_attention_w64_asm_mask.inc"
            : VS_W64_OPERANDS
            :
#include "vs_attention_w64_clobbers.inc"
        );
    } else if constexpr (ABL == 0) {          // only the last tile can be ragged
        asm volatile(
            "s_nop 4\n"
#include "vs_attention_w64_asm.inc"
            : VS_W64_OPERANDS
            :
#include "vs_attention_w64_clobbers.inc"
        );
    }
#ifdef VS_WITH_DIAG
    else if constexpr (ABL == 1) {
        asm volatile(
            "s_nop 4\n"
#include "vs_attention_w64_asm_abl1.inc"
            : VS_W64_OPERANDS
            :
#include "vs_attention_w64_clobbers.inc"
        );
    }
    else if constexpr (ABL == 2) {
        asm volatile(
            "s_nop 4\n"
#include "vs_attention_w64_asm_abl2.inc"
            : VS_W64_OPERANDS
            :
#include "vs_attention_w64_clobbers.inc"
        );
    }
    else if constexpr (ABL == 4) {
        asm volatile(
            "s_nop 4\n"
#include "vs_attention_w64_asm_abl4.inc"
            : VS_W64_OPERANDS
            :
#include "vs_attention_w64_clobbers.inc"
        );
    }
    else if constexpr (ABL == 8) {
        asm volatile(
            "s_nop 4\n"
#include "vs_attention_w64_asm_abl8.inc"
            : VS_W64_OPERANDS
            :
#include "vs_attention_w64_clobbers.inc"
        );
    }
    else if constexpr (ABL == 14) {
        asm volatile(
            "s_nop 4\n"
#include "vs_attention_w64_asm_abl14.inc"
            : VS_W64_OPERANDS
            :
#include "vs_attention_w64_clobbers.inc"
        );
    }
    else if constexpr (ABL == 15) {
        asm volatile(
            "s_nop 4\n"
#include "vs_attention_w64_asm_abl15.inc"
            : VS_W64_OPERANDS
            :
#include "vs_attention_w64_clobbers.inc"
        );
    }
    else if constexpr (ABL == 512) {
        asm volatile(
            "s_nop 4\n"
#include "vs_attention_w64_asm_abl512.inc"
            : VS_W64_OPERANDS
            :
#include "vs_attention_w64_clobbers.inc"
        );
    }
    else if constexpr (ABL == 1024) {
        asm volatile(
            "s_nop 4\n"
#include "vs_attention_w64_asm_abl1024.inc"
            : VS_W64_OPERANDS
            :
#include "vs_attention_w64_clobbers.inc"
        );
    }
#endif
#undef VS_W64_OPERANDS
}

}  // namespace

// bf16 q (pre-scaled by scale * log2 e) / k / v planes [B*H][T][64] in, bf16 [B*T][H*64] out.  Returns -1 when the
// shape does not fit this kernel (the caller then uses attn_fwd_lp_pipe).
int vsk_attention_bf16_w64(const void *q, const void *k, const void *v, const uint8_t *mask, void *out, int B, int H, int T,
                           hipStream_t st) {
    const int BH = B * H, nq = (T + 255) / 256, ntiles = (T + 63) / 64;
    dim3 grid(8 * ((BH + 7) / 8) * nq);
    if (mask != nullptr) {
        // key-bias table of the whole row + one flag byte per tile behind the 48 KiB of K / V ring: beyond the 64 KiB a
        // launch may use by default, so the limit of this kernel is raised once (gfx950: 160 KiB per workgroup)
        const size_t dyn = (size_t)ntiles * 64 * sizeof(float) + (size_t)((ntiles + 15) / 16 * 16);
        if (dyn > 96 * 1024 || ntiles > 2048) return -1;
        static std::atomic<int> raised{0};
        if (!raised.load(std::memory_order_acquire)) {
            if (hipFuncSetAttribute((const void *)attn_fwd_bf16_w64<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) {
                (void)hipGetLastError();
                return -1;
            }
            raised.store(1, std::memory_order_release);
        }
        hipLaunchKernelGGL((attn_fwd_bf16_w64<false, true>), grid, dim3(256), dyn, st, (const h16 *)q, (const h16 *)k, (const h16 *)v,
                           mask, (h16 *)out, H, T, BH, nullptr, nullptr, 0, 1);
    } else {
#ifdef VS_WITH_DIAG
#define VS_W64_LAUNCH_ABL(A_) case A_: hipLaunchKernelGGL((attn_fwd_bf16_w64<false, false, A_>), grid, dim3(256), 256, st, (const h16 *)q, (const h16 *)k, (const h16 *)v, nullptr, (h16 *)out, H, T, BH, nullptr, nullptr, 0, 0); break;
        switch (vsk_options().attn_w64_abl) {
            VS_W64_LAUNCH_ABL(1) VS_W64_LAUNCH_ABL(2) VS_W64_LAUNCH_ABL(4) VS_W64_LAUNCH_ABL(8) VS_W64_LAUNCH_ABL(14) VS_W64_LAUNCH_ABL(15) VS_W64_LAUNCH_ABL(512) VS_W64_LAUNCH_ABL(1024)
            default:
#endif
        hipLaunchKernelGGL((attn_fwd_bf16_w64<false, false>), grid, dim3(256), 256, st, (const h16 *)q, (const h16 *)k, (const h16 *)v,
                           nullptr, (h16 *)out, H, T, BH, nullptr, nullptr, 0, vsk_options().attn_w64_checked ? 1 : 0);
#ifdef VS_WITH_DIAG
        }
#undef VS_W64_LAUNCH_ABL
#endif
    }
    VSK_CHECK_LAUNCH();
    return 0;
}

// packed ragged batch: planes [H][Mtot][64], cu [B+1] row offsets, work[nwork] = (video, query tile of 256 rows)
int vsk_attention_bf16_w64_packed(const void *q, const void *k, const void *v, void *out, int H, int Mtot, const int *cu,
                                  const int *work, int nwork, hipStream_t st) {
    if (nwork <= 0) return 0;
    hipLaunchKernelGGL((attn_fwd_bf16_w64<true, false>), dim3(nwork, H), dim3(256), 256, st, (const h16 *)q, (const h16 *)k,
                       (const h16 *)v, nullptr, (h16 *)out, H, 0, 0, cu, (const int2 *)work, Mtot, vsk_options().attn_w64_checked ? 1 : 0);
    VSK_CHECK_LAUNCH();
    return 0;
}
