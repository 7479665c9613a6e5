// vs_train_device.h — device helpers shared by the training-path kernels (vs_train_kernels.hip, vs_train_attention.hip).
#pragma once
#include "vs_device.h"
#include "vs_train_device_sites.h"

namespace {

// ------------------------------------------------------------------------------------------
// Dropout: a counter-based hash instead of a stateful generator, so the backward kernels REBUILD every keep
// decision from (seed, site, row, column) and no mask is ever stored (the attention mask alone would be
// B*H*T*T bytes per layer).  Two rounds of the murmur3 32-bit finaliser:
//     rowkey  = fmix32(fmix32(seed_lo ^ site * 0x9E3779B1) ^ seed_hi ^ row * 0x85EBCA77)
//     pair    = fmix32(rowkey ^ (col >> 1) * 0x27D4EB2F)     one draw per TWO columns (round 3: the hash was 13 instructions per
//     element = the low (even col) / high (odd col) 16 bits   element in every dropout epilogue and 2/3 of attn_dropout_bits)
//     keep  <=>  element >= threshold = round(p * 2^16)       (p is honoured to 1 / 65 536)
// `site` numbers the dropout module (embedding; per layer: attention weights, dropout1, mlp.dropout, dropout2 —
// reference simnet.py:237, 159, 107, 181, 110); for the attention weights row = (video*H + head)*T + query and
// col = key, elsewhere row = frame index and col = feature index.  The stream differs from torch's Philox stream
// (the reference's masks cannot be reproduced by any re-implementation); what is pinned by tests is the keep rate,
// the independence across sites/rows, and that forward and backward use the SAME mask (finite differences and an
// explicit-mask torch model).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned fmix32(unsigned h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
struct DropSite {            // per (seed, site) constants, computed once per kernel
    unsigned base, seed_hi, thresh;
    float scale;             // 1 / (1 - p)
};
__device__ __forceinline__ DropSite drop_site(unsigned long long seed, unsigned site, float p) {
    DropSite s;
    s.base = fmix32((unsigned)seed ^ (site * 0x9E3779B1u));
    s.seed_hi = (unsigned)(seed >> 32);
    const double t = (double)p * 65536.0;
    s.thresh = p <= 0.f ? 0u : (t >= 65535.0 ? 0xFFFFu : (unsigned)(t + 0.5));
    s.scale = p < 1.f ? 1.0f / (1.0f - p) : 0.f;
    return s;
}
__device__ __forceinline__ unsigned drop_rowkey(const DropSite &s, unsigned row) {
    return fmix32(s.base ^ s.seed_hi ^ (row * 0x85EBCA77u));
}
__device__ __forceinline__ bool drop_keep(const DropSite &s, unsigned rowkey, unsigned col) {
    const unsigned pair = fmix32(rowkey ^ ((col >> 1) * 0x27D4EB2Fu));
    return ((col & 1u) ? pair >> 16 : pair & 0xFFFFu) >= s.thresh;
}

// sum over the 64 lanes of a wave
__device__ __forceinline__ float wave_sum(float v) {
    v += __shfl_xor(v, 32);
    return half_sum(v);
}

}  // namespace
