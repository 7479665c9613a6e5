// vs_train_device.h — device helpers shared by the training-path kernels (vs_train_kernels.hip, vs_train_attention.hip).
#pragma once
#include "vs_device.h"
#include "vs_train_device_sites.h"

namespace {

// ------------------------------------------------------------------------------------------
// Dropout: a counter-based hash instead of a stateful generator, so the backward kernels REBUILD every keep
// decision from (seed, site, row, column) and no mask is ever stored (the attention mask alone would be
// B*H*T*T bytes per layer).
//     rowkey  = fmix32(fmix32(seed_lo ^ site * 0x9E3779B1) ^ seed_hi ^ row * 0x85EBCA77)     (murmur3 finaliser, per row)
//     element = mix1(rowkey + col * 0x9E3779B1),  mix1(h): h ^= h >> 16; h *= 0x7FEB352D; h ^= h >> 15
//     keep  <=>  element >= threshold = round(p * 2^32)
// The per-ELEMENT step is one xorshift-multiply round on a golden-ratio-stride counter offset by the fully mixed row
// key: one 32-bit multiply (quarter rate on this chip) instead of the three of a second fmix32 - the hash runs 16 times
// per 32x32 attention tile beside fp32 MFMAs that share the VALU issue port (measured: the two-round form cost the
// attention backward ~8 %).  In the hot loops `col * stride` is carried incrementally (drop_keep_at).
// `site` numbers the dropout module (embedding; per layer: attention weights, dropout1, mlp.dropout, dropout2 —
// reference simnet.py:237, 159, 107, 181, 110); for the attention weights row = (video*H + head)*T + query and
// col = key, elsewhere row = frame index and col = feature index.  The stream differs from torch's Philox stream
// (the reference's masks cannot be reproduced by any re-implementation); what is pinned by tests is the keep rate,
// the independence across rows / columns / sites / seeds, and that forward and backward use the SAME mask (an
// explicit-mask float64 torch model fed with the masks dumped by vs_train_dropout_mask_*).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned fmix32(unsigned h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
constexpr unsigned DROP_COL_STRIDE = 0x9E3779B1u;
struct DropSite {            // per (seed, site) constants, computed once per kernel
    unsigned base, seed_hi, thresh;
    float scale;             // 1 / (1 - p)
};
__device__ __forceinline__ DropSite drop_site(unsigned long long seed, unsigned site, float p) {
    DropSite s;
    s.base = fmix32((unsigned)seed ^ (site * 0x9E3779B1u));
    s.seed_hi = (unsigned)(seed >> 32);
    const double t = (double)p * 4294967296.0;
    s.thresh = p <= 0.f ? 0u : (t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)(t + 0.5));
    s.scale = p < 1.f ? 1.0f / (1.0f - p) : 0.f;
    return s;
}
__device__ __forceinline__ unsigned drop_rowkey(const DropSite &s, unsigned row) {
    return fmix32(s.base ^ s.seed_hi ^ (row * 0x85EBCA77u));
}
// keep decision from the pre-offset counter  c = rowkey + col * DROP_COL_STRIDE
__device__ __forceinline__ bool drop_keep_at(const DropSite &s, unsigned c) {
    c ^= c >> 16; c *= 0x7FEB352Du; c ^= c >> 15;
    return c >= s.thresh;
}
__device__ __forceinline__ bool drop_keep(const DropSite &s, unsigned rowkey, unsigned col) {
    return drop_keep_at(s, rowkey + col * DROP_COL_STRIDE);
}

// sum over the 64 lanes of a wave
__device__ __forceinline__ float wave_sum(float v) {
    v += __shfl_xor(v, 32);
    return half_sum(v);
}

}  // namespace
