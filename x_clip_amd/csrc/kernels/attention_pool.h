// attention_pool.h -- attention for ONE query row per (sample, head): the last layer of a tower whose caller reads a single token row
// (the CLS head, x_clip.py:708 `enc_text[:, 0]`; functional.py stack_forward `pool_row`).  Reference arithmetic: Attention.forward
// x_clip.py:213-245 restricted to that query -- sim = scale q k^T over every key of the sample, key mask, softmax, out = attn v.
//
//   q    [batch, heads * HS]        the pooled rows' queries (to_qkv's first third applied to those rows only)
//   kv   [batch, n, 2 * heads * HS] keys | values of EVERY row (to_qkv's other two thirds)
//   out  [batch, heads * HS], lse [batch, heads] (natural log, of the scaled scores)
//   backward: dq [batch, heads * HS], dkv [batch, n, 2 * heads * HS] fully written (a key's dK = dS q, dV = P dO: one query's outer products)
//
// HBM-bound by construction (every key / value row is read once, every dK / dV row written once; 4 n HS flops per head): one wave per
// (sample, head), no MFMA.  A wave-wide load fetches KPL = 64 / CH whole key rows (CH = 16-byte chunks per row: 8 for 64 bf16 features);
// the CH lanes of a key reduce its dot product with CH-lane butterflies, and every lane group runs its OWN online softmax over the keys it
// sees (no cross-group traffic inside the loop); the KPL partial results are merged once at the end.
#pragma once
#include "common.h"

namespace xc {

struct AttnPoolParams {
    const void* q; const void* kv; const unsigned char* mask;
    void* out; float* lse;                       // forward results (backward: inputs)
    const void* dout; void* dq; void* dkv;       // backward
    int batch, n, heads;
    int nvis;                                    // keys [0, nvis) are visible to the pooled query (n, or row + 1 under a causal mask)
    float scale;
};

// sum over the CH lanes of a key group (CH a power of two <= 64; groups are aligned lane ranges)
template <int CH>
XC_DEV float group_sum(float v) {
#pragma unroll
    for (int m = 1; m < CH; m <<= 1) v += shfl_xor(v, m);
    return v;
}
// sum / max over the KPL groups (lanes with equal chunk index)
template <int CH>
XC_DEV float cross_sum(float v) {
#pragma unroll
    for (int m = CH; m < 64; m <<= 1) v += shfl_xor(v, m);
    return v;
}
template <int CH>
XC_DEV float cross_max(float v) {
#pragma unroll
    for (int m = CH; m < 64; m <<= 1) v = fmaxf(v, shfl_xor(v, m));
    return v;
}

template <typename T, int HS>
__global__ __launch_bounds__(256) void attn_pool_fwd_kernel(AttnPoolParams p) {
    constexpr int VEC = Elem<T>::VEC, CH = HS / VEC, KPL = 64 / CH;
    static_assert(CH >= 1 && CH <= 64 && (CH & (CH - 1)) == 0, "chunks per key row must be a power of two");
    const int lane = lane_id();
    const long bh = (long)blockIdx.x * 4 + wave_id();
    if (bh >= (long)p.batch * p.heads) return;                 // whole wave leaves; no barriers below
    const int b = (int)(bh / p.heads), h = (int)(bh % p.heads);
    const int inner = p.heads * HS;
    const int g = lane / CH, c = lane % CH;                    // key within the wave-wide load, 16-byte chunk of its row
    const T* kv = static_cast<const T*>(p.kv) + (long)b * p.n * 2 * inner + (long)h * HS + c * VEC;
    float qv[VEC];
    load_vec<T>(static_cast<const T*>(p.q) + (long)b * inner + (long)h * HS + c * VEC, qv);
#pragma unroll
    for (int e = 0; e < VEC; ++e) qv[e] *= p.scale;
    float m = -INFINITY, l = 0.f, acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    for (int j0 = 0; j0 < p.nvis; j0 += KPL) {
        const int j = j0 + g;
        const bool live = j < p.nvis && (p.mask == nullptr || p.mask[(long)b * p.n + j] != 0);
        const int jr = j < p.n ? j : p.n - 1;                  // (rows past the end: a valid address, the value is not used)
        float kf[VEC], vf[VEC];
        load_vec<T>(kv + (long)jr * 2 * inner, kf);
        load_vec<T>(kv + (long)jr * 2 * inner + inner, vf);
        float part = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) part += qv[e] * kf[e];
        const float s = group_sum<CH>(part);
        if (live) {                                            // (uniform over the CH lanes of a key)
            const float mn = fmaxf(m, s);
            const float fac = fast_exp(m - mn), pj = fast_exp(s - mn);      // m = -inf at a group's first key: fac = 0
            l = l * fac + pj;
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = acc[e] * fac + pj * vf[e];
            m = mn;
        }
    }
    // merge the KPL groups: a group that saw no key has m = -inf, l = 0
    const float mt = cross_max<CH>(m);
    const float w = (m == -INFINITY) ? 0.f : fast_exp(m - mt);
    const float lt = cross_sum<CH>(l * w);
    const float inv = lt > 0.f ? 1.f / lt : 0.f;               // no visible key at all: output 0 (as the dense kernels)
    float o[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = cross_sum<CH>(acc[e] * w) * inv;
    if (g == 0) store_vec<T>(static_cast<T*>(p.out) + (long)b * inner + (long)h * HS + c * VEC, o);
    if (lane == 0) p.lse[bh] = lt > 0.f ? mt + logf(lt) : 0.f;
}

template <typename T, int HS>
__global__ __launch_bounds__(256) void attn_pool_bwd_kernel(AttnPoolParams p) {
    constexpr int VEC = Elem<T>::VEC, CH = HS / VEC, KPL = 64 / CH;
    const int lane = lane_id();
    const long bh = (long)blockIdx.x * 4 + wave_id();
    if (bh >= (long)p.batch * p.heads) return;
    const int b = (int)(bh / p.heads), h = (int)(bh % p.heads);
    const int inner = p.heads * HS;
    const int g = lane / CH, c = lane % CH;
    const long hoff = (long)h * HS + c * VEC;
    const T* kv = static_cast<const T*>(p.kv) + (long)b * p.n * 2 * inner + hoff;
    T* dkv = static_cast<T*>(p.dkv) + (long)b * p.n * 2 * inner + hoff;
    float qv[VEC], dov[VEC], ov[VEC];
    load_vec<T>(static_cast<const T*>(p.q) + (long)b * inner + hoff, qv);
    load_vec<T>(static_cast<const T*>(p.dout) + (long)b * inner + hoff, dov);
    load_vec<T>(static_cast<const T*>(p.out) + (long)b * inner + hoff, ov);
    float dpart = 0.f;
#pragma unroll
    for (int e = 0; e < VEC; ++e) dpart += dov[e] * ov[e];
    const float delta = group_sum<CH>(dpart);                  // rowsum(dO o) = rowsum(P dP)
    const float lse = p.lse[bh];
    float dq[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) dq[e] = 0.f;
    for (int j0 = 0; j0 < p.n; j0 += KPL) {                    // every row of dkv is written (hidden keys: zeros)
        const int j = j0 + g;
        const bool inb = j < p.n;
        const bool live = j < p.nvis && (p.mask == nullptr || p.mask[(long)b * p.n + (inb ? j : 0)] != 0);
        const int jr = inb ? j : p.n - 1;
        float kf[VEC], vf[VEC];
        load_vec<T>(kv + (long)jr * 2 * inner, kf);
        load_vec<T>(kv + (long)jr * 2 * inner + inner, vf);
        float sp = 0.f, dp = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { sp += qv[e] * kf[e]; dp += dov[e] * vf[e]; }
        const float s = group_sum<CH>(sp) * p.scale;
        const float dpj = group_sum<CH>(dp);
        const float pj = live ? fast_exp(s - lse) : 0.f;
        const float ds = pj * (dpj - delta) * p.scale;         // d loss / d (q . k_j)
        float dk[VEC], dv[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            dq[e] += ds * kf[e];
            dk[e] = ds * qv[e];
            dv[e] = pj * dov[e];
        }
        if (inb) {
            store_vec<T>(dkv + (long)j * 2 * inner, dk);
            store_vec<T>(dkv + (long)j * 2 * inner + inner, dv);
        }
    }
    float dqt[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) dqt[e] = cross_sum<CH>(dq[e]);
    if (g == 0) store_vec<T>(static_cast<T*>(p.dq) + (long)b * inner + hoff, dqt);
}

}  // namespace xc
