// common.h -- element-type traits and 16-byte vector access shared by every kernel.
//
// All tensors are row-major with 16-byte aligned bases and leading dimensions; every HBM / LDS access
// the hot kernels issue is one 16-byte transaction per lane (8 bf16 or 4 fp32), the coalescing sweet
// spot on gfx950 (cdna_hip_programming.md Guideline 13).
#pragma once
#include "xc_device.h"

namespace xc {

template <typename T>
struct Elem;
template <>
struct Elem<float> {
    static constexpr int VEC = 4;            // elements per 16-byte chunk
};
template <>
struct Elem<bf16_t> {
    static constexpr int VEC = 8;
};

XC_DEV u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
XC_DEV void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
XC_DEV u32x4 zero16() {
    u32x4 z = {0u, 0u, 0u, 0u};
    return z;
}
XC_DEV float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
XC_DEV uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }

// ---- dropout (reference nn.Dropout in Attention / FeedForward, x_clip.py:185-212,241) -------------------------------------------------
// keep-mask of element `idx` of the tensor a (seed, stream) pair names: a stateless 32-bit mix (the "lowbias32" finaliser over the seed
// words and the 64-bit index), so the forward, the backward and a checkpointed re-run of the forward regenerate the SAME mask from
// three integers, and a test can rebuild it on the host (oracle/clip_oracle.py dropout_keep).  keep iff hash >= thresh = p * 2^32.
XC_HOST_DEV uint32_t drop_hash(uint64_t seed, uint64_t idx) {
    uint32_t h = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x85ebca77u) ^ ((uint32_t)idx * 0x9e3779b1u) ^ ((uint32_t)(idx >> 32) * 0xc2b2ae3du);
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
XC_HOST_DEV uint32_t drop_thresh(float p) { return p <= 0.f ? 0u : (p >= 1.f ? 0xffffffffu : (uint32_t)((double)p * 4294967296.0)); }

// 16 bytes -> VEC floats
XC_DEV void unpack(const u32x4& r, float (&f)[4], float*) {
    f[0] = u2f(r[0]); f[1] = u2f(r[1]); f[2] = u2f(r[2]); f[3] = u2f(r[3]);
}
XC_DEV void unpack(const u32x4& r, float (&f)[8], bf16_t*) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = u2f(r[i] << 16);
        f[2 * i + 1] = u2f(r[i] & 0xffff0000u);
    }
}
XC_DEV u32x4 pack(const float (&f)[4], float*) {
    u32x4 r = {f2u(f[0]), f2u(f[1]), f2u(f[2]), f2u(f[3])};
    return r;
}
XC_DEV u32x4 pack(const float (&f)[8], bf16_t*) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (uint32_t)f2bf(f[2 * i]) | ((uint32_t)f2bf(f[2 * i + 1]) << 16);
    return r;
}

// NT: the access carries the non-temporal hint (rows streamed once through a kernel: first use is last use)
template <typename T, bool NT = false>
XC_DEV void load_vec(const T* p, float (&f)[Elem<T>::VEC]) {
    unpack(NT ? ld16_nt(p) : ld16(p), f, (T*)nullptr);
}
template <typename T, bool NT = false>
XC_DEV void store_vec(T* p, const float (&f)[Elem<T>::VEC]) {
    if (NT) st16_nt(p, pack(f, (T*)nullptr)); else st16(p, pack(f, (T*)nullptr));
}

XC_DEV float to_f32(float v) { return v; }
XC_DEV float to_f32(bf16_t v) { return bf2f(v); }
template <typename T>
XC_DEV T from_f32(float v);
template <>
XC_DEV float from_f32<float>(float v) { return v; }
template <>
XC_DEV bf16_t from_f32<bf16_t>(float v) { return f2bf(v); }

// erf GELU and its derivative, as torch.nn.functional.gelu (reference x_clip.py:183).  Phi(x) and phi(x) share ONE
// exponential: erf(z) = 1 - (a1 t + .. + a5 t^5) exp(-z^2), t = 1 / (1 + p z) (Abramowitz-Stegun 7.1.26, |error| <=
// 1.5e-7, i.e. fp32 round-off class), and with z = |x| / sqrt(2) that exponential is exp(-x^2 / 2) = sqrt(2 pi) phi(x).
// ~12 VALU ops + 1 exp + 1 rcp for both, instead of an erff() call per use.
XC_DEV void gelu_parts(float x, float& cdf, float& pdf) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = fast_rcp(1.0f + 0.3275911f * z);
    const float e = fast_exp(-z * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * e;
    cdf = 0.5f * (1.0f + (x < 0.f ? -erf_abs : erf_abs));
    pdf = 0.39894228040143268f * e;
}
XC_DEV float gelu_erf(float x) {
    float cdf, pdf;
    gelu_parts(x, cdf, pdf);
    return x * cdf;
}

// Register transpose of a 4 (contraction rows) x VEC (outer elements) block: r[i] is the 16-byte chunk of
// contraction row k0+i; writes, for every outer element e, the 4 contraction-consecutive values to
// tile[(outer0 + e) * ld + k0 .. k0+3]  (8 bytes for bf16, 16 bytes for fp32).
XC_DEV void tr4_store(bf16_t* tile, int ld, int outer0, int k0, const u32x4 (&r)[4]) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {              // dword w of a chunk holds outer elements 2w (low half) and 2w+1 (high half)
        u32x2 lo, hi;
        lo[0] = (r[0][w] & 0xffffu) | (r[1][w] << 16);
        lo[1] = (r[2][w] & 0xffffu) | (r[3][w] << 16);
        hi[0] = (r[0][w] >> 16) | (r[1][w] & 0xffff0000u);
        hi[1] = (r[2][w] >> 16) | (r[3][w] & 0xffff0000u);
        *reinterpret_cast<u32x2*>(tile + (outer0 + 2 * w) * ld + k0) = lo;
        *reinterpret_cast<u32x2*>(tile + (outer0 + 2 * w + 1) * ld + k0) = hi;
    }
}
XC_DEV void tr4_store(float* tile, int ld, int outer0, int k0, const u32x4 (&r)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u32x4 v = {r[0][j], r[1][j], r[2][j], r[3][j]};
        st16(tile + (outer0 + j) * ld + k0, v);
    }
}

// One "k-block" of matrix-core work: 2*VEC contraction elements, 16 bytes per lane per operand; lane half
// h = lane>>5 supplies elements [h*VEC, (h+1)*VEC) of the block.  bf16: one v_mfma_f32_32x32x16_bf16;
// fp32: four v_mfma_f32_32x32x2_f32 (element q of both halves pairs up -- any k order is fine as long as A
// and B agree).  A rows / B cols = lane & 31; D: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
XC_DEV f32x16 mma_kblock(u32x4 a, u32x4 b, f32x16 c, bf16_t*) {
    return mfma_32x32x16_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), c);
}
XC_DEV f32x16 mma_kblock(u32x4 a, u32x4 b, f32x16 c, float*) {
#pragma unroll
    for (int q = 0; q < 4; ++q) c = mfma_32x32x2_f32(u2f(a[q]), u2f(b[q]), c);
    return c;
}
XC_DEV int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// XCD-aware, bijective work-group remap: hardware places block b on XCD b % 8; give every XCD one
// contiguous run of tiles so neighbouring tiles share operand panels in that XCD's private L2
// (cdna_hip_programming.md T1).
XC_DEV int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace xc
