// attention.h -- fused multi-head attention for the transformer blocks (reference Attention.forward,
// x_clip.py:213-245): softmax(scale * q k^T, key padding mask) v, forward and backward, never materialising
// the [b, h, n, n] score tensor.  Works directly on the packed QKV projection output [b, n, 3, h, 64] and
// writes the merged-head layout [b, n, h*64] the out-projection GEMM consumes -- no head split/merge copies.
//
// Shape of the computation (gfx950, wave64, v_mfma_f32_32x32x16_bf16 / 32x32x2_f32):
//   * one wave owns 32 query rows (forward, dQ) or 32 key rows (dK/dV); NW waves per work-group share the
//     staged 64-row tiles of the other side (K, V / Q, dO) in LDS.
//   * the score tile is computed TRANSPOSED (S^T = K Q^T, mfma A = K fragment, B = Q fragment), so a lane
//     holds 16 scores of ONE query (column = lane & 31): the row max / row sum are 15 in-lane ops + one
//     cross-half shuffle, and the softmax statistics m, l and the rescale factor are per-lane scalars.
//   * the probabilities go straight from the accumulator registers into the next MFMA as its B operand
//     (contraction slot (half, j) <-> key (j&3) + 8*(j>>2) + 4*half (+16*blk): A and B only have to agree on
//     the slot -> key map); the matching A operand is read from a TRANSPOSED LDS image of V (keys contiguous),
//     built with a register transpose while staging.
//   * O^T / dQ^T / dK^T / dV^T accumulate in registers (2 x 16 per 64-wide head), are scaled per lane and
//     leave through an LDS staging tile as coalesced 16-byte row stores.
//   * softmax in fp32 with the running max (reference: fp32 softmax, x_clip.py:238); masked keys get
//     probability exactly 0 (reference: masked_fill(-finfo.max) then softmax -> 0 as well).
//   * backward = two kernels (dQ over query tiles, dK/dV over key tiles), each recomputing its score tile from
//     Q, K and the saved log-sum-exp: no atomics, deterministic.
// Head width: 64 (the reference default and the value in every BASELINE config) is what every layout above is built for.  Heads of
// up to 128 dimensions (reference Attention accepts any dim_head, x_clip.py:201-212) run through the same kernels with NH = 2: a
// 128-wide head is two 64-wide halves -- two K / V (Q / dO) tile sets, scores accumulated over both halves' k-blocks, twice the
// output accumulators -- and the packed layout is [b, n, 3, h, 128].  Narrower widths are zero-padded by the caller (functional.py).
#pragma once
#include "common.h"

namespace xc {

constexpr int ATT_DH = 64;
constexpr float ATT_NEG = -3.0e38f;

struct AttnParams {
    const void* qkv;             // [batch, n, 3, heads, 64]   (wide heads: 128)
    const unsigned char* mask;   // [batch, n], 1 = attend, or null
    void* out;                   // [batch, n, heads*64]
    float* lse;                  // [batch, heads, n]   log-sum-exp of the scaled, masked scores
    const void* dout;            // [batch, n, heads*64]
    float* delta;                // [batch, heads, n]   sum_d dO * O
    void* dqkv;                  // [batch, n, 3, heads, 64]
    int batch, n, heads;
    float scale;
    int chunks;                  // row chunks (of NW*32) per (batch, head)
    int causal;                  // != 0: key j is visible to query i only if j <= i (reference Attention.forward, x_clip.py:231-234)
    int stagger_10ns;            // head-resident kernels: start delay of a CU's second work-group (attention3.h a3_stagger), 0 = none
    int first_round;             // ... applied to work-groups [0, first_round) = the first dispatch round: 2 x the device's CUs
    // attention dropout (reference Attention.dropout on the softmax probabilities, x_clip.py:241): the tiled kernels of this file only.
    // keep-mask of probability (batch, head, query i, key j) = drop_hash(drop_seed, ((batch * heads + head) * n + i) * n + j) >= drop_thresh
    uint32_t drop_thresh;        // 0: no dropout
    float drop_scale;            // 1 / (1 - p)
    uint64_t drop_seed;
};

template <typename T>
struct AttCfg {
    static constexpr int VEC = Elem<T>::VEC;
    static constexpr int KB = 2 * VEC;            // contraction elements per k-block
    static constexpr int DKB = ATT_DH / KB;       // k-blocks along the head dim
    static constexpr int NKB = 32 / KB;           // k-blocks along a 32-row sub-tile
    static constexpr int LD = 64 + VEC;           // padded LDS row, both tile orientations
    static constexpr int TILE = 64 * LD;
    static constexpr int CH = 64 / VEC;           // 16-byte chunks per 64-wide row
};

// rows [r0, r0+64) x 64 columns of X (row stride ldx) -> LDS tile[row][col]; rows >= nrows are zero
template <typename T, int NT>
XC_DEV void stage_rows(T* tile, const T* X, long ldx, int r0, int nrows, int tid) {
    typedef AttCfg<T> C;
    for (int u = tid; u < 64 * C::CH; u += NT) {
        const int row = u / C::CH, c = u % C::CH;
        st16(tile + row * C::LD + c * C::VEC, (r0 + row < nrows) ? ld16(X + (long)(r0 + row) * ldx + c * C::VEC) : zero16());
    }
}
// same source -> LDS tile[col][row] (rows contiguous), via 4-row register transposes
template <typename T, int NT>
XC_DEV void stage_rows_transposed(T* tile, const T* X, long ldx, int r0, int nrows, int tid) {
    typedef AttCfg<T> C;
    for (int u = tid; u < 16 * C::CH; u += NT) {
        const int rg = u / C::CH, c = u % C::CH;
        u32x4 r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + rg * 4 + i;
            r[i] = (row < nrows) ? ld16(X + (long)row * ldx + c * C::VEC) : zero16();
        }
        tr4_store(tile, C::LD, c * C::VEC, rg * 4, r);
    }
}

// accumulator registers of one 32x32 tile -> B operand of the next MFMA (k-block blk of the 32 rows)
XC_DEV u32x4 frag_from_acc(const f32x16& acc, int blk, bf16_t*) {
    u32x4 f;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        f[w] = f2bf_pk(acc[8 * blk + 2 * w], acc[8 * blk + 2 * w + 1]);            // (one v_cvt_pk_bf16_f32 per pair)
    return f;
}
XC_DEV u32x4 frag_from_acc(const f32x16& acc, int blk, float*) {
    u32x4 f = {f2u(acc[4 * blk]), f2u(acc[4 * blk + 1]), f2u(acc[4 * blk + 2]), f2u(acc[4 * blk + 3])};
    return f;
}
// matching A operand from a transposed tile row (one output feature, rows contiguous): the 32-row sub-tile
// starting at `base`, k-block blk, lane half h  ->  rows (j&3) + 8*(j>>2) + 4h + 16*blk (bf16) / j + 4h + 8*blk (fp32)
XC_DEV u32x4 load_tr_frag(const bf16_t* rowp, int base, int blk, int h) {
    const u32x2 lo = *reinterpret_cast<const u32x2*>(rowp + base + 16 * blk + 4 * h);
    const u32x2 hi = *reinterpret_cast<const u32x2*>(rowp + base + 16 * blk + 4 * h + 8);
    u32x4 f = {lo[0], lo[1], hi[0], hi[1]};
    return f;
}
XC_DEV u32x4 load_tr_frag(const float* rowp, int base, int blk, int h) { return ld16(rowp + base + 8 * blk + 4 * h); }

// acc (rows = feature d, col = this lane's row) -> staging tile -> coalesced rows of a [.., 64]-wide destination
template <typename T, int NW>
XC_DEV void store_rows_via_lds(T* Os, const f32x16 (&acc)[2], float mul, T* dst, long ldd, int row0, int nrows, int lane,
                               int wave) {
    typedef AttCfg<T> C;
    const int c31 = lane & 31;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            Os[(wave * 32 + c31) * C::LD + db * 32 + mfma_row(r, lane)] = from_f32<T>(acc[db][r] * mul);
    sync();
    for (int u = lane; u < 32 * C::CH; u += 64) {
        const int row = u / C::CH, c = u % C::CH;
        if (row0 + row < nrows) st16(dst + (long)(row0 + row) * ldd + c * C::VEC, ld16(Os + (wave * 32 + row) * C::LD + c * C::VEC));
    }
    sync();
}

// ---- forward ------------------------------------------------------------------------------------------------
template <typename T, int NW, int NH = 1>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(AttnParams p) {
    typedef AttCfg<T> C;
    constexpr int NT = NW * 64;
    constexpr int DH = ATT_DH * NH;                    // head width: NH halves of 64
    XC_LDS_DYNAMIC(lds);
    T* Ks = reinterpret_cast<T*>(lds);                 // NH x [64 keys][LD]
    T* Vt = Ks + NH * C::TILE;                         // NH x [64 d][LD]  (keys contiguous)
    T* Os = Vt + NH * C::TILE;                         // [NW*32][LD]
    unsigned char* Ms = reinterpret_cast<unsigned char*>(Os + NW * 32 * C::LD);   // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c31 = lane & 31;
    const int logical = xcd_remap(blockIdx.x, p.batch * p.heads * p.chunks);
    const int qc = logical % p.chunks, bh = logical / p.chunks;
    const int hh = bh % p.heads, bi = bh / p.heads;
    const int n = p.n;
    const long ldq = 3L * p.heads * DH;
    const T* Qb = reinterpret_cast<const T*>(p.qkv) + (long)bi * n * ldq + hh * DH;
    const T* Kb = Qb + (long)p.heads * DH;
    const T* Vb = Kb + (long)p.heads * DH;
    const int q0 = (qc * NW + wave) * 32;
    const int qrow = q0 + c31;
    const int qld = qrow < n ? qrow : n - 1;
    const int qlim = p.causal ? qrow : 0x7fffffff;         // last key this lane's query may attend to
    u32x4 qf[NH][C::DKB];
#pragma unroll
    for (int e = 0; e < NH; ++e)
#pragma unroll
        for (int kb = 0; kb < C::DKB; ++kb) qf[e][kb] = ld16(Qb + (long)qld * ldq + e * ATT_DH + kb * C::KB + h * C::VEC);

    f32x16 o[NH][2];
#pragma unroll
    for (int e = 0; e < NH; ++e)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[e][db][r] = 0.f;
    float m = ATT_NEG, l = 0.f;

    for (int kt0 = 0; kt0 < n; kt0 += 64) {
        sync();                                        // the previous tile has been consumed by every wave
#pragma unroll
        for (int e = 0; e < NH; ++e) {
            stage_rows<T, NT>(Ks + e * C::TILE, Kb + e * ATT_DH, ldq, kt0, n, tid);
            stage_rows_transposed<T, NT>(Vt + e * C::TILE, Vb + e * ATT_DH, ldq, kt0, n, tid);
        }
        if (tid < 64) Ms[tid] = (kt0 + tid < n) && (p.mask == nullptr || p.mask[(long)bi * n + kt0 + tid] != 0);
        sync();
        f32x16 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
            for (int e = 0; e < NH; ++e)
#pragma unroll
                for (int kb = 0; kb < C::DKB; ++kb)
                    s[t] = mma_kblock(ld16(Ks + e * C::TILE + (t * 32 + c31) * C::LD + kb * C::KB + h * C::VEC), qf[e][kb], s[t], (T*)nullptr);
        }
        float mx = ATT_NEG;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kj = t * 32 + mfma_row(r, lane);
                    const float sv = (Ms[kj] && kt0 + kj <= qlim) ? s[t][r] * p.scale : ATT_NEG;
                s[t][r] = sv;
                mx = fmaxf(mx, sv);
            }
        mx = fmaxf(mx, shfl_xor(mx, 32));
        const float m_new = fmaxf(m, mx);
        const float alpha = fast_exp(m - m_new);
        float rs = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float pv = (s[t][r] > 0.5f * ATT_NEG) ? fast_exp(s[t][r] - m_new) : 0.f;
                rs += pv;                                                          // (the normaliser sums the UNdropped probabilities)
                if (p.drop_thresh) {
                    const uint64_t e = ((uint64_t)bh * n + (uint32_t)qld) * n + (uint32_t)(kt0 + t * 32 + mfma_row(r, lane));
                    pv = drop_hash(p.drop_seed, e) >= p.drop_thresh ? pv * p.drop_scale : 0.f;
                }
                s[t][r] = pv;
            }
        rs += shfl_xor(rs, 32);
        l = l * alpha + rs;
        m = m_new;
#pragma unroll
        for (int e = 0; e < NH; ++e)
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[e][db][r] *= alpha;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int blk = 0; blk < C::NKB; ++blk) {
                const u32x4 pf = frag_from_acc(s[t], blk, (T*)nullptr);
#pragma unroll
                for (int e = 0; e < NH; ++e)
#pragma unroll
                    for (int db = 0; db < 2; ++db)
                        o[e][db] = mma_kblock(load_tr_frag(Vt + e * C::TILE + (db * 32 + c31) * C::LD, t * 32, blk, h), pf, o[e][db], (T*)nullptr);
            }
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    T* out = reinterpret_cast<T*>(p.out) + (long)bi * n * p.heads * DH + hh * DH;
#pragma unroll
    for (int e = 0; e < NH; ++e)
        store_rows_via_lds<T, NW>(Os, o[e], inv, out + e * ATT_DH, (long)p.heads * DH, q0, n, lane, wave);
    if (h == 0 && qrow < n) p.lse[((long)bi * p.heads + hh) * n + qrow] = m + logf(l);
}

// ---- delta_i = sum_d dO[i, d] O[i, d] per (batch, head, row) ----------------------------------------------
template <typename T, int NH = 1>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* __restrict__ o, const T* __restrict__ dout,
                                                         float* __restrict__ delta, int batch, int n, int heads) {
    constexpr int VEC = Elem<T>::VEC, LPH = NH * ATT_DH / VEC;     // lanes per head
    const int lane = lane_id();
    const long row = (long)blockIdx.x * 4 + wave_id();
    if (row >= (long)batch * n) return;
    const int width = heads * NH * ATT_DH;
    for (int c0 = 0; c0 < width / VEC; c0 += 64) {
        const int c = c0 + lane;
        float acc = 0.f;
        if (c < width / VEC) {
            float a[VEC], b[VEC];
            load_vec<T>(o + row * width + c * VEC, a);
            load_vec<T>(dout + row * width + c * VEC, b);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc += a[k] * b[k];
        }
#pragma unroll
        for (int s = 1; s < LPH; s <<= 1) acc += shfl_xor(acc, s);
        if (c < width / VEC && (lane % LPH) == 0) {
            const int hh = c / LPH;
            const long bi = row / n, qi = row % n;
            delta[(bi * heads + hh) * n + qi] = acc;
        }
    }
}

// ---- dQ -----------------------------------------------------------------------------------------------------
template <typename T, int NW, int NH = 1>
__global__ __launch_bounds__(NW * 64) void attn_dq_kernel(AttnParams p) {
    typedef AttCfg<T> C;
    constexpr int NT = NW * 64;
    constexpr int DH = ATT_DH * NH;
    XC_LDS_DYNAMIC(lds);
    T* Ks = reinterpret_cast<T*>(lds);                 // NH x [64 keys][LD]
    T* Kt = Ks + NH * C::TILE;                         // NH x [64 d][LD]
    T* Vs = Kt + NH * C::TILE;                         // NH x [64 keys][LD]
    T* Os = Vs + NH * C::TILE;                         // [NW*32][LD]
    unsigned char* Ms = reinterpret_cast<unsigned char*>(Os + NW * 32 * C::LD);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c31 = lane & 31;
    const int logical = xcd_remap(blockIdx.x, p.batch * p.heads * p.chunks);
    const int qc = logical % p.chunks, bh = logical / p.chunks;
    const int hh = bh % p.heads, bi = bh / p.heads;
    const int n = p.n;
    const long ldq = 3L * p.heads * DH, ldo = (long)p.heads * DH;
    const T* Qb = reinterpret_cast<const T*>(p.qkv) + (long)bi * n * ldq + hh * DH;
    const T* Kb = Qb + (long)p.heads * DH;
    const T* Vb = Kb + (long)p.heads * DH;
    const T* dOb = reinterpret_cast<const T*>(p.dout) + (long)bi * n * ldo + hh * DH;
    const int q0 = (qc * NW + wave) * 32;
    const int qrow = q0 + c31;
    const int qld = qrow < n ? qrow : n - 1;
    const int qlim = p.causal ? qrow : 0x7fffffff;         // last key this lane's query may attend to
    u32x4 qf[NH][C::DKB], dof[NH][C::DKB];
#pragma unroll
    for (int e = 0; e < NH; ++e)
#pragma unroll
        for (int kb = 0; kb < C::DKB; ++kb) {
            qf[e][kb] = ld16(Qb + (long)qld * ldq + e * ATT_DH + kb * C::KB + h * C::VEC);
            dof[e][kb] = ld16(dOb + (long)qld * ldo + e * ATT_DH + kb * C::KB + h * C::VEC);
        }
    const float lse_q = p.lse[((long)bi * p.heads + hh) * n + qld];
    const float delta_q = p.delta[((long)bi * p.heads + hh) * n + qld];
    f32x16 dq[NH][2];
#pragma unroll
    for (int e = 0; e < NH; ++e)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[e][db][r] = 0.f;

    for (int kt0 = 0; kt0 < n; kt0 += 64) {
        sync();
#pragma unroll
        for (int e = 0; e < NH; ++e) {
            stage_rows<T, NT>(Ks + e * C::TILE, Kb + e * ATT_DH, ldq, kt0, n, tid);
            stage_rows_transposed<T, NT>(Kt + e * C::TILE, Kb + e * ATT_DH, ldq, kt0, n, tid);
            stage_rows<T, NT>(Vs + e * C::TILE, Vb + e * ATT_DH, ldq, kt0, n, tid);
        }
        if (tid < 64) Ms[tid] = (kt0 + tid < n) && (p.mask == nullptr || p.mask[(long)bi * n + kt0 + tid] != 0);
        sync();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int e = 0; e < NH; ++e)
#pragma unroll
                for (int kb = 0; kb < C::DKB; ++kb) {
                    s = mma_kblock(ld16(Ks + e * C::TILE + (t * 32 + c31) * C::LD + kb * C::KB + h * C::VEC), qf[e][kb], s, (T*)nullptr);
                    dp = mma_kblock(ld16(Vs + e * C::TILE + (t * 32 + c31) * C::LD + kb * C::KB + h * C::VEC), dof[e][kb], dp, (T*)nullptr);
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kj = t * 32 + mfma_row(r, lane);
                    const float pv = (Ms[kj] && kt0 + kj <= qlim) ? fast_exp(s[r] * p.scale - lse_q) : 0.f;
                float dpr = dp[r];
                if (p.drop_thresh) {                                           // d loss / d P = mask / (1 - p) o (dO V^T)
                    const uint64_t e = ((uint64_t)bh * n + (uint32_t)qld) * n + (uint32_t)(kt0 + kj);
                    dpr = drop_hash(p.drop_seed, e) >= p.drop_thresh ? dpr * p.drop_scale : 0.f;
                }
                s[r] = pv * (dpr - delta_q) * p.scale;                         // dS^T (already times the q scale)
            }
#pragma unroll
            for (int blk = 0; blk < C::NKB; ++blk) {
                const u32x4 df = frag_from_acc(s, blk, (T*)nullptr);
#pragma unroll
                for (int e = 0; e < NH; ++e)
#pragma unroll
                    for (int db = 0; db < 2; ++db)
                        dq[e][db] = mma_kblock(load_tr_frag(Kt + e * C::TILE + (db * 32 + c31) * C::LD, t * 32, blk, h), df, dq[e][db], (T*)nullptr);
            }
        }
    }
    T* dQ = reinterpret_cast<T*>(p.dqkv) + (long)bi * n * ldq + hh * DH;
#pragma unroll
    for (int e = 0; e < NH; ++e) store_rows_via_lds<T, NW>(Os, dq[e], 1.0f, dQ + e * ATT_DH, ldq, q0, n, lane, wave);
}

// ---- dK, dV -------------------------------------------------------------------------------------------------
template <typename T, int NW, int NH = 1>
__global__ __launch_bounds__(NW * 64) void attn_dkv_kernel(AttnParams p) {
    typedef AttCfg<T> C;
    constexpr int NT = NW * 64;
    constexpr int DH = ATT_DH * NH;
    XC_LDS_DYNAMIC(lds);
    T* Qs = reinterpret_cast<T*>(lds);                 // NH x [64 q][LD]
    T* Qt = Qs + NH * C::TILE;                         // NH x [64 d][LD]
    T* dOs = Qt + NH * C::TILE;                        // NH x [64 q][LD]
    T* dOt = dOs + NH * C::TILE;                       // NH x [64 d][LD]
    // (wide heads: eight fp32 tiles are 139 KB, so the output staging tile lies over the first tiles once the loop is done)
    T* Os = NH == 1 ? dOt + C::TILE : Qs;              // [NW*32][LD]
    float* Ls = reinterpret_cast<float*>(dOt + NH * C::TILE + (NH == 1 ? NW * 32 * C::LD : 0));   // [64] lse of the staged queries
    float* Ds = Ls + 64;                               // [64] delta
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c31 = lane & 31;
    const int logical = xcd_remap(blockIdx.x, p.batch * p.heads * p.chunks);
    const int kc = logical % p.chunks, bh = logical / p.chunks;
    const int hh = bh % p.heads, bi = bh / p.heads;
    const int n = p.n;
    const long ldq = 3L * p.heads * DH, ldo = (long)p.heads * DH;
    const T* Qb = reinterpret_cast<const T*>(p.qkv) + (long)bi * n * ldq + hh * DH;
    const T* Kb = Qb + (long)p.heads * DH;
    const T* Vb = Kb + (long)p.heads * DH;
    const T* dOb = reinterpret_cast<const T*>(p.dout) + (long)bi * n * ldo + hh * DH;
    const float* lse = p.lse + ((long)bi * p.heads + hh) * n;
    const float* delta = p.delta + ((long)bi * p.heads + hh) * n;
    const int k0 = (kc * NW + wave) * 32;
    const int krow = k0 + c31;
    const int kld = krow < n ? krow : n - 1;
    const bool kvalid = krow < n && (p.mask == nullptr || p.mask[(long)bi * n + kld] != 0);
    const int kmin = p.causal ? krow : 0;                  // first query that may attend to this lane's key
    u32x4 kf[NH][C::DKB], vf[NH][C::DKB];
#pragma unroll
    for (int e = 0; e < NH; ++e)
#pragma unroll
        for (int kb = 0; kb < C::DKB; ++kb) {
            kf[e][kb] = ld16(Kb + (long)kld * ldq + e * ATT_DH + kb * C::KB + h * C::VEC);
            vf[e][kb] = ld16(Vb + (long)kld * ldq + e * ATT_DH + kb * C::KB + h * C::VEC);
        }
    f32x16 dk[NH][2], dv[NH][2];
#pragma unroll
    for (int e = 0; e < NH; ++e)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[e][db][r] = 0.f; dv[e][db][r] = 0.f; }

    for (int qt0 = 0; qt0 < n; qt0 += 64) {
        sync();
#pragma unroll
        for (int e = 0; e < NH; ++e) {
            stage_rows<T, NT>(Qs + e * C::TILE, Qb + e * ATT_DH, ldq, qt0, n, tid);
            stage_rows_transposed<T, NT>(Qt + e * C::TILE, Qb + e * ATT_DH, ldq, qt0, n, tid);
            stage_rows<T, NT>(dOs + e * C::TILE, dOb + e * ATT_DH, ldo, qt0, n, tid);
            stage_rows_transposed<T, NT>(dOt + e * C::TILE, dOb + e * ATT_DH, ldo, qt0, n, tid);
        }
        if (tid < 64) {
            const bool v = qt0 + tid < n;
            Ls[tid] = v ? lse[qt0 + tid] : 0.f;
            Ds[tid] = v ? delta[qt0 + tid] : 0.f;
        }
        sync();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int e = 0; e < NH; ++e)
#pragma unroll
                for (int kb = 0; kb < C::DKB; ++kb) {
                    s = mma_kblock(ld16(Qs + e * C::TILE + (t * 32 + c31) * C::LD + kb * C::KB + h * C::VEC), kf[e][kb], s, (T*)nullptr);
                    dp = mma_kblock(ld16(dOs + e * C::TILE + (t * 32 + c31) * C::LD + kb * C::KB + h * C::VEC), vf[e][kb], dp, (T*)nullptr);
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = t * 32 + mfma_row(r, lane);
                const float pv = (kvalid && qt0 + ql < n && qt0 + ql >= kmin) ? fast_exp(s[r] * p.scale - Ls[ql]) : 0.f;
                float keep = 1.f;
                if (p.drop_thresh) {
                    const int qi = qt0 + ql < n ? qt0 + ql : n - 1;
                    const uint64_t e = ((uint64_t)bh * n + (uint32_t)qi) * n + (uint32_t)kld;
                    keep = drop_hash(p.drop_seed, e) >= p.drop_thresh ? p.drop_scale : 0.f;
                }
                s[r] = pv * keep;                                              // P o mask / (1 - p): what multiplied V in the forward
                dp[r] = pv * (dp[r] * keep - Ds[ql]) * p.scale;                // dS (times the q scale)
            }
#pragma unroll
            for (int blk = 0; blk < C::NKB; ++blk) {
                const u32x4 pf = frag_from_acc(s, blk, (T*)nullptr);
                const u32x4 df = frag_from_acc(dp, blk, (T*)nullptr);
#pragma unroll
                for (int e = 0; e < NH; ++e)
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        dv[e][db] = mma_kblock(load_tr_frag(dOt + e * C::TILE + (db * 32 + c31) * C::LD, t * 32, blk, h), pf, dv[e][db], (T*)nullptr);
                        dk[e][db] = mma_kblock(load_tr_frag(Qt + e * C::TILE + (db * 32 + c31) * C::LD, t * 32, blk, h), df, dk[e][db], (T*)nullptr);
                    }
            }
        }
    }
    if (NH > 1) sync();                                // the staging tile lies over Qs / Qt: every wave must be through with them
    T* dK = reinterpret_cast<T*>(p.dqkv) + (long)bi * n * ldq + (long)p.heads * DH + hh * DH;
    T* dV = dK + (long)p.heads * DH;
#pragma unroll
    for (int e = 0; e < NH; ++e) {
        store_rows_via_lds<T, NW>(Os, dk[e], 1.0f, dK + e * ATT_DH, ldq, k0, n, lane, wave);
        store_rows_via_lds<T, NW>(Os, dv[e], 1.0f, dV + e * ATT_DH, ldq, k0, n, lane, wave);
    }
}

template <typename T, int NW, int NH = 1>
constexpr int attn_fwd_lds_bytes() { return (2 * NH * AttCfg<T>::TILE + NW * 32 * AttCfg<T>::LD) * (int)sizeof(T) + 64; }
template <typename T, int NW, int NH = 1>
constexpr int attn_dq_lds_bytes() { return (3 * NH * AttCfg<T>::TILE + NW * 32 * AttCfg<T>::LD) * (int)sizeof(T) + 64; }
template <typename T, int NW, int NH = 1>
constexpr int attn_dkv_lds_bytes() { return (4 * NH * AttCfg<T>::TILE + (NH == 1 ? NW * 32 * AttCfg<T>::LD : 0)) * (int)sizeof(T) + 512; }

}  // namespace xc
