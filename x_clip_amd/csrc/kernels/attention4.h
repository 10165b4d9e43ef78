// attention4.h -- head-resident bf16 attention for WIDE heads: head slots of 128 features = two 64-wide halves (reference Attention accepts
// any dim_head, x_clip.py:201-212; ViT widths 80 / 96 / 128 run zero-padded to the slot), n <= 288.
//
// Through round 3 such heads took the tiled kernels of attention.h (a 64-key tile staged per step with register transposes, one barrier pair
// per tile: "correctness paths, not tuned"), and round 4's first timing of them (profiles/r04_e_wide_heads.log, b = 1024, n = 257, 4 heads of
// 128) read 1019 us forward / 4397 us backward where the 64-wide head-resident kernels of attention3.h take 369 / 916 for the same model
// width.  These kernels are attention3.h's design with two half-images per operand: every operand of the head is brought into LDS once by LDS
// DMA as swizzled 128-byte-row images (half e of operand X at X_img + e * npad * 128, each half exactly an attention3.h image, so all of its
// fragment readers apply to a half by its base address), one barrier, and an uninterrupted loop over 32-row sub-tiles.  A score is a
// contraction over 128 features = two chains of four MFMAs; an output accumulator is [2 halves][2 x 32 features].
//   forward : K and V images (4 halves x npad x 128 B = 147 KB at n = 257 ... 288: ONE work-group per CU), wave w owns queries [32 w, 32 w + 32);
//             a short tail (n = 32 q + 1, 2) is processed cooperatively as in attention3.h, its (tail query x tail key) pair included.
//   backward: one kernel, delta in the prologue, phase A (K, V images: dQ of the wave's query blocks), phase B (Q, dO images DMA'd over them:
//             dK, dV of the wave's key blocks), four waves (one per SIMD: 64 + 128 accumulator and ~100 fragment / score registers per lane
//             in phase B do not fit 256 -- the overflow lives in the AGPR half of the 512-entry file the single wave of a SIMD owns).
// Numerics are those of attention3.h (fp32 online softmax in the base-2 domain of the scaled scores, probabilities and dS rounded to bf16
// for the second MFMA, masked keys get probability exactly 0).  No dropout here (the tiled kernels keep it).
#pragma once
#include "attention3.h"

namespace xc {

constexpr int A4_NH = 2;                                       // 64-wide halves per head slot
constexpr int A4_DH = A4_NH * ATT_DH;
constexpr int A4_TAIL_REC = 2 + A4_DH;                         // floats per (wave, tail row) in the forward: m, l, O[128]

// both halves of an operand: rows [0, npad) x 128 features of X (row stride ldx) -> X_img, X_img + himg
XC_DEV void a4_dma_images(unsigned char* img, int himg, const bf16_t* X, long ldx, int n, int npad, int wave, int nwaves, int lane) {
#pragma unroll
    for (int e = 0; e < A4_NH; ++e) a3_dma_image(img + e * himg, X + e * ATT_DH, ldx, n, npad, wave, nwaves, lane);
}
// this lane's row fragments of row r of X, both halves, straight from global memory
XC_DEV void a4_row_frags(const bf16_t* X, long ldx, int r, int lane, u32x4 (&f)[A4_NH][4]) {
#pragma unroll
    for (int e = 0; e < A4_NH; ++e) a3_row_frags(X + e * ATT_DH, ldx, r, lane, f[e]);
}
XC_DEV void a4_zero(f32x16 (&a)[A4_NH][2]) {
#pragma unroll
    for (int e = 0; e < A4_NH; ++e)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) a[e][db][r] = 0.f;
}
// S^T (+)= X_t (rows of sub-tile t, both halves) . f^T : the 8-MFMA contraction over 128 features
XC_DEV void a4_scores(const unsigned char* img, int himg, int t, const u32x4 (&f)[A4_NH][4], int lane, f32x16& s) {
#pragma unroll
    for (int e = 0; e < A4_NH; ++e)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) s = mma_kblock(a3_row_frag(img + e * himg, t, kb, lane), f[e][kb], s, (bf16_t*)nullptr);
}
// acc^T[d, col] += X_t^T (columns of sub-tile t, both halves) . w  (w: the packed 32 x 32 weights of the sub-tile, accumulator layout)
XC_DEV void a4_accumulate(const unsigned char* img, int himg, int t, const f32x16& w, int lane, f32x16 (&acc)[A4_NH][2]) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const u32x4 wf = a2_pack_acc(w, blk);
#pragma unroll
        for (int e = 0; e < A4_NH; ++e)
#pragma unroll
            for (int db = 0; db < 2; ++db) acc[e][db] = mma_kblock(a3_col_frag(img + e * himg, t, blk, db, lane), wf, acc[e][db], (bf16_t*)nullptr);
    }
}

// one 32-key sub-tile of the online-softmax forward (attention3.h a3_fwd_step with two halves)
template <bool MASKED, bool CAUSAL>
XC_DEV void a4_fwd_step(const unsigned char* Ks, const unsigned char* Vs, int himg, const unsigned char* Ms, int t, const u32x4 (&qf)[A4_NH][4],
                        float scale2, int lane, int qidx, f32x16 (&o)[A4_NH][2], float& m2, float& l) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    a4_scores(Ks, himg, t, qf, lane, s);
    bool valid[16];
    float mx = ATT_NEG;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        valid[r] = (!MASKED || Ms[t * 32 + mfma_row(r, lane)] != 0) && (!CAUSAL || t * 32 + mfma_row(r, lane) <= qidx);
        mx = fmaxf(mx, valid[r] ? s[r] : ATT_NEG);
    }
    mx = fmaxf(mx, shfl_xor(mx, 32));
    const float m_new = fmaxf(m2, mx > 0.5f * ATT_NEG ? mx * scale2 : ATT_NEG);
    const float alpha = fast_exp2(m2 - m_new);
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float pv = fast_exp2(s[r] * scale2 - m_new);
        if (MASKED || CAUSAL) pv = valid[r] ? pv : 0.f;
        s[r] = pv;
        rs += pv;
    }
    rs += shfl_xor(rs, 32);
    l = l * alpha + rs;
    m2 = m_new;
#pragma unroll
    for (int e = 0; e < A4_NH; ++e)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[e][db][r] *= alpha;
    a4_accumulate(Vs, himg, t, s, lane, o);
}
template <bool CAUSAL>
XC_DEV void a4_fwd_step_auto(const unsigned char* Ks, const unsigned char* Vs, int himg, const unsigned char* Ms, int t, const u32x4 (&qf)[A4_NH][4],
                             float scale2, int lane, int qlo, f32x16 (&o)[A4_NH][2], float& m2, float& l) {
    const int qidx = qlo + (lane & 31);
    if (CAUSAL) {
        if (t * 32 > qlo + 31) return;
        if (t * 32 + 31 > qlo) { a4_fwd_step<true, true>(Ks, Vs, himg, Ms, t, qf, scale2, lane, qidx, o, m2, l); return; }
    }
    const bool kv = Ms[t * 32 + (lane & 31)] != 0;
    if (wave_all(kv)) a4_fwd_step<false, false>(Ks, Vs, himg, Ms, t, qf, scale2, lane, qidx, o, m2, l);
    else if (wave_any(kv)) a4_fwd_step<true, false>(Ks, Vs, himg, Ms, t, qf, scale2, lane, qidx, o, m2, l);
}

XC_HOST_DEV int a4_fwd_waves(int n) { const int b = a3_waves(n); return b < 8 ? b : 8; }
// K and V images, key validity, tail partials (results leave straight from the registers)
inline int attn4_fwd_lds_bytes(int n) {
    const int npad = (n + 31) & ~31, nw = a4_fwd_waves(n);
    return 2 * A4_NH * npad * 128 + npad + nw * A3_TAIL_MAX * A4_TAIL_REC * 4 + 64;
}
inline int attn4_bwd_lds_bytes(int n) {
    const int npad = (n + 31) & ~31;
    return 2 * A4_NH * npad * 128 + npad + 2 * npad * 4 + (a3_coop_tail(n) ? a3_bwd_waves(n) * A3_TAIL_MAX * 2 * A4_DH * 4 : 0) + 64;
}

// ---- forward ----------------------------------------------------------------------------------------------------------
template <bool CAUSAL>
__global__ __launch_bounds__(512) void attn4_fwd_kernel(AttnParams p) {
    XC_LDS_DYNAMIC(lds);
    const int n = p.n, npad = (n + 31) & ~31, himg = npad * 128;
    unsigned char* Ks = lds;
    unsigned char* Vs = Ks + A4_NH * himg;
    unsigned char* Ms = Vs + A4_NH * himg;                     // [npad] key validity
    float* Ts = reinterpret_cast<float*>(Ms + npad);           // [nwaves][A3_TAIL_MAX][A4_TAIL_REC] tail partials
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c31 = lane & 31;
    const int wave = uniform(tid >> 6), nwaves = blockDim.x >> 6;
    const int bh = xcd_remap(blockIdx.x, p.batch * p.heads);
    const int hh = bh % p.heads, bi = bh / p.heads;
    const long ldq = 3L * p.heads * A4_DH, ldo = (long)p.heads * A4_DH;
    const bf16_t* Qb = reinterpret_cast<const bf16_t*>(p.qkv) + (long)bi * n * ldq + hh * A4_DH;
    const bf16_t* Kb = Qb + (long)p.heads * A4_DH;
    const bf16_t* Vb = Kb + (long)p.heads * A4_DH;
    bf16_t* out = reinterpret_cast<bf16_t*>(p.out) + (long)bi * n * ldo + hh * A4_DH;
    float* lse_out = p.lse + ((long)bi * p.heads + hh) * n;
    a4_dma_images(Ks, himg, Kb, ldq, n, npad, wave, nwaves, lane);
    a4_dma_images(Vs, himg, Vb, ldq, n, npad, wave, nwaves, lane);
    a3_key_validity(Ms, p.mask, (long)bi * n, n, npad);
    const bool coop = a3_coop_tail(n);
    const int tail0 = (n >> 5) << 5, ntail = n & 31;
    u32x4 qf[A4_NH][4];
    f32x16 o[A4_NH][2];
    wait_vmem();
    sync();
    const int nsub = npad >> 5;
    const float scale2 = p.scale * 1.4426950408889634f;
    if (coop) {                                                // tail queries x this wave's share of the key sub-tiles
        const int trow = tail0 + c31 < n ? tail0 + c31 : n - 1;
        a4_row_frags(Qb, ldq, trow, lane, qf);
        a4_zero(o);
        float m = ATT_NEG, l = 0.f;
        for (int t = wave; t < nsub; t += nwaves) a4_fwd_step_auto<CAUSAL>(Ks, Vs, himg, Ms, t, qf, scale2, lane, tail0, o, m, l);
        if (c31 < ntail) {
            float* rec = Ts + ((long)wave * A3_TAIL_MAX + c31) * A4_TAIL_REC;
            if (h == 0) { rec[0] = m; rec[1] = l; }
#pragma unroll
            for (int e = 0; e < A4_NH; ++e) a3_put_col(rec + 2 + e * ATT_DH, o[e], lane);
        }
    }
    sync();                                                    // (the tail partials are in)
    // (at most eight waves -- two per SIMD, 256 registers each: 64 output accumulators + 32 query fragment registers do not fit the 168 a
    //  ninth wave would leave -- so a 9-block sequence gives wave 0 a second query block)
    const int nblk = a3_waves(n);
    for (int qb = wave; qb < nblk; qb += nwaves) {
        const int q0 = qb * 32, qrow = q0 + c31;
        const int qld = qrow < n ? qrow : n - 1;
        a4_row_frags(Qb, ldq, qld, lane, qf);
        a4_zero(o);
        float m = ATT_NEG, l = 0.f;
        for (int t = 0; t < nsub; ++t) a4_fwd_step_auto<CAUSAL>(Ks, Vs, himg, Ms, t, qf, scale2, lane, q0, o, m, l);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
        for (int e = 0; e < A4_NH; ++e) a3_store_rows_direct(o[e], out + e * ATT_DH, ldo, q0, n, lane, inv);   // (no LDS left for a staging tile)
        if (h == 0 && qrow < n) lse_out[qrow] = m * 0.6931471805599453f + logf(l);    // m is in log2 units
    }
    if (coop && wave == 0) {                                   // merge the nwaves partials of every tail row; lane = feature d of a half
        for (int q = 0; q < ntail; ++q) {
            float M = ATT_NEG;
            for (int w = 0; w < nwaves; ++w) M = fmaxf(M, Ts[((long)w * A3_TAIL_MAX + q) * A4_TAIL_REC]);
            float L = 0.f, acc[A4_NH] = {0.f, 0.f};
            for (int w = 0; w < nwaves; ++w) {
                const float* rec = Ts + ((long)w * A3_TAIL_MAX + q) * A4_TAIL_REC;
                const float f = fast_exp2(rec[0] - M);
                L += rec[1] * f;
#pragma unroll
                for (int e = 0; e < A4_NH; ++e) acc[e] += rec[2 + e * ATT_DH + lane] * f;
            }
#pragma unroll
            for (int e = 0; e < A4_NH; ++e) out[(long)(tail0 + q) * ldo + e * ATT_DH + lane] = f2bf(L > 0.f ? acc[e] / L : 0.f);
            if (lane == 0) lse_out[tail0 + q] = M * 0.6931471805599453f + logf(L);
        }
    }
}

// ---- backward ---------------------------------------------------------------------------------------------------------
// phase A body: dQ^T += K^T dS^T of key sub-tile t for the 32 queries whose fragments are (qf, dof)
template <bool CAUSAL>
XC_DEV void a4_bwd_dq_step(const unsigned char* Ks, const unsigned char* Vs, int himg, const unsigned char* Ms, int t, const u32x4 (&qf)[A4_NH][4],
                           const u32x4 (&dof)[A4_NH][4], float lse2_q, float delta_q, float scale2, int lane, int qidx, bool masked,
                           f32x16 (&dq)[A4_NH][2]) {
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = -delta_q; }       // (dP - delta out of the MFMA chain)
    a4_scores(Ks, himg, t, qf, lane, s);
    a4_scores(Vs, himg, t, dof, lane, dp);
    if (masked || CAUSAL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kj = t * 32 + mfma_row(r, lane);
            const float pv = (Ms[kj] && (!CAUSAL || kj <= qidx)) ? fast_exp2(s[r] * scale2 - lse2_q) : 0.f;
            s[r] = pv * dp[r];                                           // dS^T / scale
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = fast_exp2(s[r] * scale2 - lse2_q) * dp[r];
    }
    a4_accumulate(Ks, himg, t, s, lane, dq);
}
// phase B body: dK^T, dV^T of query sub-tile t for the 32 keys whose fragments are (kf, vf)
template <bool CAUSAL>
XC_DEV void a4_bwd_dkv_step(const unsigned char* Qs, const unsigned char* dOs, int himg, const float* Ls2, const float* Ds, int t, int n,
                            const u32x4 (&kf)[A4_NH][4], const u32x4 (&vf)[A4_NH][4], bool kvalid, float scale2, int lane, int kidx, bool masked,
                            f32x16 (&dk)[A4_NH][2], f32x16 (&dv)[A4_NH][2]) {
    const int h = lane >> 5;
    f32x16 s, dp;
    float l2[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u32x4 a = ld16(Ls2 + t * 32 + 8 * q + 4 * h), b = ld16(Ds + t * 32 + 8 * q + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) { l2[4 * q + e] = u2f(a[e]); s[4 * q + e] = 0.f; dp[4 * q + e] = -u2f(b[e]); }
    }
    a4_scores(Qs, himg, t, kf, lane, s);
    a4_scores(dOs, himg, t, vf, lane, dp);
    if (masked || CAUSAL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ql = t * 32 + mfma_row(r, lane);
            float pv = fast_exp2(s[r] * scale2 - l2[r]);
            pv = (kvalid && ql < n && (!CAUSAL || ql >= kidx)) ? pv : 0.f;
            s[r] = pv;                                                   // P
            dp[r] = pv * dp[r];                                          // dS / scale
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = fast_exp2(s[r] * scale2 - l2[r]);
            dp[r] = s[r] * dp[r];
        }
    }
    a4_accumulate(dOs, himg, t, s, lane, dv);
    a4_accumulate(Qs, himg, t, dp, lane, dk);
}

template <bool CAUSAL>
__global__ __launch_bounds__(256) void attn4_bwd_kernel(AttnParams p) {
    XC_LDS_DYNAMIC(lds);
    const int n = p.n, npad = (n + 31) & ~31, himg = npad * 128;
    unsigned char* R0 = lds;                                   // K, then Q  (two halves each)
    unsigned char* R1 = R0 + A4_NH * himg;                     // V, then dO
    unsigned char* Ms = R1 + A4_NH * himg;                     // [npad] key validity
    float* Ls = reinterpret_cast<float*>(Ms + npad);           // [npad] lse log2(e) per query
    float* Ds = Ls + npad;                                     // [npad] delta per query
    float* Tp = Ds + npad;                                     // [nwaves][A3_TAIL_MAX][2 * 128] tail partials: dQ (phase A), dK | dV (phase B)
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c31 = lane & 31;
    const int wave = uniform(tid >> 6), nwaves = blockDim.x >> 6;
    const int bh = xcd_remap(blockIdx.x, p.batch * p.heads);
    const int hh = bh % p.heads, bi = bh / p.heads;
    const long ldq = 3L * p.heads * A4_DH, ldo = (long)p.heads * A4_DH;
    const bf16_t* Qb = reinterpret_cast<const bf16_t*>(p.qkv) + (long)bi * n * ldq + hh * A4_DH;
    const bf16_t* Kb = Qb + (long)p.heads * A4_DH;
    const bf16_t* Vb = Kb + (long)p.heads * A4_DH;
    const bf16_t* dOb = reinterpret_cast<const bf16_t*>(p.dout) + (long)bi * n * ldo + hh * A4_DH;
    const bf16_t* Ob = reinterpret_cast<const bf16_t*>(p.out) + (long)bi * n * ldo + hh * A4_DH;
    bf16_t* dQ = reinterpret_cast<bf16_t*>(p.dqkv) + (long)bi * n * ldq + hh * A4_DH;
    bf16_t* dK = dQ + (long)p.heads * A4_DH;
    bf16_t* dV = dK + (long)p.heads * A4_DH;
    a4_dma_images(R0, himg, Kb, ldq, n, npad, wave, nwaves, lane);
    a4_dma_images(R1, himg, Vb, ldq, n, npad, wave, nwaves, lane);
    a3_key_validity(Ms, p.mask, (long)bi * n, n, npad);
    const bool coop = a3_coop_tail(n);
    const int tail0 = (n >> 5) << 5, ntail = n & 31;
    const int nblk = a3_waves(n);                              // 32-row blocks owned by single waves (without a cooperative tail)
    const int nsub = npad >> 5;
    const float scale2 = p.scale * 1.4426950408889634f;
    // delta_i = sum_d dO[i, d] O[i, d] over 128 features and lse_i log2(e): lane (i = c31, half-wave h) covers 64 of them
    for (int blk = wave; blk < nsub; blk += nwaves) {
        const int row_ = blk * 32 + c31;
        const int rl = row_ < n ? row_ : n - 1;
        const float lse_r = p.lse[((long)bi * p.heads + hh) * n + rl];
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float a[8], b[8];
            load_vec<bf16_t>(Ob + (long)rl * ldo + h * 64 + c * 8, a);
            load_vec<bf16_t>(dOb + (long)rl * ldo + h * 64 + c * 8, b);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += a[k] * b[k];
        }
        acc += shfl_xor(acc, 32);
        if (h == 0) {
            Ds[row_] = row_ < n ? acc : 0.f;
            Ls[row_] = row_ < n ? lse_r * 1.4426950408889634f : 0.f;
        }
    }
    wait_vmem();
    sync();
    uint32_t plain_bits = 0;                                   // bit t: every key of sub-tile t is valid (npad <= 288: 9 sub-tiles)
    for (int t = 0; t < nsub; ++t) plain_bits |= wave_all(Ms[t * 32 + c31] != 0) ? (1u << t) : 0u;

    // ---- phase A: dQ^T[d, query] for the wave's query blocks, streaming the key sub-tiles of the K / V images ----
    u32x4 f0[A4_NH][4], f1[A4_NH][4];                          // the block's own rows: (Q, dO) in phase A, (K, V) in phase B
    f32x16 g0[A4_NH][2], g1[A4_NH][2];                         // dQ in phase A; dK, dV in phase B
    if (coop) {                                                // tail queries first: this wave's share of the key sub-tiles
        const int trow = tail0 + c31 < n ? tail0 + c31 : n - 1;
        a4_row_frags(Qb, ldq, trow, lane, f0);
        a4_row_frags(dOb, ldo, trow, lane, f1);
        a4_zero(g0);
        const float lq = Ls[tail0 + c31], dl = Ds[tail0 + c31];
        for (int t = wave; t < nsub; t += nwaves)
            if (!CAUSAL || t * 32 <= tail0 + 31)
                a4_bwd_dq_step<CAUSAL>(R0, R1, himg, Ms, t, f0, f1, lq, dl, scale2, lane, tail0 + c31, !((plain_bits >> t) & 1u), g0);
        if (c31 < ntail) {
#pragma unroll
            for (int e = 0; e < A4_NH; ++e) a3_put_col(Tp + ((long)wave * A3_TAIL_MAX + c31) * 2 * A4_DH + e * ATT_DH, g0[e], lane);
        }
    }
    for (int rb = wave; rb < nblk; rb += nwaves) {
        const int row = rb * 32 + c31;
        const int rl = row < n ? row : n - 1;
        a4_row_frags(Qb, ldq, rl, lane, f0);
        a4_row_frags(dOb, ldo, rl, lane, f1);
        a4_zero(g0);
        const float lse_q = Ls[row], delta_q = Ds[row];
        const int tend = CAUSAL ? (rb + 1 < nsub ? rb + 1 : nsub) : nsub;      // key sub-tiles above the diagonal contribute nothing
        for (int t = 0; t < tend; ++t)
            a4_bwd_dq_step<CAUSAL>(R0, R1, himg, Ms, t, f0, f1, lse_q, delta_q, scale2, lane, row, !((plain_bits >> t) & 1u), g0);
#pragma unroll
        for (int e = 0; e < A4_NH; ++e) a3_store_rows_direct(g0[e], dQ + e * ATT_DH, ldq, rb * 32, n, lane, p.scale);
    }
    sync();                                                    // every wave is done with the K / V images; the tail partials are complete
    a4_dma_images(R0, himg, Qb, ldq, n, npad, wave, nwaves, lane);
    a4_dma_images(R1, himg, dOb, ldo, n, npad, wave, nwaves, lane);
    if (coop && wave == 0) {                                   // tail dQ = sum of the waves' partials; lane = feature d of a half
        for (int q = 0; q < ntail; ++q)
#pragma unroll
            for (int e = 0; e < A4_NH; ++e) {
                float acc = 0.f;
                for (int w = 0; w < nwaves; ++w) acc += Tp[((long)w * A3_TAIL_MAX + q) * 2 * A4_DH + e * ATT_DH + lane];
                dQ[(long)(tail0 + q) * ldq + e * ATT_DH + lane] = f2bf(acc * p.scale);
            }
    }
    wait_vmem();
    sync();                                                    // Q / dO images in place; Tp may be reused

    // ---- phase B: dK^T, dV^T for the wave's key blocks, streaming the query sub-tiles of the Q / dO images ----
    for (int rb = wave; rb < nblk; rb += nwaves) {
        const int row = rb * 32 + c31;
        const int rl = row < n ? row : n - 1;
        a4_row_frags(Kb, ldq, rl, lane, f0);
        a4_row_frags(Vb, ldq, rl, lane, f1);
        const bool kvalid = Ms[row] != 0;
        a4_zero(g0);
        a4_zero(g1);
        const bool keys_plain = wave_all(kvalid);              // (uniform: no padding among this block's keys)
        for (int t = CAUSAL ? rb : 0; t < nsub; ++t)           // (query sub-tiles below the diagonal see none of these keys)
            a4_bwd_dkv_step<CAUSAL>(R0, R1, himg, Ls, Ds, t, n, f0, f1, kvalid, scale2, lane, row, !(keys_plain && t * 32 + 32 <= n), g0, g1);
#pragma unroll
        for (int e = 0; e < A4_NH; ++e) {
            a3_store_rows_direct(g0[e], dK + e * ATT_DH, ldq, rb * 32, n, lane, p.scale);
            a3_store_rows_direct(g1[e], dV + e * ATT_DH, ldq, rb * 32, n, lane);
        }
    }
    if (coop) {                                                // tail keys: this wave's share of the query sub-tiles
        const int trow = tail0 + c31 < n ? tail0 + c31 : n - 1;
        a4_row_frags(Kb, ldq, trow, lane, f0);
        a4_row_frags(Vb, ldq, trow, lane, f1);
        const bool tvalid = Ms[tail0 + c31] != 0;
        a4_zero(g0);
        a4_zero(g1);
        for (int t = wave; t < nsub; t += nwaves)
            a4_bwd_dkv_step<CAUSAL>(R0, R1, himg, Ls, Ds, t, n, f0, f1, tvalid, scale2, lane, tail0 + c31, true, g0, g1);
        if (c31 < ntail) {
            float* rec = Tp + ((long)wave * A3_TAIL_MAX + c31) * 2 * A4_DH;
#pragma unroll
            for (int e = 0; e < A4_NH; ++e) {
                a3_put_col(rec + e * ATT_DH, g0[e], lane);
                a3_put_col(rec + A4_DH + e * ATT_DH, g1[e], lane);
            }
        }
        sync();
        if (wave == 0) {
            for (int q = 0; q < ntail; ++q)
#pragma unroll
                for (int e = 0; e < A4_NH; ++e) {
                    float ak = 0.f, av = 0.f;
                    for (int w = 0; w < nwaves; ++w) {
                        const float* rec = Tp + ((long)w * A3_TAIL_MAX + q) * 2 * A4_DH;
                        ak += rec[e * ATT_DH + lane];
                        av += rec[A4_DH + e * ATT_DH + lane];
                    }
                    dK[(long)(tail0 + q) * ldq + e * ATT_DH + lane] = f2bf(ak * p.scale);
                    dV[(long)(tail0 + q) * ldq + e * ATT_DH + lane] = f2bf(av);
                }
        }
    }
}

}  // namespace xc
