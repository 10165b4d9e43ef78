// attention3.h -- head-resident bf16 attention for the sequence lengths of the CLIP encoders (n <= 288: text 257 / 78,
// vision 32 .. 288 tokens; reference Attention.forward, x_clip.py:213-245).
//
// At n = 257, d_head = 64 the fused attention is HBM / latency bound, not MFMA bound (forward: 132 KB of traffic against
// 17 MFLOP per head), so the layout of the work matters more than the inner loop:
//   * one work-group per (batch, head); wave w owns query rows (and, in the backward, key rows) [32 w, 32 w + 32);
//   * every operand of the head is brought into LDS exactly once by LDS DMA (global_load_lds, 16 B per lane, all pieces
//     in flight together) as the swizzled 128-byte-row images of attention2.h, followed by ONE barrier and an
//     uninterrupted compute loop over 32-row sub-tiles -- no per-tile load / barrier / compute round trips;
//   * the backward is ONE kernel: delta = rowsum(dO o O) is computed in the prologue, phase A accumulates dQ for the
//     wave's queries, phase B accumulates dK / dV for the wave's keys (scores are recomputed per phase, nothing is
//     exchanged between waves, no atomics, deterministic).  Q, K, V, dO and O are read once, dQ / dK / dV written once:
//     264 KB of traffic per head instead of the 363 KB (+ the delta pass) of the dq / dkv kernel pair.
// Numerics are those of attention.h / attention2.h (fp32 online softmax, probabilities rounded to bf16 for the second
// MFMA, masked keys get probability exactly 0).
#pragma once
#include "attention2.h"

namespace xc {

constexpr int A3_MAX_N = 288;

// DMA rows [0, npad) x 64 d of X (row stride ldx) into a swizzled image; rows >= n are clamped to row n-1 (finite data:
// their probabilities are exactly 0, so they contribute 0 and nothing of them is stored)
XC_DEV void a3_dma_image(unsigned char* img, const bf16_t* X, long ldx, int n, int npad, int wave, int nwaves, int lane) {
    const int pieces = npad >> 3;                              // 1 KiB = 8 rows per wave-instruction
    for (int pc = wave; pc < pieces; pc += nwaves) {
        const int row = pc * 8 + (lane >> 3);
        const int chunk = a2_slot(row, lane & 7);              // logical chunk stored at slot lane & 7 (involution)
        const int g = row < n ? row : n - 1;
        glds16(X + (long)g * ldx + chunk * 8, img + pc * 1024);
    }
}

// ---- forward ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(576) void attn3_fwd_kernel(AttnParams p) {
    XC_LDS_DYNAMIC(lds);
    const int n = p.n, npad = (n + 31) & ~31;
    unsigned char* Ks = lds;
    unsigned char* Vs = Ks + npad * 128;
    unsigned char* Ms = Vs + npad * 128;                       // [npad] key validity
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c31 = lane & 31;
    const int wave = uniform(tid >> 6), nwaves = blockDim.x >> 6;
    const int bh = xcd_remap(blockIdx.x, p.batch * p.heads);
    const int hh = bh % p.heads, bi = bh / p.heads;
    const long ldq = 3L * p.heads * ATT_DH;
    const bf16_t* Qb = reinterpret_cast<const bf16_t*>(p.qkv) + (long)bi * n * ldq + hh * ATT_DH;
    const bf16_t* Kb = Qb + (long)p.heads * ATT_DH;
    const bf16_t* Vb = Kb + (long)p.heads * ATT_DH;
    a3_dma_image(Ks, Kb, ldq, n, npad, wave, nwaves, lane);
    a3_dma_image(Vs, Vb, ldq, n, npad, wave, nwaves, lane);
    for (int k = tid; k < npad; k += blockDim.x) Ms[k] = (k < n) && (p.mask == nullptr || p.mask[(long)bi * n + k] != 0);
    const int q0 = wave * 32;
    const int qrow = q0 + c31;
    const int qld = qrow < n ? qrow : n - 1;
    u32x4 qf[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) qf[kb] = ld16(Qb + (long)qld * ldq + kb * 16 + h * 8);
    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m = ATT_NEG, l = 0.f;
    wait_vmem();
    sync();
    const int nsub = npad >> 5;
    for (int t = 0; t < nsub; ++t) {
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) s = mma_kblock(a2_row_frag(Ks, t * 32 + c31, kb, h), qf[kb], s, (bf16_t*)nullptr);
        float mx = ATT_NEG;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float sv = Ms[t * 32 + mfma_row(r, lane)] ? s[r] * p.scale : ATT_NEG;
            s[r] = sv;
            mx = fmaxf(mx, sv);
        }
        mx = fmaxf(mx, shfl_xor(mx, 32));
        const float m_new = fmaxf(m, mx);
        const float alpha = fast_exp(m - m_new);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = (s[r] > 0.5f * ATT_NEG) ? fast_exp(s[r] - m_new) : 0.f;
            s[r] = pv;
            rs += pv;
        }
        rs += shfl_xor(rs, 32);
        l = l * alpha + rs;
        m = m_new;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const u32x4 pf = a2_pack_acc(s, blk);
#pragma unroll
            for (int db = 0; db < 2; ++db) o[db] = mma_kblock(a2_col_frag(Vs, t, blk, db, lane), pf, o[db], (bf16_t*)nullptr);
        }
    }
    sync();                                                    // every wave is done with the K / V images
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    bf16_t* out = reinterpret_cast<bf16_t*>(p.out) + (long)bi * n * p.heads * ATT_DH + hh * ATT_DH;
    a2_store_rows(lds + wave * 32 * 144, o, inv, out, (long)p.heads * ATT_DH, q0, n, lane);
    if (h == 0 && qrow < n) p.lse[((long)bi * p.heads + hh) * n + qrow] = m + logf(l);
}

// ---- backward (dQ, dK, dV and delta in one kernel) ------------------------------------------------------------------------
__global__ __launch_bounds__(576) void attn3_bwd_kernel(AttnParams p) {
    XC_LDS_DYNAMIC(lds);
    const int n = p.n, npad = (n + 31) & ~31;
    const int img = npad * 128;
    unsigned char* Qs = lds;
    unsigned char* dOs = Qs + img;
    unsigned char* Ks = dOs + img;
    unsigned char* Vs = Ks + img;
    unsigned char* Ms = Vs + img;                              // [npad] key validity
    float* Ls = reinterpret_cast<float*>(Ms + npad);           // [npad] lse per query      (npad is a multiple of 32)
    float* Ds = Ls + npad;                                     // [npad] delta per query
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c31 = lane & 31;
    const int wave = uniform(tid >> 6), nwaves = blockDim.x >> 6;
    const int bh = xcd_remap(blockIdx.x, p.batch * p.heads);
    const int hh = bh % p.heads, bi = bh / p.heads;
    const long ldq = 3L * p.heads * ATT_DH, ldo = (long)p.heads * ATT_DH;
    const bf16_t* Qb = reinterpret_cast<const bf16_t*>(p.qkv) + (long)bi * n * ldq + hh * ATT_DH;
    const bf16_t* Kb = Qb + (long)p.heads * ATT_DH;
    const bf16_t* Vb = Kb + (long)p.heads * ATT_DH;
    const bf16_t* dOb = reinterpret_cast<const bf16_t*>(p.dout) + (long)bi * n * ldo + hh * ATT_DH;
    const bf16_t* Ob = reinterpret_cast<const bf16_t*>(p.out) + (long)bi * n * ldo + hh * ATT_DH;
    a3_dma_image(Ks, Kb, ldq, n, npad, wave, nwaves, lane);
    a3_dma_image(Vs, Vb, ldq, n, npad, wave, nwaves, lane);
    a3_dma_image(Qs, Qb, ldq, n, npad, wave, nwaves, lane);
    a3_dma_image(dOs, dOb, ldo, n, npad, wave, nwaves, lane);
    for (int k = tid; k < npad; k += blockDim.x) Ms[k] = (k < n) && (p.mask == nullptr || p.mask[(long)bi * n + k] != 0);
    // delta_i = sum_d dO[i, d] O[i, d] and lse_i for this wave's 32 rows: lane (i = c31, half h) covers 32 of the 64 d
    const int r0 = wave * 32;
    const int row = r0 + c31;
    const int rld = row < n ? row : n - 1;
    {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a[8], b[8];
            load_vec<bf16_t>(Ob + (long)rld * ldo + h * 32 + c * 8, a);
            load_vec<bf16_t>(dOb + (long)rld * ldo + h * 32 + c * 8, b);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += a[k] * b[k];
        }
        acc += shfl_xor(acc, 32);
        if (h == 0) {
            Ds[row] = row < n ? acc : 0.f;
            Ls[row] = row < n ? p.lse[((long)bi * p.heads + hh) * n + rld] : 0.f;
        }
    }
    wait_vmem();
    sync();
    const int nsub = npad >> 5;

    // ---- phase A: dQ^T[d, query] for the wave's queries, streaming the key sub-tiles ----
    {
        u32x4 qf[4], dof[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            qf[kb] = a2_row_frag(Qs, row, kb, h);
            dof[kb] = a2_row_frag(dOs, row, kb, h);
        }
        const float lse_q = Ls[row], delta_q = Ds[row];
        f32x16 dq[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
        for (int t = 0; t < nsub; ++t) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                s = mma_kblock(a2_row_frag(Ks, t * 32 + c31, kb, h), qf[kb], s, (bf16_t*)nullptr);
                dp = mma_kblock(a2_row_frag(Vs, t * 32 + c31, kb, h), dof[kb], dp, (bf16_t*)nullptr);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = Ms[t * 32 + mfma_row(r, lane)] ? fast_exp(s[r] * p.scale - lse_q) : 0.f;
                s[r] = pv * (dp[r] - delta_q) * p.scale;                       // dS^T (already times the q scale)
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const u32x4 df = a2_pack_acc(s, blk);
#pragma unroll
                for (int db = 0; db < 2; ++db) dq[db] = mma_kblock(a2_col_frag(Ks, t, blk, db, lane), df, dq[db], (bf16_t*)nullptr);
            }
        }
        // this wave's own K / V rows (phase B operands) leave the images before they are recycled as staging space
        u32x4 kf[4], vf[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            kf[kb] = a2_row_frag(Ks, row, kb, h);
            vf[kb] = a2_row_frag(Vs, row, kb, h);
        }
        const bool kvalid = Ms[row] != 0;
        sync();                                                // all waves are done reading the K / V images
        unsigned char* stage = Ks + wave * 32 * 144;
        bf16_t* dQ = reinterpret_cast<bf16_t*>(p.dqkv) + (long)bi * n * ldq + hh * ATT_DH;
        a2_store_rows(stage, dq, 1.0f, dQ, ldq, r0, n, lane);

        // ---- phase B: dK^T, dV^T for the wave's keys, streaming the query sub-tiles ----
        f32x16 dk[2], dv[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
        for (int t = 0; t < nsub; ++t) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                s = mma_kblock(a2_row_frag(Qs, t * 32 + c31, kb, h), kf[kb], s, (bf16_t*)nullptr);
                dp = mma_kblock(a2_row_frag(dOs, t * 32 + c31, kb, h), vf[kb], dp, (bf16_t*)nullptr);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = t * 32 + mfma_row(r, lane);
                const float pv = (kvalid && ql < n) ? fast_exp(s[r] * p.scale - Ls[ql]) : 0.f;
                s[r] = pv;                                                     // P
                dp[r] = pv * (dp[r] - Ds[ql]) * p.scale;                       // dS (times the q scale)
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const u32x4 pf = a2_pack_acc(s, blk);
                const u32x4 df = a2_pack_acc(dp, blk);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dv[db] = mma_kblock(a2_col_frag(dOs, t, blk, db, lane), pf, dv[db], (bf16_t*)nullptr);
                    dk[db] = mma_kblock(a2_col_frag(Qs, t, blk, db, lane), df, dk[db], (bf16_t*)nullptr);
                }
            }
        }
        bf16_t* dK = reinterpret_cast<bf16_t*>(p.dqkv) + (long)bi * n * ldq + (long)p.heads * ATT_DH + hh * ATT_DH;
        bf16_t* dV = dK + (long)p.heads * ATT_DH;
        a2_store_rows(stage, dk, 1.0f, dK, ldq, r0, n, lane);
        a2_store_rows(stage, dv, 1.0f, dV, ldq, r0, n, lane);
    }
}

inline int attn3_fwd_lds_bytes(int n) {
    const int npad = (n + 31) & ~31, nw = npad / 32;
    const int a = 2 * npad * 128 + npad, b = nw * 32 * 144;
    return (a > b ? a : b) + 64;
}
inline int attn3_bwd_lds_bytes(int n) {
    const int npad = (n + 31) & ~31;
    return 4 * npad * 128 + npad + 2 * npad * 4 + 64;
}

}  // namespace xc
