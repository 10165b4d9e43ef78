// attention2.h -- the bf16 production attention kernels (forward, dQ, dK/dV) for the transformer blocks (reference
// Attention.forward, x_clip.py:213-245).  Same algorithm, interfaces and numerics as attention.h (which stays as the
// fp32 path): softmax(scale * q k^T + key mask) v with fp32 online softmax, scores never materialised, packed QKV in,
// merged heads out, deterministic two-kernel backward.  What changes is the data path on gfx950:
//
//   * every staged 64-row tile (K, V, Q, dO) lives in LDS ONCE, as [64 rows][64 d] bf16 with 128-byte rows whose 16-byte
//     chunk c of row r sits at slot c ^ rot3((r >> 1) & 7) (rot3 = rotate the 3 bits right by one).  That single image
//     serves both kinds of MFMA operand read, conflict free:
//       - "row" fragments (8 consecutive d of one row; contraction over d: S = K Q^T, dP = V dO^T) are one ds_read_b128;
//         the 16 rows of a ds_read_b128 lane group fall on 16 distinct 16-byte slots of the 256-byte bank row;
//       - "column" fragments (8 rows of one d column; contraction over keys / queries: O^T += V^T P, dQ^T += K^T dS,
//         dV^T += dO^T P, dK^T += Q^T dS) are two ds_read_b64_tr_b16 -- the hardware transposes, no transposed copy of
//         the tile and no register transposes while staging (those cost 16-way conflicting ds_write_b64 in attention.h);
//         the 4 rows x 64 bytes a half-wave reads cover the 256-byte bank row exactly once.
//   * 32-key (or 32-query) halves of a staged tile that lie entirely beyond n are skipped (n = 257 = 4 * 64 + 1 pays
//     for 9 sub-tiles instead of 10).
//   * results leave through an LDS staging tile with packed 8-byte writes and coalesced 16-byte row stores.
#pragma once
#include "attention.h"

namespace xc {

constexpr int A2_TILE_BYTES = 64 * 128;

XC_DEV int a2_rot3(int x) { return ((x & 1) << 2) | (x >> 1); }
XC_DEV int a2_slot(int row, int chunk) { return chunk ^ a2_rot3((row >> 1) & 7); }

// rows [r0, r0+64) x 64 d of X (row stride ldx elements) -> swizzled LDS image; rows >= nrows are zero
template <int NT>
XC_DEV void a2_stage(unsigned char* tile, const bf16_t* X, long ldx, int r0, int nrows, int tid) {
    for (int u = tid; u < 64 * 8; u += NT) {
        const int row = u >> 3, c = u & 7;
        const u32x4 v = (r0 + row < nrows) ? ld16(X + (long)(r0 + row) * ldx + c * 8) : zero16();
        st16(tile + row * 128 + a2_slot(row, c) * 16, v);
    }
}
// row fragment: tile row `row`, d block kb (16 d), lane half h -> d = 16 kb + 8 h + 0..7
XC_DEV u32x4 a2_row_frag(const unsigned char* tile, int row, int kb, int h) {
    return ld16(tile + row * 128 + a2_slot(row, kb * 2 + h) * 16);
}
// column fragment: rows 32 t + 16 blk + {4h + 0..3, 8 + 4h + 0..3} (the contraction slots of frag_from_acc) of the d
// columns 32 db + (lane & 31)
XC_DEV u32x4 a2_col_frag(const unsigned char* tile, int t, int blk, int db, int lane) {
    const int g = lane >> 4, tt = lane & 15;
    const int row = 32 * t + 16 * blk + 4 * (g >> 1) + (tt >> 2);
    const int col = 32 * db + 16 * (g & 1) + (tt & 3) * 4;
    const int within = (col & 7) * 2;
    const s16x4 lo = lds_read_tr16(tile + row * 128 + a2_slot(row, col >> 3) * 16 + within);
    const s16x4 hi = lds_read_tr16(tile + (row + 8) * 128 + a2_slot(row + 8, col >> 3) * 16 + within);
    const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
    u32x4 f = {a[0], a[1], b[0], b[1]};
    return f;
}
XC_DEV u32x4 a2_pack_acc(const f32x16& acc, int blk) { return frag_from_acc(acc, blk, (bf16_t*)nullptr); }

// acc[db] (rows = d = 32 db + mfma_row, col = this lane's row c31) * mul -> this wave's 32 x 64 staging block (row pitch
// 144 bytes) -> coalesced 16-byte row stores to dst rows [row0, row0+32)
XC_DEV void a2_store_rows(unsigned char* stage, const f32x16 (&acc)[2], float mul, bf16_t* dst, long ldd, int row0, int nrows,
                          int lane) {
    const int c31 = lane & 31, h = lane >> 5;
    wave_sync();                                   // a previous use of this wave's block has been read out
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32x2 w;
            w[0] = (uint32_t)f2bf(acc[db][4 * q] * mul) | ((uint32_t)f2bf(acc[db][4 * q + 1] * mul) << 16);
            w[1] = (uint32_t)f2bf(acc[db][4 * q + 2] * mul) | ((uint32_t)f2bf(acc[db][4 * q + 3] * mul) << 16);
            *reinterpret_cast<u32x2*>(stage + c31 * 144 + (db * 32 + 8 * q + 4 * h) * 2) = w;
        }
    // the block is private to this wave: no work-group barrier; a wave's LDS operations complete in order
    wave_sync();
    for (int u = lane; u < 32 * 8; u += 64) {
        const int row = u >> 3, c = u & 7;
        if (row0 + row < nrows) st16(dst + (long)(row0 + row) * ldd + c * 8, ld16(stage + row * 144 + c * 16));
    }
}

// ---- forward -------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64) void attn2_fwd_kernel(AttnParams p) {
    constexpr int NT = NW * 64;
    XC_LDS_DYNAMIC(lds);
    unsigned char* Ks = lds;                                   // [64 keys] swizzled image
    unsigned char* Vs = Ks + A2_TILE_BYTES;
    unsigned char* Os = Vs + A2_TILE_BYTES;                    // [NW][32 x 144 B] output staging
    unsigned char* Ms = Os + NW * 32 * 144;                    // [64] key validity
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c31 = lane & 31;
    const int logical = xcd_remap(blockIdx.x, p.batch * p.heads * p.chunks);
    const int qc = logical % p.chunks, bh = logical / p.chunks;
    const int hh = bh % p.heads, bi = bh / p.heads;
    const int n = p.n;
    const long ldq = 3L * p.heads * ATT_DH;
    const bf16_t* Qb = reinterpret_cast<const bf16_t*>(p.qkv) + (long)bi * n * ldq + hh * ATT_DH;
    const bf16_t* Kb = Qb + (long)p.heads * ATT_DH;
    const bf16_t* Vb = Kb + (long)p.heads * ATT_DH;
    const int q0 = (qc * NW + wave) * 32;
    const int qrow = q0 + c31;
    const int qld = qrow < n ? qrow : n - 1;
    const int qlim = p.causal ? qrow : 0x7fffffff;         // last key this lane's query may attend to
    u32x4 qf[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) qf[kb] = ld16(Qb + (long)qld * ldq + kb * 16 + h * 8);

    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m = ATT_NEG, l = 0.f;

    for (int kt0 = 0; kt0 < n; kt0 += 64) {
        sync();
        a2_stage<NT>(Ks, Kb, ldq, kt0, n, tid);
        a2_stage<NT>(Vs, Vb, ldq, kt0, n, tid);
        if (tid < 64) Ms[tid] = (kt0 + tid < n) && (p.mask == nullptr || p.mask[(long)bi * n + kt0 + tid] != 0);
        sync();
        const int nsub = (kt0 + 32 < n) ? 2 : 1;               // the second 32-key half may lie entirely beyond n
        f32x16 s[2];
        float mx = ATT_NEG;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t < nsub) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) s[t] = mma_kblock(a2_row_frag(Ks, t * 32 + c31, kb, h), qf[kb], s[t], (bf16_t*)nullptr);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kj = t * 32 + mfma_row(r, lane);
                    const float sv = (Ms[kj] && kt0 + kj <= qlim) ? s[t][r] * p.scale : ATT_NEG;
                    s[t][r] = sv;
                    mx = fmaxf(mx, sv);
                }
            }
        }
        mx = fmaxf(mx, shfl_xor(mx, 32));
        const float m_new = fmaxf(m, mx);
        const float alpha = fast_exp(m - m_new);
        float rs = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t < nsub) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = (s[t][r] > 0.5f * ATT_NEG) ? fast_exp(s[t][r] - m_new) : 0.f;
                    s[t][r] = pv;
                    rs += pv;
                }
            }
        }
        rs += shfl_xor(rs, 32);
        l = l * alpha + rs;
        m = m_new;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t < nsub) {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const u32x4 pf = a2_pack_acc(s[t], blk);
#pragma unroll
                    for (int db = 0; db < 2; ++db) o[db] = mma_kblock(a2_col_frag(Vs, t, blk, db, lane), pf, o[db], (bf16_t*)nullptr);
                }
            }
        }
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    bf16_t* out = reinterpret_cast<bf16_t*>(p.out) + (long)bi * n * p.heads * ATT_DH + hh * ATT_DH;
    a2_store_rows(Os + wave * 32 * 144, o, inv, out, (long)p.heads * ATT_DH, q0, n, lane);
    if (h == 0 && qrow < n) p.lse[((long)bi * p.heads + hh) * n + qrow] = m + logf(l);
}

// ---- dQ --------------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64) void attn2_dq_kernel(AttnParams p) {
    constexpr int NT = NW * 64;
    XC_LDS_DYNAMIC(lds);
    unsigned char* Ks = lds;
    unsigned char* Vs = Ks + A2_TILE_BYTES;
    unsigned char* Os = Vs + A2_TILE_BYTES;
    unsigned char* Ms = Os + NW * 32 * 144;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c31 = lane & 31;
    const int logical = xcd_remap(blockIdx.x, p.batch * p.heads * p.chunks);
    const int qc = logical % p.chunks, bh = logical / p.chunks;
    const int hh = bh % p.heads, bi = bh / p.heads;
    const int n = p.n;
    const long ldq = 3L * p.heads * ATT_DH, ldo = (long)p.heads * ATT_DH;
    const bf16_t* Qb = reinterpret_cast<const bf16_t*>(p.qkv) + (long)bi * n * ldq + hh * ATT_DH;
    const bf16_t* Kb = Qb + (long)p.heads * ATT_DH;
    const bf16_t* Vb = Kb + (long)p.heads * ATT_DH;
    const bf16_t* dOb = reinterpret_cast<const bf16_t*>(p.dout) + (long)bi * n * ldo + hh * ATT_DH;
    const int q0 = (qc * NW + wave) * 32;
    const int qrow = q0 + c31;
    const int qld = qrow < n ? qrow : n - 1;
    const int qlim = p.causal ? qrow : 0x7fffffff;         // last key this lane's query may attend to
    u32x4 qf[4], dof[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        qf[kb] = ld16(Qb + (long)qld * ldq + kb * 16 + h * 8);
        dof[kb] = ld16(dOb + (long)qld * ldo + kb * 16 + h * 8);
    }
    const float lse_q = p.lse[((long)bi * p.heads + hh) * n + qld];
    const float delta_q = p.delta[((long)bi * p.heads + hh) * n + qld];
    f32x16 dq[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;

    for (int kt0 = 0; kt0 < n; kt0 += 64) {
        sync();
        a2_stage<NT>(Ks, Kb, ldq, kt0, n, tid);
        a2_stage<NT>(Vs, Vb, ldq, kt0, n, tid);
        if (tid < 64) Ms[tid] = (kt0 + tid < n) && (p.mask == nullptr || p.mask[(long)bi * n + kt0 + tid] != 0);
        sync();
        const int nsub = (kt0 + 32 < n) ? 2 : 1;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t < nsub) {
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
                    const int kj = t * 32 + mfma_row(r, lane);
                    const float pv = (Ms[kj] && kt0 + kj <= qlim) ? fast_exp(s[r] * p.scale - lse_q) : 0.f;
                    s[r] = pv * (dp[r] - delta_q) * p.scale;                   // dS^T (already times the q scale)
                }
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const u32x4 df = a2_pack_acc(s, blk);
#pragma unroll
                    for (int db = 0; db < 2; ++db) dq[db] = mma_kblock(a2_col_frag(Ks, t, blk, db, lane), df, dq[db], (bf16_t*)nullptr);
                }
            }
        }
    }
    bf16_t* dQ = reinterpret_cast<bf16_t*>(p.dqkv) + (long)bi * n * ldq + hh * ATT_DH;
    a2_store_rows(Os + wave * 32 * 144, dq, 1.0f, dQ, ldq, q0, n, lane);
}

// ---- dK, dV ------------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64) void attn2_dkv_kernel(AttnParams p) {
    constexpr int NT = NW * 64;
    XC_LDS_DYNAMIC(lds);
    unsigned char* Qs = lds;
    unsigned char* dOs = Qs + A2_TILE_BYTES;
    unsigned char* Os = dOs + A2_TILE_BYTES;
    float* Ls = reinterpret_cast<float*>(Os + NW * 32 * 144);  // [64] lse of the staged queries
    float* Ds = Ls + 64;                                       // [64] delta
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c31 = lane & 31;
    const int logical = xcd_remap(blockIdx.x, p.batch * p.heads * p.chunks);
    const int kc = logical % p.chunks, bh = logical / p.chunks;
    const int hh = bh % p.heads, bi = bh / p.heads;
    const int n = p.n;
    const long ldq = 3L * p.heads * ATT_DH, ldo = (long)p.heads * ATT_DH;
    const bf16_t* Qb = reinterpret_cast<const bf16_t*>(p.qkv) + (long)bi * n * ldq + hh * ATT_DH;
    const bf16_t* Kb = Qb + (long)p.heads * ATT_DH;
    const bf16_t* Vb = Kb + (long)p.heads * ATT_DH;
    const bf16_t* dOb = reinterpret_cast<const bf16_t*>(p.dout) + (long)bi * n * ldo + hh * ATT_DH;
    const float* lse = p.lse + ((long)bi * p.heads + hh) * n;
    const float* delta = p.delta + ((long)bi * p.heads + hh) * n;
    const int k0 = (kc * NW + wave) * 32;
    const int krow = k0 + c31;
    const int kld = krow < n ? krow : n - 1;
    const bool kvalid = krow < n && (p.mask == nullptr || p.mask[(long)bi * n + kld] != 0);
    const int kmin = p.causal ? krow : 0;                  // first query that may attend to this lane's key
    u32x4 kf[4], vf[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        kf[kb] = ld16(Kb + (long)kld * ldq + kb * 16 + h * 8);
        vf[kb] = ld16(Vb + (long)kld * ldq + kb * 16 + h * 8);
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }

    for (int qt0 = 0; qt0 < n; qt0 += 64) {
        sync();
        a2_stage<NT>(Qs, Qb, ldq, qt0, n, tid);
        a2_stage<NT>(dOs, dOb, ldo, qt0, n, tid);
        if (tid < 64) {
            const bool v = qt0 + tid < n;
            Ls[tid] = v ? lse[qt0 + tid] : 0.f;
            Ds[tid] = v ? delta[qt0 + tid] : 0.f;
        }
        sync();
        const int nsub = (qt0 + 32 < n) ? 2 : 1;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t < nsub) {
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
                    const float pv = (kvalid && qt0 + ql < n && qt0 + ql >= kmin) ? fast_exp(s[r] * p.scale - Ls[ql]) : 0.f;
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
        }
    }
    bf16_t* dK = reinterpret_cast<bf16_t*>(p.dqkv) + (long)bi * n * ldq + (long)p.heads * ATT_DH + hh * ATT_DH;
    bf16_t* dV = dK + (long)p.heads * ATT_DH;
    a2_store_rows(Os + wave * 32 * 144, dk, 1.0f, dK, ldq, k0, n, lane);
    a2_store_rows(Os + wave * 32 * 144, dv, 1.0f, dV, ldq, k0, n, lane);
}

template <int NW>
constexpr int attn2_lds_bytes() { return 2 * A2_TILE_BYTES + NW * 32 * 144 + 512; }

}  // namespace xc
