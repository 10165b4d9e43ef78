// attention5.h -- the head-resident attention backward as ONE pass over the (query block, key block) pairs.
//
// attention3.h's backward visits every pair twice: phase A (dQ for the wave's queries) and phase B (dK, dV for the wave's keys) each
// recompute S and dP and their exponentials -- 28 MFMAs and two rounds of score arithmetic per 32 x 32 pair.  Here a pair is computed
// once, by the wave that owns its KEY block (dK, dV in registers as in phase B: 16 MFMAs), and its dS goes -- rounded to bf16, as the
// dK product uses it -- through a 4 KiB LDS tile to the wave that owns its QUERY block, which adds dS K to its dQ accumulators
// (4 MFMAs): 20 MFMAs and one round of score arithmetic per pair.  The schedule is a rotation: in step s wave w computes the pair
// (query block (w + s) mod nb, key block w), so the nb waves work on nb different query blocks and every query-block owner finds exactly
// one new tile behind the step's barrier -- no atomics, no fp32 dQ tile in LDS, deterministic.
//
// One work-group of nb = n / 32 <= 8 waves per head: Q, dO and K images (the K image is the A operand of the dQ product) + nb exchange
// tiles = 146 KiB at n = 257.  Sequences of 32 nb + 1 tokens (the text encoder's CLS + 256) keep the tail row out of the MFMA blocks the way
// attention3.h does (a3_tail_dot / a3_tail_outer: the tail row's contribution is the accumulators' initial value); the sums over a wave's 32
// lanes that the tail's own gradients need (dQ_tail = sum_k dS k, dK_tail = sum_q dS q, dV_tail = sum_q P dO) are MFMAs whose B operand has
// one live column.  Non-causal, no dropout, 64-wide head slots; everything else stays on attention3.h.
#pragma once
#include "attention3.h"

namespace xc {

constexpr int A5_MIN_BLOCKS = 7;
constexpr int A5_TILE = 32 * 64;                               // a dS exchange tile: 32 key rows x 32 queries, bf16

// The exchange tile: row = key (64-byte pitch), 16-byte chunk c (queries 8 c .. 8 c + 7) at chunk position c ^ ((row >> 2) & 3) -- the
// 8-byte writes of a half-wave's 32 rows then spread over all banks.  The writer owns a key row; the reader wants the MFMA operand whose
// contraction index runs over the KEYS and whose lane index is the query: a transposing read, as a3_col_frag does on the operand images.
XC_DEV void a5_tile_put(unsigned char* tile, int c31, int h, int g, u32x2 v) {
    *reinterpret_cast<u32x2*>(tile + c31 * 64 + ((g ^ ((c31 >> 2) & 3)) << 4) + 8 * h) = v;
}
XC_DEV u32x4 a5_tile_frag(const unsigned char* tile, int blk, int lane) {
    const int g = lane >> 4, tt = lane & 15;
    const int r0 = 16 * blk + 4 * (g >> 1) + (tt >> 2);        // + 8 for the second half of the k-block
    const int col = 16 * (g & 1) + (tt & 3) * 4;               // query column (a multiple of 4)
    const int lo_off = r0 * 64 + (((col >> 3) ^ ((r0 >> 2) & 3)) << 4) + (col & 7) * 2;
    const int hi_off = (r0 + 8) * 64 + (((col >> 3) ^ (((r0 + 8) >> 2) & 3)) << 4) + (col & 7) * 2;
    const s16x4 lo = lds_read_tr16(tile + lo_off);
    const s16x4 hi = lds_read_tr16(tile + hi_off);
    const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
    u32x4 f = {a[0], a[1], b[0], b[1]};
    return f;
}

// (measured, b = 1024, 8 heads: n = 256 755 against 783 us, n = 257 899 against 907; n = 129 508 against 413 -- four waves in one work-group
//  leave every SIMD with one wave: only the eight-block sequences go this way; profiles/r05_l_attn_bwd_*.log)
XC_HOST_DEV bool a5_takes(int n, int causal) { return !causal && ((n & 31) == 0 || (n & 31) == 1) && (n >> 5) >= 2 && (n >> 5) <= 8; }
XC_HOST_DEV bool a5_prefers(int n) { return (n >> 5) >= A5_MIN_BLOCKS; }
inline int attn5_bwd_lds_bytes(int n) {
    const int npad = (n + 31) & ~31, nb = n >> 5;
    return 3 * npad * 128 + 2 * nb * A5_TILE + npad + 2 * npad * 4 + nb * 192 * 4 + nb * 64 * 4 + 64;
}

// <f (this lane's row fragments: 8 features per k-block for its half), row>: `rowp` = a row of 64 features in global memory (a
// broadcast load: every lane of a half-wave reads the same 16 bytes)
XC_DEV float a5_row_dot(const bf16_t* rowp, const u32x4 (&f)[4], int lane) {
    const int h = lane >> 5;
    float acc = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const u32x4 g = ld16(rowp + kb * 16 + h * 8);
#pragma unroll
        for (int w = 0; w < 4; ++w) acc = dot2_bf16(f[kb][w], g[w], acc);
    }
    return acc + shfl_xor(acc, 32);
}
// the MFMA B operand whose column 0 holds v[k] (k = the 32 contraction slots of a 32-row block, one value per lane c31 = k in `mine`)
// and whose other 31 columns are zero: the lanes exchange their values through 32 floats of LDS (`sc`, this wave's own)
XC_DEV void a5_column_operand(float* sc, float mine, int lane, u32x4 (&bf)[2]) {
    const int c31 = lane & 31, h = lane >> 5;
    wave_sync();
    if (h == 0) sc[c31] = mine;
    wave_sync();
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const u32x4 lo = ld16(sc + 16 * blk + 4 * h), hi = ld16(sc + 16 * blk + 8 + 4 * h);
        const u32x4 v = {f2bf_pk(u2f(lo[0]), u2f(lo[1])), f2bf_pk(u2f(lo[2]), u2f(lo[3])), f2bf_pk(u2f(hi[0]), u2f(hi[1])), f2bf_pk(u2f(hi[2]), u2f(hi[3]))};
        bf[blk] = c31 == 0 ? v : zero16();
    }
    wave_sync();
}
// out[d] (64 floats, lane c31 = 0 of both halves writes) = sum_k X^T[d, k] v[k] over the 32 rows of sub-tile t of image X
XC_DEV void a5_weighted_row_sum(const unsigned char* X, int t, float* sc, float mine, int lane, float* out) {
    u32x4 bf[2];
    a5_column_operand(sc, mine, lane, bf);
    f32x16 acc[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int db = 0; db < 2; ++db) acc[db] = mma_kblock(a3_col_frag(X, t, blk, db, lane), bf[blk], acc[db], (bf16_t*)nullptr);
    if ((lane & 31) == 0) a3_put_col(out, acc, lane);
}

// a wave's 32 x 64 accumulator block to global rows as WHOLE 128-byte lines: a3_store_rows_direct's store instruction carries 32 rows x 32
// bytes -- a quarter of 32 different lines, 4 requests per line at the L2 -- this one 8 rows x 128 bytes.  The rows go through `tile` (2 KiB of
// LDS private to the wave: 16 rows x 128 bytes, chunk j of row r at chunk position j ^ (r & 7)) in two halves.
XC_DEV void a5_store_rows_lines(const f32x16 (&acc)[2], bf16_t* dst, long ldd, int row0, int nrows, int lane, unsigned char* tile, float mul = 1.f) {
    const int c31 = lane & 31, h = lane >> 5, r16 = c31 & 15;
    u32x4 ch[2][2];
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        uint32_t pk[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            pk[q][0] = f2bf_pk(acc[db][4 * q] * mul, acc[db][4 * q + 1] * mul);
            pk[q][1] = f2bf_pk(acc[db][4 * q + 2] * mul, acc[db][4 * q + 3] * mul);
        }
#pragma unroll
        for (int qq = 0; qq < 4; qq += 2) {
            permlane32_swap(pk[qq][0], pk[qq + 1][0]);
            permlane32_swap(pk[qq][1], pk[qq + 1][1]);
            ch[db][qq >> 1] = u32x4{pk[qq][0], pk[qq][1], pk[qq + 1][0], pk[qq + 1][1]};       // chunk 4 db + qq + h of row c31
        }
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        wave_sync();
        if ((c31 >> 4) == hf) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int j = 4 * db + 2 * q2 + h;
                    *reinterpret_cast<u32x4*>(tile + r16 * 128 + ((j ^ (r16 & 7)) << 4)) = ch[db][q2];
                }
        }
        wave_sync();
#pragma unroll
        for (int jr = 0; jr < 2; ++jr) {
            const int r = (lane >> 3) + 8 * jr, j = lane & 7;
            const u32x4 v = *reinterpret_cast<const u32x4*>(tile + r * 128 + ((j ^ (r & 7)) << 4));
            const int row = row0 + 16 * hf + r;
            if (row < nrows) st16(dst + (long)row * ldd + j * 8, v);
        }
    }
}

// (One work-group of 146 KiB per CU: a head's phases -- images and rows in, pairs, gradients out -- are serial on its CU.  Measured on the
//  persistent twin attention7.h by leaving phases out (profiles/r06_o_attn7_abl.log, n = 256, b = 1024: 900 us = loads 250 + delta pass 115 +
//  pairs 285 + stores 194, additive; 170 us of stores here).  Starting the CUs of the first dispatch round in 2 / 4 / 8 groups a fraction of a
//  head apart -- so that one group's memory phases meet another's pairs -- changes nothing, 817 ... 835 us at every offset
//  (profiles/r06_p_attn5_stagger.log): the phases are not queueing behind each other at the memory system, each is its own latency chain.)
__global__ __launch_bounds__(512) void attn5_bwd_kernel(AttnParams p) {
    XC_LDS_DYNAMIC(lds);
    const int n = p.n, nb = n >> 5, tail = n & 31, npad = (n + 31) & ~31, nsub = npad >> 5;
    const int img = npad * 128;
    unsigned char* Qs = lds;
    unsigned char* dOs = Qs + img;
    unsigned char* Ks = dOs + img;
    unsigned char* Xs = Ks + img;                              // [2][nb] dS exchange tiles (a5_tile_put / a5_tile_frag), two steps deep
    unsigned char* Ms = Xs + 2 * nb * A5_TILE;                 // [npad] key validity
    float* Ls = reinterpret_cast<float*>(Ms + npad);           // [npad] lse log2(e) per query
    float* Ds = Ls + npad;                                     // [npad] delta per query
    float* Tp = Ds + npad;                                     // [nb][3][64] the waves' partials of the tail row's dQ | dK | dV
    float* Sc = Tp + nb * 192;                                 // [nb][64] per-wave scratch of a5_column_operand
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c31 = lane & 31;
    const int wave = uniform(tid >> 6), nwaves = nb;
    const int bh = xcd_remap(blockIdx.x, p.batch * p.heads);
    const int hh = bh % p.heads, bi = bh / p.heads;
    const long ldq = 3L * p.heads * ATT_DH, ldo = (long)p.heads * ATT_DH;
    const float scale2 = p.scale * 1.4426950408889634f;
    const int row = wave * 32 + c31;                           // this lane's key (and query) of the wave's block (< n: a full block)
    // (A persistent form -- one work-group per CU walking heads, the next head's Q / dO images requested once the last pair is through, its K
    //  image behind the last dQ product, its delta pass under this head's stores -- was built and measured SLOWER, 958 against 899 us at
    //  n = 257: the prefetch can only start at the very end of a head, and the next prologue then waits for this head's stores too
    //  (profiles/r05_m_attn_bwd_single_pass_v3_persistent_prefetch.log).  What a head costs beside its pairs -- three image DMAs, the delta
    //  pass, 36 stores per lane: 0.41 ms of the launch -- stays exposed with ONE resident work-group of 146 KiB per CU; attention3.h's two
    //  80 KiB work-groups per CU overlap some of it, which is why 29 % fewer MFMAs buy only 1 - 4 %.)
    const bf16_t* Qb = reinterpret_cast<const bf16_t*>(p.qkv) + (long)bi * n * ldq + hh * ATT_DH;
    const bf16_t* Kb = Qb + (long)p.heads * ATT_DH;
    const bf16_t* Vb = Kb + (long)p.heads * ATT_DH;
    const bf16_t* dOb = reinterpret_cast<const bf16_t*>(p.dout) + (long)bi * n * ldo + hh * ATT_DH;
    const bf16_t* Ob = reinterpret_cast<const bf16_t*>(p.out) + (long)bi * n * ldo + hh * ATT_DH;
    bf16_t* dQ = reinterpret_cast<bf16_t*>(p.dqkv) + (long)bi * n * ldq + hh * ATT_DH;
    bf16_t* dK = dQ + (long)p.heads * ATT_DH;
    bf16_t* dV = dK + (long)p.heads * ATT_DH;
    const float* const lse_h = p.lse + ((long)bi * p.heads + hh) * n;
    // Round 6, two changes of order that leave the result bit-identical (profiles/r06_q_attn5_var.log, r06_r_attn5_var_lines.log; b = 1024, 8
    // heads, one box: n = 257 1004 -> 965 us, n = 256 827 -> 803):
    //   * the delta pass's rows (O, dO of the wave's own block) are requested BEFORE the three images: loads return in order, so the row dots
    //     run while the images land instead of behind them;
    //   * dQ / dK / dV leave as whole 128-byte lines (a5_store_rows_lines).
    // The measurement build keeps the round-5 forms selectable for the A/B: p.chunks = 1 quarter-line stores, 2 images first, 4 no stores
    // (timing only).  (Also measured: the non-temporal hint on the quarter-line stores, 1005 -> 1143 ... 1200 us -- partial lines written
    // through to HBM; the tail key's rows -- this wave's Q / dO rows in operand layout, V's tail row -- requested with the delta rows instead
    // of behind the barrier: 961 - 965 against 948 - 958 us, twelve more loads per lane in front of the images cost more than the round trip
    // they hide, profiles/r06_w_attn5_tail_rows_prefetch_not_kept.log.)
#ifdef XCLIP_MEASURE
    const int var = uniform(p.chunks);
#else
    constexpr int var = 0;
#endif
    auto delta_of = [&](int row_, const u32x4 (&a)[4], const u32x4 (&b)[4], float lse_r) {      // delta_i = sum_d dO[i, d] O[i, d]; lse_i log2(e)
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float fa[8], fb[8];
            unpack(a[c], fa, (bf16_t*)nullptr);
            unpack(b[c], fb, (bf16_t*)nullptr);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += fa[k] * fb[k];
        }
        acc += shfl_xor(acc, 32);
        if (h == 0) {
            Ds[row_] = row_ < n ? acc : 0.f;
            Ls[row_] = row_ < n ? lse_r * 1.4426950408889634f : 0.f;
        }
    };
    auto delta_rows = [&](int row_, u32x4 (&a)[4], u32x4 (&b)[4]) {
        const int rl = row_ < n ? row_ : n - 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            a[c] = ld16(Ob + (long)rl * ldo + h * 32 + c * 8);
            b[c] = ld16(dOb + (long)rl * ldo + h * 32 + c * 8);
        }
        return lse_h[rl];
    };
    {
        u32x4 a[4], b[4];
        float lse_r = 0.f;
        if (!(var & 2)) lse_r = delta_rows(row, a, b);
        a3_dma_image(Qs, Qb, ldq, n, npad, wave, nwaves, lane);
        a3_dma_image(dOs, dOb, ldo, n, npad, wave, nwaves, lane);
        a3_dma_image(Ks, Kb, ldq, n, npad, wave, nwaves, lane);
        a3_key_validity(Ms, p.mask, (long)bi * n, n, npad);
        if (var & 2) lse_r = delta_rows(row, a, b);
        delta_of(row, a, b, lse_r);
        for (int blk = wave + nwaves; blk < nsub; blk += nwaves) {             // (n = 32 nb + 1: the tail row's block, wave 0)
            const float lse_t = delta_rows(blk * 32 + c31, a, b);
            delta_of(blk * 32 + c31, a, b, lse_t);
        }
    }
    // the wave's own key block: K, V rows straight from global memory (L2 hits: the image DMA asks for the same lines)
    u32x4 kf[4], vf[4];
    a3_row_frags(Kb, ldq, row, lane, kf);
    a3_row_frags(Vb, ldq, row, lane, vf);
    wait_vmem();
    sync();
    const bool kvalid = Ms[row] != 0;
    const bool masked = !wave_all(kvalid);                     // (uniform) padding among this block's keys
    f32x16 dk[2], dv[2], dq[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; dq[db][r] = 0.f; }
    float* const sc = Sc + wave * 64;
    if (tail) {                                                // (uniform) n = 32 nb + 1: the tail row, sub-tile nb row 0 of the images
        const int trow = 32 * nb;
        // the tail QUERY against this wave's keys (a lane owns a key): dK, dV start at its contribution; dQ_tail's partial sum
        {
            const float st = a3_tail_dot(Qs, nb, 0, kf, lane), dpt = a3_tail_dot(dOs, nb, 0, vf, lane);
            const float pt = kvalid ? fast_exp2(st * scale2 - Ls[trow]) : 0.f;
            const float ds = pt * (dpt - Ds[trow]);            // dS / scale
            a3_tail_outer(Qs, nb, 0, ds, lane, dk);
            a3_tail_outer(dOs, nb, 0, pt, lane, dv);
            a5_weighted_row_sum(Ks, wave, sc, ds, lane, Tp + (wave * 3 + 0) * 64);          // sum_k dS[k] K[k]
        }
        // the tail KEY against this wave's queries (a lane owns a query): dQ starts at its contribution; dK_tail / dV_tail partial sums
        {
            u32x4 qf[4], dof[4];
            a3_row_frags(Qb, ldq, row, lane, qf);
            a3_row_frags(dOb, ldo, row, lane, dof);
            const float st = a3_tail_dot(Ks, nb, 0, qf, lane), dpt = a5_row_dot(Vb + (long)trow * ldq, dof, lane);
            const float pt = Ms[trow] != 0 ? fast_exp2(st * scale2 - Ls[row]) : 0.f;
            const float ds = pt * (dpt - Ds[row]);
            a3_tail_outer(Ks, nb, 0, ds, lane, dq);
            a5_weighted_row_sum(Qs, wave, sc, ds, lane, Tp + (wave * 3 + 1) * 64);          // sum_q dS[q] Q[q]
            a5_weighted_row_sum(dOs, wave, sc, pt, lane, Tp + (wave * 3 + 2) * 64);         // sum_q P[q] dO[q]
        }
    }
    // step s: wave w computes the pair (query block (w + s) mod nb, key block w) -- dK, dV, and dS into buffer s & 1 -- and, one barrier
    // later and in the same stretch of code as the NEXT pair's chains, adds the tile the owner of key block (w - s) mod nb left for its
    // own query block to dQ.  One barrier per step: the tiles of step s - 1 were consumed before it, so step s + 1 may overwrite them.
    auto produce = [&](int s) {
        const int t = wave + s < nb ? wave + s : wave + s - nb;         // the query block of this step's pair (uniform)
        unsigned char* const myX = Xs + ((s & 1) * nb + wave) * A5_TILE;
        u32x4 qa[4], da[4];
        a3_tile_rows(Qs, t, lane, qa);
        a3_tile_rows(dOs, t, lane, da);
        f32x16 sv, dp;
        float l2[16], dl[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32x4 a = ld16(Ls + t * 32 + 8 * q + 4 * h), b = ld16(Ds + t * 32 + 8 * q + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) { l2[4 * q + e] = u2f(a[e]); dl[4 * q + e] = u2f(b[e]); }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { sv[r] = 0.f; dp[r] = -dl[r]; }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            sv = mma_kblock(qa[kb], kf[kb], sv, (bf16_t*)nullptr);
            dp = mma_kblock(da[kb], vf[kb], dp, (bf16_t*)nullptr);
        }
        if (masked) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = kvalid ? fast_exp2(sv[r] * scale2 - l2[r]) : 0.f;
                sv[r] = pv;
                dp[r] = pv * dp[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sv[r] = fast_exp2(sv[r] * scale2 - l2[r]);
                dp[r] = sv[r] * dp[r];
            }
        }
        u32x4 df[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const u32x4 pf = a2_pack_acc(sv, blk);
            df[blk] = a2_pack_acc(dp, blk);
            // dS (bf16, as the dK product uses it) to the owner of query block t: row = this lane's key, quad g = queries 8 g + 4 h + 0..3
            // (the packed operand df[blk] holds exactly the quads 2 blk, 2 blk + 1)
            a5_tile_put(myX, c31, h, 2 * blk, u32x2{df[blk][0], df[blk][1]});
            a5_tile_put(myX, c31, h, 2 * blk + 1, u32x2{df[blk][2], df[blk][3]});
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                dv[db] = mma_kblock(a3_col_frag(dOs, t, blk, db, lane), pf, dv[db], (bf16_t*)nullptr);
                dk[db] = mma_kblock(a3_col_frag(Qs, t, blk, db, lane), df[blk], dk[db], (bf16_t*)nullptr);
            }
        }
    };
    auto consume = [&](int s) {
        const int pw = wave - s >= 0 ? wave - s : wave - s + nb;        // who computed (query block `wave`, key block pw)
        const unsigned char* const X = Xs + ((s & 1) * nb + pw) * A5_TILE;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const u32x4 dsf = a5_tile_frag(X, blk, lane);
#pragma unroll
            for (int db = 0; db < 2; ++db) dq[db] = mma_kblock(a3_col_frag(Ks, pw, blk, db, lane), dsf, dq[db], (bf16_t*)nullptr);
        }
    };
    produce(0);
    for (int s = 0; s < nb; ++s) {
        sync();                                                // the tiles of step s are in place; those of step s - 1 have been consumed
        if (s + 1 < nb) produce(s + 1);
        consume(s);
    }
    if (var & 4) {
#pragma unroll
        for (int db = 0; db < 2; ++db) { reg_keep(dq[db]); reg_keep(dk[db]); reg_keep(dv[db]); }
    } else if (var & 1) {
        a3_store_rows_direct(dq, dQ, ldq, wave * 32, n, lane, p.scale);
        a3_store_rows_direct(dk, dK, ldq, wave * 32, n, lane, p.scale);
        a3_store_rows_direct(dv, dV, ldq, wave * 32, n, lane);
    } else {
        // through the wave's own exchange tile of the parity no one reads any more: the tiles of step nb - 2 were consumed before the loop's last
        // barrier, and only this wave ever writes this one
        // (an opaque lane id: the store addresses are computed here, not ahead of the pair loop and carried through it in spilled registers)
        unsigned char* const mine = Xs + ((nb & 1) * nb + wave) * A5_TILE;
        const int lane_s = (int)opaque((uint32_t)lane);
        a5_store_rows_lines(dq, dQ, ldq, wave * 32, n, lane_s, mine, p.scale);
        a5_store_rows_lines(dk, dK, ldq, wave * 32, n, lane_s, mine, p.scale);
        a5_store_rows_lines(dv, dV, ldq, wave * 32, n, lane_s, mine);
    }
    if (tail && wave == 0) {                                   // the tail row's own gradients; lane = feature d (the partials are in: the loop's barriers)
        const int trow = 32 * nb;
        const float qv = bf2f(Qb[(long)trow * ldq + lane]), kv = bf2f(Kb[(long)trow * ldq + lane]), vv = bf2f(Vb[(long)trow * ldq + lane]);
        const float dov = bf2f(dOb[(long)trow * ldo + lane]);
        const float st = wave_sum(qv * kv), dpt = wave_sum(dov * vv);           // tail query x tail key
        const float pt = Ms[trow] != 0 ? fast_exp2(st * scale2 - Ls[trow]) : 0.f;
        const float ds = pt * (dpt - Ds[trow]);
        float aq = ds * kv, ak = ds * qv, av = pt * dov;
        for (int w = 0; w < nb; ++w) {
            aq += Tp[(w * 3 + 0) * 64 + lane];
            ak += Tp[(w * 3 + 1) * 64 + lane];
            av += Tp[(w * 3 + 2) * 64 + lane];
        }
        dQ[(long)trow * ldq + lane] = f2bf(aq * p.scale);
        dK[(long)trow * ldq + lane] = f2bf(ak * p.scale);
        dV[(long)trow * ldq + lane] = f2bf(av);
    }
}

}  // namespace xc
