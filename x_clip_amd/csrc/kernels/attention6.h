// attention6.h -- the head-resident attention backward as a STREAM: one persistent work-group per CU walks heads, and nothing of a head
// is resident but its K / V block fragments ("a CU always has a second head's operands in flight").
//
// What attention5.h could not hide (profiles/r05_w_sq_attn5_bwd.txt: 41 % of the wave cycles in counted waits): a head is three image
// DMAs + a delta pass, THEN 64 block pairs, THEN 36 stores per lane, and with one 146 KiB work-group per CU those three phases of
// consecutive heads cannot overlap -- 0.41 of the 0.90 ms per text layer is the exposed load + store skeleton.  Its rotation schedule
// (wave w works on query block (w + s) mod 8 in step s) is what forces the whole Q / dO images to be resident.
//
// Here every wave still owns a KEY block (K, V row fragments and the K column fragments of the dQ product in registers for the head,
// dK / dV accumulators in registers), but all eight waves work on the SAME query block t in step t:
//   * Q_t | dO_t | O_t (12 KiB) arrive through a three-slot LDS ring, requested TWO steps ahead -- across head boundaries too -- with a
//     counted wait at the step's barrier that leaves the pieces requested in this step in flight (requested one step ahead they had to
//     land within a step: the first version ran at the DMA's loaded latency, 3.5 us per step);
//     delta_t = rowsum(dO_t o O_t) comes out of the slot (every wave, 16 v_dot2 per lane: no pass over the head, no second read);
//   * the pair (t, w) is attention5.h's: S, dP, one round of exponentials, dV, dK (16 MFMAs); dS goes through the wave's OWN 2 KiB
//     exchange tile (no barrier) back into the wave as the transposed operand, and dQ_t's partial dS K_w (4 MFMAs, K^T fragments
//     from registers) is added into a [32][65] tile with ds_add_u32 IN FIXED POINT (a lane owns a query row: consecutive banks, no
//     conflicts).  Not ds_add_f32: the first version used it and ran 6.5 x slower than attention5.h -- the fp32 LDS atomic is served one
//     lane at a time on gfx950, 192 cycles per wave-instruction against 4.1 for the integer form (tools/probes/lds_atomic_rate_probe.hip,
//     profiles/r06_c_lds_atomic_rate_probe.txt).  A row's scale is a power of two from a bound every wave computes identically:
//     |dQ_t[q, d]| <= sum_k P |dP - delta| |K| <= 2 |dO_q| max_k |V_k| max_k |K_k| (row 2-norms; sum_k P = 1, O a convex
//     combination of V rows), so 2^30 / bound never overflows an int32 whatever subset of the keys a partial sum covers, and what is
//     lost is 2^-30 of the BOUND -- far below the bf16 the result is rounded to.  Integer adds commute exactly: dQ is bit-reproducible;
//   * behind the step's barrier the tile is complete: in step t + 1 every wave converts four rows of it, stores them (whole 128-byte
//     lines) and clears them -- the head's dQ leaves 8 KiB per step instead of 32 KiB at the end;
//   * the NEXT head's K / V images are requested piece by piece during this head's steps (a wave overwrites only the rows of its own
//     block, which it has already taken into registers), its log-sum-exps and key mask wait in one register per thread;
//   * dK / dV leave from registers after the last step and stay in flight under the next head's first step.
// One barrier per step, as attention5.h.
// The 257th token: as a QUERY it is a ninth ring item and one VALU step per wave (dK / dV += its outer products; attention3.h's
// a3_tail_dot / a3_tail_outer); as a KEY it is handled by wave t in step t on the fragments that wave holds anyway (its dQ term is the
// rank-1 update ds_tail[q] k_tail the finishing pass adds).  Non-causal, no dropout, 64-wide head slots, n = 256 / 257.
#pragma once
#include "attention5.h"

namespace xc {

constexpr int A6_NB = 8;                                       // key blocks = waves
constexpr int A6_SLOT = 3 * 4096;                              // Q_t | dO_t | O_t sub-tile images of one ring item
constexpr int A6_KV_ROWS = 264;                                // 256 rows + the 8-row piece that holds the tail row
constexpr int A6_KV_IMG = A6_KV_ROWS * 128;
constexpr int A6_DQ_LD = 65;                                   // floats per row of a dQ tile: lane q, fixed d -> bank (q + d) mod 32
constexpr int A6_DQ_TILE = 32 * A6_DQ_LD;                      // floats
constexpr int A6_RING = 3;                                     // ring slots: an item is requested two steps before it is used
constexpr int A6_LDS_BYTES = A6_RING * A6_SLOT + 2 * A6_KV_IMG + A6_NB * A5_TILE + 2 * A6_DQ_TILE * 4 + 264 * 4 + 256 * 4 + A6_NB * 192 * 4 +
                             A6_NB * 64 * 4 + 128 + 128 + 272 + 64 + 256;

XC_HOST_DEV bool a6_takes(int n, int causal) { return !causal && (n == 256 || n == 257); }

// a head = three element offsets (the pointers are formed where they are used: ten 64-bit pointers per head x two heads in flight
// were 40 scalar registers the compiler spilled into vector lanes)
struct A6Head {
    long qo;                                                   // into qkv / dqkv: first element of the head's Q rows
    long oo;                                                   // into out / dout
    long lo;                                                   // into lse
    long mo;                                                   // into mask (row of the sample)
};

// per-lane byte offset of piece pc (rows 8 pc .. 8 pc + 7 of a 32-row block, 16-byte chunk lane & 7 of the row at its swizzled slot:
// attention2.h layout) inside an operand with row stride ld elements; the block's first row travels in the scalar offset
XC_DEV uint32_t a6_voff(int pc, long ld, int lane) {
    const int lr = pc * 8 + (lane >> 3);
    return ((uint32_t)lr * (uint32_t)ld + (uint32_t)a2_slot(lr, lane & 7) * 8u) * 2u;
}
// acc[db][r] += mul * X[row 0 of sub-image img][d = 32 db + mfma_row(r, lane)]   (attention3.h a3_tail_outer, accumulating)
XC_DEV void a6_tail_outer_acc(const unsigned char* img, float mul, int lane, f32x16 (&acc)[2]) {
    const int h = lane >> 5;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const u32x2 v = *reinterpret_cast<const u32x2*>(img + (4 * db + g) * 16 + 8 * h);
            acc[db][4 * g + 0] += mul * u2f(v[0] << 16);
            acc[db][4 * g + 1] += mul * u2f(v[0] & 0xffff0000u);
            acc[db][4 * g + 2] += mul * u2f(v[1] << 16);
            acc[db][4 * g + 3] += mul * u2f(v[1] & 0xffff0000u);
        }
}
// <row 0 of sub-image a, row 0 of sub-image b> (64 features; the same value in every lane)
XC_DEV float a6_rows_dot(const unsigned char* a, const unsigned char* b, int lane) {
    const int h = lane >> 5;
    float acc = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const u32x4 x = ld16(a + (kb * 2 + h) * 16), y = ld16(b + (kb * 2 + h) * 16);
#pragma unroll
        for (int w = 0; w < 4; ++w) acc = dot2_bf16(x[w], y[w], acc);
    }
    return acc + shfl_xor(acc, 32);
}

__global__ __launch_bounds__(512) void attn6_bwd_kernel(AttnParams p) {
    XC_LDS_DYNAMIC(lds);
    const int n = p.n, tail = n & 31;
    unsigned char* const Ring = lds;                           // [A6_RING][A6_SLOT]
    unsigned char* const Ks = Ring + A6_RING * A6_SLOT;              // the head's K image (rows 0 .. 263), then the next head's
    unsigned char* const Vs = Ks + A6_KV_IMG;
    unsigned char* const XW = Vs + A6_KV_IMG;                  // [8] the waves' own dS exchange tiles
    int* const DQ = reinterpret_cast<int*>(XW + A6_NB * A5_TILE);        // [2][32][65] fixed point
    float* const Ls = reinterpret_cast<float*>(DQ + 2 * A6_DQ_TILE);     // [264] lse log2(e) of the head's queries
    float* const DST = Ls + 264;                               // [256] dS / scale of (query, tail key)
    float* const Tp = DST + 256;                               // [8][3][64] partials of the tail row's dQ | dK | dV
    float* const Sc = Tp + A6_NB * 192;                        // [8][64] per-wave scratch: a5_column_operand | the step's deltas
    unsigned char* const KT = reinterpret_cast<unsigned char*>(Sc + A6_NB * 64);   // the tail key's K row, V row (128 bytes each)
    unsigned char* const VT = KT + 128;
    unsigned char* const Ms = VT + 128;                        // [264] key validity
    float* const Nrm = reinterpret_cast<float*>(Ms + 272);     // [8] max_k |K_k|^2 per wave, [8] max_k |V_k|^2
    float* const Inv = Nrm + 16;                               // [2][32] 1 / scale of the dQ tiles' rows
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c31 = lane & 31;
    const int wave = uniform(tid >> 6);
    const int total = p.batch * p.heads, G = gridDim.x;
    const long ldq = 3L * p.heads * ATT_DH, ldo = (long)p.heads * ATT_DH;
    const float scale2 = p.scale * 1.4426950408889634f, inv_scale2 = 1.0f / scale2;
    // measurement build only (XCLIP_ATTN6_ABL -> p.chunks; 0 in the product; results are garbage): 1 no atomics, 2 constant fixed-point scale,
    // 4 no delta, 8 no dQ partial at all, 16 no finishing pass, 32 no tail-key work, 64 no requests after the cold start
    const int abl = p.chunks;
    const int items = 8 + (tail ? 1 : 0);                      // ring items per head: the 8 query blocks (+ the tail rows)
    const int row = wave * 32 + c31;                           // this lane's key
    unsigned char* const myX = XW + wave * A5_TILE;
    float* const sc = Sc + wave * 64;

    const bf16_t* const QKV = reinterpret_cast<const bf16_t*>(p.qkv);
    const bf16_t* const DOUT = reinterpret_cast<const bf16_t*>(p.dout);
    const bf16_t* const OUT = reinterpret_cast<const bf16_t*>(p.out);
    bf16_t* const DQKV = reinterpret_cast<bf16_t*>(p.dqkv);
    const long kofs = (long)p.heads * ATT_DH;                  // K (V) of a row = Q + kofs (+ 2 kofs)
    auto head = [&](int bh) {
        const int hh = bh % p.heads, bi = bh / p.heads;
        A6Head H;
        H.qo = (long)bi * n * ldq + hh * ATT_DH;
        H.oo = (long)bi * n * ldo + hh * ATT_DH;
        H.lo = ((long)bi * p.heads + hh) * n;
        H.mo = (long)bi * n;
        return H;
    };
    auto logical = [&](int k) {                                // the k-th head of this work-group, or -1
        const int L = (int)blockIdx.x + k * G;
        return L < total ? xcd_remap(L, total) : -1;
    };
    // Every request goes through a buffer descriptor of the head's operand (rows past n - 1 read as zero: the tail piece covers rows
    // 256 .. 263) as an asm-issued LDS DMA the compiler does not see (xc_device.h buf_glds16_raw): a lane's offset inside a piece is ONE
    // loop-invariant register per row stride, the block's first row a scalar offset.
    const uint32_t vq = a6_voff(wave & 3, ldq, lane), vo = a6_voff(wave & 3, ldo, lane);
    const uint32_t extq = (uint32_t)(n - 1) * (uint32_t)ldq * 2u + 128u, exto = (uint32_t)(n - 1) * (uint32_t)ldo * 2u + 128u;
    // this wave's share of ring item `it` of head H into `slot`: items 0 .. 7 = the query blocks (Q and O pieces by waves 0 .. 3, dO pieces
    // by waves 4 .. 7), item 8 = the tail rows (one 8-row piece per operand, waves 0 .. 2)
    auto ring_issue = [&](const A6Head& H, int it, unsigned char* slot) {
        if (it < 8) {
            if (wave < 4) {
                buf_glds16_raw(make_rsrc(uniform_ptr(QKV + H.qo), extq), vq, (uint32_t)it * 64u * (uint32_t)ldq, slot + wave * 1024);
                buf_glds16_raw(make_rsrc(uniform_ptr(OUT + H.oo), exto), vo, (uint32_t)it * 64u * (uint32_t)ldo, slot + 8192 + wave * 1024);
            } else {
                buf_glds16_raw(make_rsrc(uniform_ptr(DOUT + H.oo), exto), vo, (uint32_t)it * 64u * (uint32_t)ldo, slot + 4096 + (wave - 4) * 1024);
            }
        } else if (wave < 3) {
            const long ld = wave == 0 ? ldq : ldo;
            const bf16_t* X = wave == 0 ? QKV + H.qo : (wave == 1 ? DOUT + H.oo : OUT + H.oo);
            buf_glds16_raw(make_rsrc(uniform_ptr(X), wave == 0 ? extq : exto), a6_voff(0, ld, lane), 512u * (uint32_t)ld, slot + wave * 4096);
        }
    };
    // one 8-row piece of this wave's OWN rows of the K / V image of head H: step 0 .. 3 -> K pieces, 4 .. 7 -> V pieces; "step 8": the tail piece
    // (requested with step 7's, so that a head's last step requests no K / V piece: behind its counted wait the images are complete)
    auto kv_issue = [&](const A6Head& H, int step) {
        if (step < 8) {
            unsigned char* img = step < 4 ? Ks : Vs;
            const bf16_t* X = QKV + H.qo + (step < 4 ? kofs : 2 * kofs);
            buf_glds16_raw(make_rsrc(uniform_ptr(X), extq), a6_voff(step & 3, ldq, lane), (uint32_t)wave * 64u * (uint32_t)ldq, img + wave * 4096 + (step & 3) * 1024);
        } else if (wave < 2) {
            buf_glds16_raw(make_rsrc(uniform_ptr(QKV + H.qo + (wave == 0 ? kofs : 2 * kofs)), extq), a6_voff(0, ldq, lane), 512u * (uint32_t)ldq,
                           (wave == 0 ? Ks : Vs) + 8 * 4096);
        }
    };

    int bh = logical(0);
    if (bh < 0) return;                                        // (uniform)
    A6Head H = head(bh);
    // ---- cold start: the first head's K / V images, ring item 0, lse / mask registers; the dQ tiles start at zero ----
#pragma unroll 1
    for (int s = 0; s < 9; ++s) kv_issue(H, s);
    ring_issue(H, 0, Ring);
    ring_issue(H, 1, Ring + A6_SLOT);
    const int me = tid < n ? tid : n - 1;
    float lse_reg = p.lse[H.lo + me];
    unsigned char mask_reg = p.mask != nullptr ? p.mask[H.mo + me] : (unsigned char)1;
    for (int u = tid; u < 2 * A6_DQ_TILE; u += 512) DQ[u] = 0;
    wait_vmem();
    sync();

    int gs = 0;                                                // the ring slot of the item this step consumes (items in order, slots mod 3)
    // how many DMA pieces this wave requests with ring item `it` / K-V step `st` (uniform per wave): the counted wait of a step
    auto ring_count = [&](int it) { return it < 8 ? (wave < 4 ? 2 : 1) : (wave < 3 ? 1 : 0); };
    auto wait_older_than = [&](int pieces) {                   // everything but the `pieces` youngest vector-memory operations has completed
        switch (pieces) {
            case 0: XC_WAIT_VMEM_LE(0); break;
            case 1: XC_WAIT_VMEM_LE(1); break;
            case 2: XC_WAIT_VMEM_LE(2); break;
            case 3: XC_WAIT_VMEM_LE(3); break;
            default: XC_WAIT_VMEM_LE(4); break;
        }
    };
    for (int k = 0;; ++k) {
        const int nbh = logical(k + 1);
        const bool has_next = nbh >= 0;                        // (uniform)
        const A6Head N = head(has_next ? nbh : bh);
        // ---- the head's resident part: K / V row fragments, K^T fragments of the dQ product, the tail key's rows, lse, mask ----
        u32x4 kf[4], vf[4], kc[2][2];
        a3_tile_rows(Ks, wave, lane, kf);
        a3_tile_rows(Vs, wave, lane, vf);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int db = 0; db < 2; ++db) kc[blk][db] = a3_col_frag(Ks, wave, blk, db, lane);
        if (wave < 2 && lane < 8) st16((wave == 0 ? KT : VT) + lane * 16, ld16((wave == 0 ? Ks : Vs) + 8 * 4096 + lane * 16));
        {                                                      // largest squared row norm of this wave's K and V rows (the fixed-point bound)
            float k2 = 0.f, v2 = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int w = 0; w < 4; ++w) { k2 = dot2_bf16(kf[kb][w], kf[kb][w], k2); v2 = dot2_bf16(vf[kb][w], vf[kb][w], v2); }
            k2 += shfl_xor(k2, 32);
            v2 += shfl_xor(v2, 32);
            k2 = wave_max(k2);
            v2 = wave_max(v2);
            if (lane == 0) { Nrm[wave] = k2; Nrm[8 + wave] = v2; }
        }
        if (tid < 264) {
            Ls[tid] = tid < n ? lse_reg * 1.4426950408889634f : 0.f;
            Ms[tid] = (tid < n) && mask_reg != 0;
        }
        lse_reg = p.lse[N.lo + me];
        mask_reg = p.mask != nullptr ? p.mask[N.mo + me] : (unsigned char)1;
        f32x16 dk[2], dv[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
        barrier_nodrain();                                     // every wave holds its fragments; KT / VT / Ls / Ms are in place
        const bool kvalid = Ms[row] != 0;
        const bool masked = !wave_all(kvalid);                 // (uniform) padding among this block's keys
        const bool tkey = tail && Ms[256] != 0;
        float kvn;                                             // max_k |K_k|^2 max_k |V_k|^2 over the head (the tail key's rows included)
        {
            const u32x4 a0 = ld16(Nrm), a1 = ld16(Nrm + 4), b0 = ld16(Nrm + 8), b1 = ld16(Nrm + 12);
            float k2 = fmaxf(fmaxf(fmaxf(u2f(a0[0]), u2f(a0[1])), fmaxf(u2f(a0[2]), u2f(a0[3]))), fmaxf(fmaxf(u2f(a1[0]), u2f(a1[1])), fmaxf(u2f(a1[2]), u2f(a1[3]))));
            float v2 = fmaxf(fmaxf(fmaxf(u2f(b0[0]), u2f(b0[1])), fmaxf(u2f(b0[2]), u2f(b0[3]))), fmaxf(fmaxf(u2f(b1[0]), u2f(b1[1])), fmaxf(u2f(b1[2]), u2f(b1[3]))));
            if (tail) {
                k2 = fmaxf(k2, a6_rows_dot(KT, KT, lane));
                v2 = fmaxf(v2, a6_rows_dot(VT, VT, lane));
            }
            kvn = k2 * v2;
        }
        // the finishing pass of a dQ tile: this lane's row 4 wave + lane / 16, features 4 (lane % 16) ... + 3
        const int frow = 4 * wave + (lane >> 4), fd = 4 * (lane & 15);
        float ktf[4] = {0.f, 0.f, 0.f, 0.f};
        if (tail) {
            const u32x2 kq = *reinterpret_cast<const u32x2*>(KT + fd * 2);
            ktf[0] = u2f(kq[0] << 16); ktf[1] = u2f(kq[0] & 0xffff0000u); ktf[2] = u2f(kq[1] << 16); ktf[3] = u2f(kq[1] & 0xffff0000u);
        }
        auto finish_tile = [&](int t) {                        // dQ rows 32 t .. 32 t + 31 out of tile t & 1, which is cleared
            int* const at = DQ + (t & 1) * A6_DQ_TILE + frow * A6_DQ_LD + fd;
            const float inv = (abl & 2) ? 1.0f / 1024.f : Inv[(t & 1) * 32 + frow];
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (float)at[e] * inv;
            if (tail) {
                const float ds = DST[32 * t + frow];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += ds * ktf[e];
            }
            const u32x2 o = {f2bf_pk(v[0] * p.scale, v[1] * p.scale), f2bf_pk(v[2] * p.scale, v[3] * p.scale)};
            *reinterpret_cast<u32x2*>(DQKV + H.qo + (long)(32 * t + frow) * ldq + fd) = o;
#pragma unroll
            for (int e = 0; e < 4; ++e) at[e] = 0;
        };

#pragma unroll 1
        for (int t = 0; t < 8; ++t) {
            unsigned char* const S = Ring + gs * A6_SLOT;
            unsigned char* const S2 = Ring + (gs == 0 ? 2 : gs - 1) * A6_SLOT;      // slot of the item two ahead = the one step t - 1 consumed
            gs = gs == 2 ? 0 : gs + 1;
            // the finished tile first (its store is OLDER than this step's requests), then the requests: ring item t + 2 (of this head, or of
            // the next), one K / V piece of the next head
            if (t > 0 && !(abl & 16)) finish_tile(t - 1);
            int pieces = 0;
            if (abl & 64) {}
            else if (t + 2 < items) { ring_issue(H, t + 2, S2); pieces += ring_count(t + 2); }
            else if (has_next) { ring_issue(N, t + 2 - items, S2); pieces += ring_count(t + 2 - items); }
            if (has_next && !(abl & 64)) {
                kv_issue(N, t);
                ++pieces;
                if (t == 7 && tail) { kv_issue(N, 8); pieces += wave < 2 ? 1 : 0; }
            }
            const unsigned char* const Qs = S;
            const unsigned char* const dOs = S + 4096;
            const unsigned char* const Os = S + 8192;
            u32x4 qa[4], da[4];
            a3_tile_rows(Qs, 0, lane, qa);
            a3_tile_rows(dOs, 0, lane, da);
            // delta of this lane's query (row c31 of the block): its half of dO . O, then the other half's
            float dlt = 0.f;
            if (!(abl & 4))
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const u32x4 of = a3_row_frag(Os, 0, kb, lane);
#pragma unroll
                for (int w = 0; w < 4; ++w) dlt = dot2_bf16(da[kb][w], of[w], dlt);
            }
            dlt += shfl_xor(dlt, 32);
            // this query ROW's fixed-point scale 2^e <= 2^29 / (2 |dO_q| max |V_k| max |K_k|) -- a row of the tile is only ever added to by the
            // lanes that own it (c31 = q in every wave, which all see the same dO_q: the same scale), so no maximum over the block is needed
            // (the first version took one: six cross-lane exchanges per step, 80 us of a launch)
            float fx = 1024.f;
            if (!(abl & 2)) {
                float d2 = 0.f;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int w = 0; w < 4; ++w) d2 = dot2_bf16(da[kb][w], da[kb][w], d2);
                d2 += shfl_xor(d2, 32);
                const float bound = 2.0f * __builtin_sqrtf(d2 * kvn);
                int ex = (int)((f2u(bound) >> 23) & 0xffu) - 127;              // floor(log2 bound) (bound >= 0)
                ex = ex < -60 ? -60 : (ex > 60 ? 60 : ex);
                fx = u2f((uint32_t)(28 - ex + 127) << 23);                     // bound 2^(28 - ex) < 2^29
                if (wave == 0 && h == 0) Inv[(t & 1) * 32 + c31] = u2f((uint32_t)(ex - 28 + 127) << 23);
            }
            wave_sync();
            if (h == 0) sc[32 + c31] = dlt;
            wave_sync();
            // both chains start at their offsets: S at -lse / scale (so that P = exp2(S scale2) is one multiply + one v_exp_f32 per score,
            // nothing of the 16 log-sum-exps stays live across the MFMAs), dP at -delta
            f32x16 sv, dp;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 a = ld16(Ls + t * 32 + 8 * q + 4 * h), b = ld16(sc + 32 + 8 * q + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) { sv[4 * q + e] = -u2f(a[e]) * inv_scale2; dp[4 * q + e] = -u2f(b[e]); }
            }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                sv = mma_kblock(qa[kb], kf[kb], sv, (bf16_t*)nullptr);
                dp = mma_kblock(da[kb], vf[kb], dp, (bf16_t*)nullptr);
            }
            if (masked) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = kvalid ? fast_exp2(sv[r] * scale2) : 0.f;
                    sv[r] = pv;
                    dp[r] = pv * dp[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sv[r] = fast_exp2(sv[r] * scale2);
                    dp[r] = sv[r] * dp[r];
                }
            }
            u32x4 df[2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const u32x4 pf = a2_pack_acc(sv, blk);
                df[blk] = a2_pack_acc(dp, blk);
                a5_tile_put(myX, c31, h, 2 * blk, u32x2{df[blk][0], df[blk][1]});
                a5_tile_put(myX, c31, h, 2 * blk + 1, u32x2{df[blk][2], df[blk][3]});
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dv[db] = mma_kblock(a3_col_frag(dOs, 0, blk, db, lane), pf, dv[db], (bf16_t*)nullptr);
                    dk[db] = mma_kblock(a3_col_frag(Qs, 0, blk, db, lane), df[blk], dk[db], (bf16_t*)nullptr);
                }
            }
            lds_fence();
            // dQ_t's partial over this wave's keys: the wave's own dS tile back as the transposed operand
            if (!(abl & 8)) {
            f32x16 dq[2];
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const u32x4 dsf = a5_tile_frag(myX, blk, lane);
#pragma unroll
                for (int db = 0; db < 2; ++db) dq[db] = mma_kblock(kc[blk][db], dsf, dq[db], (bf16_t*)nullptr);
            }
            if (!(abl & 1)) {
                int* const tq = DQ + (t & 1) * A6_DQ_TILE + c31 * A6_DQ_LD;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) lds_atomic_add(tq + 32 * db + mfma_row(r, lane), (int)(dq[db][r] * fx));
            } else { reg_keep(dq[0]); reg_keep(dq[1]); }
            }
            lds_fence();
            // the tail KEY against this query block: wave t, on the fragments it holds (uniform branch)
            if (tail && wave == t && !(abl & 32)) {
                u32x4 qr[4], dr[4];                            // (read again: held across the step they would cost every wave 32 registers)
                a3_tile_rows(Qs, 0, lane, qr);
                a3_tile_rows(dOs, 0, lane, dr);
                const float st = a3_tail_dot(KT, 0, 0, qr, lane), dpt = a3_tail_dot(VT, 0, 0, dr, lane);
                const float pt = tkey ? fast_exp2(st * scale2 - Ls[t * 32 + c31]) : 0.f;
                const float ds = pt * (dpt - dlt);                                     // dS / scale
                if (h == 0) DST[t * 32 + c31] = ds;
                a5_weighted_row_sum(Qs, 0, sc, ds, lane, Tp + (wave * 3 + 1) * 64);    // sum_q dS[q] Q[q]
                a5_weighted_row_sum(dOs, 0, sc, pt, lane, Tp + (wave * 3 + 2) * 64);   // sum_q P[q] dO[q]
            }
            // the item of the NEXT step has landed (it was requested a step ago; this step's requests may stay in flight -- except behind a
            // head's last step when it carried K / V pieces: the next prologue reads the images)
            if (t == 7 && !tail) wait_vmem(); else wait_older_than(pieces);
            barrier_nodrain();
        }
        if (tail) {
            // ---- ring item 8: the tail QUERY against this wave's keys (a lane owns a key) ----
            unsigned char* const S = Ring + gs * A6_SLOT;
            unsigned char* const S2 = Ring + (gs == 0 ? 2 : gs - 1) * A6_SLOT;
            gs = gs == 2 ? 0 : gs + 1;
            finish_tile(7);
            int pieces = 0;
            if (has_next) { ring_issue(N, 1, S2); pieces = ring_count(1); }
            const float dlt = a6_rows_dot(S + 4096, S + 8192, lane);
            const float st = a3_tail_dot(S, 0, 0, kf, lane), dpt = a3_tail_dot(S + 4096, 0, 0, vf, lane);
            const float pt = kvalid ? fast_exp2(st * scale2 - Ls[256]) : 0.f;
            const float ds = pt * (dpt - dlt);
            a6_tail_outer_acc(S, ds, lane, dk);
            a6_tail_outer_acc(S + 4096, pt, lane, dv);
            {                                                  // sum_k dS[k] K[k] over this wave's keys: the K^T fragments against a one-column operand
                u32x4 bf[2];
                a5_column_operand(sc, ds, lane, bf);
                f32x16 acc[2];
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int db = 0; db < 2; ++db) acc[db] = mma_kblock(kc[blk][db], bf[blk], acc[db], (bf16_t*)nullptr);
                if (c31 == 0) a3_put_col(Tp + (wave * 3 + 0) * 64, acc, lane);
            }
            // the tail row's own gradients (wave 0; lane = feature d): everything but the waves' partials is read BEFORE the barrier -- behind
            // it the other waves are already writing the next head's KT / VT / Ls and requesting ring items into this slot
            float aq = 0.f, ak = 0.f, av = 0.f;
            if (wave == 0) {
                const float qv = bf2f(*reinterpret_cast<const bf16_t*>(S + 2 * lane)), dov = bf2f(*reinterpret_cast<const bf16_t*>(S + 4096 + 2 * lane));
                const float ov = bf2f(*reinterpret_cast<const bf16_t*>(S + 8192 + 2 * lane));
                const float kv = bf2f(*reinterpret_cast<const bf16_t*>(KT + 2 * lane)), vv = bf2f(*reinterpret_cast<const bf16_t*>(VT + 2 * lane));
                const float s2 = wave_sum(qv * kv), dp2 = wave_sum(dov * vv), dl2 = wave_sum(dov * ov);
                const float pt2 = tkey ? fast_exp2(s2 * scale2 - Ls[256]) : 0.f;
                const float ds2 = pt2 * (dp2 - dl2);
                aq = ds2 * kv; ak = ds2 * qv; av = pt2 * dov;
            }
            wait_older_than(pieces);
            barrier_nodrain();
            if (wave == 0) {                                   // (Tp is next written by wave 0 itself, in the next head's step 0)
                for (int w = 0; w < A6_NB; ++w) {
                    aq += Tp[(w * 3 + 0) * 64 + lane];
                    ak += Tp[(w * 3 + 1) * 64 + lane];
                    av += Tp[(w * 3 + 2) * 64 + lane];
                }
                bf16_t* const trow = DQKV + H.qo + 256L * ldq + lane;
                trow[0] = f2bf(aq * p.scale);
                trow[kofs] = f2bf(ak * p.scale);
                trow[2 * kofs] = f2bf(av);
            }
        } else {
            finish_tile(7);
        }
        a3_store_rows_direct(dk, DQKV + H.qo + kofs, ldq, wave * 32, n, lane, p.scale);
        a3_store_rows_direct(dv, DQKV + H.qo + 2 * kofs, ldq, wave * 32, n, lane);
        if (!has_next) break;                                  // (uniform)
        bh = nbh;
        H = N;
    }
}

}  // namespace xc
