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

// A sequence of 32 q + r rows with a SHORT tail (1 <= r <= A3_TAIL_MAX, e.g. the default text length 257 = 8 * 32 + 1) is not
// given a ninth wave: 9 waves on 4 SIMDs put 3 on one SIMD and cost 1.44x (measured: n = 256 runs in 0.34 ms, n = 257 in
// 0.69 ms).  Instead the q full tiles get one wave each and the tail rows are processed COOPERATIVELY: wave w evaluates the
// tail queries against key sub-tiles w, w + nwaves, ...; the partial (max, sum, O) triples meet in a small LDS scratch.
constexpr int A3_TAIL_MAX = 2;
XC_HOST_DEV bool a3_coop_tail(int n) { return (n & 31) != 0 && (n & 31) <= A3_TAIL_MAX && n >= 64; }
XC_HOST_DEV int a3_waves(int n) { return a3_coop_tail(n) ? n / 32 : (n + 31) / 32; }
constexpr int A3_TAIL_REC = 66;                                // floats per (wave, tail row): m, l, O[64]

// Operand reads of 32-row sub-tile t of an image (attention2.h layout), written so that everything but `t * 4096` depends on the
// lane alone: the swizzle of a row only looks at bits 1-3 of the row number, which a sub-tile's base (a multiple of 16) does not
// touch.  The generic a2_row_frag / a2_col_frag recompute the swizzle from the full row number, and the compiler could not hoist that
// out of the sub-tile loops: ~50 of the ~115 vector instructions of a backward step were address arithmetic
// (profiles/r02_run19_attention_pmc.txt: the kernels are VALU-bound, 58 % VALU-busy against 28 % MFMA-busy).
XC_DEV u32x4 a3_row_frag(const unsigned char* img, int t, int kb, int lane) {
    const int c31 = lane & 31, h = lane >> 5;
    return ld16(img + t * 4096 + (c31 * 128 + a2_slot(c31, kb * 2 + h) * 16));
}
XC_DEV u32x4 a3_col_frag(const unsigned char* img, int t, int blk, int db, int lane) {
    const int g = lane >> 4, tt = lane & 15;
    const int r0 = 4 * (g >> 1) + (tt >> 2);
    const int col = 32 * db + 16 * (g & 1) + (tt & 3) * 4;
    const int lo_off = r0 * 128 + a2_slot(r0, col >> 3) * 16 + (col & 7) * 2;
    const int hi_off = (r0 + 8) * 128 + a2_slot(r0 + 8, col >> 3) * 16 + (col & 7) * 2;
    const unsigned char* base = img + t * 4096 + blk * 2048;
    const s16x4 lo = lds_read_tr16(base + lo_off);
    const s16x4 hi = lds_read_tr16(base + hi_off);
    const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
    u32x4 f = {a[0], a[1], b[0], b[1]};
    return f;
}

// one 32-key sub-tile of the online-softmax forward for the 32 queries whose fragments are qf (shared by both passes).
// The kernels are VALU-bound (the first version spent ~250 VALU instructions per 8 MFMAs here), so the softmax bookkeeping is
// kept in the base-2 domain of the SCALED scores: m2 = running max of s * scale2 (scale2 = scale log2 e, the max is taken
// over the raw accumulators), p = exp2(fma(s, scale2, -m2)) is one fma + one bare v_exp_f32 per score, and the key-validity
// selects are only compiled into the MASKED variant -- the caller votes per sub-tile (all keys valid: plain path; none: the
// sub-tile is skipped, its probabilities are exactly 0).
// CAUSAL (reference Attention.forward, x_clip.py:231-234: key j is hidden from query i when j > i) adds the comparison against
// the lane's own query index qidx -- compiled only into the sub-tiles that straddle the diagonal.
template <bool MASKED, bool CAUSAL>
XC_DEV void a3_fwd_step(const unsigned char* Ks, const unsigned char* Vs, const unsigned char* Ms, int t, const u32x4 (&qf)[4],
                        float scale2, int lane, int qidx, f32x16 (&o)[2], float& m2, float& l) {
    const int h = lane >> 5, c31 = lane & 31;
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) s = mma_kblock(a3_row_frag(Ks, t, kb, lane), qf[kb], s, (bf16_t*)nullptr);
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
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const u32x4 pf = a2_pack_acc(s, blk);
#pragma unroll
        for (int db = 0; db < 2; ++db) o[db] = mma_kblock(a3_col_frag(Vs, t, blk, db, lane), pf, o[db], (bf16_t*)nullptr);
    }
}
// votes on the sub-tile's key validity and runs the matching variant; with CAUSAL the wave's queries are [qlo, qlo + 32): sub-tiles
// entirely above the diagonal are skipped, the ones straddling it take the per-element comparison
template <bool CAUSAL>
XC_DEV void a3_fwd_step_auto(const unsigned char* Ks, const unsigned char* Vs, const unsigned char* Ms, int t, const u32x4 (&qf)[4],
                             float scale2, int lane, int qlo, f32x16 (&o)[2], float& m2, float& l) {
    const int qidx = qlo + (lane & 31);
    if (CAUSAL) {
        if (t * 32 > qlo + 31) return;
        if (t * 32 + 31 > qlo) { a3_fwd_step<true, true>(Ks, Vs, Ms, t, qf, scale2, lane, qidx, o, m2, l); return; }
    }
    const bool kv = Ms[t * 32 + (lane & 31)] != 0;
    if (wave_all(kv)) a3_fwd_step<false, false>(Ks, Vs, Ms, t, qf, scale2, lane, qidx, o, m2, l);
    else if (wave_any(kv)) a3_fwd_step<true, false>(Ks, Vs, Ms, t, qf, scale2, lane, qidx, o, m2, l);
}

// Two work-groups share a CU and every head takes the same time, so left alone they load together and compute together.  The one
// that was given the upper part of the CU's LDS in the FIRST round of work-groups waits half a head's time once: from then on one
// streams its images from HBM while the other is in its MFMA phases.
XC_DEV void a3_stagger(int ticks_10ns, int first_round) {
    // first_round = 2 x the device's CUs (the host passes it: work-groups of the first dispatch round); the wait is bounded by an
    // iteration count as well as by the clock, so a misbehaving realtime counter cannot hold a work-group
    if (ticks_10ns > 0 && (int)blockIdx.x < first_round && lds_base_granule() != 0) {
        const uint64_t until = realtime_10ns() + (uint64_t)ticks_10ns;
        for (int spin = 0; spin < 4 * ticks_10ns && realtime_10ns() < until; ++spin) nap();
    }
}
// ---- a single tail row (n = 32 q + 1: the text encoder's 257 = CLS + 256 tokens) without a 33rd MFMA block ------------------------------
// As the LAST sub-tile of every sweep the tail row costs a whole masked 32 x 32 step -- 8 to 16 MFMAs and ~110 vector instructions for one
// useful row or column of 32 (n = 257 ran 23 % longer than n = 256).  A lane owns a query (or key) of its wave's block, so the one score
// it has against the tail row is a 64-long dot product of two rows it can reach directly: its own row fragments (registers) and the tail
// row of the resident image (LDS, the same address in every lane of a half-wave: a broadcast read) -- 16 v_dot2c_f32_bf16 and one
// exchange between the half-waves.  Processed FIRST, the tail row's contribution is the INITIAL value of the accumulators (forward: the
// probability of the first key seen is exactly 1, so m = s, l = 1, O = v_tail; backward: dQ = dS k_tail, dK = dS q_tail, dV = P dO_tail)
// instead of zeros, and the sweeps run over the full sub-tiles only.
// score of the lane's row (fragments f: 8 features per k-block for its half h) against row `row` of sub-tile `t` of an image
XC_DEV float a3_tail_dot(const unsigned char* img, int t, int row, const u32x4 (&f)[4], int lane) {
    const int h = lane >> 5;
    float acc = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const u32x4 g = ld16(img + t * 4096 + (row * 128 + a2_slot(row, kb * 2 + h) * 16));
#pragma unroll
        for (int w = 0; w < 4; ++w) acc = dot2_bf16(f[kb][w], g[w], acc);
    }
    return acc + shfl_xor(acc, 32);
}
// acc[db][r] = mul * X[row][d = 32 db + mfma_row(r, lane)]: row `row` of sub-tile t of an image in the transposed accumulator layout
XC_DEV void a3_tail_outer(const unsigned char* img, int t, int row, float mul, int lane, f32x16 (&acc)[2]) {
    const int h = lane >> 5;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const u32x2 v = *reinterpret_cast<const u32x2*>(img + t * 4096 + (row * 128 + a2_slot(row, 4 * db + g) * 16 + 8 * h));
            acc[db][4 * g + 0] = mul * u2f(v[0] << 16);
            acc[db][4 * g + 1] = mul * u2f(v[0] & 0xffff0000u);
            acc[db][4 * g + 2] = mul * u2f(v[1] << 16);
            acc[db][4 * g + 3] = mul * u2f(v[1] & 0xffff0000u);
        }
}
// (abl & 4, measurement build: the round-3 form -- the tail row as the last sub-tile of every sweep)
XC_HOST_DEV bool a3_single_tail(int n, int abl = 0) { return a3_coop_tail(n) && (n & 31) == 1 && !(abl & 4); }

// key validity bytes of a head into LDS: at most two positions per thread in the shapes these kernels take (npad <= 288; the forward
// runs 2 npad threads, the backward 256), both loads unconditional (index clamped) and in flight together.  Written as
// `Ms[k] = (k < n) && (mask == nullptr || mask[k])` every position was a branch around its own load and a full drain of the memory
// counter -- which also counts the operand images requested just before (tools/isa_scan.py, round 3: five such pairs in the backward).
XC_DEV void a3_key_validity(unsigned char* Ms, const unsigned char* mask, long row0, int n, int npad) {
    const int k0 = threadIdx.x, k1 = threadIdx.x + blockDim.x;
    unsigned char m0 = 1, m1 = 1;
    if (mask != nullptr) {                                     // (uniform)
        m0 = mask[row0 + (k0 < n ? k0 : n - 1)];
        m1 = mask[row0 + (k1 < n ? k1 : n - 1)];
    }
    if (k0 < npad) Ms[k0] = (k0 < n) && m0 != 0;
    if (k1 < npad) Ms[k1] = (k1 < n) && m1 != 0;
    for (int k = k1 + blockDim.x; k < npad; k += blockDim.x)   // (never in the shapes above)
        Ms[k] = (k < n) && (mask == nullptr || mask[row0 + k] != 0);
}

// ---- forward ----------------------------------------------------------------------------------------------------------
// (at most 128 VGPRs: two 8-wave work-groups of ~78 KB LDS share a CU)
template <bool CAUSAL>
__global__ __launch_bounds__(576) XC_FOUR_WAVES_PER_SIMD void attn3_fwd_kernel(AttnParams p) {
    XC_LDS_DYNAMIC(lds);
    const int n = p.n, npad = (n + 31) & ~31;
    unsigned char* Ks = lds;
    unsigned char* Vs = Ks + npad * 128;
    unsigned char* Ms = Vs + npad * 128;                       // [npad] key validity
    float* Ts = reinterpret_cast<float*>(Ms + npad);           // [nwaves][A3_TAIL_MAX][A3_TAIL_REC] tail partials
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c31 = lane & 31;
    const int wave = uniform(tid >> 6), nwaves = blockDim.x >> 6;
    a3_stagger(p.stagger_10ns, p.first_round);
    const int bh = xcd_remap(blockIdx.x, p.batch * p.heads);
    const int hh = bh % p.heads, bi = bh / p.heads;
    const long ldq = 3L * p.heads * ATT_DH;
    const bf16_t* Qb = reinterpret_cast<const bf16_t*>(p.qkv) + (long)bi * n * ldq + hh * ATT_DH;
    const bf16_t* Kb = Qb + (long)p.heads * ATT_DH;
    const bf16_t* Vb = Kb + (long)p.heads * ATT_DH;
    bf16_t* out = reinterpret_cast<bf16_t*>(p.out) + (long)bi * n * p.heads * ATT_DH + hh * ATT_DH;
    float* lse_out = p.lse + ((long)bi * p.heads + hh) * n;
    const bool coop = a3_coop_tail(n);
    const int tail0 = (n >> 5) << 5, ntail = n & 31;
    const int q0 = wave * 32;
    const int qrow = q0 + c31;
    const int qld = qrow < n ? qrow : n - 1;
    // (round 6: the first query rows a wave uses -- the tail rows, or its own block where there is no tail -- are requested BEFORE the images:
    //  loads return in order, so they are there when the images are, instead of one more round trip behind the barrier.  Holding BOTH sets
    //  across the tail phase spills under this kernel's 128 registers.)
    u32x4 qf[4];
    const int qfirst = coop ? (tail0 + c31 < n ? tail0 + c31 : n - 1) : qld;
#ifdef XCLIP_MEASURE
    const bool q_late = (p.chunks & 8) != 0;                   // (measurement build, XCLIP_ATTN_ABL=8: the round-5 order, for the A/B)
#else
    constexpr bool q_late = false;
#endif
    if (!q_late) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) qf[kb] = ld16(Qb + (long)qfirst * ldq + kb * 16 + h * 8);
    }
    a3_dma_image(Ks, Kb, ldq, n, npad, wave, nwaves, lane);
    a3_dma_image(Vs, Vb, ldq, n, npad, wave, nwaves, lane);
    a3_key_validity(Ms, p.mask, (long)bi * n, n, npad);
    f32x16 o[2];
    wait_vmem();
    sync();
    if (q_late) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) qf[kb] = ld16(Qb + (long)qfirst * ldq + kb * 16 + h * 8);
    }
    const int nsub = npad >> 5;
    const float scale2 = p.scale * 1.4426950408889634f;
    if (coop) {                                                // tail queries x this wave's share of the key sub-tiles
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
        float m = ATT_NEG, l = 0.f;
        int tcoop = nsub;
        if (a3_single_tail(n, p.chunks)) {                     // (uniform) the tail query against the tail key: no sub-tile of its own --
            tcoop = nsub - 1;                                  // the last wave's partial starts from it (wave 0 used to walk TWO sub-tiles)
            if (wave == nwaves - 1 && Ms[tail0] != 0) {
                m = a3_tail_dot(Ks, tcoop, 0, qf, lane) * scale2;
                l = 1.f;
                a3_tail_outer(Vs, tcoop, 0, 1.f, lane, o);
            }
        }
        for (int t = wave; t < tcoop; t += nwaves) a3_fwd_step_auto<CAUSAL>(Ks, Vs, Ms, t, qf, scale2, lane, tail0, o, m, l);
        if (c31 < ntail) {
            float* rec = Ts + ((long)wave * A3_TAIL_MAX + c31) * A3_TAIL_REC;
            if (h == 0) { rec[0] = m; rec[1] = l; }
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) rec[2 + db * 32 + mfma_row(r, lane)] = o[db][r];
        }
    }
    if (coop) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) qf[kb] = ld16(Qb + (long)qld * ldq + kb * 16 + h * 8);
    }
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m = ATT_NEG, l = 0.f;
    int tend = nsub;
    if (a3_single_tail(n, p.chunks)) {                         // (uniform) the 257th key first: it initialises (m, l, O) -- see a3_tail_dot
        tend = nsub - 1;
        if (!CAUSAL && Ms[tail0] != 0) {                       // (causal: the last key is hidden from every query of a full block)
            m = a3_tail_dot(Ks, tend, 0, qf, lane) * scale2;
            l = 1.f;
            a3_tail_outer(Vs, tend, 0, 1.f, lane, o);
        }
    }
    for (int t = 0; t < tend; ++t) a3_fwd_step_auto<CAUSAL>(Ks, Vs, Ms, t, qf, scale2, lane, q0, o, m, l);
    sync();                                                    // every wave is done with the K / V images (and the tail partials are in)
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    a2_store_rows(lds + wave * 32 * 144, o, inv, out, (long)p.heads * ATT_DH, q0, n, lane);
    if (h == 0 && qrow < n) lse_out[qrow] = m * 0.6931471805599453f + logf(l);    // m is in log2 units
    if (coop && wave == 0) {                                   // merge the nwaves partials of every tail row; lane = feature d
        for (int q = 0; q < ntail; ++q) {
            float M = ATT_NEG;
            for (int w = 0; w < nwaves; ++w) M = fmaxf(M, Ts[((long)w * A3_TAIL_MAX + q) * A3_TAIL_REC]);
            float L = 0.f, acc = 0.f;
            for (int w = 0; w < nwaves; ++w) {
                const float* rec = Ts + ((long)w * A3_TAIL_MAX + q) * A3_TAIL_REC;
                const float f = fast_exp2(rec[0] - M);
                L += rec[1] * f;
                acc += rec[2 + lane] * f;
            }
            out[(long)(tail0 + q) * p.heads * ATT_DH + lane] = f2bf(L > 0.f ? acc / L : 0.f);
            if (lane == 0) lse_out[tail0 + q] = M * 0.6931471805599453f + logf(L);
        }
    }
}

// ---- backward (dQ, dK, dV and delta in one kernel) ------------------------------------------------------------------------
// phase A body: dQ^T[d, query] += K^T dS^T for the 32 queries whose fragments are (qf, dof) against key sub-tile t
// (lse2_q = lse_q log2(e) and scale2 = scale log2(e): the probability is one fma + a bare v_exp_f32).
// The backward is VALU-bound like the forward, so per score it is kept to fma, exp, mul: the dP accumulator STARTS at -delta
// (dP - delta comes out of the MFMA chain), the factor `scale` of dS = P (dP - delta) scale is applied once to the finished dQ / dK
// (a power of two for dim_head 64: the bf16 rounding of dS is unchanged), and the key-validity / causal selects with their LDS mask
// reads are only compiled into the MASKED variant -- the caller votes per sub-tile, as the forward does.
// (`masked` is wave-uniform and only switches between the two score loops: one copy of the MFMA chains, so the register allocation
//  stays that of a single variant -- two inlined instances of the whole step cost 2.7 x the registers and half the occupancy)
// The step is one dependent chain (fragment reads -> 8 MFMAs -> 16 exp -> pack -> 4 MFMAs) and a SIMD holds two waves, so the row
// fragments of sub-tile t + 1 are requested by the caller BEFORE the chain of sub-tile t starts (kr / vr = the fragments of t).
// this lane's row fragments (all four k-blocks) of sub-tile t of an operand image
XC_DEV void a3_tile_rows(const unsigned char* img, int t, int lane, u32x4 (&f)[4]) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) f[kb] = a3_row_frag(img, t, kb, lane);
}
// S^T and dP^T - delta of one sub-tile: the two MFMA chains only (the caller issues them one sub-tile AHEAD of the score arithmetic)
XC_DEV void a3_bwd_dq_scores(const u32x4 (&kr)[4], const u32x4 (&vr)[4], const u32x4 (&qf)[4], const u32x4 (&dof)[4], float delta_q,
                             f32x16& s, f32x16& dp) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = -delta_q; }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        s = mma_kblock(kr[kb], qf[kb], s, (bf16_t*)nullptr);
        dp = mma_kblock(vr[kb], dof[kb], dp, (bf16_t*)nullptr);
    }
}
// probabilities, dS^T and dQ^T += K^T dS^T of sub-tile t from its finished score accumulators
template <bool CAUSAL>
XC_DEV void a3_bwd_dq_finish(const unsigned char* Ks, const unsigned char* Ms, int t, f32x16& s, const f32x16& dp, float lse2_q, float scale2,
                             int lane, int qidx, f32x16 (&dq)[2], bool masked) {
    if (masked || CAUSAL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kj = t * 32 + mfma_row(r, lane);
            const float pv = (Ms[kj] && (!CAUSAL || kj <= qidx)) ? fast_exp2(s[r] * scale2 - lse2_q) : 0.f;
            s[r] = pv * dp[r];                                                 // dS^T / scale
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = fast_exp2(s[r] * scale2 - lse2_q) * dp[r];
    }
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const u32x4 df = a2_pack_acc(s, blk);
#pragma unroll
        for (int db = 0; db < 2; ++db) dq[db] = mma_kblock(a3_col_frag(Ks, t, blk, db, lane), df, dq[db], (bf16_t*)nullptr);
    }
}
// sub-tiles t0, t0 + dt, ... < tend of the K / V images against one query block, software-pipelined: while the score arithmetic of
// sub-tile t runs on the VALU, the MFMA chains of t + dt are already in the matrix pipe.  plain_bits: bit t set = all 32 keys of sub-tile t are valid (voted once per head)
template <bool CAUSAL>
XC_DEV void a3_bwd_dq_sweep(const unsigned char* Ks, const unsigned char* Vs, const unsigned char* Ms, int t0, int tend, int dt,
                            uint32_t plain_bits, const u32x4 (&qf)[4], const u32x4 (&dof)[4], float lse2_q, float delta_q, float scale2,
                            int lane, int qidx, f32x16 (&dq)[2]) {
    if (t0 >= tend) return;
    // (one fragment buffer, two score sets: fragments AND scores two deep need 254 + 24 registers -- one wave per SIMD)
    u32x4 kr[4], vr[4];
    f32x16 sa, da, sb, db;
    a3_tile_rows(Ks, t0, lane, kr);
    a3_tile_rows(Vs, t0, lane, vr);
    a3_bwd_dq_scores(kr, vr, qf, dof, delta_q, sa, da);
    for (int t = t0; t < tend; t += 2 * dt) {
        const int t1 = t + dt, t2 = t + 2 * dt;
        if (t1 < tend) {
            a3_tile_rows(Ks, t1, lane, kr);
            a3_tile_rows(Vs, t1, lane, vr);
            a3_bwd_dq_scores(kr, vr, qf, dof, delta_q, sb, db);
        }
        a3_bwd_dq_finish<CAUSAL>(Ks, Ms, t, sa, da, lse2_q, scale2, lane, qidx, dq, !((plain_bits >> t) & 1u));
        if (t1 < tend) {
            if (t2 < tend) {
                a3_tile_rows(Ks, t2, lane, kr);
                a3_tile_rows(Vs, t2, lane, vr);
                a3_bwd_dq_scores(kr, vr, qf, dof, delta_q, sa, da);
            }
            a3_bwd_dq_finish<CAUSAL>(Ks, Ms, t1, sb, db, lse2_q, scale2, lane, qidx, dq, !((plain_bits >> t1) & 1u));
        }
    }
}
// phase B body: dK^T, dV^T for the 32 keys whose fragments are (kf, vf) against query sub-tile t.  masked (wave-uniform): some of
// the wave's keys are padding, or the query sub-tile runs past n
template <bool CAUSAL>
XC_DEV void a3_bwd_dkv_step(const unsigned char* Qs, const unsigned char* dOs, const float* Ls2, const float* Ds, int t, int n,
                            const u32x4 (&qr)[4], const u32x4 (&dor)[4], const u32x4 (&kf)[4], const u32x4 (&vf)[4], bool kvalid,
                            float scale2, int lane, int kidx, f32x16 (&dk)[2], f32x16 (&dv)[2], bool masked) {
    const int h = lane >> 5;
    f32x16 s, dp;
    // lse log2(e) and delta of the 16 queries this lane's accumulator registers belong to: rows 8 q + 4 h + (0..3) of the
    // sub-tile, i.e. four 16-byte LDS reads each -- issued up front, unconditionally (rows >= n hold 0 and are masked by a
    // select below; a conditional load here compiled to sixteen exec-masked branches with an LDS round trip each)
    float l2[16], dl[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u32x4 a = ld16(Ls2 + t * 32 + 8 * q + 4 * h), b = ld16(Ds + t * 32 + 8 * q + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) { l2[4 * q + e] = u2f(a[e]); dl[4 * q + e] = u2f(b[e]); }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = -dl[r]; }                 // (dP - delta out of the MFMA chain)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        s = mma_kblock(qr[kb], kf[kb], s, (bf16_t*)nullptr);
        dp = mma_kblock(dor[kb], vf[kb], dp, (bf16_t*)nullptr);
    }
    if (masked || CAUSAL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ql = t * 32 + mfma_row(r, lane);
            float pv = fast_exp2(s[r] * scale2 - l2[r]);
            pv = (kvalid && ql < n && (!CAUSAL || ql >= kidx)) ? pv : 0.f;
            s[r] = pv;                                                         // P
            dp[r] = pv * dp[r];                                                // dS / scale
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = fast_exp2(s[r] * scale2 - l2[r]);
            dp[r] = s[r] * dp[r];
        }
    }
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const u32x4 pf = a2_pack_acc(s, blk);
        const u32x4 df = a2_pack_acc(dp, blk);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            dv[db] = mma_kblock(a3_col_frag(dOs, t, blk, db, lane), pf, dv[db], (bf16_t*)nullptr);
            dk[db] = mma_kblock(a3_col_frag(Qs, t, blk, db, lane), df, dk[db], (bf16_t*)nullptr);
        }
    }
}
// query sub-tiles t0, t0 + dt, ... < tend of the Q / dO images against one key block (as a3_bwd_dq_sweep)
template <bool CAUSAL>
XC_DEV void a3_bwd_dkv_sweep(const unsigned char* Qs, const unsigned char* dOs, const float* Ls2, const float* Ds, int t0, int tend, int dt,
                             int n, bool keys_plain, const u32x4 (&kf)[4], const u32x4 (&vf)[4], bool kvalid, float scale2, int lane,
                             int kidx, f32x16 (&dk)[2], f32x16 (&dv)[2]) {
    // (no fragment prefetch here: with dK and dV both live it does not fit in 256 registers -- 254 + 16 measured -- and a third
    //  wave-slot's worth of registers would halve the occupancy)
    for (int t = t0; t < tend; t += dt) {
        u32x4 qa[4], da[4];
        a3_tile_rows(Qs, t, lane, qa);
        a3_tile_rows(dOs, t, lane, da);
        a3_bwd_dkv_step<CAUSAL>(Qs, dOs, Ls2, Ds, t, n, qa, da, kf, vf, kvalid, scale2, lane, kidx, dk, dv, !(keys_plain && t * 32 + 32 <= n));
    }
}
// this lane's 32 of the 64 feature values of its column (query / key c31) -> rec[0..63]
XC_DEV void a3_put_col(float* rec, const f32x16 (&acc)[2], int lane) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) rec[db * 32 + mfma_row(r, lane)] = acc[db][r];
}

// acc[db] (rows = d = 32 db + mfma_row, column = this lane's row c31) -> dst rows [row0, row0 + 32), straight from the
// registers: a lane owns a row, v_permlane32_swap pairs the 4-column quads into 16-byte stores (as the GEMM epilogue)
XC_DEV void a3_store_rows_direct(const f32x16 (&acc)[2], bf16_t* dst, long ldd, int row0, int nrows, int lane, float mul = 1.f) {
    const int c31 = lane & 31, h = lane >> 5;
    const int row = row0 + c31;
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
            if (row < nrows) {
                u32x4 o = {pk[qq][0], pk[qq][1], pk[qq + 1][0], pk[qq + 1][1]};
                st16(dst + (long)row * ldd + db * 32 + qq * 8 + 8 * h, o);
            }
        }
    }
}
// this lane's 16-byte pieces of row `r` of X (the operand layout a2_row_frag returns), straight from global memory
XC_DEV void a3_row_frags(const bf16_t* X, long ldx, int r, int lane, u32x4 (&f)[4]) {
    const int h = lane >> 5;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) f[kb] = ld16(X + (long)r * ldx + kb * 16 + h * 8);
}

// Two operand images at a time: phase A needs the K and V images (its own Q / dO rows come straight from global memory,
// L2 hits: the image DMA of the other work-group / phase asks for the same lines), phase B the Q and dO images (own K / V
// rows from global), so the second pair is DMA'd over the first between the phases and a head needs ~80 KB of LDS instead
// of 160: TWO work-groups of four waves per CU (each wave takes every fourth 32-row block), one computing while the other
// waits for HBM.  (The one-work-group-per-CU version measured load + store skeleton 405 us, phase A 225, phase B 395, total
// 1060 = their sum at n = 256.)  p.chunks is a measurement switch here (XCLIP_ATTN_ABL: 1 = skip phase A, 2 = skip phase B, 4 = the tail row as a 33rd block).
template <bool CAUSAL>
__global__ __launch_bounds__(256) void attn3_bwd_kernel(AttnParams p) {
    XC_LDS_DYNAMIC(lds);
    const int n = p.n, npad = (n + 31) & ~31;
    const int img = npad * 128;
    unsigned char* R0 = lds;                                   // K, then Q
    unsigned char* R1 = R0 + img;                              // V, then dO
    unsigned char* Ms = R1 + img;                              // [npad] key validity
    float* Ls = reinterpret_cast<float*>(Ms + npad);           // [npad] lse log2(e) per query (npad is a multiple of 32)
    float* Ds = Ls + npad;                                     // [npad] delta per query
    float* Tp = Ds + npad;                                     // [nwaves][A3_TAIL_MAX][128] tail partials: dQ (phase A), dK | dV (phase B)
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c31 = lane & 31;
    const int wave = uniform(tid >> 6), nwaves = blockDim.x >> 6;
    a3_stagger(p.stagger_10ns, p.first_round);
    const int bh = xcd_remap(blockIdx.x, p.batch * p.heads);
    const int hh = bh % p.heads, bi = bh / p.heads;
    const long ldq = 3L * p.heads * ATT_DH, ldo = (long)p.heads * ATT_DH;
    const bf16_t* Qb = reinterpret_cast<const bf16_t*>(p.qkv) + (long)bi * n * ldq + hh * ATT_DH;
    const bf16_t* Kb = Qb + (long)p.heads * ATT_DH;
    const bf16_t* Vb = Kb + (long)p.heads * ATT_DH;
    const bf16_t* dOb = reinterpret_cast<const bf16_t*>(p.dout) + (long)bi * n * ldo + hh * ATT_DH;
    const bf16_t* Ob = reinterpret_cast<const bf16_t*>(p.out) + (long)bi * n * ldo + hh * ATT_DH;
    bf16_t* dQ = reinterpret_cast<bf16_t*>(p.dqkv) + (long)bi * n * ldq + hh * ATT_DH;
    bf16_t* dK = dQ + (long)p.heads * ATT_DH;
    bf16_t* dV = dK + (long)p.heads * ATT_DH;
    a3_dma_image(R0, Kb, ldq, n, npad, wave, nwaves, lane);
    a3_dma_image(R1, Vb, ldq, n, npad, wave, nwaves, lane);
    a3_key_validity(Ms, p.mask, (long)bi * n, n, npad);
    const bool coop = a3_coop_tail(n);
    const int tail0 = (n >> 5) << 5, ntail = n & 31;
    const int nblk = a3_waves(n);                              // 32-row blocks owned by single waves (without a cooperative tail)
    const int nsub = npad >> 5;
    const float scale2 = p.scale * 1.4426950408889634f;
    // delta_i = sum_d dO[i, d] O[i, d] and lse_i log2(e): lane (i = c31, half h) covers 32 of the 64 d
    for (int blk = wave; blk < nsub; blk += nwaves) {
        const int row_ = blk * 32 + c31;
        const int rl = row_ < n ? row_ : n - 1;
        const float lse_r = p.lse[((long)bi * p.heads + hh) * n + rl];      // (unconditional, with the row's O / dO chunks: one round trip)
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a[8], b[8];
            load_vec<bf16_t>(Ob + (long)rl * ldo + h * 32 + c * 8, a);
            load_vec<bf16_t>(dOb + (long)rl * ldo + h * 32 + c * 8, b);
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
    u32x4 f0[4], f1[4];                                        // the block's own rows: (Q, dO) in phase A, (K, V) in phase B
    f32x16 g0[2], g1[2];                                       // dQ in phase A; dK, dV in phase B
    if (!(p.chunks & 1)) {
        if (coop) {                                            // tail queries first: this wave's share of the key sub-tiles
            const int trow = tail0 + c31 < n ? tail0 + c31 : n - 1;
            a3_row_frags(Qb, ldq, trow, lane, f0);
            a3_row_frags(dOb, ldo, trow, lane, f1);
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) g0[db][r] = 0.f;
            const float lq = Ls[tail0 + c31], dl = Ds[tail0 + c31];
            int tcoop = nsub;
            if (a3_single_tail(n, p.chunks)) {                 // (uniform) tail query x tail key: the last wave's partial starts from it
                tcoop = nsub - 1;
                const float st = a3_tail_dot(R0, tcoop, 0, f0, lane), dt_ = a3_tail_dot(R1, tcoop, 0, f1, lane);
                if (wave == nwaves - 1 && Ms[tail0] != 0) a3_tail_outer(R0, tcoop, 0, fast_exp2(st * scale2 - lq) * (dt_ - dl), lane, g0);
            }
            a3_bwd_dq_sweep<CAUSAL>(R0, R1, Ms, wave, tcoop, nwaves, plain_bits, f0, f1, lq, dl, scale2, lane, tail0 + c31, g0);
            if (c31 < ntail) a3_put_col(Tp + ((long)wave * A3_TAIL_MAX + c31) * 128, g0, lane);
        }
        for (int rb = wave; rb < nblk; rb += nwaves) {
            const int row = rb * 32 + c31;
            const int rl = row < n ? row : n - 1;
            a3_row_frags(Qb, ldq, rl, lane, f0);
            a3_row_frags(dOb, ldo, rl, lane, f1);
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) g0[db][r] = 0.f;
            const float lse_q = Ls[row], delta_q = Ds[row];
            int tend = CAUSAL ? (rb + 1 < nsub ? rb + 1 : nsub) : nsub;            // key sub-tiles above the diagonal contribute nothing
            if (a3_single_tail(n, p.chunks) && !CAUSAL) {                           // (uniform) the 257th key: dQ starts at dS k_tail
                tend = nsub - 1;
                if (Ms[tail0] != 0) {
                    const float pt = fast_exp2(a3_tail_dot(R0, tend, 0, f0, lane) * scale2 - lse_q);
                    const float ds = pt * (a3_tail_dot(R1, tend, 0, f1, lane) - delta_q);        // dS / scale
                    a3_tail_outer(R0, tend, 0, ds, lane, g0);
                }
            }
            a3_bwd_dq_sweep<CAUSAL>(R0, R1, Ms, 0, tend, 1, plain_bits, f0, f1, lse_q, delta_q, scale2, lane, row, g0);
            a3_store_rows_direct(g0, dQ, ldq, rb * 32, n, lane, p.scale);
        }
    }
    sync();                                                    // every wave is done with the K / V images; the tail partials are complete
    a3_dma_image(R0, Qb, ldq, n, npad, wave, nwaves, lane);
    a3_dma_image(R1, dOb, ldo, n, npad, wave, nwaves, lane);
    if (coop && wave == 0 && !(p.chunks & 1)) {                // tail dQ = sum of the waves' partials; lane = feature d
        for (int q = 0; q < ntail; ++q) {
            float acc = 0.f;
            for (int w = 0; w < nwaves; ++w) acc += Tp[((long)w * A3_TAIL_MAX + q) * 128 + lane];
            dQ[(long)(tail0 + q) * ldq + lane] = f2bf(acc * p.scale);
        }
    }
    wait_vmem();
    sync();                                                    // Q / dO images in place; Tp may be reused

    // ---- phase B: dK^T, dV^T for the wave's key blocks, streaming the query sub-tiles of the Q / dO images ----
    if (!(p.chunks & 2)) {
        for (int rb = wave; rb < nblk; rb += nwaves) {
            const int row = rb * 32 + c31;
            const int rl = row < n ? row : n - 1;
            a3_row_frags(Kb, ldq, rl, lane, f0);
            a3_row_frags(Vb, ldq, rl, lane, f1);
            const bool kvalid = Ms[row] != 0;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) { g0[db][r] = 0.f; g1[db][r] = 0.f; }
            const bool keys_plain = wave_all(kvalid);                              // (uniform: no padding among this block's keys)
            int qend = nsub;
            if (a3_single_tail(n, p.chunks)) {                                     // (uniform) the 257th query: dK / dV start at its contribution
                qend = nsub - 1;                                                   // (causal: the last query sees every key)
                const float st = a3_tail_dot(R0, qend, 0, f0, lane);               // (every lane: the dot product ends in a cross-lane exchange)
                const float pt = kvalid ? fast_exp2(st * scale2 - Ls[tail0]) : 0.f;
                const float ds = pt * (a3_tail_dot(R1, qend, 0, f1, lane) - Ds[tail0]);          // dS / scale
                a3_tail_outer(R0, qend, 0, ds, lane, g0);
                a3_tail_outer(R1, qend, 0, pt, lane, g1);
            }
            // (query sub-tiles below the diagonal see none of these keys)
            a3_bwd_dkv_sweep<CAUSAL>(R0, R1, Ls, Ds, CAUSAL ? rb : 0, qend, 1, n, keys_plain, f0, f1, kvalid, scale2, lane, row, g0, g1);
            a3_store_rows_direct(g0, dK, ldq, rb * 32, n, lane, p.scale);
            a3_store_rows_direct(g1, dV, ldq, rb * 32, n, lane);
        }
        if (coop) {                                            // tail keys: this wave's share of the query sub-tiles
            const int trow = tail0 + c31 < n ? tail0 + c31 : n - 1;
            a3_row_frags(Kb, ldq, trow, lane, f0);
            a3_row_frags(Vb, ldq, trow, lane, f1);
            const bool tvalid = Ms[tail0 + c31] != 0;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) { g0[db][r] = 0.f; g1[db][r] = 0.f; }
            int tcoop = nsub;
            if (a3_single_tail(n, p.chunks)) {                 // (uniform) tail key x tail query: the last wave's partial starts from it
                tcoop = nsub - 1;
                const float st = a3_tail_dot(R0, tcoop, 0, f0, lane), dt_ = a3_tail_dot(R1, tcoop, 0, f1, lane);
                if (wave == nwaves - 1) {
                    const float pt = tvalid ? fast_exp2(st * scale2 - Ls[tail0]) : 0.f;
                    a3_tail_outer(R0, tcoop, 0, pt * (dt_ - Ds[tail0]), lane, g0);
                    a3_tail_outer(R1, tcoop, 0, pt, lane, g1);
                }
            }
            a3_bwd_dkv_sweep<CAUSAL>(R0, R1, Ls, Ds, wave, tcoop, nwaves, n, false, f0, f1, tvalid, scale2, lane, tail0 + c31, g0, g1);
            if (c31 < ntail) {
                float* rec = Tp + ((long)wave * A3_TAIL_MAX + c31) * 128;
                a3_put_col(rec, g0, lane);
                a3_put_col(rec + 64, g1, lane);
            }
        }
    }
    if (coop) {
        sync();
        if (wave == 0 && !(p.chunks & 2)) {
            for (int q = 0; q < ntail; ++q) {
                float ak = 0.f, av = 0.f;
                for (int w = 0; w < nwaves; ++w) {
                    const float* rec = Tp + ((long)w * A3_TAIL_MAX + q) * 128;
                    ak += rec[lane];
                    av += rec[64 + lane];
                }
                dK[(long)(tail0 + q) * ldq + lane] = f2bf(ak * p.scale);
                dV[(long)(tail0 + q) * ldq + lane] = f2bf(av);
            }
        }
    }
}

constexpr int A3_BWD_WAVES = 4;
XC_HOST_DEV int a3_bwd_waves(int n) { const int b = a3_waves(n); return b < A3_BWD_WAVES ? b : A3_BWD_WAVES; }

inline int attn3_fwd_lds_bytes(int n) {
    const int npad = (n + 31) & ~31, nw = a3_waves(n);
    const int a = 2 * npad * 128 + npad + nw * A3_TAIL_MAX * A3_TAIL_REC * 4, b = nw * 32 * 144;
    return (a > b ? a : b) + 64;
}
inline int attn3_bwd_lds_bytes(int n) {
    const int npad = (n + 31) & ~31;
    return 2 * npad * 128 + npad + 2 * npad * 4 + (a3_coop_tail(n) ? a3_bwd_waves(n) * A3_TAIL_MAX * 128 * 4 : 0) + 64;
}

}  // namespace xc
