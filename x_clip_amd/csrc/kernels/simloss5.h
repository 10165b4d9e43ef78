// simloss5.h -- the contrastive head (simloss3.h: S = scale * Q K^T reduced to log-sum-exp partials in the forward, turned into the
// gradient factor G in the backward; reference x_clip.py:813-847) on the PRODUCTION GEMM loop of gemm4.h (g5_run): the Q operand --
// the one that streams, K's column panel is re-read from L2 by every tile of its column -- in a ring of three LDS stages, descriptor-
// addressed LDS DMA, the counted waits.  At the configs[2] per-rank block 4096 x 32768 x 512 (the same-shape plain GEMM that WRITES the
// logits: 144 - 151 us): forward (log-sum-exp partials, nothing stored) 177 -> 154 - 158 us = 893 TFLOP/s; G 362 -> 170 us
// (profiles/r03_b_*, r03_o_* ... r03_u_*; DESIGN_APPENDIX.md section 3 has the steps).  The first ring form of G (whole-line epilogue with the
// general tile in the same function: 423 us) lives on in the measurement build.
#pragma once
#include "gemm4.h"
#include "simloss3.h"

namespace xc {

// forward: no stores at all -- the g5_run protocol's "nothing left in flight" epilogue
struct Sim5LseEpilogue {
    const SimParams& p;
    float scale;                 // sim_scale(p), read once per work-group (simloss3.h)
    // an interior tile leaves exactly 8 small stores behind (its rows' partial maxima and sums); they are younger than the operand pieces
    // the next tile's first K step waits for and may stay in flight over it, like the 16 line stores of a GEMM tile
    static constexpr bool LOOSE8 = true;
    XC_DEV void finish() {}
    XC_DEV bool packs_lines(int, int) const { return false; }
    XC_DEV void pack_lines(f32x16 (&)[4][2], unsigned char*, u32x4 (&)[4][4], int, int) const {}
    template <bool NT = false> XC_DEV void store_lines(const u32x4 (&)[4][4], int, int) const {}
    // (measured and not kept: re-reading the epilogue's parameters from the kernarg segment per tile -- params_in_memory(), 54 -> 35
    //  spilled SGPRs -- made the kernel SLOWER, 194 against 157 us: every field access became its own scalar load + wait;
    //  profiles/r03_f_sim_kernels_32k.log)
    XC_DEV int with_scratch(f32x16 (&acc)[4][2], int m0, int n0, unsigned char*) const { return Sim3LseEpilogue{p, scale}(acc, m0, n0); }   // 8 or 0
};

// ---- backward: G on the ring loop -----------------------------------------------------------------------------------------------------
// Why G was slow (362 us at 4096 x 32768 x 512 where the plain GEMM that stores the same 268 MB takes 144 - 151): simloss3.h keeps the
// general tile -- per-element range tests, masked loads -- in the same function as the interior one, and around the 128 accumulators
// the compiler then spills 100 - 240 vector registers INTO THE INTERIOR PATH: ~23 us of epilogue per tile where the exponentials and
// the stores account for ~9.  Here every FULL tile (on the diagonal or off it) is handled by a kernel of its own on the ring loop: the
// accumulators become G in place, column quad by column quad (the four per-column factors live only that long), and leave through the
// plain GEMM's line exchange as whole-line stores -- 226 registers, nothing spilled; across the K loops it carries one register (the
// lane's share of sum G o S: per tile it would be 16k same-address atomics at 32k x 4k, each one in front of the next tile's first
// counted wait -- measured: 258 against 165 us for this launch).  Tiles at a ragged edge are walked by a second, small launch of
// simloss3.h's kernel over a tile LIST (Sim5EdgeTiles); batch sizes that are multiples of 256 never need it.
// Measured at 4096 x 32768 x 512 (profiles/r03_q*_sim_g_variants.log, r03_r_*): 362 -> 292 (two launches, atomics per tile) -> 210
// (atomic per wave) -> 191 - 197 (diagonal tiles in this kernel too) -> 182 (banded tile order) -> 170 (lse loads unconditional and ahead).
XC_DEV bool sim5_full_tile(const SimParams& p, int m0, int n0) { return (m0 + G2_BM <= p.nq) && (n0 + G2_BN <= p.nk); }
XC_DEV bool sim5_off_diagonal(const SimParams& p, int m0, int n0) { return m0 + p.diag_off + G2_BM <= n0 || m0 + p.diag_off >= n0 + G2_BN; }

// (STREAM -- non-temporal stores for a G the L2s cannot hold anyway -- is a template parameter: as a run-time branch around the 16
//  stores it cost this kernel 102 spilled registers)
// (VAR: measurement build only, XCLIP_SIMG -- 1 = without DEFER_FRAGS, 2 = without the spread vote / exact form (round-3 arithmetic), 3 = both,
//  4 = the exchange addresses hoisted and spilled as before round 6)
template <bool STREAM, int VAR = 0>
struct Sim5FastGradEpilogue {
    const SimParams& p;
    const Gemm2Params& gp;       // C = G, ldc = ldg, alpha = 1: what the line stores address
    float scale, gmul;           // sim_scale(p) and *p.gmul (or 1), read once per work-group
    float dt_acc = 0.f;          // this lane's share of sum G o acc over the work-group's tiles (one register across the K loops)
    // the epilogue works on all 128 accumulators at once beside ~50 registers of its own: it takes the 24 the loop would hold for the next
    // tile's first fragments (gemm4.h g5_defer_frags) -- without them every further term of the arithmetic tipped the kernel into
    // 30 - 250 spilled registers (tests/test_isa_guard.py)
    static constexpr bool DEFER_FRAGS = !(VAR & 1);
    XC_DEV void finish() {
        const float dt = wave_sum(dt_acc) * (scale / (p.g_times_scale ? scale : 1.0f));
        if ((threadIdx.x & 63) == 0 && p.dtau != nullptr) atomic_add(p.dtau, dt);
    }
    XC_DEV bool packs_lines(int, int) const { return false; }
    XC_DEV void pack_lines(f32x16 (&)[4][2], unsigned char*, u32x4 (&)[4][4], int, int) const {}
    template <bool NT = false> XC_DEV void store_lines(const u32x4 (&)[4][4], int, int) const {}
    XC_DEV int with_scratch(f32x16 (&acc)[4][2], int m0, int n0, unsigned char* scratch) {
        if (!sim5_full_tile(p, m0, n0)) return 0;                   // (uniform) the edge launch's tile
        to_g(acc, m0, n0);
        // (an opaque lane id: what is derived from it below is recomputed per tile, not carried across the K loop in spilled registers --
        //  round 6: twelve scratch reloads per tile, each behind s_waitcnt vmcnt(0), i.e. a drain of the next tile's DMA pieces and of the
        //  previous group's stores; profiles/r06_k_*)
        const int lane = (VAR & 4) ? (int)(threadIdx.x & 63) : (int)opaque((uint32_t)(threadIdx.x & 63));      // (VAR 4: the round-5 form, for the A/B)
        const int wave = uniform(threadIdx.x >> 6), wm = wave >> 2, wn = wave & 3;
        const G4GemmEpilogue<G4_PLAIN> lines{gp};
        const BufRsrc rc = make_rsrc(gp.C + (long)m0 * gp.ldc + n0, 255u * (uint32_t)gp.ldc * 2u + 512u);
        const uint32_t vc = ((uint32_t)(wm * 128 + (lane >> 3)) * (uint32_t)gp.ldc + (uint32_t)(wn * 64 + 8 * (lane & 7))) * 2u;
        const uint32_t s8 = (uint32_t)gp.ldc * 16u;                  // 8 rows * ldc * 2 bytes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x4 o[4];
            lines.template pack_lines_i<true>(acc[i], scratch, o, lane);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (STREAM) buf_st16_nt<0>(rc, vc, s8 * (uint32_t)(4 * i + k), o[k]);
                else buf_st16<0>(rc, vc, s8 * (uint32_t)(4 * i + k), o[k]);
            }
        }
        // the 16 stores are YOUNGER than the A / B pieces the next tile's first K step waits for and may stay in flight over it
        // (g5_run: in_flight == 16)
        return 16;
    }
    // The accumulators of a full tile become G in place.  Per logit: G = exp(s - R) (a' + c') with s = acc * scale,
    // a' = gs a exp(R - lse_q), c' = gs c exp(R - lse_k) -- ONE exponential per logit, in the base-2 domain: one fma + a bare v_exp_f32;
    // then an add, an fma, and an fma for sum G o acc (d tau = that sum x scale / gs, applied once per wave).
    // The reference point R is one of the wave block's own lse values (its first row's, or its first column's when a = 0): exp(R - lse)
    // and exp(s - R) then stay inside fp32 as long as every lse of the block (its 128 rows' when a != 0, its 64 columns' when c != 0)
    // lies within 60 of it.  (History: R = scale -- valid since |cos| <= 1 -- overflowed exp(scale - lse) at exp(tau) = 200 with small
    // cosines; R = the block's first row's lse failed as soon as the lse values themselves spread: exp(tau) = 200 with a few perfectly
    // matched pairs among unrelated ones puts matched rows at ~200 and the others at ~40, and 0 x inf = NaN filled 65519 of 65536
    // entries of the ADVICE r3 reproducer.)  When some lse lies farther away than that (a wave vote) the tile takes the EXACT form
    // G = gs a exp(s - lse_q) + gs c exp(s - lse_k), two exponentials per logit whose arguments are <= 0 wherever the term matters -- as a
    // small block in front of each column quad's arithmetic (see `exact` below; how it got there: a second copy of the loop nest spilled
    // 59 - 269 registers, an if / else around the quad's logits 41, around each row's four logits it cost the overlap of the exponentials).
    // A tile that holds a piece of the positive diagonal (uniform test; O(tiles_m) tiles) corrects that logit per column
    // quad, in two small blocks around the quad's arithmetic: G = (dcl ? 0 : the above) - gs e.  (As a second copy of the whole loop for
    // those tiles the function spilled 255 registers; as a per-logit select in the one loop it would tax every tile.)
    XC_DEV void to_g(f32x16 (&acc)[4][2], int m0, int n0) {
        const int lane = threadIdx.x & 63, h = lane >> 5;
        const int wave = uniform(threadIdx.x >> 6), wm = wave >> 2, wn = wave & 3;
        const float gm_ = gmul;
        const float a = p.a * gm_, c = p.c * gm_;
        const float gs = p.g_times_scale ? scale : 1.0f;
        const float egs = p.e * gm_ * gs, keep = p.dcl ? 0.f : 1.f;
        constexpr float LOG2E = 1.4426950408889634f;
        // Every load of the tile is unconditional and requested ahead of its use -- the four row lse values, the wave block's 64
        // column values (one per lane, for the spread), then the column quads one quad ahead of the arithmetic.  Written as
        // `a != 0 ? ... lse_q[gm] : 0` each of them was a branch with its own load and full drain of the memory counter: 12 serialized
        // round trips per tile (tools/isa_scan.py).
        const float* const kcol = p.lse_k + n0 + wn * 64 + 4 * h;
        float lq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) lq[i] = p.lse_q[m0 + wm * 128 + i * 32 + (lane & 31)];
        const float lk1 = p.lse_k[n0 + wn * 64 + lane];
        // the reference point: one of the block's own lse values (its first row's, or its first column's when a = 0) -- a uniform load;
        // whether every other lse of the block lies within 60 of it is ONE vote (no cross-lane reduction: a min / max butterfly in front
        // of every tile's arithmetic cost 27 us of 170 at 4096 x 32768, profiles/r04_a_sim_g_after_exact_path.log)
        const float Rq = p.lse_q[m0 + wm * 128], Rk = p.lse_k[n0 + wn * 64];
        u32x4 t = ld16(kcol);
        auto sgpr = [](float v) { return u2f((uint32_t)uniform((int)f2u(v))); };
        const float R = sgpr((a != 0.f) ? Rq : Rk);
        bool far = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) far = far || (a != 0.f && !(fabsf(lq[i] - R) <= 60.f));
        far = far || (c != 0.f && !(fabsf(lk1 - R) <= 60.f));
        const bool exact = (VAR & 2) ? false : wave_any(far);        // (uniform)
        const float scale2 = sgpr(scale * LOG2E), R2 = sgpr(R * LOG2E);
        const float cx = sgpr((c != 0.f) ? gs * c : 0.f);
        const bool on_diag = !sim5_off_diagonal(p, m0, n0);          // (uniform)
        float dt = 0.f;
        const float ax = sgpr((a != 0.f) ? gs * a : 0.f);
        // the exact form rides on the fast one: a small block in front of each column quad (uniform branch, rare) computes the quad's 16
        // values of G with two exponentials each and REPLACES the accumulators by raw' = log2(G) / scale2 -- the fast arithmetic below,
        // run with R = 0 and the factors (1, 0), then reproduces G = exp2(raw' scale2) (zero for the clamped log of 0).  The hot path is
        // the round-3 code plus that branch: as the other side of an if / else around the quad's logits the kernel spilled 41 registers,
        // around each row's four logits it lost the overlap of the exponentials (G 170 -> 206 us, profiles/r04_b_sim_g.log).
        // What robustness costs, same box (profiles/r04_d_sim_g_variants.log, 4096 x 32768 x 512): 181.7 us without the vote and this block
        // (XCLIP_SIMG=2), 196.2 us with them (one vote + ONE copy of the loop nest per tile: 269 spilled registers); without DEFER_FRAGS the
        // spills sit in the tile loop: 220.8 us.  Round 3's 170 us form returned NaN for every wave block whose lse values spread beyond 88.
        const float R2f = exact ? 0.f : R2;                           // (uniform)
        float rowv[4];                                               // the row's factor gs a exp(R - lse_q) -- exact form: 1
        float lq2[4];                                                // (exact form only: the row's lse_q, base 2)
        int dl[4];                                                   // the row's diagonal column, relative to the lane's first column
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gm = m0 + wm * 128 + i * 32 + (lane & 31);
            const float f = (a != 0.f) ? ax * fast_exp(R - lq[i]) : 0.f;
            rowv[i] = exact ? 1.f : f;
            lq2[i] = (a != 0.f) ? lq[i] * LOG2E : 3.0e38f;            // (a = 0: exp2(-huge) = 0, not 0 x exp2(junk))
            dl[i] = gm + p.diag_off - (n0 + wn * 64 + 4 * h);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u32x4 tn = t;
                if (j * 4 + q < 7) tn = ld16(kcol + (j * 4 + q + 1 < 4 ? 0 : 32) + 8 * ((j * 4 + q + 1) & 3));   // the next quad's four lse_k
                float ek[4];                                         // gs c exp(R - lse_k) of the quad's four columns -- exact form: 0
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float f = (c != 0.f) ? cx * fast_exp(R - u2f(t[k])) : 0.f;
                    ek[k] = exact ? 0.f : f;
                }
                float rawd[4] = {0.f, 0.f, 0.f, 0.f};                // the row's positive logit (unscaled), if it lies in this quad
                if (on_diag) {
                    // (DCL: the positive is not part of its row's / column's lse and may exceed it by any amount -- as exp(s - lse) it
                    //  would be inf, and inf x 0 = NaN below: the logit leaves through the loop as exp2(-huge) = 0 instead)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const bool sel = dl[i] == j * 32 + 8 * q + k;
                            rawd[i] = sel ? acc[i][j][4 * q + k] : rawd[i];
                            if (p.dcl) acc[i][j][4 * q + k] = sel ? -1.0e30f : acc[i][j][4 * q + k];
                        }
                }
                if (exact) {                                         // (uniform; rare)
                    const float inv2 = 1.0f / scale2;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float raw = acc[i][j][4 * q + k];
                            const float s2 = raw * scale2;
                            const float lk2 = (c != 0.f) ? u2f(t[k]) * LOG2E : 3.0e38f;
                            const float g = ax * fast_exp2(s2 - lq2[i]) + cx * fast_exp2(s2 - lk2);
                            const float rp = fmaxf(__builtin_log2f(g) * inv2, -1.0e30f);             // raw': exp2(raw' scale2) = g
                            dt += g * raw - fast_exp2(rp * scale2) * rp;                             // (what the loop below adds is taken out)
                            acc[i][j][4 * q + k] = rp;
                        }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float raw = acc[i][j][4 * q + k];
                        const float g = fast_exp2(raw * scale2 - R2f) * (rowv[i] + ek[k]);
                        dt += g * raw;
                        acc[i][j][4 * q + k] = g;
                    }
                if (on_diag) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float delta = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float g = acc[i][j][4 * q + k];
                            const bool sel = dl[i] == j * 32 + 8 * q + k;
                            const float gd = g * keep - egs;
                            delta += sel ? gd - g : 0.f;
                            acc[i][j][4 * q + k] = sel ? gd : g;
                        }
                        dt += delta * rawd[i];
                    }
                }
                t = tn;
                sched_fence();
            }
        dt_acc += dt;
    }
};

// the tiles Sim5FastGradEpilogue skips, as a list: the ragged last column tile of every row tile (if the columns are ragged), then
// every column tile of the ragged last row tile (if the rows are).  The corner both name is a hole in the first part (m0 = rows of the
// padded grid: nothing is stored for it).
struct Sim5EdgeTiles {
    const SimParams& s;
    XC_DEV int count(const Gemm2Params& p) const { return ((s.nk % G2_BN) ? p.tiles_m : 0) + ((s.nq % G2_BM) ? p.tiles_n : 0); }
    XC_DEV void origin(const Gemm2Params& p, int id, int& m0, int& n0) const {
        const bool ragged_rows = (s.nq % G2_BM) != 0, ragged_cols = (s.nk % G2_BN) != 0;
        const int ncol = ragged_cols ? p.tiles_m : 0;
        if (id < ncol) {
            const bool corner = ragged_rows && id == p.tiles_m - 1;
            m0 = (corner ? p.tiles_m : id) * G2_BM;
            n0 = corner ? 0 : (p.tiles_n - 1) * G2_BN;
        } else {
            m0 = (p.tiles_m - 1) * G2_BM;
            n0 = (id - ncol) * G2_BN;
        }
    }
};

template <bool STREAM, int VAR = 0>
__global__ __launch_bounds__(G2_THREADS, 2) void sim5_grad_fast_kernel(SimParams p) {
    XC_LDS_DYNAMIC(lds);
    Gemm2Params g = sim3_gemm_params(p);
    g.C = reinterpret_cast<bf16_t*>(p.G);
    g.ldc = p.ldg;
    g.stream_out = STREAM;
    g5_run<false, false, Sim5FastGradEpilogue<STREAM, VAR>>(g, lds, Sim5FastGradEpilogue<STREAM, VAR>{p, g, sim_scale(p), p.gmul != nullptr ? *p.gmul : 1.0f});
}
__global__ __launch_bounds__(G2_THREADS, 2) void sim5_grad_edge_kernel(SimParams p) {
    XC_LDS_DYNAMIC(lds);
    const Gemm2Params g = sim3_gemm_params(p);
    g3_run<false, false, 0>(g, lds, Sim3GradEpilogue{p}, Sim5EdgeTiles{p});
}

#ifdef XCLIP_MEASURE
// backward (measurement build only, XCLIP_SIM=5: measured SLOWER than simloss3.h's form, see xclip_simloss_grad): interior tiles off the diagonal (all but O(tiles_m) of them) turn their accumulators into G in place and leave through
// the plain GEMM's pack_lines / store_lines; the others keep simloss3.h's row-per-lane form with its range and diagonal tests
struct Sim5GradEpilogue {
    const SimParams& p;
    const Gemm2Params& gp;       // C = G, ldc = ldg, alpha = 1: what the line stores address
    Sim3GradEpilogue slow;
    XC_DEV void finish() { slow.finish(); }
    XC_DEV bool packs_lines(int m0, int n0) const {
        return (m0 + G2_BM <= p.nq) && (n0 + G2_BN <= p.nk) && (m0 + p.diag_off + G2_BM <= n0 || m0 + p.diag_off >= n0 + G2_BN);
    }
    XC_DEV void pack_lines(f32x16 (&acc)[4][2], unsigned char* scratch, u32x4 (&o)[4][4], int m0, int n0) {
        const int lane = threadIdx.x & 63, h = lane >> 5;
        const int wave = uniform(threadIdx.x >> 6), wm = wave >> 2, wn = wave & 3;
        const float scale = sim_scale(p);
        const float gm_ = p.gmul != nullptr ? *p.gmul : 1.0f;
        const float a = p.a * gm_, c = p.c * gm_;
        const float gs = p.g_times_scale ? scale : 1.0f;
        const float R = (a != 0.f) ? p.lse_q[m0 + wm * 128] : p.lse_k[n0 + wn * 64];      // (uniform) see Sim5FastGradEpilogue::to_g
        float dt = 0.f;
        float ek[2][4][4];                                           // c exp(R - lse_k) of the lane's 32 columns, once per tile
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 t = ld16(p.lse_k + n0 + wn * 64 + j * 32 + 4 * h + 8 * q);
#pragma unroll
                for (int k = 0; k < 4; ++k) ek[j][q][k] = (c != 0.f) ? c * fast_exp(R - u2f(t[k])) : 0.f;
            }
        const G4GemmEpilogue<G4_PLAIN> lines{gp};
        // one 32-row group at a time: its 32 accumulators become G in place and go straight into the line exchange, so the group's
        // registers are free again before the next group's exponentials (all four groups first: 220 spilled registers)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gm = m0 + wm * 128 + i * 32 + (lane & 31);
            const float eq = (a != 0.f) ? a * fast_exp(R - p.lse_q[gm]) : 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // exp(s - lse) = exp(s - R) exp(R - lse): ONE exponential per logit
                        const float s_ = acc[i][j][4 * q + k] * scale;
                        const float v = fast_exp(s_ - R) * (eq + ek[j][q][k]);
                        dt += v * s_;
                        acc[i][j][4 * q + k] = v * gs;
                    }
            lines.template pack_lines_i<true>(acc[i], scratch, o[i]);
        }
        slow.dt_acc += dt;
    }
    template <bool NT = false> XC_DEV void store_lines(const u32x4 (&o)[4][4], int m0, int n0) const {
        G4GemmEpilogue<G4_PLAIN>{gp}.template store_lines<NT>(o, m0, n0);
    }
    XC_DEV int with_scratch(f32x16 (&acc)[4][2], int m0, int n0, unsigned char*) { return slow(acc, m0, n0) == 16 ? 16 : 0; }
};

#endif

__global__ __launch_bounds__(G2_THREADS, 2) void sim5_lse_kernel(SimParams p) {
    XC_LDS_DYNAMIC(lds);
    const Gemm2Params g = sim3_gemm_params(p);
    g5_run<false, false, Sim5LseEpilogue>(g, lds, Sim5LseEpilogue{p, sim_scale(p)});
}
#ifdef XCLIP_MEASURE
__global__ __launch_bounds__(G2_THREADS, 2) void sim5_grad_kernel(SimParams p) {
    XC_LDS_DYNAMIC(lds);
    Gemm2Params g = sim3_gemm_params(p);
    g.C = reinterpret_cast<bf16_t*>(p.G);
    g.ldc = p.ldg;
    g.stream_out = (long)p.nq * p.ldg * 2 > (48L << 20);             // G larger than the L2s can hold anyway: streamed stores
    g5_run<false, false, Sim5GradEpilogue>(g, lds, Sim5GradEpilogue{p, g, Sim3GradEpilogue{p}});
}
#endif

}  // namespace xc
