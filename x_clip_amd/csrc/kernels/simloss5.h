// simloss5.h -- the contrastive head (simloss3.h: S = scale * Q K^T reduced to log-sum-exp partials in the forward, turned into the
// gradient factor G in the backward; reference x_clip.py:813-847) on the PRODUCTION GEMM loop of gemm4.h (g5_run): the Q operand --
// the one that streams, K's column panel is re-read from L2 by every tile of its column -- in a ring of three LDS stages, descriptor-
// addressed LDS DMA, the counted waits.  Forward (log-sum-exp partials, nothing stored): 177 -> 157 us = 876 TFLOP/s at the configs[2]
// per-rank block 4096 x 32768 x 512 (the same-shape plain GEMM that WRITES the logits: 144 us; profiles/r03_b_sim_kernels_32k.log).
// The G kernel was moved too (whole-line epilogue, the next tile's early Q pieces) and got slower, 423 against 362 us: it stays on
// simloss3.h in the product and lives on here for the measurement build only.
#pragma once
#include "gemm4.h"
#include "simloss3.h"

namespace xc {

// forward: no stores at all -- the g5_run protocol's "nothing left in flight" epilogue
struct Sim5LseEpilogue {
    const SimParams& p;
    XC_DEV void finish() {}
    XC_DEV bool packs_lines(int, int) const { return false; }
    XC_DEV void pack_lines(f32x16 (&)[4][2], unsigned char*, u32x4 (&)[4][4], int, int) const {}
    template <bool NT = false> XC_DEV void store_lines(const u32x4 (&)[4][4], int, int) const {}
    // (measured and not kept: re-reading the epilogue's parameters from the kernarg segment per tile -- params_in_memory(), 54 -> 35
    //  spilled SGPRs -- made the kernel SLOWER, 194 against 157 us: every field access became its own scalar load + wait;
    //  profiles/r03_f_sim_kernels_32k.log)
    XC_DEV int with_scratch(f32x16 (&acc)[4][2], int m0, int n0, unsigned char*) const { return Sim3LseEpilogue{p}(acc, m0, n0); }
};

// ---- backward: G in two launches -----------------------------------------------------------------------------------------------------
// Why G was slow (362 us at 4096 x 32768 x 512 where the plain GEMM that stores the same 268 MB takes 144): both earlier forms keep the
// general tile -- per-element range and diagonal tests -- in the same function as the interior one, and around the 128 accumulators the
// compiler then spills 100 - 240 vector registers INTO THE INTERIOR PATH: ~23 us of epilogue per tile where the exponentials and the
// stores account for ~9.  Here the interior tiles off the diagonal (all but O(tiles_m) of them) have a kernel of their own on the ring
// loop: the accumulators become G in place, column quad by column quad (the four per-column factors live only that long), and leave
// through the plain GEMM's line exchange as whole-line stores -- 217 registers, nothing spilled, no state carried across the K loop
// (sum G o S: one atomic per wave and tile).  The tiles it skips -- on the diagonal, at a ragged edge -- are walked by a second,
// small launch of simloss3.h's kernel over a tile LIST (Sim5EdgeTiles).
XC_DEV bool sim5_plain_tile(const SimParams& p, int m0, int n0) {
    return (m0 + G2_BM <= p.nq) && (n0 + G2_BN <= p.nk) && (m0 + p.diag_off + G2_BM <= n0 || m0 + p.diag_off >= n0 + G2_BN);
}

// (STREAM -- non-temporal stores for a G the L2s cannot hold anyway -- is a template parameter: as a run-time branch around the 16
//  stores it cost this kernel 102 spilled registers)
// (LINES: the tile leaves through g5_run's pack_lines / store_lines pair -- the next tile's first four A pieces are issued between the
//  two, in front of the stores -- instead of storing from with_scratch, group by group)
template <bool STREAM, bool LINES = false>
struct Sim5FastGradEpilogue {
    const SimParams& p;
    const Gemm2Params& gp;       // C = G, ldc = ldg, alpha = 1: what the line stores address
    float dt_acc = 0.f;          // this lane's share of sum G o acc over the work-group's tiles (one register across the K loops)
    XC_DEV void finish() {
        const float dt = wave_sum(dt_acc) * (sim_scale(p) / (p.g_times_scale ? sim_scale(p) : 1.0f));
        if ((threadIdx.x & 63) == 0 && p.dtau != nullptr) atomic_add(p.dtau, dt);
    }
    XC_DEV bool packs_lines(int m0, int n0) const { return LINES && sim5_plain_tile(p, m0, n0); }
    XC_DEV void pack_lines(f32x16 (&acc)[4][2], unsigned char* scratch, u32x4 (&o)[4][4], int m0, int n0) {
        to_g(acc, m0, n0);
        const G4GemmEpilogue<G4_PLAIN> lines{gp};
#pragma unroll
        for (int i = 0; i < 4; ++i) lines.template pack_lines_i<true>(acc[i], scratch, o[i]);
    }
    template <bool NT = false> XC_DEV void store_lines(const u32x4 (&o)[4][4], int m0, int n0) const {
        G4GemmEpilogue<G4_PLAIN>{gp}.template store_lines<STREAM>(o, m0, n0);
    }
    XC_DEV int with_scratch(f32x16 (&acc)[4][2], int m0, int n0, unsigned char* scratch) {
        if (LINES || !sim5_plain_tile(p, m0, n0)) return 0;         // (uniform) the edge launch's tile
        to_g(acc, m0, n0);
        const int lane = threadIdx.x & 63;
        const int wave = uniform(threadIdx.x >> 6), wm = wave >> 2, wn = wave & 3;
        const G4GemmEpilogue<G4_PLAIN> lines{gp};
        const BufRsrc rc = make_rsrc(gp.C + (long)m0 * gp.ldc + n0, 255u * (uint32_t)gp.ldc * 2u + 512u);
        const uint32_t vc = ((uint32_t)(wm * 128 + (lane >> 3)) * (uint32_t)gp.ldc + (uint32_t)(wn * 64 + 8 * (lane & 7))) * 2u;
        const uint32_t s8 = (uint32_t)gp.ldc * 16u;                  // 8 rows * ldc * 2 bytes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x4 o[4];
            lines.template pack_lines_i<true>(acc[i], scratch, o);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (STREAM) buf_st16_nt<0>(rc, vc, s8 * (uint32_t)(4 * i + k), o[k]);
                else buf_st16<0>(rc, vc, s8 * (uint32_t)(4 * i + k), o[k]);
            }
        }
        // the 16 stores are YOUNGER than the A / B pieces the next tile's first K step waits for and
        // may stay in flight over it (g5_run: in_flight == 16) -- returning 0 here made that step wait for G to reach memory
        return 16;
    }
    // the accumulators of an interior tile off the diagonal become G in place
    XC_DEV void to_g(f32x16 (&acc)[4][2], int m0, int n0) {
        const int lane = threadIdx.x & 63, h = lane >> 5;
        const int wave = uniform(threadIdx.x >> 6), wm = wave >> 2, wn = wave & 3;
        const float scale = sim_scale(p);
        const float gm_ = p.gmul != nullptr ? *p.gmul : 1.0f;
        const float a = p.a * gm_, c = p.c * gm_;
        const float gs = p.g_times_scale ? scale : 1.0f;
        // Per logit: G = exp(s - scale) (a' + c') with s = acc * scale, a' = gs a exp(scale - lse_q), c' = gs c exp(scale - lse_k) -- ONE
        // exponential per logit (|cos| <= 1, so s <= scale), in the base-2 domain: one fma + a bare v_exp_f32; then an add, a multiply,
        // and an fma for sum G o acc (d tau = that sum x scale / gs, applied once per tile).  Four vector instructions + the exponential
        // (the first form spent eight: the epilogue's cost over the plain GEMM's was ~12 us per tile, all of it VALU).
        const float scale2 = scale * 1.4426950408889634f;
        float dt = 0.f;
        float eq[4];                                                 // gs a exp(scale - lse_q) of the lane's row in each 32-row group
#pragma unroll
        for (int i = 0; i < 4; ++i) eq[i] = gs * a * fast_exp(scale - p.lse_q[m0 + wm * 128 + i * 32 + (lane & 31)]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 t = ld16(p.lse_k + n0 + wn * 64 + j * 32 + 4 * h + 8 * q);
                float ek[4];                                         // gs c exp(scale - lse_k) of the quad's four columns
#pragma unroll
                for (int k = 0; k < 4; ++k) ek[k] = gs * c * fast_exp(scale - u2f(t[k]));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float raw = acc[i][j][4 * q + k];
                        const float g = fast_exp2(raw * scale2 - scale2) * (eq[i] + ek[k]);
                        dt += g * raw;
                        acc[i][j][4 * q + k] = g;
                    }
            }
        dt_acc += dt;
    }
};

// the tiles Sim5FastGradEpilogue skips, as a list with a fixed number of slots: per row tile m the one or two column tiles its
// diagonal segment [m0 + off, m0 + off + 256) touches (slots 0, 1) and the ragged last column tile (slot 2); then, if the last row
// tile is ragged, all of its column tiles.  A slot whose tile does not exist, is interior after all, or is already named by an
// earlier slot is a hole (m0 = rows of the padded grid: nothing is stored for it).
struct Sim5EdgeTiles {
    const SimParams& s;
    XC_DEV int count(const Gemm2Params& p) const { return 3 * p.tiles_m + ((s.nq % G2_BM) ? p.tiles_n : 0); }
    XC_DEV void origin(const Gemm2Params& p, int id, int& m0, int& n0) const {
        const int hole = p.tiles_m * G2_BM;
        const bool ragged_rows = (s.nq % G2_BM) != 0, ragged_cols = (s.nk % G2_BN) != 0;
        int m, c;
        if (id >= 3 * p.tiles_m) {                                   // the ragged last row tile: every column tile
            m = p.tiles_m - 1;
            c = id - 3 * p.tiles_m;
        } else {
            m = id / 3;
            const int slot = id - 3 * m;
            const int d0 = m * G2_BM + s.diag_off;                   // first diagonal column of the row tile
            const int lo = d0 >= 0 ? d0 / G2_BN : -1, hi = d0 + G2_BM - 1 >= 0 ? (d0 + G2_BM - 1) / G2_BN : -1;
            if (slot == 0) c = lo;
            else if (slot == 1) c = hi != lo ? hi : -1;
            else c = (ragged_cols && p.tiles_n - 1 != lo && p.tiles_n - 1 != hi) ? p.tiles_n - 1 : -1;
            if (ragged_rows && m == p.tiles_m - 1) c = -1;           // that row tile is listed whole below
        }
        if (c < 0 || c >= p.tiles_n || sim5_plain_tile(s, m * G2_BM, c * G2_BN)) { m0 = hole; n0 = 0; return; }
        m0 = m * G2_BM;
        n0 = c * G2_BN;
    }
};

template <bool STREAM, bool LINES = false>
__global__ __launch_bounds__(G2_THREADS, 2) void sim5_grad_fast_kernel(SimParams p) {
    XC_LDS_DYNAMIC(lds);
    Gemm2Params g = sim3_gemm_params(p);
    g.C = reinterpret_cast<bf16_t*>(p.G);
    g.ldc = p.ldg;
    g.stream_out = STREAM;
    g5_run<false, false, Sim5FastGradEpilogue<STREAM, LINES>>(g, lds, Sim5FastGradEpilogue<STREAM, LINES>{p, g});
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
        float dt = 0.f;
        float ek[2][4][4];                                           // c exp(scale - lse_k) of the lane's 32 columns, once per tile
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 t = ld16(p.lse_k + n0 + wn * 64 + j * 32 + 4 * h + 8 * q);
#pragma unroll
                for (int k = 0; k < 4; ++k) ek[j][q][k] = (c != 0.f) ? c * fast_exp(scale - u2f(t[k])) : 0.f;
            }
        const G4GemmEpilogue<G4_PLAIN> lines{gp};
        // one 32-row group at a time: its 32 accumulators become G in place and go straight into the line exchange, so the group's
        // registers are free again before the next group's exponentials (all four groups first: 220 spilled registers)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gm = m0 + wm * 128 + i * 32 + (lane & 31);
            const float eq = (a != 0.f) ? a * fast_exp(scale - p.lse_q[gm]) : 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // exp(s - lse) = exp(s - scale) exp(scale - lse): ONE exponential per logit (|cos| <= 1, so s <= scale)
                        const float s_ = acc[i][j][4 * q + k] * scale;
                        const float v = fast_exp(s_ - scale) * (eq + ek[j][q][k]);
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
    g5_run<false, false, Sim5LseEpilogue>(g, lds, Sim5LseEpilogue{p});
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
