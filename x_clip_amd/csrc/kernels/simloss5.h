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
