// simloss3.h -- the contrastive head (simloss.h: S = scale * Q K^T reduced to log-sum-exp partials in the forward, turned into
// the gradient factor G in the backward; reference x_clip.py:813-847) on the production GEMM loop of gemm3.h: persistent
// 256 x 256 tiles, LDS-DMA staged operands, and -- because the MFMAs are issued transposed -- a lane that owns ONE logit row:
//   forward : max and sum-of-exp over the lane's 32 columns are in-lane, one shuffle merges the two half-waves, each wave
//             writes a (max, sum) partial per row for its 64-column slot; the logits are never stored;
//   backward: G = gmul {[a exp(S - lse_q) + c exp(S - lse_k)] (1 - dcl d_ij) - e d_ij} (optionally x scale) leaves with the
//             GEMM's 16-byte row stores; sum G o S accumulates into d tau.
// Measured at the configs[2] per-rank block (4096 x 32768 x 512): see profiles/ (the 128 x 128 register-staged versions in
// simloss.h ran at 338 / 245 TF/s and remain as the fp32 / small-problem path).
#pragma once
#include "gemm3.h"
#include "simloss.h"

namespace xc {

struct Sim3LseEpilogue {
    XC_DEV void finish() {}
    const SimParams& p;
    // scale = exp(tau) x host factor, read ONCE per work-group (the kernels pass sim_scale(p)): as a per-tile read of *log_scale it was
    // a vector load + a full drain of the memory counter in front of every tile's epilogue (the compiler cannot use a scalar load past
    // the previous tile's stores), the only load the interior path has
    float scale;
    XC_DEV int operator()(f32x16 (&acc)[4][2], int m0, int n0) const {
        const int lane = threadIdx.x & 63, h = lane >> 5;
        const int wave = uniform(threadIdx.x >> 6), wm = wave >> 2, wn = wave & 3;
        const int c0 = n0 + wn * 64;                               // this wave's 64-column slot
        if (c0 >= p.nk) return 0;
        const long slot = c0 >> 6;
        // interior tiles that do not touch the diagonal (all but O(tiles_m) of them) need no range / diagonal tests
        const bool plain = (m0 + G2_BM <= p.nq) && (n0 + G2_BN <= p.nk) &&
                           (m0 + p.diag_off + G2_BM <= n0 || m0 + p.diag_off >= n0 + G2_BN);
        if (plain) {
            const float scale2 = scale * 1.4426950408889634f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gm = m0 + wm * 128 + i * 32 + (lane & 31);
                float mx = SIM_NEG;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[i][j][r]);
                const float mx2 = mx * scale2;                       // (base-2 domain: one fma + a bare v_exp_f32 per logit)
                mx *= scale;                                         // scale = exp(tau) x host factor > 0
                float l = 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) l += fast_exp2(acc[i][j][r] * scale2 - mx2);
                const float m2 = shfl_xor(mx, 32), l2 = shfl_xor(l, 32);
                const float mm = fmaxf(mx, m2);
                const float ll = l * fast_exp(mx - mm) + l2 * fast_exp(m2 - mm);
                if (h == 0) {
                    p.part_m[slot * p.nq + gm] = mm;
                    p.part_l[slot * p.nq + gm] = ll;
                }
            }
            return 8;                                                // vector-memory instructions left behind (g5_run: LOOSE8)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gm = m0 + wm * 128 + i * 32 + (lane & 31);
            const bool valid = gm < p.nq;
            const int dcol = gm + p.diag_off;
            float mx = SIM_NEG;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gn = c0 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float s = acc[i][j][r] * scale;
                    acc[i][j][r] = s;
                    if (gn < p.nk) {
                        if (gn == dcol && valid) p.pos[gm] = s;
                        if (!(p.dcl && gn == dcol)) mx = fmaxf(mx, s);
                    }
                }
            float l = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gn = c0 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (gn < p.nk && !(p.dcl && gn == dcol)) l += fast_exp(acc[i][j][r] - mx);
                }
            const float m2 = shfl_xor(mx, 32), l2 = shfl_xor(l, 32);
            const float mm = fmaxf(mx, m2);
            const float ll = l * fast_exp(mx - mm) + l2 * fast_exp(m2 - mm);
            if (h == 0 && valid) {
                p.part_m[slot * p.nq + gm] = mm;
                p.part_l[slot * p.nq + gm] = ll;
            }
        }
        return 0;
    }
};

struct Sim3GradEpilogue {
    const SimParams& p;
    float dt_acc = 0.f;          // this lane's share of sum G o S over ALL tiles of the work-group: one atomic per wave at the end
                                 // (per tile it was 16k same-address atomics at 32k x 4k, each on the next barrier's critical path)
    XC_DEV void finish() {
        const float dt = wave_sum(dt_acc);
        if ((threadIdx.x & 63) == 0 && p.dtau != nullptr) atomic_add(p.dtau, dt);
    }
    XC_DEV int operator()(f32x16 (&acc)[4][2], int m0, int n0) {
        const int lane = threadIdx.x & 63, h = lane >> 5;
        const int wave = uniform(threadIdx.x >> 6), wm = wave >> 2, wn = wave & 3;
        const float scale = sim_scale(p);
        const float gm_ = p.gmul != nullptr ? *p.gmul : 1.0f;
        const float a = p.a * gm_, c = p.c * gm_, e = p.e * gm_;
        const float gs = p.g_times_scale ? scale : 1.0f;
        const int nkp = (p.nk + 7) & ~7;                           // G rows are padded to a whole 16-byte chunk with zeros
        const bool full = (m0 + G2_BM <= p.nq) && (n0 + G2_BN <= p.nk);
        bf16_t* G = reinterpret_cast<bf16_t*>(p.G);
        float dt = 0.f;
        const bool plain = full && (m0 + p.diag_off + G2_BM <= n0 || m0 + p.diag_off >= n0 + G2_BN);
        if (plain) {
            // the lane's 32 columns (and their lse_k) are the same for all four row blocks: 8 float4 loads per tile
            const float R = (a != 0.f) ? p.lse_q[m0 + wm * 128] : p.lse_k[n0 + wn * 64];      // (uniform) a log-sum-exp of the tile itself
            float ek[2][4][4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x4 t = ld16(p.lse_k + n0 + wn * 64 + j * 32 + 4 * h + 8 * q);
#pragma unroll
                    // exp(s - lse) = exp(s - R) exp(R - lse): ONE exponential per logit; the per-row / per-column factors are computed
                    // once per tile (simloss5.h to_g: why R is one of the tile's own lse values and not `scale`)
                    for (int k = 0; k < 4; ++k) ek[j][q][k] = (c != 0.f) ? c * fast_exp(R - u2f(t[k])) : 0.f;
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gm = m0 + wm * 128 + i * 32 + (lane & 31);
                const float eq = (a != 0.f) ? a * fast_exp(R - p.lse_q[gm]) : 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int nb = n0 + wn * 64 + j * 32;
                    uint32_t pk[4][2];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float g[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float s_ = acc[i][j][4 * q + k] * scale;
                            const float v = fast_exp(s_ - R) * (eq + ek[j][q][k]);
                            dt += v * s_;
                            g[k] = v * gs;
                        }
                        pk[q][0] = (uint32_t)f2bf(g[0]) | ((uint32_t)f2bf(g[1]) << 16);
                        pk[q][1] = (uint32_t)f2bf(g[2]) | ((uint32_t)f2bf(g[3]) << 16);
                    }
#pragma unroll
                    for (int qq = 0; qq < 4; qq += 2) {
                        permlane32_swap(pk[qq][0], pk[qq + 1][0]);
                        permlane32_swap(pk[qq][1], pk[qq + 1][1]);
                        u32x4 o = {pk[qq][0], pk[qq][1], pk[qq + 1][0], pk[qq + 1][1]};
                        st16(G + (long)gm * p.ldg + nb + qq * 8 + 8 * h, o);
                    }
                }
            }
            dt_acc += dt;
            return 16;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gm = m0 + wm * 128 + i * 32 + (lane & 31);
            const bool row_ok = full || gm < p.nq;
            const float lq = p.lse_q[row_ok ? gm : p.nq - 1];
            const int dcol = gm + p.diag_off;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nb = n0 + wn * 64 + j * 32;
                uint32_t pk[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float g[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int gn = nb + 4 * h + 8 * q + k;
                        const float s = acc[i][j][4 * q + k] * scale;
                        const bool diag = gn == dcol;
                        float v = 0.f;
                        if (row_ok && (full || gn < p.nk)) {
                            if (!(p.dcl && diag)) {
                                if (a != 0.f) v += a * fast_exp(s - lq);
                                if (c != 0.f) v += c * fast_exp(s - p.lse_k[gn]);
                            }
                            if (diag) v -= e;
                            dt += v * s;
                        }
                        g[k] = v * gs;
                    }
                    pk[q][0] = (uint32_t)f2bf(g[0]) | ((uint32_t)f2bf(g[1]) << 16);
                    pk[q][1] = (uint32_t)f2bf(g[2]) | ((uint32_t)f2bf(g[3]) << 16);
                }
#pragma unroll
                for (int qq = 0; qq < 4; qq += 2) {
                    permlane32_swap(pk[qq][0], pk[qq + 1][0]);
                    permlane32_swap(pk[qq][1], pk[qq + 1][1]);
                    const int gn = nb + qq * 8 + 8 * h;
                    if (row_ok && (full || gn < nkp)) {
                        u32x4 o = {pk[qq][0], pk[qq][1], pk[qq + 1][0], pk[qq + 1][1]};
                        st16(G + (long)gm * p.ldg + gn, o);
                    }
                }
            }
        }
        dt_acc += dt;
        return full ? 16 : 0;
    }
};

XC_DEV Gemm2Params sim3_gemm_params(const SimParams& p) {
    Gemm2Params g;
    g.A = reinterpret_cast<const bf16_t*>(p.Q); g.B = reinterpret_cast<const bf16_t*>(p.K); g.C = nullptr;
    g.lda = p.d; g.ldb = p.d; g.ldc = 0;
    g.M = p.nq; g.N = p.nk; g.K = p.d; g.alpha = 1.f;
    g.bias = nullptr; g.residual = nullptr; g.ldr = 0; g.addrows = nullptr; g.rowidx = nullptr; g.ld_add = 0;
    g.partial = nullptr; g.k_per_split = p.d;
    g.tiles_m = (p.nq + G2_BM - 1) / G2_BM; g.tiles_n = (p.nk + G2_BN - 1) / G2_BN;
    g.stream_out = 0;
    // ring-loop kernels (simloss5.h on g5_run): banded tile order, as the GEMM's (xclip_gemm) -- an XCD keeps a band of K panels in its L2
    // and walks the row tiles against it.  In plain order each XCD owns whole row tiles and streams EVERY K panel past them: at
    // 4096 x 32768 x 512 the fabric delivered 557 MB per launch for 37.7 MB of operands (profiles/r03_x_sim_hbm_traffic_pmc.txt)
    g.band_n = 0;
    if (g.tiles_n > 8)
        for (int b = 8; b >= 4 && g.band_n == 0; --b)
            if (g.tiles_n % b == 0) g.band_n = b;
    return g;
}

__global__ __launch_bounds__(G2_THREADS, 2) void sim3_lse_kernel(SimParams p) {
    XC_LDS_DYNAMIC(lds);
    const Gemm2Params g = sim3_gemm_params(p);
    g3_run<false, false, 0>(g, lds, Sim3LseEpilogue{p, sim_scale(p)});
}
__global__ __launch_bounds__(G2_THREADS, 2) void sim3_grad_kernel(SimParams p) {
    XC_LDS_DYNAMIC(lds);
    const Gemm2Params g = sim3_gemm_params(p);
    g3_run<false, false, 0>(g, lds, Sim3GradEpilogue{p});
}

}  // namespace xc
