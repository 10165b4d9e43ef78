// simloss.h -- the contrastive head: similarity matrix + InfoNCE / DCL (reference x_clip.py:813-847), forward
// and backward, without ever materialising the logits.
//
// One primitive covers text->image, image->text, the CLOOB extra-projection variant, multiview pairs and the
// rank-sharded row/column blocks of the multi-GPU path:
//
//   forward   S = scale * Q K^T   (Q: [nq, d] "own" latents, K: [nk, d] the other modality, scale = exp(tau))
//             lse_i = log sum_j exp(S_ij)      (j = i + diag_off left out when dcl)
//             pos_i = S_{i, i + diag_off}
//             loss += coef * sum_i (lse_i - pos_i)
//   backward  G_ij = gmul {[a exp(S_ij - lse_q[i]) + c exp(S_ij - lse_k[j])] (1 - dcl * delta) - e * delta}     (delta = [j == i + diag_off])
//             G (optionally pre-multiplied by scale) is written once (storage dtype) and consumed by two ordinary GEMMs:
//             dQ = scale * G K ("NN"),  dK = scale * G^T Q ("TN");   dtau += sum_ij G_ij S_ij.
//   scale = host float x exp(device scalar): the temperature and the upstream loss gradient never visit the host.
//
// The S tile is produced by the GEMM main loop (128x128 per work-group, MFMA, fp32 in LDS); the forward
// epilogue reduces each tile row to online-softmax partials (max, sum) per 64-column slot -- two threads per row -- and a
// second tiny kernel folds the per-slot partials.  Algorithmic HBM traffic of the forward: (nq + nk) * d * e read,
// 2 * nq * tiles_n * 4 written; the backward adds nq * nk * e for G (written once, read twice).
#pragma once
#include "gemm.h"

namespace xc {

struct SimParams {
    const void* Q; const void* K;
    int nq, nk, d;
    float scale;                           // S = scale * exp(*log_scale) * Q K^T
    const float* log_scale;                // device scalar (the temperature parameter tau, x_clip.py:574,736) or null
    const float* gmul;                     // device scalar multiplying G (the upstream d loss) or null
    int g_times_scale;                     // store scale * G instead of G (folds temp into the dQ / dK GEMMs)
    int diag_off, dcl;
    int tiles_m, tiles_n;
    // forward
    float* part_m; float* part_l;          // [tiles_n][nq]
    float* pos;                            // [nq]
    // backward
    const float* lse_q; const float* lse_k;
    float a, c, e;
    void* G; long ldg;
    float* dtau;                           // scalar accumulator
};

constexpr float SIM_NEG = -3.0e38f;

XC_DEV float sim_scale(const SimParams& p) { return p.log_scale != nullptr ? p.scale * expf(*p.log_scale) : p.scale; }

template <typename T>
__global__ __launch_bounds__(256) void sim_lse_partial_kernel(SimParams p) {
    constexpr int LDC = GemmCfg<T>::LDC;
    XC_LDS_DYNAMIC(lds);
    const float* Cs = reinterpret_cast<const float*>(lds);
    const int tid = threadIdx.x;
    const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tn = tile % p.tiles_n;
    const int m0 = (tile / p.tiles_n) * GEMM_BM, n0 = tn * GEMM_BN;
    gemm_mainloop<T, false, false>(reinterpret_cast<const T*>(p.Q), p.d, reinterpret_cast<const T*>(p.K), p.d, p.nq, p.nk,
                                   m0, n0, 0, p.d, lds);
    // two threads per row, 64 columns each, then one shuffle to merge the pair
    const float scale = sim_scale(p);
    const int row = tid >> 1, half = tid & 1;
    const int gm = m0 + row;
    const int dcol = gm + p.diag_off;
    float m = SIM_NEG, l = 0.f;
    for (int c4 = 0; c4 < 16; ++c4) {
        const int col = half * 64 + c4 * 4;
        float v[4];
        load_vec<float>(Cs + row * LDC + col, v);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int gn = n0 + col + k;
            const float s = v[k] * scale;
            if (gn < p.nk) {
                if (gn == dcol && gm < p.nq) p.pos[gm] = s;
                if (!(p.dcl && gn == dcol)) {
                    if (s > m) {
                        l = l * fast_exp(m - s) + 1.0f;
                        m = s;
                    } else {
                        l += fast_exp(s - m);
                    }
                }
            }
        }
    }
    // one (max, sum) partial per row and 64-column SLOT (two slots per 128-column tile: one per thread of the row pair)
    if (gm < p.nq && n0 + half * 64 < p.nk) {
        const long slot = (long)tn * 2 + half;
        p.part_m[slot * p.nq + gm] = m;
        p.part_l[slot * p.nq + gm] = l;
    }
}

// Fold the per-slot partials:  lse_i ; loss += coef * sum_i (lse_i - pos_i).
// Work-group = 64 rows x 16 waves: lane = row (every load is a coalesced 256-byte row segment of part_m / part_l [slots][nq]),
// wave w folds slots w, w + 16, ... online, the 16 per-wave (max, sum) pairs of a row meet in LDS.  (One thread per row
// walking all slots -- the first version -- took longer than the MFMA kernel it follows at nk = 32768.)
__global__ __launch_bounds__(1024) void sim_lse_combine_kernel(const float* __restrict__ part_m, const float* __restrict__ part_l,
                                                               const float* __restrict__ pos, float* __restrict__ lse,
                                                               float* __restrict__ loss, int nq, int slots, float coef) {
    XC_LDS_DYNAMIC(lds);
    float* red_m = reinterpret_cast<float*>(lds);           // [16][64]
    float* red_l = red_m + 16 * 64;
    const int lane = lane_id(), wave = wave_id();
    const int i = blockIdx.x * 64 + lane;
    float m = SIM_NEG, l = 0.f;
    if (i < nq) {
        for (int t = wave; t < slots; t += 16) {
            const float pm = part_m[(long)t * nq + i], pl = part_l[(long)t * nq + i];
            const float mm = fmaxf(m, pm);
            l = l * fast_exp(m - mm) + pl * fast_exp(pm - mm);
            m = mm;
        }
    }
    red_m[wave * 64 + lane] = m;
    red_l[wave * 64 + lane] = l;
    sync();
    if (wave == 0) {
        float contrib = 0.f;
        if (i < nq) {
            float mm = SIM_NEG;
#pragma unroll
            for (int w = 0; w < 16; ++w) mm = fmaxf(mm, red_m[w * 64 + lane]);
            float ll = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) ll += red_l[w * 64 + lane] * fast_exp(red_m[w * 64 + lane] - mm);
            const float v = (ll > 0.f) ? mm + logf(ll) : logf(1e-20f);     // reference: log(sum + 1e-20)
            lse[i] = v;
            contrib = coef * (v - pos[i]);
        }
        contrib = wave_sum(contrib);
        if (lane == 0 && loss != nullptr) atomic_add(loss, contrib);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void sim_grad_kernel(SimParams p) {
    constexpr int VEC = Elem<T>::VEC, LDC = GemmCfg<T>::LDC;
    XC_LDS_DYNAMIC(lds);
    const float* Cs = reinterpret_cast<const float*>(lds);
    const int tid = threadIdx.x;
    const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int m0 = (tile / p.tiles_n) * GEMM_BM, n0 = (tile % p.tiles_n) * GEMM_BN;
    gemm_mainloop<T, false, false>(reinterpret_cast<const T*>(p.Q), p.d, reinterpret_cast<const T*>(p.K), p.d, p.nq, p.nk,
                                   m0, n0, 0, p.d, lds);
    constexpr int CPR = 128 / VEC;
    T* G = reinterpret_cast<T*>(p.G);
    const float scale = sim_scale(p);
    const float gm = p.gmul != nullptr ? *p.gmul : 1.0f;
    const float a = p.a * gm, c = p.c * gm, e = p.e * gm;
    const float gs = p.g_times_scale ? scale : 1.0f;
    float dt = 0.f;
    for (int id = tid; id < 128 * CPR; id += GEMM_THREADS) {
        const int row = id / CPR, col = (id % CPR) * VEC;
        const int gm = m0 + row, gn0 = n0 + col;
        if (gm < p.nq && gn0 < p.nk) {                      // G rows are padded to a whole chunk: columns >= nk get 0
            const float lq = p.lse_q[gm];
            const int dcol = gm + p.diag_off;
            float g[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const int gn = gn0 + k;
                const float s = Cs[row * LDC + col + k] * scale;
                const bool diag = (gn == dcol);
                float v = 0.f;
                if (gn < p.nk) {
                    if (!(p.dcl && diag)) {                 // a zero coefficient switches its term off (no exp -> no inf * 0)
                        if (a != 0.f) v += a * fast_exp(s - lq);
                        if (c != 0.f) v += c * fast_exp(s - p.lse_k[gn]);
                    }
                    if (diag) v -= e;
                    dt += v * s;
                }
                g[k] = v * gs;
            }
            store_vec<T>(G + (long)gm * p.ldg + gn0, g);
        }
    }
    dt = wave_sum(dt);
    if (lane_id() == 0 && p.dtau != nullptr) atomic_add(p.dtau, dt);
}

}  // namespace xc
