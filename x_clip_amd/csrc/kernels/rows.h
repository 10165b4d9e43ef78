// rows.h -- HBM-bound row kernels: gain-only LayerNorm (plain / +residual / fused GEGLU prologue) and
// l2-normalisation, forward and backward.
//
// Layout: one wave64 per row, the whole row held in registers (lane owns the 16-byte chunks
// c = lane + 64*i), every global access a coalesced 16-byte transaction, all reductions are wave
// shuffles -- no LDS in the forward kernels.  Algorithmic HBM traffic: forward 2 x rows x D x e
// (+ residual read), backward 3 x rows x D x e.  A row is read once and written once: the streamed accesses carry the
// non-temporal hint where the launcher's template argument NT says so (xclip_api.hip ROWS_NT: measured per kernel).
//
// Semantics follow the reference LayerNorm (x_clip.py:112-121: biased variance, gain only, the caller
// passes eps = 1e-5 for fp32 and 1e-3 otherwise), GEGLU (x_clip.py:180-183: value = first half, gate =
// second half, erf GELU) and F.normalize (x_clip.py:54-55, eps 1e-12).  Arithmetic is fp32 whatever the
// storage type.
#pragma once
#include "common.h"

namespace xc {

// ---- row <-> registers --------------------------------------------------------------------------------
// Loads row `xr` (width D); with GEGLU the source row is [value(D) | gate(D)] and the loaded value is
// value * gelu(gate) (u/gt return the raw halves for the backward).
template <typename T, int MAXC, bool GEGLU, bool NT = false>
XC_DEV void load_row(const T* xr, int D, int lane, float (&v)[MAXC][Elem<T>::VEC]) {
    constexpr int VEC = Elem<T>::VEC;
    const int nch = D / VEC;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            load_vec<T, NT>(xr + c * VEC, v[i]);
            if (GEGLU) {
                float gt[VEC];
                load_vec<T, NT>(xr + D + c * VEC, gt);
#pragma unroll
                for (int j = 0; j < VEC; ++j) v[i][j] *= gelu_erf(gt[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[i][j] = 0.f;
        }
    }
}

template <typename T, int MAXC>
XC_DEV void row_stats(const float (&v)[MAXC][Elem<T>::VEC], int D, int lane, float& mean, float& var) {
    constexpr int VEC = Elem<T>::VEC;
    const int nch = D / VEC;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
        for (int j = 0; j < VEC; ++j) s += v[i][j];            // padding chunks hold zeros
    mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        if (lane + 64 * i < nch) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float d = v[i][j] - mean;
                q += d * d;
            }
        }
    }
    var = wave_sum(q) / (float)D;
}

// ---- LayerNorm forward:  y = (x - mean) * rstd * g (+ res) ---------------------------------------------
template <typename T, int MAXC, bool GEGLU, bool NT = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ g,
                                                     const T* __restrict__ res, T* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     int rows, int D, float eps, long ldy, int y_grp) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = lane_id();
    const long row = (long)blockIdx.x * 4 + wave_id();
    if (row >= rows) return;                       // whole wave leaves together; no barriers below
    const int nch = D / VEC;
    // output row: `y_grp` > 0 leaves one row free in front of every group of y_grp rows (the CLS slot of
    // the vision encoder output, x_clip.py:389-390), so the final LayerNorm writes straight into [b, 1+n, D]
    const long yrow = y_grp > 0 ? row + row / y_grp + 1 : row;
    float v[MAXC][VEC];
    load_row<T, MAXC, GEGLU, NT>(x + row * ldx, D, lane, v);
    float mean, var;
    row_stats<T, MAXC>(v, D, lane, mean, var);
    const float rstd = fast_rsqrt(var + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            float gv[VEC], o[VEC];
            load_vec<T>(g + c * VEC, gv);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = (v[i][j] - mean) * rstd * gv[j];
            if (res != nullptr) {
                float rv[VEC];
                load_vec<T, NT>(res + row * (long)D + c * VEC, rv);
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] += rv[j];
            }
            store_vec<T, NT>(y + yrow * ldy + c * VEC, o);
        }
    }
    if (lane == 0) {
        mean_out[row] = mean;
        rstd_out[row] = rstd;
    }
}

#ifdef XCLIP_MEASURE
// (measurement build only: measured, not faster -- see launch_ln_fwd)  The same LayerNorm forward (no GEGLU) with RPW rows per wave and every load of a row pair -- x, the residual, the gain -- requested
// before the first reduction: with one 1 KiB row per wave the D = 512 kernels keep too little in flight (4.0 TB/s at 263 k rows where
// the 8 KiB-row kernels reach 5.2) and every row pays the load -> reduce -> load gain / residual -> store chain in full.
template <typename T, int MAXC, int RPW>
__global__ __launch_bounds__(256) void ln_fwd_rows_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ g,
                                                          const T* __restrict__ res, T* __restrict__ y,
                                                          float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                          int rows, int D, float eps, long ldy, int y_grp) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = lane_id();
    const long row0 = ((long)blockIdx.x * 4 + wave_id()) * RPW;
    if (row0 >= rows) return;                      // whole wave leaves together; no barriers below
    const int nch = D / VEC;
    float v[RPW][MAXC][VEC], rv[RPW][MAXC][VEC], gv[MAXC][VEC];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const long row = row0 + r < rows ? row0 + r : rows - 1;                  // (a clamped duplicate of the last row: never stored)
        load_row<T, MAXC, false>(x + row * ldx, D, lane, v[r]);
        if (res != nullptr) load_row<T, MAXC, false>(res + row * (long)D, D, lane, rv[r]);
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
        if (lane + 64 * i < nch) load_vec<T>(g + (lane + 64 * i) * VEC, gv[i]);
    float mean[RPW], var[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) row_stats<T, MAXC>(v[r], D, lane, mean[r], var[r]);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const long row = row0 + r;
        if (row < rows) {
            const float rstd = fast_rsqrt(var[r] + eps);
            const long yrow = y_grp > 0 ? row + row / y_grp + 1 : row;
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = lane + 64 * i;
                if (c < nch) {
                    float o[VEC];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) o[j] = (v[r][i][j] - mean[r]) * rstd * gv[i][j];
                    if (res != nullptr) {
#pragma unroll
                        for (int j = 0; j < VEC; ++j) o[j] += rv[r][i][j];
                    }
                    store_vec<T>(y + yrow * ldy + c * VEC, o);
                }
            }
            if (lane == 0) {
                mean_out[row] = mean[r];
                rstd_out[row] = rstd;
            }
        }
    }
}

#endif

// ---- two chained LayerNorms in one pass over the rows ----------------------------------------------------------------------------
// The residual block boundary of the Transformer (x_clip.py:245,288-289 then :126): x1 = LN(p) g1 + res (the attention block's
// to_out LayerNorm + skip) is immediately followed by h2 = LN(x1) g2 (the feed-forward PreNorm).  Both are row-wise over the same
// rows, so one kernel reads p and res once and writes x1, h2 and both pairs of statistics: the second LayerNorm no longer
// re-reads x1.  The second one sees x1 exactly as stored (rounded to T), so results are those of the two separate calls.
template <typename T, int MAXC, bool NT = false>
__global__ __launch_bounds__(256) void ln_chain_fwd_kernel(const T* __restrict__ p, const T* __restrict__ g1, const T* __restrict__ res,
                                                           T* __restrict__ x1, float* __restrict__ mean1, float* __restrict__ rstd1,
                                                           const T* __restrict__ g2, T* __restrict__ h2, float* __restrict__ mean2,
                                                           float* __restrict__ rstd2, int rows, int D, float eps) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = lane_id();
    const long row = (long)blockIdx.x * 4 + wave_id();
    if (row >= rows) return;
    const int nch = D / VEC;
    float v[MAXC][VEC], rv[MAXC][VEC];
    load_row<T, MAXC, false, NT>(p + row * (long)D, D, lane, v);
    load_row<T, MAXC, false, NT>(res + row * (long)D, D, lane, rv);    // (both rows requested before the first reduction)
    float mean, var;
    row_stats<T, MAXC>(v, D, lane, mean, var);
    const float rstd = fast_rsqrt(var + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            float gv[VEC];
            load_vec<T>(g1 + c * VEC, gv);
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[i][j] = to_f32(from_f32<T>((v[i][j] - mean) * rstd * gv[j] + rv[i][j]));   // x1 as stored
            store_vec<T, NT>(x1 + row * (long)D + c * VEC, v[i]);
        }
    }
    float m2, var2;
    row_stats<T, MAXC>(v, D, lane, m2, var2);                  // (padding chunks still hold zeros)
    const float r2 = fast_rsqrt(var2 + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            float gv[VEC], o[VEC];
            load_vec<T>(g2 + c * VEC, gv);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = (v[i][j] - m2) * r2 * gv[j];
            store_vec<T, NT>(h2 + row * (long)D + c * VEC, o);
        }
    }
    if (lane == 0) {
        mean1[row] = mean; rstd1[row] = rstd;
        mean2[row] = m2; rstd2[row] = r2;
    }
}

// backward of the same pair: dx1 = LN2'(dh2) + dres (the gradient of x1: still written, the next residual junction adds it), then
// dp = LN1'(dx1) from the registers.  dg2 / dg1 partial rows per work-group as in ln_bwd_kernel: partial [gridDim.x, 2 D].
template <typename T, int MAXC, bool NT = false>
__global__ __launch_bounds__(256) void ln_chain_bwd_kernel(const T* __restrict__ dh2, const T* __restrict__ x1, const T* __restrict__ g2,
                                                           const float* __restrict__ mean2, const float* __restrict__ rstd2,
                                                           const T* __restrict__ dres, T* __restrict__ dx1, const T* __restrict__ p,
                                                           const T* __restrict__ g1, const float* __restrict__ mean1,
                                                           const float* __restrict__ rstd1, T* __restrict__ dp,
                                                           float* __restrict__ dg_partial, int rows, int D) {
    constexpr int VEC = Elem<T>::VEC;
    XC_LDS_DYNAMIC(lds);
    float* red = reinterpret_cast<float*>(lds);            // [3][2 D]
    const int lane = lane_id(), wave = wave_id();
    const int nch = D / VEC;
    float g2v[MAXC][VEC], g1v[MAXC][VEC], dg2[MAXC][VEC], dg1[MAXC][VEC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { g2v[i][j] = 0.f; g1v[i][j] = 0.f; dg2[i][j] = 0.f; dg1[i][j] = 0.f; }
        if (c < nch) { load_vec<T>(g2 + c * VEC, g2v[i]); load_vec<T>(g1 + c * VEC, g1v[i]); }
    }
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const float m2 = mean2[row], r2 = rstd2[row], m1 = mean1[row], r1 = rstd1[row];
        float xh[MAXC][VEC], dy[MAXC][VEC], rv[MAXC][VEC], pv[MAXC][VEC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {                           // all four rows requested before the first reduction
            const int c = lane + 64 * i;
            if (c < nch) {
                load_vec<T, NT>(dres + row * (long)D + c * VEC, rv[i]);
                load_vec<T, NT>(p + row * (long)D + c * VEC, pv[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
#pragma unroll
            for (int j = 0; j < VEC; ++j) { xh[i][j] = 0.f; dy[i][j] = 0.f; }
            if (c < nch) {
                load_vec<T, NT>(x1 + row * (long)D + c * VEC, xh[i]);
                load_vec<T, NT>(dh2 + row * (long)D + c * VEC, dy[i]);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    xh[i][j] = (xh[i][j] - m2) * r2;
                    dg2[i][j] += dy[i][j] * xh[i][j];
                    dy[i][j] *= g2v[i][j];
                    s1 += dy[i][j];
                    s2 += dy[i][j] * xh[i][j];
                }
            }
        }
        float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
        s1 = 0.f; s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float d1 = r2 * (dy[i][j] - c1 - xh[i][j] * c2) + rv[i][j];  // gradient of x1
                    dy[i][j] = d1;
                    const float ph = (pv[i][j] - m1) * r1;
                    xh[i][j] = ph;
                    dg1[i][j] += d1 * ph;
                    const float dyg = d1 * g1v[i][j];
                    s1 += dyg;
                    s2 += dyg * ph;
                }
                store_vec<T, NT>(dx1 + row * (long)D + c * VEC, dy[i]);
            }
        }
        c1 = wave_sum(s1) / (float)D;
        c2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                float o[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] = r1 * (dy[i][j] * g1v[i][j] - c1 - xh[i][j] * c2);
                store_vec<T, NT>(dp + row * (long)D + c * VEC, o);
            }
        }
    }
    // fold the 4 waves' partials: columns [0, D) = dg2, [D, 2 D) = dg1; row blockIdx.x of dg_partial [gridDim.x, 2 D]
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch)
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    red[(wave - 1) * 2 * D + c * VEC + j] = dg2[i][j];
                    red[(wave - 1) * 2 * D + D + c * VEC + j] = dg1[i][j];
                }
        }
    }
    sync();
    if (wave == 0) {
        float* out = dg_partial + (long)blockIdx.x * 2 * D;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch)
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const int col = c * VEC + j;
                    out[col] = dg2[i][j] + red[col] + red[2 * D + col] + red[4 * D + col];
                    out[D + col] = dg1[i][j] + red[D + col] + red[3 * D + col] + red[5 * D + col];
                }
        }
    }
}

// ---- LayerNorm backward --------------------------------------------------------------------------------
// xhat = (x - mean) rstd ; dyg = dy g ; dx = rstd (dyg - mean(dyg) - xhat mean(dyg xhat)) ; dg += dy xhat.
// With GEGLU the LayerNorm input was a = u gelu(t): du = da gelu(t), dt = da u gelu'(t), written to the
// [rows, 2D] gradient of the FF1 output.  Waves walk the rows grid-stride and keep their dg partials in
// registers; one LDS fold + one row of per-work-group partial sums at the end.  `dres` (optional, [rows, D]) is
// added to dx: the pre-norm residual blocks x + f(LN(x)) hand their skip-path gradient straight to this kernel.
// FFN (round 6; not with GEGLU): the dx this kernel writes is the gradient dOut of the residual block BELOW it, whose fused feed-forward
// backward (gemm9.h) starts from four per-row constants {rstd, -mean rstd, s1 / F rstd, s2 / F rstd} with s1 = dOut . (W2 gamma) and
// s2 = dOut . (x2 - x1) -- x2 being THIS kernel's own input row x (the block's output), x1 the block's input.  The row is in registers
// here: one more row read (x1) and two more reductions replace ffn_rowstats_kernel's pass over dOut, x2 and x1 (0.76 ms per step of
// configs[1]).  dOut enters the dot products as stored (rounded to T), chunk for chunk as that kernel summed it.
struct LnFfnStats {
    const void* x1;              // [rows, D] (ld1) the lower block's input
    long ld1;
    const float* wg;             // [D] W2 gamma of the lower block (ffn_wgamma_kernel)
    const float* mean4;          // [rows] statistics of the lower block's inner LayerNorm
    const float* rstd4;
    float inv_f;                 // 1 / F
    float* rowc;                 // [rows, 4] out
};
template <typename T, int MAXC, bool GEGLU, bool NT = false, bool FFN = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, long ldx,
                                                     const T* __restrict__ g, const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in, const T* __restrict__ dres,
                                                     T* __restrict__ dx, long lddx, float* __restrict__ dg_partial, int rows,
                                                     int D, LnFfnStats fs = LnFfnStats{}) {
    constexpr int VEC = Elem<T>::VEC;
    XC_LDS_DYNAMIC(lds);
    float* red = reinterpret_cast<float*>(lds);            // [3][D]
    const int lane = lane_id(), wave = wave_id();
    const int nch = D / VEC;
    float gv[MAXC][VEC], dgacc[MAXC][VEC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { gv[i][j] = 0.f; dgacc[i][j] = 0.f; }
        if (c < nch) load_vec<T>(g + c * VEC, gv[i]);
    }
    constexpr int FC = (FFN && !GEGLU) ? MAXC : 1;             // (register arrays of the statistics: one dummy chunk without them)
    float wgv[FC][VEC];
    if (FFN && !GEGLU) {
#pragma unroll
        for (int i = 0; i < FC; ++i) {
            const int c = lane + 64 * i;
#pragma unroll
            for (int j = 0; j < VEC; ++j) wgv[i][j] = 0.f;
            if (c < nch) {
#pragma unroll
                for (int j = 0; j < VEC; j += 4) {
                    const u32x4 w = ld16(fs.wg + c * VEC + j);
                    wgv[i][j] = u2f(w[0]); wgv[i][j + 1] = u2f(w[1]); wgv[i][j + 2] = u2f(w[2]); wgv[i][j + 3] = u2f(w[3]);
                }
            }
        }
    }
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const float mean = mean_in[row], rstd = rstd_in[row];
        float rs4 = 0.f, m4 = 0.f;                                 // (FFN) requested with the row's own statistics: one round trip
        if (FFN && !GEGLU) { rs4 = fs.rstd4[row]; m4 = fs.mean4[row]; }
        float xh[MAXC][VEC], dyv[MAXC][VEC];
        load_row<T, MAXC, GEGLU, NT>(x + row * ldx, D, lane, xh);
        float xr[FC][VEC], x1v[FC][VEC];                           // (FFN) the raw row = the lower block's output, and its input
        if (FFN && !GEGLU) {
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                const int c = lane + 64 * i;
#pragma unroll
                for (int j = 0; j < VEC; ++j) { xr[i][j] = xh[i][j]; x1v[i][j] = 0.f; }
                if (c < nch) load_vec<T, NT>(reinterpret_cast<const T*>(fs.x1) + row * fs.ld1 + c * VEC, x1v[i]);
            }
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                load_vec<T, NT>(dy + row * (long)D + c * VEC, dyv[i]);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    xh[i][j] = (xh[i][j] - mean) * rstd;
                    dgacc[i][j] += dyv[i][j] * xh[i][j];
                    dyv[i][j] *= gv[i][j];                         // dyg
                    s1 += dyv[i][j];
                    s2 += dyv[i][j] * xh[i][j];
                }
            }
        }
        const float c1 = wave_sum(s1) / (float)D;
        const float c2 = wave_sum(s2) / (float)D;
        float a1 = 0.f, a2 = 0.f;                                  // (FFN) dOut . wg and dOut . (x2 - x1) of this row
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                float da[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) da[j] = rstd * (dyv[i][j] - c1 - xh[i][j] * c2);
                if (GEGLU) {
                    float u[VEC], t[VEC], du[VEC], dt[VEC];
                    load_vec<T, NT>(x + row * ldx + c * VEC, u);
                    load_vec<T, NT>(x + row * ldx + D + c * VEC, t);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        float cdf, pdf;
                        gelu_parts(t[j], cdf, pdf);
                        du[j] = da[j] * (t[j] * cdf);                          // d/du [u gelu(t)]
                        dt[j] = da[j] * u[j] * (cdf + t[j] * pdf);             // d/dt [u gelu(t)]
                    }
                    store_vec<T, NT>(dx + row * lddx + c * VEC, du);
                    store_vec<T, NT>(dx + row * lddx + D + c * VEC, dt);
                } else {
                    if (dres != nullptr) {                         // gradient arriving over the residual branch
                        float rv[VEC];
                        load_vec<T, NT>(dres + row * (long)D + c * VEC, rv);
#pragma unroll
                        for (int j = 0; j < VEC; ++j) da[j] += rv[j];
                    }
                    store_vec<T, NT>(dx + row * lddx + c * VEC, da);
                    if (FFN) {
                        const int fi = i < FC ? i : 0;
#pragma unroll
                        for (int j = 0; j < VEC; ++j) {
                            const float d = to_f32(from_f32<T>(da[j]));            // dOut as stored
                            a1 += d * wgv[fi][j];
                            a2 += d * (xr[fi][j] - x1v[fi][j]);
                        }
                    }
                }
            }
        }
        if (FFN && !GEGLU) {
            a1 = wave_sum(a1);
            a2 = wave_sum(a2);
            if (lane == 0) {
                const u32x4 v = {f2u(rs4), f2u(-m4 * rs4), f2u(a1 * fs.inv_f * rs4), f2u(a2 * fs.inv_f * rs4)};
                st16(fs.rowc + row * 4, v);
            }
        }
    }
    // fold the 4 waves' dg partials; every work-group stores ITS column sums as row blockIdx.x of dg_partial
    // [gridDim.x, D] (no atomics: thousands of work-groups adding into the same D floats serialise in L2 and cost a
    // fixed ~0.4 ms per call); colsum_fold_kernel adds the rows into the caller's accumulator.
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch)
#pragma unroll
                for (int j = 0; j < VEC; ++j) red[(wave - 1) * D + c * VEC + j] = dgacc[i][j];
        }
    }
    sync();
    if (wave == 0) {
        float* out = dg_partial + (long)blockIdx.x * D;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch)
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const int col = c * VEC + j;
                    out[col] = dgacc[i][j] + red[col] + red[D + col] + red[2 * D + col];
                }
        }
    }
}

// GEGLU variant, a = u gelu(t) feeding the LayerNorm.  The generic kernel above evaluates the GELU twice per element (once
// for a, once for its derivative) and at 2048-wide rows that makes it VALU-bound (~75 ops per element against 20 bytes of
// traffic).  Here one pass produces gelu(t), u gelu'(t) and the normalised a and keeps them in registers (4 x MAXC x VEC
// floats per lane) until the row's two reductions are known; gamma sits in LDS.  One exp + one rcp per element.
// SPLIT waves share a row (each MAXC chunks per lane of its D / SPLIT columns) so that wide rows stay under 168 VGPRs
// (three waves per SIMD); their two partial sums meet in LDS, double-buffered so one barrier per row suffices.
template <typename T, int MAXC, int SPLIT, bool NT = false>
__global__ __launch_bounds__(256) void ln_geglu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, long ldx,
                                                           const T* __restrict__ g, const float* __restrict__ mean_in,
                                                           const float* __restrict__ rstd_in, T* __restrict__ dx, long lddx,
                                                           float* __restrict__ dg_partial, int rows, int D) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int RPB = 4 / SPLIT;                         // rows per work-group pass
    XC_LDS_DYNAMIC(lds);
    float* red = reinterpret_cast<float*>(lds);            // [4][D / SPLIT] dg partials of the four waves
    float* sred = red + 4 * (D / SPLIT);                   // [2][4][2] row sums (s1, s2) per wave, two buffers
    T* gs = reinterpret_cast<T*>(sred + 16);               // [D] gamma
    const int lane = lane_id(), wave = wave_id();
    const int part = wave % SPLIT, slot = wave / SPLIT;
    const int nch = D / VEC / SPLIT, c0 = part * nch;      // this wave's chunks: c0 + [0, nch)
    for (int c = threadIdx.x; c < D / VEC; c += blockDim.x) st16(gs + c * VEC, ld16(g + c * VEC));
    float dgacc[MAXC][VEC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
        for (int j = 0; j < VEC; ++j) dgacc[i][j] = 0.f;
    sync();
    int it = 0;
    for (long base = (long)blockIdx.x * RPB; base < rows; base += (long)gridDim.x * RPB, ++it) {
        const long row = base + slot;
        const bool live = row < rows;
        const long rr = live ? row : rows - 1;
        const float rstd = rstd_in[rr], shift = -mean_in[rr] * rstd;
        float ge[MAXC][VEC], udge[MAXC][VEC], ah[MAXC][VEC], dyg[MAXC][VEC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                float u[VEC], t[VEC], gv[VEC];
                load_vec<T, NT>(x + rr * ldx + (c0 + c) * VEC, u);
                load_vec<T, NT>(x + rr * ldx + D + (c0 + c) * VEC, t);
                load_vec<T, NT>(dy + rr * (long)D + (c0 + c) * VEC, dyg[i]);
                load_vec<T>(gs + (c0 + c) * VEC, gv);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    float cdf, pdf;
                    gelu_parts(t[j], cdf, pdf);
                    ge[i][j] = t[j] * cdf;                                     // gelu(t)
                    udge[i][j] = u[j] * (cdf + t[j] * pdf);                    // u gelu'(t)
                    ah[i][j] = u[j] * ge[i][j] * rstd + shift;                 // normalised a
                    if (live) dgacc[i][j] += dyg[i][j] * ah[i][j];
                    dyg[i][j] *= gv[j];
                    s1 += dyg[i][j];
                    s2 += dyg[i][j] * ah[i][j];
                }
            }
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (SPLIT > 1) {
            float* buf = sred + (it & 1) * 8;
            if (lane == 0) { buf[wave * 2] = s1; buf[wave * 2 + 1] = s2; }
            sync();
            s1 = 0.f; s2 = 0.f;
#pragma unroll
            for (int q = 0; q < SPLIT; ++q) { s1 += buf[(slot * SPLIT + q) * 2]; s2 += buf[(slot * SPLIT + q) * 2 + 1]; }
        }
        const float k1 = s1 / (float)D * rstd;
        const float k2 = s2 / (float)D * rstd;
        if (live) {
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = lane + 64 * i;
                if (c < nch) {
                    float du[VEC], dt[VEC];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const float da = dyg[i][j] * rstd - k1 - ah[i][j] * k2;
                        du[j] = da * ge[i][j];
                        dt[j] = da * udge[i][j];
                    }
                    store_vec<T, NT>(dx + row * lddx + (c0 + c) * VEC, du);
                    store_vec<T, NT>(dx + row * lddx + D + (c0 + c) * VEC, dt);
                }
            }
        }
    }
    // every wave parks its dg partials; the first SPLIT waves fold the RPB row slots of their column part
    const int Dw = D / SPLIT;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch)
#pragma unroll
            for (int j = 0; j < VEC; ++j) red[wave * Dw + c * VEC + j] = dgacc[i][j];
    }
    sync();
    if (wave < SPLIT) {
        float* out = dg_partial + (long)blockIdx.x * D + wave * Dw;
        for (int col = lane; col < Dw; col += 64) {
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < RPB; ++q) acc += red[(q * SPLIT + wave) * Dw + col];
            out[col] = acc;
        }
    }
}

// accum[c] += sum_r partial[r, c]  for partial [nrows, D] fp32 (row stride ldp).  grid = (ceil(D / 64), slices); each wave sums a strip
// of rows for 64 columns (coalesced 256-byte row segments), the 4 waves fold through LDS, one atomic per column per
// work-group (slices-way contention only).
__global__ __launch_bounds__(256) void colsum_fold_kernel(const float* __restrict__ partial, long ldp, float* __restrict__ accum,
                                                          int nrows, int D) {
    XC_LDS_DYNAMIC(lds);                                 // 4 x 64 floats
    float (*red)[64] = reinterpret_cast<float (*)[64]>(lds);
    const int lane = lane_id(), wave = wave_id();
    const int col = blockIdx.x * 64 + lane;
    const int strips = gridDim.y * 4;
    const int per = (nrows + strips - 1) / strips;
    const int r0 = (blockIdx.y * 4 + wave) * per;
    const int r1 = r0 + per < nrows ? r0 + per : nrows;
    float s = 0.f;
    if (col < D)
        for (int r = r0; r < r1; ++r) s += partial[(long)r * ldp + col];
    red[wave][lane] = s;
    sync();
    if (wave == 0 && col < D) atomic_add(accum + col, red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
}

// the same for TWO accumulators whose partial rows are interleaved ([nrows, 2 D]: blockIdx.z picks the half) -- the chained LayerNorm
// backward folds both gains' partials in one launch
__global__ __launch_bounds__(256) void colsum_fold2_kernel(const float* __restrict__ partial, float* __restrict__ accum_a,
                                                           float* __restrict__ accum_b, int nrows, int D) {
    XC_LDS_DYNAMIC(lds);                                 // 4 x 64 floats
    float (*red)[64] = reinterpret_cast<float (*)[64]>(lds);
    const int lane = lane_id(), wave = wave_id();
    const int col = blockIdx.x * 64 + lane;
    const int strips = gridDim.y * 4;
    const int per = (nrows + strips - 1) / strips;
    const int r0 = (blockIdx.y * 4 + wave) * per;
    const int r1 = r0 + per < nrows ? r0 + per : nrows;
    const float* src = partial + (long)blockIdx.z * D;
    float* accum = blockIdx.z ? accum_b : accum_a;
    float s = 0.f;
    if (col < D)
        for (int r = r0; r < r1; ++r) s += src[(long)r * (2 * D) + col];
    red[wave][lane] = s;
    sync();
    if (wave == 0 && col < D) atomic_add(accum + col, red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
}

// ---- l2 normalisation -------------------------------------------------------------------------------------
// y = x / max(||x||, 1e-12) ; rnorm saved for the backward: dx = (dy - y <y, dy>) rnorm.
template <typename T, int MAXC>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                         float* __restrict__ rnorm_out, int rows, int D) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = lane_id();
    const long row = (long)blockIdx.x * 4 + wave_id();
    if (row >= rows) return;
    const int nch = D / VEC;
    float v[MAXC][VEC];
    load_row<T, MAXC, false>(x + row * (long)D, D, lane, v);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
        for (int j = 0; j < VEC; ++j) q += v[i][j] * v[i][j];
    const float nrm = sqrtf(wave_sum(q));
    const float rn = 1.0f / fmaxf(nrm, 1e-12f);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            float o[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = v[i][j] * rn;
            store_vec<T>(y + row * (long)D + c * VEC, o);
        }
    }
    if (lane == 0) rnorm_out[row] = rn;
}

// out[r] = <a[r, :], b[r, :]>: the similarity of matched pairs, the reference's inference return einsum('b d, b d -> b') (x_clip.py:744-746)
template <typename T, int MAXC>
__global__ __launch_bounds__(256) void rowdot_kernel(const T* __restrict__ a, long lda, const T* __restrict__ b, long ldb, T* __restrict__ out,
                                                     int rows, int D) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = lane_id();
    const long row = (long)blockIdx.x * 4 + wave_id();
    if (row >= rows) return;
    float u[MAXC][VEC], w[MAXC][VEC];
    load_row<T, MAXC, false>(a + row * lda, D, lane, u);
    load_row<T, MAXC, false>(b + row * ldb, D, lane, w);
    const int nch = D / VEC;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
        if (lane + 64 * i < nch)
#pragma unroll
            for (int j = 0; j < VEC; ++j) q += u[i][j] * w[i][j];
    q = wave_sum(q);
    if (lane == 0) out[row] = from_f32<T>(q);
}

template <typename T, int MAXC>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                         const float* __restrict__ rnorm, T* __restrict__ dx,
                                                         int rows, int D) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = lane_id();
    const long row = (long)blockIdx.x * 4 + wave_id();
    if (row >= rows) return;
    const int nch = D / VEC;
    float yv[MAXC][VEC], dv[MAXC][VEC];
    load_row<T, MAXC, false>(y + row * (long)D, D, lane, yv);
    load_row<T, MAXC, false>(dy + row * (long)D, D, lane, dv);
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
        for (int j = 0; j < VEC; ++j) dot += yv[i][j] * dv[i][j];
    dot = wave_sum(dot);
    const float rn = rnorm[row];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            float o[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = (dv[i][j] - yv[i][j] * dot) * rn;
            store_vec<T>(dx + row * (long)D + c * VEC, o);
        }
    }
}

}  // namespace xc
