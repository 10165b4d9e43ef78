// colnorm.h -- the pieces of the visual self-supervision head that are not GEMMs (reference x_clip/visual_ssl.py):
// BatchNorm1d over the rows of a [R, C] activation (+ the ReLU that follows it in the projector / predictor MLPs,
// visual_ssl.py:112-136) and the negative-cosine loss of SimSiam (loss_fn, visual_ssl.py:104-107).  All HBM-bound.
//
// BatchNorm statistics run down the columns of a row-major tensor, so a lane OWNS one 16-byte chunk of columns and walks
// the rows: a wave reads cw consecutive chunks (cw = 64 when the tensor is at least 64 chunks wide, i.e. 1 KiB coalesced
// per row; narrower tensors put 64 / cw rows side by side in a wave).  grid = (column slabs, row slices); the slices'
// partial sums meet in a small [slices, 2, C] workspace and a one-thread-per-column finalize kernel turns them into the
// statistics -- no atomics, deterministic.  The variance is accumulated around a per-column shift (row 0 of the tensor),
// var = (S2 - S1^2 / R) / R over d = x - shift, which removes the cancellation of E[x^2] - E[x]^2 when |mean| >> std.
#pragma once
#include "common.h"

namespace xc {

struct ColNormGeom {
    int cw;                      // lanes (16-byte chunks) per row within a wave: power of two <= 64
    int slabs;                   // gridDim.x: ceil(chunks / cw)
    int slices;                  // gridDim.y
};

template <typename T>
XC_DEV void cn_coords(int C, int cw, int& chunk, int& sub, bool& live) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = lane_id();
    chunk = blockIdx.x * cw + (lane & (cw - 1));
    sub = lane / cw;
    live = chunk < C / VEC;
}

// fold the 256 threads' (a[VEC], b[VEC]) through LDS: thread (wave 0, sub 0) of every chunk gets the work-group total
template <int VEC>
XC_DEV bool cn_fold(unsigned char* lds, int cw, float (&a)[VEC], float (&b)[VEC]) {
    float* red = reinterpret_cast<float*>(lds);          // [256][2 VEC]
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[tid * 2 * VEC + j] = a[j]; red[tid * 2 * VEC + VEC + j] = b[j]; }
    sync();
    if (tid >= cw) return false;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { a[j] = 0.f; b[j] = 0.f; }
    for (int t = tid; t < 256; t += cw)
#pragma unroll
        for (int j = 0; j < VEC; ++j) { a[j] += red[t * 2 * VEC + j]; b[j] += red[t * 2 * VEC + VEC + j]; }
    return true;
}

// ---- forward statistics: partial[slice, 0, c] = sum_r (x - x[0, c]),  partial[slice, 1, c] = sum_r (x - x[0, c])^2 ----------
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, float* __restrict__ partial, int R, int C, int cw) {
    constexpr int VEC = Elem<T>::VEC;
    XC_LDS_DYNAMIC(lds);
    int chunk, sub;
    bool live;
    cn_coords<T>(C, cw, chunk, sub, live);
    const int rpw = 64 / cw;
    float a[VEC], b[VEC], sh[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { a[j] = 0.f; b[j] = 0.f; sh[j] = 0.f; }
    if (live) {
        load_vec<T>(x + chunk * VEC, sh);
        for (long r = ((long)blockIdx.y * 4 + wave_id()) * rpw + sub; r < R; r += (long)gridDim.y * 4 * rpw) {
            float v[VEC];
            load_vec<T>(x + r * C + chunk * VEC, v);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { const float d = v[j] - sh[j]; a[j] += d; b[j] += d * d; }
        }
    }
    if (cn_fold<VEC>(lds, cw, a, b) && live) {
        float* out = partial + (long)blockIdx.y * 2 * C + chunk * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { out[j] = a[j]; out[C + j] = b[j]; }
    }
}

// work-group = 64 columns x 4 waves, wave w sums slices w, w + 4, ... (a single thread walking 256 slices measured 67 us per call);
// the partial pairs meet in LDS and wave 0 finishes the column
XC_DEV bool cn_slice_sums(unsigned char* lds, const float* __restrict__ partial, int slices, int C, float& s1, float& s2, int& c) {
    float (*red)[4][64] = reinterpret_cast<float (*)[4][64]>(lds);      // [2][4][64]
    const int lane = lane_id(), wave = wave_id();
    c = blockIdx.x * 64 + lane;
    s1 = 0.f;
    s2 = 0.f;
    if (c < C)
        for (int s = wave; s < slices; s += 4) { s1 += partial[(long)s * 2 * C + c]; s2 += partial[(long)s * 2 * C + C + c]; }
    red[0][wave][lane] = s1;
    red[1][wave][lane] = s2;
    sync();
    if (wave != 0 || c >= C) return false;
    s1 = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
    s2 = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
    return true;
}

// batch mean / rstd per column (saved for the backward) and the running-statistics update of nn.BatchNorm1d
// (momentum m: running = (1 - m) running + m stat, with the UNBIASED variance, as torch does)
template <typename T>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const T* __restrict__ x, const float* __restrict__ partial, int slices, int R, int C,
                                                          float eps, float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var) {
    XC_LDS_DYNAMIC(lds);
    float s1, s2;
    int c;
    if (!cn_slice_sums(lds, partial, slices, C, s1, s2, c)) return;
    const float inv = 1.0f / (float)R;
    const float m1 = s1 * inv;
    const float var = fmaxf(s2 * inv - m1 * m1, 0.f);
    const float mu = to_f32(x[c]) + m1;
    mean[c] = mu;
    rstd[c] = 1.0f / sqrtf(var + eps);
    if (running_mean != nullptr) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * ((float)R / (float)(R > 1 ? R - 1 : 1));
    }
}

// evaluation mode: the normalisation constants are the running statistics
__global__ __launch_bounds__(256) void bn_eval_stats_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var, float eps,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    mean[c] = running_mean[c];
    rstd[c] = 1.0f / sqrtf(running_var[c] + eps);
}

// ---- forward apply: y = relu?((x - mean) rstd gamma + beta) -------------------------------------------------------------------
template <typename T, bool RELU>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y,
                                                       int R, int C, int cw) {
    constexpr int VEC = Elem<T>::VEC;
    int chunk, sub;
    bool live;
    cn_coords<T>(C, cw, chunk, sub, live);
    if (!live) return;
    const int rpw = 64 / cw;
    float mu[VEC], rs[VEC], g[VEC], be[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int c = chunk * VEC + j;
        mu[j] = mean[c];
        rs[j] = rstd[c];
        g[j] = gamma != nullptr ? gamma[c] : 1.f;
        be[j] = beta != nullptr ? beta[c] : 0.f;
    }
    for (long r = ((long)blockIdx.y * 4 + wave_id()) * rpw + sub; r < R; r += (long)gridDim.y * 4 * rpw) {
        float v[VEC];
        load_vec<T>(x + r * C + chunk * VEC, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            v[j] = (v[j] - mu[j]) * rs[j] * g[j] + be[j];          // the expression the backward re-evaluates for the ReLU mask
            if (RELU) v[j] = fmaxf(v[j], 0.f);
        }
        store_vec<T>(y + r * C + chunk * VEC, v);
    }
}

// ---- backward --------------------------------------------------------------------------------------------------------------
// dz = dy (z > 0 with the ReLU, z = xhat gamma + beta recomputed from x);  partial[slice, 0, c] = sum dz,  [slice, 1, c] = sum dz xhat
struct BnCols {
    const float *mean, *rstd, *gamma, *beta;
};
template <typename T, bool RELU>
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const T* __restrict__ x, const T* __restrict__ dy, BnCols p,
                                                          float* __restrict__ partial, int R, int C, int cw) {
    constexpr int VEC = Elem<T>::VEC;
    XC_LDS_DYNAMIC(lds);
    int chunk, sub;
    bool live;
    cn_coords<T>(C, cw, chunk, sub, live);
    const int rpw = 64 / cw;
    float a[VEC], b[VEC], mu[VEC], rs[VEC], g[VEC], be[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { a[j] = 0.f; b[j] = 0.f; }
    if (live) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = chunk * VEC + j;
            mu[j] = p.mean[c];
            rs[j] = p.rstd[c];
            g[j] = p.gamma != nullptr ? p.gamma[c] : 1.f;
            be[j] = p.beta != nullptr ? p.beta[c] : 0.f;
        }
        for (long r = ((long)blockIdx.y * 4 + wave_id()) * rpw + sub; r < R; r += (long)gridDim.y * 4 * rpw) {
            float v[VEC], d[VEC];
            load_vec<T>(x + r * C + chunk * VEC, v);
            load_vec<T>(dy + r * C + chunk * VEC, d);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float xh = (v[j] - mu[j]) * rs[j];
                const float dz = (RELU && xh * g[j] + be[j] <= 0.f) ? 0.f : d[j];
                a[j] += dz;
                b[j] += dz * xh;
            }
        }
    }
    if (cn_fold<VEC>(lds, cw, a, b) && live) {
        float* out = partial + (long)blockIdx.y * 2 * C + chunk * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { out[j] = a[j]; out[C + j] = b[j]; }
    }
}

// per column: dbeta = sum dz, dgamma = sum dz xhat (written, not accumulated) and the two means the dx pass needs;
// with the running statistics (training = 0) the normalisation constants do not depend on x and the means drop out
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int slices, int R, int C, int training,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ coef) {
    XC_LDS_DYNAMIC(lds);
    float s1, s2;
    int c;
    if (!cn_slice_sums(lds, partial, slices, C, s1, s2, c)) return;
    if (dbeta != nullptr) dbeta[c] = s1;
    if (dgamma != nullptr) dgamma[c] = s2;
    coef[c] = training ? s1 / (float)R : 0.f;
    coef[C + c] = training ? s2 / (float)R : 0.f;
}

// dx = gamma rstd (dz - mean(dz) - xhat mean(dz xhat))
template <typename T, bool RELU>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, BnCols p,
                                                           const float* __restrict__ coef, T* __restrict__ dx, int R, int C, int cw) {
    constexpr int VEC = Elem<T>::VEC;
    int chunk, sub;
    bool live;
    cn_coords<T>(C, cw, chunk, sub, live);
    if (!live) return;
    const int rpw = 64 / cw;
    float mu[VEC], rs[VEC], g[VEC], be[VEC], c1[VEC], c2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int c = chunk * VEC + j;
        mu[j] = p.mean[c];
        rs[j] = p.rstd[c];
        g[j] = p.gamma != nullptr ? p.gamma[c] : 1.f;
        be[j] = p.beta != nullptr ? p.beta[c] : 0.f;
        c1[j] = coef[c];
        c2[j] = coef[C + c];
    }
    for (long r = ((long)blockIdx.y * 4 + wave_id()) * rpw + sub; r < R; r += (long)gridDim.y * 4 * rpw) {
        float v[VEC], d[VEC];
        load_vec<T>(x + r * C + chunk * VEC, v);
        load_vec<T>(dy + r * C + chunk * VEC, d);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float xh = (v[j] - mu[j]) * rs[j];
            const float dz = (RELU && xh * g[j] + be[j] <= 0.f) ? 0.f : d[j];
            v[j] = g[j] * rs[j] * (dz - c1[j] - xh * c2[j]);
        }
        store_vec<T>(dx + r * C + chunk * VEC, v);
    }
}

// ---- SimSiam loss (visual_ssl.py:104-107,249-259): sum_r coef (2 - 2 <p_r, z_r> / (max(|p_r|, eps) max(|z_r|, eps))) ---------------
// one wave per row; cosv / rp / rz [rows] are kept for the backward (the target z is a constant: visual_ssl.py:243-249)
template <typename T>
__global__ __launch_bounds__(256) void neg_cosine_fwd_kernel(const T* __restrict__ p, const T* __restrict__ z, int rows, int D, float coef,
                                                             float* __restrict__ cosv, float* __restrict__ rp, float* __restrict__ rz,
                                                             float* __restrict__ loss_accum) {
    constexpr int VEC = Elem<T>::VEC;
    XC_LDS_DYNAMIC(lds);
    float* red = reinterpret_cast<float*>(lds);           // [4]
    const int lane = lane_id(), wave = wave_id();
    float term = 0.f;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {     // capped grid: one atomic per work-group
        float pp = 0.f, zz = 0.f, pz = 0.f;
        for (int c = lane; c < D / VEC; c += 64) {
            float a[VEC], b[VEC];
            load_vec<T>(p + row * D + c * VEC, a);
            load_vec<T>(z + row * D + c * VEC, b);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { pp += a[j] * a[j]; zz += b[j] * b[j]; pz += a[j] * b[j]; }
        }
        pp = wave_sum(pp);
        zz = wave_sum(zz);
        pz = wave_sum(pz);
        const float ip = 1.0f / fmaxf(sqrtf(pp), 1e-12f), iz = 1.0f / fmaxf(sqrtf(zz), 1e-12f);
        const float cs = pz * ip * iz;
        if (lane == 0) { cosv[row] = cs; rp[row] = ip; rz[row] = iz; }
        term += 2.f - 2.f * cs;
    }
    if (lane == 0) red[wave] = term;
    sync();
    if (threadIdx.x == 0) atomic_add(loss_accum, coef * (red[0] + red[1] + red[2] + red[3]));
}

// dp_r = gmul coef (-2) (zhat - cos phat) / |p|      (a row whose norm was clamped has zero gradient through the norm; the clamp
// only matters for an all-zero row, whose gradient is then -2 zhat / eps -- as F.normalize's)
template <typename T>
__global__ __launch_bounds__(256) void neg_cosine_bwd_kernel(const T* __restrict__ p, const T* __restrict__ z, const float* __restrict__ cosv,
                                                             const float* __restrict__ rp, const float* __restrict__ rz,
                                                             const float* __restrict__ gmul, float coef, T* __restrict__ dp, int rows, int D) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = lane_id();
    const long row = (long)blockIdx.x * 4 + wave_id();
    if (row >= rows) return;
    const float g = -2.f * coef * gmul[0];
    const float ip = rp[row], iz = rz[row], cs = cosv[row];
    const bool clamped = ip >= 1e12f;
    for (int c = lane; c < D / VEC; c += 64) {
        float a[VEC], b[VEC];
        load_vec<T>(p + row * D + c * VEC, a);
        load_vec<T>(z + row * D + c * VEC, b);
#pragma unroll
        for (int j = 0; j < VEC; ++j) a[j] = g * ip * (b[j] * iz - (clamped ? 0.f : cs * a[j] * ip));
        store_vec<T>(dp + row * D + c * VEC, a);
    }
}

}  // namespace xc
