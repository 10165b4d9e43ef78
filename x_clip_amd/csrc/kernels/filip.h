// filip.h -- reductions of the fine-grained (FILIP) contrastive head, reference x_clip.py:797-811,821-847 with
// use_all_token_embeds: per (text x, image y) pair the token similarity block s[t, k] = temp <T[x,t], I[y,k]> is reduced to
//     t2i[x, y] = sum_t w[x,t] max_k s[t,k] / max(sum_t w[x,t], 1e-6)          (masked_mean over text tokens, x_clip.py:40-44,805-807)
//     i2t[x, y] = mean_k max_{t : w[x,t]} s[t,k]                               (x_clip.py:809-811)
// and both [b, B] matrices enter the InfoNCE / DCL tail with rows = texts (the reference does NOT transpose i2t in this mode).
// The token blocks themselves come from the MFMA GEMM (gemm*.h) in chunks of images written to a bounded workspace
// S[(x, t), (y, k)]; these kernels are the HBM-bound passes over such a chunk:
//   filip_reduce_kernel : one wave per (x, y): both reductions + the arg-max positions the backward routes through
//   filip_route_kernel  : builds the chunk of the routing matrix P = d loss / d s (two non-zeros families per pair), which the
//                         backward multiplies with the token matrices by two ordinary GEMMs (dT += P I, dI = P^T T)
//   rowlse_kernel / rowgrad_kernel : log-sum-exp over the rows of a materialised [rows, cols] fp32 logit matrix and its gradient
// `temp` arrives as a device scalar (log-temperature), like in simloss.h.
#pragma once
#include "common.h"

namespace xc {

constexpr float FILIP_NEG = -3.0e38f;

// S: [bx * nt rows, lds] (storage T), chunk columns (y, k) with y in [0, yc), k in [0, ni).  mask: [bx, nt] bytes (1 = real token).
// t2i / i2t: [bx, ldo] fp32 at column y0 + y.  kmax: [bx, nt, yc_total] int16 at (.., y0 + y); tmax: [bx, yc_total, ni] int16.
template <typename T>
__global__ __launch_bounds__(256) void filip_reduce_kernel(const T* __restrict__ S, long lds_, const unsigned char* __restrict__ mask,
                                                           const float* __restrict__ log_temp, float* __restrict__ t2i,
                                                           float* __restrict__ i2t, long ldo, short* __restrict__ kmax,
                                                           short* __restrict__ tmax, float* __restrict__ cnt, int bx, int nt, int yc,
                                                           int ni, int y0, int ytotal) {
    const int lane = lane_id();
    const long pair = (long)blockIdx.x * 4 + wave_id();
    if (pair >= (long)bx * yc) return;
    const int x = (int)(pair / yc), y = (int)(pair % yc);
    const float temp = expf(*log_temp);
    constexpr int MAXK = 4;                                   // image tokens per lane: ni <= 256
    float colmax[MAXK];
    int colarg[MAXK];
#pragma unroll
    for (int q = 0; q < MAXK; ++q) { colmax[q] = FILIP_NEG; colarg[q] = 0; }
    float wsum = 0.f, acc = 0.f;
    for (int t = 0; t < nt; ++t) {
        const bool w = mask[(long)x * nt + t] != 0;
        const T* row = S + ((long)x * nt + t) * lds_ + (long)y * ni;
        float best = FILIP_NEG;
        int bestk = 0;
#pragma unroll
        for (int q = 0; q < MAXK; ++q) {
            const int k = lane + 64 * q;
            if (k < ni) {
                const float v = to_f32(row[k]) * temp;
                if (v > best) { best = v; bestk = k; }
                if (w && v > colmax[q]) { colmax[q] = v; colarg[q] = t; }
            }
        }
        // wave arg-max over k (first index wins ties, like torch.max)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float ob = shfl_xor(best, m);
            const int ok = shfl_xor(bestk, m);
            if (ob > best || (ob == best && ok < bestk)) { best = ob; bestk = ok; }
        }
        if (lane == 0) kmax[((long)x * nt + t) * ytotal + y0 + y] = (short)bestk;
        if (w) { acc += best; wsum += 1.f; }
    }
    float csum = 0.f;
#pragma unroll
    for (int q = 0; q < MAXK; ++q) {
        const int k = lane + 64 * q;
        if (k < ni) {
            csum += colmax[q];
            tmax[((long)x * ytotal + y0 + y) * ni + k] = (short)colarg[q];
        }
    }
    csum = wave_sum(csum);
    if (lane == 0) {
        t2i[(long)x * ldo + y0 + y] = acc / fmaxf(wsum, 1e-6f);
        if (y0 + y == 0) cnt[x] = wsum;                       // number of real text tokens of sample x
        i2t[(long)x * ldo + y0 + y] = csum / (float)ni;
    }
}

// P[(x,t),(y,k)] = temp * ( g1[x,y0+y] * w[x,t] / cnt[x] * [k == kmax] + g2[x,y0+y] / ni * [t == tmax[.., k]] )
// one thread per 16-byte output chunk of a row; rows are padded with zeros up to ldp
template <typename T>
__global__ __launch_bounds__(256) void filip_route_kernel(T* __restrict__ P, long ldp, const unsigned char* __restrict__ mask,
                                                          const float* __restrict__ log_temp, const float* __restrict__ g1,
                                                          const float* __restrict__ g2, long ldg, const short* __restrict__ kmax,
                                                          const short* __restrict__ tmax, const float* __restrict__ cnt, int bx, int nt,
                                                          int yc, int ni, int y0, int ytotal) {
    constexpr int VEC = Elem<T>::VEC;
    const float temp = expf(*log_temp);
    const int nch = (int)(ldp / VEC);
    const long total = (long)bx * nt * nch;
    for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(id % nch);
        const long row = id / nch;
        const int x = (int)(row / nt), t = (int)(row % nt);
        const bool w = mask[row] != 0;
        const float invc = 1.0f / fmaxf(cnt[x], 1e-6f);
        float v[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int col = ch * VEC + e;
            float val = 0.f;
            if (col < yc * ni) {
                const int y = col / ni, k = col % ni;
                const long gy = (long)x * ldg + y0 + y;
                if (w && kmax[((long)x * nt + t) * ytotal + y0 + y] == k) val += g1[gy] * invc;
                if (tmax[((long)x * ytotal + y0 + y) * ni + k] == t && w) val += g2[gy] / (float)ni;
                val *= temp;
            }
            v[e] = val;
        }
        store_vec<T>(P + row * ldp + ch * VEC, v);
    }
}

// lse[r] = log sum_c exp(S[r, c]) (column r + diag_off left out when dcl); pos[r] = S[r, r + diag_off];
// loss += coef * sum_r (lse[r] - pos[r]).  One wave per row, fp32 logits.
__global__ __launch_bounds__(256) void rowlse_kernel(const float* __restrict__ S, long lds_, int rows, int cols, int diag_off, int dcl,
                                                     float coef, float* __restrict__ lse, float* __restrict__ loss) {
    const int lane = lane_id();
    const long r = (long)blockIdx.x * 4 + wave_id();
    float contrib = 0.f;
    if (r < rows) {
        const float* row = S + r * lds_;
        const int dc = (int)r + diag_off;
        float m = FILIP_NEG;
        for (int c = lane; c < cols; c += 64)
            if (!(dcl && c == dc)) m = fmaxf(m, row[c]);
        m = wave_max(m);
        float l = 0.f;
        for (int c = lane; c < cols; c += 64)
            if (!(dcl && c == dc)) l += fast_exp(row[c] - m);
        l = wave_sum(l);
        const float v = l > 0.f ? m + logf(l) : logf(1e-20f);
        if (lane == 0) {
            lse[r] = v;
            contrib = coef * (v - ((dc >= 0 && dc < cols) ? row[dc] : 0.f));
        }
    }
    if (lane == 0 && loss != nullptr && r < rows) atomic_add(loss, contrib);
}
// G[r, c] = gmul * coef * ( exp(S[r,c] - lse[r]) (1 - dcl [c == r + diag_off]) - [c == r + diag_off] )
__global__ __launch_bounds__(256) void rowgrad_kernel(const float* __restrict__ S, long lds_, const float* __restrict__ lse, int rows,
                                                      int cols, int diag_off, int dcl, float coef, const float* __restrict__ gmul,
                                                      float* __restrict__ G, long ldg, float* __restrict__ dtau) {
    const float gm = (gmul != nullptr ? *gmul : 1.0f) * coef;
    float dt = 0.f;
    const long total = (long)rows * cols;
    for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
        const long r = id / cols;
        const int c = (int)(id % cols);
        const bool diag = c == (int)r + diag_off;
        float v = (dcl && diag) ? 0.f : fast_exp(S[r * lds_ + c] - lse[r]);
        if (diag) v -= 1.0f;
        G[r * ldg + c] = gm * v;
        dt += gm * v * S[r * lds_ + c];
    }
    dt = wave_sum(dt);                                         // d loss / d tau = sum G o S (every logit is linear in temp)
    if (lane_id() == 0 && dtau != nullptr) atomic_add(dtau, dt);
}

// ---- similarity regularisation (x_clip.py:773-784) -----------------------------------------------------------------------------
// D = A - C off the global diagonal, sum of squares into one fp32 accumulator.  One wave per row, 16-byte chunks, one atomic per
// work-group.
template <typename T>
__global__ __launch_bounds__(256) void simreg_diff_kernel(const T* __restrict__ A, long lda, const T* __restrict__ C, long ldc,
                                                          T* __restrict__ D, long ldd, int rows, int cols, int diag_off,
                                                          float* __restrict__ sumsq) {
    constexpr int VEC = Elem<T>::VEC;
    XC_LDS_DYNAMIC(lds);
    float* red = reinterpret_cast<float*>(lds);            // [4]
    const int lane = lane_id(), wave = wave_id();
    float acc = 0.f;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const int dcol = (int)row + diag_off;
        for (int c = lane; c < cols / VEC; c += 64) {
            float a[VEC], b[VEC];
            load_vec<T>(A + row * lda + c * VEC, a);
            load_vec<T>(C + row * ldc + c * VEC, b);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float d = (c * VEC + j == dcol) ? 0.f : a[j] - b[j];
                a[j] = d;
                acc += d * d;
            }
            store_vec<T>(D + row * ldd + c * VEC, a);
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wave] = acc;
    sync();
    if (threadIdx.x == 0 && sumsq != nullptr) atomic_add(sumsq, red[0] + red[1] + red[2] + red[3]);
}

}  // namespace xc
