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

// Row-coalesced form of the same reductions for chunk rows of at most 256 * VEC * FILIP_MAXCH columns and ni >= VEC (the host sizes
// its chunks for it): ONE work-group per text sample walks that sample's real rows, always one row of raw chunks ahead of the
// arithmetic; every row is read once as whole 16-byte chunks by all 256 threads (the per-pair kernel above reads 2 ni-byte pieces
// at a row stride of the whole chunk: 0.8 TB/s).  A thread keeps the running column maxima (max over t, for i2t) of its own
// columns in registers.  The per-row segment maxima (max over k, for t2i): a chunk overlaps at most two image segments, so every
// thread leaves a (max, arg) pair per portion in LDS (plain stores, double-buffered by row parity: one barrier per row) and the
// thread that owns segment y scans the ~ni / VEC + 1 chunks overlapping it in column order (first index wins, like torch.max).
constexpr int FILIP_MAXCH = 8;

template <typename T>
__global__ __launch_bounds__(256) void filip_reduce_rows_kernel(const T* __restrict__ S, long lds_, const unsigned char* __restrict__ mask,
                                                                const float* __restrict__ log_temp, float* __restrict__ t2i,
                                                                float* __restrict__ i2t, long ldo, short* __restrict__ kmax,
                                                                short* __restrict__ tmax, float* __restrict__ cnt, int bx, int nt,
                                                                int yc_all, int ni, int y0_all, int ytotal, int ysplit) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int SEGS = 8;                                        // segments per thread: yc <= 2048
    XC_LDS_DYNAMIC(lds);
    const int x = blockIdx.x, tid = threadIdx.x;
    // blockIdx.y selects a sub-range of `ysplit` images of the chunk (ysplit * ni is a whole number of 16-byte chunks): more
    // work-groups in flight than text samples, shorter rows per work-group
    const int ysub = blockIdx.y * ysplit;
    const int yc = (yc_all - ysub) < ysplit ? (yc_all - ysub) : ysplit;
    const int y0 = y0_all + ysub;
    S += (long)ysub * ni;
    const int ncols = yc * ni, nch = (ncols + VEC - 1) / VEC;
    float* pval = reinterpret_cast<float*>(lds);                   // [2 buffers][nch][2 portions] portion maxima of a row
    int* parg = reinterpret_cast<int*>(pval + 4 * nch);            // same shape: k of the portion maximum
    float* seg_sum = reinterpret_cast<float*>(parg + 4 * nch);     // [yc] sum over k of the column maxima (epilogue)
    const float temp = expf(*log_temp);
    float cmax[FILIP_MAXCH][VEC];
    short carg[FILIP_MAXCH][VEC];
    int ys[FILIP_MAXCH], ks[FILIP_MAXCH];                          // (image, token) of the first column of each owned chunk
#pragma unroll
    for (int i = 0; i < FILIP_MAXCH; ++i) {
        const int col0 = (tid + 256 * i) * VEC;
        ys[i] = col0 / ni;
        ks[i] = col0 - ys[i] * ni;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { cmax[i][e] = FILIP_NEG; carg[i][e] = 0; }
    }
    float t2i_acc[SEGS];
#pragma unroll
    for (int j = 0; j < SEGS; ++j) t2i_acc[j] = 0.f;
    float wsum = 0.f;
    auto next_live = [&](int t) {                                  // rows of padding tokens take no part in either direction
        while (t < nt && mask[(long)x * nt + t] == 0) ++t;
        return t;
    };
    u32x4 cur[FILIP_MAXCH], nxt[FILIP_MAXCH];
    auto fetch = [&](int t) {
        const T* row = S + ((long)x * nt + t) * lds_;
#pragma unroll
        for (int i = 0; i < FILIP_MAXCH; ++i) {
            const int c = tid + 256 * i;
            if (c < nch) nxt[i] = ld16(row + c * VEC);
        }
    };
    int t = next_live(0);
    if (t < nt) fetch(t);
    int par = 0;
    while (t < nt) {
        wsum += 1.f;
#pragma unroll
        for (int i = 0; i < FILIP_MAXCH; ++i) cur[i] = nxt[i];
        const int tn = next_live(t + 1);
        if (tn < nt) fetch(tn);                                    // a whole row of arithmetic covers the next row's latency
        float* pv = pval + par * 2 * nch;
        int* pa = parg + par * 2 * nch;
#pragma unroll
        for (int i = 0; i < FILIP_MAXCH; ++i) {
            const int c = tid + 256 * i;
            if (c < nch) {
                float v[VEC];
                unpack(cur[i], v, (T*)nullptr);
                int k = ks[i], portion = 0;
                float smax = FILIP_NEG;
                int sarg = 0;
                pv[2 * c + 1] = FILIP_NEG;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    if (c * VEC + e < ncols) {
                        const float val = v[e] * temp;
                        if (val > cmax[i][e]) { cmax[i][e] = val; carg[i][e] = (short)t; }
                        if (val > smax) { smax = val; sarg = k; }
                        if (++k == ni) {                           // the first segment of the chunk ends here
                            pv[2 * c + portion] = smax; pa[2 * c + portion] = sarg;
                            smax = FILIP_NEG; sarg = 0; k = 0; portion = 1;
                        }
                    }
                }
                if (portion < 2 && (k != 0 || portion == 0)) { pv[2 * c + portion] = smax; pa[2 * c + portion] = sarg; }
            }
        }
        sync();
#pragma unroll
        for (int j = 0; j < SEGS; ++j) {
            const int y = tid + 256 * j;
            if (y < yc) {
                const int c_lo = (y * ni) / VEC, c_hi = ((y + 1) * ni - 1) / VEC;
                int c = c_lo;
                int portion = ((y * ni) % VEC == 0) ? 0 : 1;
                float best = FILIP_NEG;
                int bk = 0;
                for (; c <= c_hi; ++c, portion = 0) {
                    const float pvv = pv[2 * c + portion];
                    if (pvv > best) { best = pvv; bk = pa[2 * c + portion]; }
                }
                kmax[((long)x * nt + t) * ytotal + y0 + y] = (short)bk;
                t2i_acc[j] += best;
            }
        }
        par ^= 1;                                                  // the next row writes the other buffer: no second barrier
        t = tn;
    }
    for (int y = tid; y < yc; y += 256) seg_sum[y] = 0.f;
    sync();
#pragma unroll
    for (int i = 0; i < FILIP_MAXCH; ++i) {
        const int c = tid + 256 * i;
        if (c < nch) {
            int y = ys[i], k = ks[i];
            float part = 0.f;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int col = c * VEC + e;
                if (col < ncols) {
                    tmax[((long)x * ytotal + y0) * ni + col] = carg[i][e];
                    part += cmax[i][e];
                    if (++k == ni) { atomic_add(seg_sum + y, part); part = 0.f; k = 0; ++y; }
                }
            }
            if (k != 0 && y < yc) atomic_add(seg_sum + y, part);
        }
    }
    sync();
#pragma unroll
    for (int j = 0; j < SEGS; ++j) {
        const int y = tid + 256 * j;
        if (y < yc) {
            t2i[(long)x * ldo + y0 + y] = t2i_acc[j] / fmaxf(wsum, 1e-6f);
            i2t[(long)x * ldo + y0 + y] = seg_sum[y] / (float)ni;
        }
    }
    if (tid == 0 && y0 == 0) cnt[x] = wsum;
}

// P[(x,t),(y,k)] = temp * ( g1[x,y0+y] * w[x,t] / cnt[x] * [k == kmax] + g2[x,y0+y] / ni * [t == tmax[.., k]] )
// One work-group per (TEXT x, slice of 2048 columns); a thread owns one 16-byte chunk COLUMN and walks the text's nt rows.  What a
// chunk needs besides the row's kmax entries does not depend on t: its eight tmax entries (one dword-aligned 16-byte load, kept in
// registers), the images it spans and their factors g1 / cnt, g2 / ni (staged in LDS once per work-group); the kmax entries of a
// batch of rows x the slice's images are staged in LDS per batch.  So the loop over the rows is loads-free: compare, select, one
// 16-byte store per row.
// History (configs[3], 1 GB chunk; profiles/r03_d / r03_h / r03_i_kernel_stats_filip.txt): a flat (row, chunk) index with two 64-bit
// divisions and ~14 narrow global loads per 16 bytes written: 0.64 ms = 1.5 TB/s; divisions gone: 0.67 ms (not the arithmetic);
// one work-group per (row, slice) with the per-image factors in LDS and one wide tmax load: 0.52 ms (276 k work-groups living ~4 us
// each: launch- and latency-bound); this form: see DESIGN_APPENDIX.md section 3.  Rows of padding tokens and the padding columns are written as zeros.
constexpr int ROUTE_MAX_IMG = 2050;                             // images a 2048-column slice can touch (ni >= 1)
constexpr int ROUTE_KM_ENTRIES = 16384;                         // staged kmax entries per batch of rows (32 KiB)
constexpr int ROUTE_LDS_BYTES = ROUTE_MAX_IMG * 8 + ROUTE_KM_ENTRIES * 2 + 16;

template <typename T>
__global__ __launch_bounds__(256) void filip_route_kernel(T* __restrict__ P, long ldp, const unsigned char* __restrict__ mask,
                                                          const float* __restrict__ log_temp, const float* __restrict__ g1,
                                                          const float* __restrict__ g2, long ldg, const short* __restrict__ kmax,
                                                          const short* __restrict__ tmax, const float* __restrict__ cnt, int bx, int nt,
                                                          int yc, int ni, int y0, int ytotal) {
    constexpr int VEC = Elem<T>::VEC;
    XC_LDS_DYNAMIC(lds);                                         // ROUTE_LDS_BYTES
    float* const s_a1 = reinterpret_cast<float*>(lds);           // [count] temp g1[x, y] / cnt[x]
    float* const s_a2 = s_a1 + ROUTE_MAX_IMG;                    // [count] temp g2[x, y] / ni
    short* const s_km = reinterpret_cast<short*>(s_a2 + ROUTE_MAX_IMG);   // [rows of the batch][count]
    const int nch = (int)(ldp / VEC);
    const int ch = blockIdx.x * 256 + threadIdx.x;
    const int x = blockIdx.y;
    const int ncols = yc * ni;
    const int slice0 = blockIdx.x * 256 * VEC;                   // first column of the work-group's slice
    int last = slice0 + 256 * VEC - 1;
    last = last < ncols - 1 ? last : ncols - 1;
    const bool live_slice = slice0 < ncols;                      // (uniform) else: padding columns only
    const int y_first = live_slice ? slice0 / ni : 0;
    const int count = live_slice ? last / ni - y_first + 1 : 0;  // images the slice spans
    if (live_slice) {
        const float temp = expf(*log_temp);
        const float invc = temp / fmaxf(cnt[x], 1e-6f), a2f = temp / (float)ni;
        const float* g1r = g1 + (long)x * ldg + y0 + y_first;
        const float* g2r = g2 + (long)x * ldg + y0 + y_first;
        for (int i = threadIdx.x; i < count; i += 256) {
            s_a1[i] = g1r[i] * invc;
            s_a2[i] = g2r[i] * a2f;
        }
    }
    // this thread's chunk: first (image, token), its eight tmax entries
    const int col0 = ch * VEC;
    const bool live = ch < nch && col0 < ncols;
    int yi0 = 0, k0 = 0;
    short tmv[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) tmv[e] = (short)-1;
    if (live) {
        const int y = col0 / ni;
        yi0 = y - y_first;
        k0 = col0 - y * ni;
        const long tbase = ((long)x * ytotal + y0) * ni + col0;
        if (VEC == 8 && (tbase & 1) == 0 && col0 + VEC <= ncols) {             // eight entries, 4-byte aligned: one 16-byte load
            typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));      // (dword-aligned 16-byte load)
            const u32x4_a4 raw = *reinterpret_cast<const u32x4_a4*>(tmax + tbase);
#pragma unroll
            for (int e = 0; e < 4; ++e) { tmv[2 * e] = (short)(raw[e] & 0xffffu); tmv[2 * e + 1] = (short)(raw[e] >> 16); }
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) if (col0 + e < ncols) tmv[e] = tmax[tbase + e];
        }
    }
    const int tb = count > 0 ? (ROUTE_KM_ENTRIES / count > 0 ? ROUTE_KM_ENTRIES / count : 1) : nt;      // rows per kmax batch
    for (int t0 = 0; t0 < nt; t0 += tb) {
        const int t1 = t0 + tb < nt ? t0 + tb : nt;
        sync();                                                  // (the previous batch's entries are no longer read; s_a1 / s_a2 are in)
        for (int i = threadIdx.x; i < (t1 - t0) * count; i += 256) {
            const int r = i / count, c = i - r * count;
            s_km[i] = kmax[((long)x * nt + t0 + r) * ytotal + y0 + y_first + c];
        }
        sync();
        if (ch < nch) {
            for (int t = t0; t < t1; ++t) {
                const long row = (long)x * nt + t;
                float v[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] = 0.f;
                if (live && mask[row] != 0) {                    // (rows of padding tokens and the padding columns stay zero)
                    const short* kmr = s_km + (t - t0) * count;
                    int yi = yi0, k = k0;
                    int kbest = kmr[yi];
                    float a1 = s_a1[yi], a2 = s_a2[yi];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        if (col0 + e < ncols) {
                            float val = (k == kbest) ? a1 : 0.f;
                            if (tmv[e] == t) val += a2;
                            v[e] = val;
                            if (++k == ni && col0 + e + 1 < ncols) {
                                k = 0; ++yi;
                                kbest = kmr[yi]; a1 = s_a1[yi]; a2 = s_a2[yi];
                            }
                        }
                    }
                }
                store_vec<T>(P + row * ldp + (long)ch * VEC, v);
            }
        }
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
