// tokens.h -- gathers / scatters either side of the transformer stacks: text embedding, image patchify
// (with the patch-dropout keep-set folded in), vision mean-pool, dtype casts.  All HBM-bound.
#pragma once
#include "common.h"

namespace xc {

// ---- text embedding (reference TextTransformer.forward, x_clip.py:320-335) ----------------------------------
// out[b, 0] = cls ; out[b, 1+j] = E[tok[b, j]] + P[j]   (cls == nullptr: no CLS row; P == nullptr: no abs-pos)
// A token id outside [0, vocab) (nn.Embedding raises IndexError / a device assert there) never touches memory: its output row is
// NaN -- the loss of that step is loudly not a number -- and *bad_flag (may be null) is set for the host to turn into the IndexError.
template <typename T>
__global__ __launch_bounds__(256) void text_embed_fwd_kernel(const long long* __restrict__ tok, const T* __restrict__ E,
                                                             const T* __restrict__ P, const T* __restrict__ cls,
                                                             T* __restrict__ out, int batch, int n, int D, long long vocab,
                                                             int* __restrict__ bad_flag) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = lane_id();
    const int npos = n + (cls != nullptr ? 1 : 0);
    const long row = (long)blockIdx.x * 4 + wave_id();
    if (row >= (long)batch * npos) return;
    const int bi = (int)(row / npos), pos = (int)(row % npos);
    const int j = pos - (cls != nullptr ? 1 : 0);
    const int nch = D / VEC;
    long long id = (j < 0) ? 0 : tok[(long)bi * n + j];
    const bool bad = id < 0 || id >= vocab;                   // (wave-uniform)
    if (bad) {
        id = 0;
        if (lane == 0 && bad_flag != nullptr) *bad_flag = 1;
    }
    const T* src = (j < 0) ? cls : E + (long)id * D;
    for (int c = lane; c < nch; c += 64) {
        float v[VEC];
        load_vec<T>(src + c * VEC, v);
        if (bad) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) v[k] = __builtin_nanf("");
        }
        if (j >= 0 && P != nullptr) {
            float pv[VEC];
            load_vec<T>(P + (long)j * D + c * VEC, pv);
#pragma unroll
            for (int k = 0; k < VEC; ++k) v[k] += pv[k];
        }
        store_vec<T>(out + row * (long)D + c * VEC, v);
    }
}

// Backward: dE[tok] += dout (fp32 atomics, vocabulary rows are hit at random), dP[j] / dcls = column sums over
// the batch (kept in registers per wave, one atomic per column per wave at the end).
// grid = (npos, batch splits); block = 4 waves, each wave strides over batch rows of its position.
template <typename T, int MAXC>
__global__ __launch_bounds__(256) void text_embed_bwd_kernel(const T* __restrict__ dout, const long long* __restrict__ tok,
                                                             float* __restrict__ dE, float* __restrict__ dP,
                                                             float* __restrict__ dcls, int batch, int n, int D, int has_cls,
                                                             long long vocab) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = lane_id();
    const int npos = n + has_cls;
    const int pos = blockIdx.x;
    const int j = pos - has_cls;
    const int nch = D / VEC;
    float acc[MAXC][VEC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[i][k] = 0.f;
    for (int bi = blockIdx.y * 4 + wave_id(); bi < batch; bi += gridDim.y * 4) {
        const T* src = dout + ((long)bi * npos + pos) * D;
        float* erow = nullptr;
        if (j >= 0 && dE != nullptr) {
            const long long id = tok[(long)bi * n + j];
            if (id >= 0 && id < vocab) erow = dE + (long)id * D;   // out-of-range ids (flagged by the forward) own no row
        }
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                float v[VEC];
                load_vec<T>(src + c * VEC, v);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    acc[i][k] += v[k];
                    if (erow != nullptr) atomic_add(erow + c * VEC + k, v[k]);
                }
            }
        }
    }
    // the four waves' column sums meet in LDS; one atomic per column per work-group (gridDim.y-way contention per address
    // instead of 4 gridDim.y: the position / CLS rows are few and every work-group of a position adds into the same one)
    XC_LDS_DYNAMIC(lds);
    float* red = reinterpret_cast<float*>(lds);            // [3][D]
    const int wave = wave_id();
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch)
#pragma unroll
                for (int k = 0; k < VEC; ++k) red[(wave - 1) * D + c * VEC + k] = acc[i][k];
        }
    }
    sync();
    float* dst = (j >= 0) ? (dP != nullptr ? dP + (long)j * D : nullptr) : dcls;
    if (dst != nullptr && wave == 0) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch)
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const int col = c * VEC + k;
                    atomic_add(dst + col, acc[i][k] + red[col] + red[D + col] + red[2 * D + col]);
                }
        }
    }
}

// ---- patchify: 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' (x_clip.py:357) with the patch-dropout keep-set applied ----
// out[(b, i), e] = img[b, c, ph*p + p1, pw*p + p2],  e = (p1*p + p2)*C + c,  patch = keep ? keep[b, i] : i.
// One 16-byte output chunk per thread; rows are padded with zeros up to ldo (K multiple of the chunk width).
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const T* __restrict__ img, const int* __restrict__ keep,
                                                       T* __restrict__ out, long ldo, int batch, int C, int H, int W,
                                                       int p, int nkeep) {
    constexpr int VEC = Elem<T>::VEC;
    const int gw = W / p;
    const int rowlen = p * p * C;
    const int nch = (int)(ldo / VEC);
    const long total = (long)batch * nkeep * nch;
    for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(id % nch);
        const long row = id / nch;
        const int bi = (int)(row / nkeep), i = (int)(row % nkeep);
        const int patch = keep != nullptr ? keep[row] : i;
        const int ph = patch / gw, pw = patch % gw;
        T vals[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int e = ch * VEC + k;
            if (e < rowlen) {
                const int c = e % C, p2 = (e / C) % p, p1 = e / (C * p);
                vals[k] = img[(((long)bi * C + c) * H + ph * p + p1) * W + pw * p + p2];
            } else {
                vals[k] = from_f32<T>(0.f);
            }
        }
        st16(out + row * ldo + ch * VEC, *reinterpret_cast<u32x4*>(vals));
    }
}

// The same for bf16 images with three channels and a patch edge that is a multiple of 8 (round 6): a thread takes 8 pixels of one patch row --
// one 16-byte load from each channel plane -- interleaves them into the 24 output elements (p1, p2 .. p2 + 7, c) and stores three 16-byte
// chunks that continue its neighbour's: whole 64-byte runs in, whole rows out.  (The element-wise kernel above issues eight 2-byte loads with
// their own index arithmetic per chunk: 223 us for the 201 MB of configs[1]'s kept patches.)
__global__ __launch_bounds__(256) void patchify_rgb8_kernel(const bf16_t* __restrict__ img, const int* __restrict__ keep, bf16_t* __restrict__ out,
                                                            long ldo, int batch, int H, int W, int p, int nkeep) {
    const int gw = W / p, g8 = p >> 3;                         // groups of 8 pixels in a patch row
    const int per_patch = p * g8;
    const long total = (long)batch * nkeep * per_patch;
    const long plane = (long)H * W;
    for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
        const int t = (int)(id % per_patch);
        const long row = id / per_patch;
        const int p1 = t / g8, q = t - p1 * g8;
        const int bi = (int)(row / nkeep), i = (int)(row - (long)bi * nkeep);
        const int patch = keep != nullptr ? keep[row] : i;
        const int ph = patch / gw, pw = patch - ph * gw;
        const bf16_t* src = img + (long)bi * 3 * plane + (long)(ph * p + p1) * W + pw * p + q * 8;
        const u32x4 r = ld16(src), g = ld16(src + plane), b = ld16(src + 2 * plane);
        // pixel k of channel X: the low (k even) or high half of X[k >> 1]; output element 3 k + c
        uint32_t o[12];
#pragma unroll
        for (int w = 0; w < 4; ++w) {                          // two pixels (2 w, 2 w + 1) -> three output words
            const uint32_t r0 = r[w] & 0xffffu, r1 = r[w] >> 16, g0 = g[w] & 0xffffu, g1 = g[w] >> 16, b0 = b[w] & 0xffffu, b1 = b[w] >> 16;
            o[3 * w] = r0 | (g0 << 16);
            o[3 * w + 1] = b0 | (r1 << 16);
            o[3 * w + 2] = g1 | (b1 << 16);
        }
        bf16_t* dst = out + row * ldo + ((long)p1 * p + q * 8) * 3;
        st16(dst, u32x4{o[0], o[1], o[2], o[3]});
        st16(dst + 8, u32x4{o[4], o[5], o[6], o[7]});
        st16(dst + 16, u32x4{o[8], o[9], o[10], o[11]});
    }
}

// ---- vision CLS pooling: mean over tokens (x_clip.py:366-370) ---------------------------------------------
// x[b, t, :] at x + b * xbs + t * D (xbs = batch stride in elements, so the tokens may sit behind a CLS slot).
// grid = (batch, ceil(nch / 64)), block = one wave.
template <typename T>
__global__ __launch_bounds__(64) void token_mean_fwd_kernel(const T* __restrict__ x, long xbs, T* __restrict__ out, int n,
                                                            int D) {
    constexpr int VEC = Elem<T>::VEC;
    const int c = blockIdx.y * 64 + threadIdx.x;
    if (c >= D / VEC) return;
    const T* src = x + (long)blockIdx.x * xbs + c * VEC;
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    for (int t = 0; t < n; ++t) {
        float v[VEC];
        load_vec<T>(src + (long)t * D, v);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] += v[k];
    }
    const float inv = 1.0f / (float)n;
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] *= inv;
    store_vec<T>(out + (long)blockIdx.x * D + c * VEC, acc);
}

// dx[b, t] = dout[b] / n (+ dsrc[b, t], dsrc rows at dsrc + b * sbs + t * D);  dx is contiguous [batch, n, D]
template <typename T>
__global__ __launch_bounds__(256) void token_mean_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ dsrc, long sbs,
                                                             T* __restrict__ dx, int batch, int n, int D) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = lane_id();
    const long row = (long)blockIdx.x * 4 + wave_id();
    if (row >= (long)batch * n) return;
    const long bi = row / n, t = row % n;
    const float inv = 1.0f / (float)n;
    for (int c = lane; c < D / VEC; c += 64) {
        float v[VEC];
        load_vec<T>(dout + bi * D + c * VEC, v);
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] *= inv;
        if (dsrc != nullptr) {
            float o[VEC];
            load_vec<T>(dsrc + bi * sbs + t * D + c * VEC, o);
#pragma unroll
            for (int k = 0; k < VEC; ++k) v[k] += o[k];
        }
        store_vec<T>(dx + row * (long)D + c * VEC, v);
    }
}

// ---- strided row copy: dst[r, :] = src[r, :]  (rows at src + r*lds / dst + r*ldd) ---------------------------
// Used to drop the CLS-row gradients into a zeroed [b, n, D] gradient buffer and similar re-layouts.
template <typename T>
__global__ __launch_bounds__(256) void copy_rows_kernel(const T* __restrict__ src, long lds_, T* __restrict__ dst, long ldd,
                                                        long rows, int D) {
    constexpr int VEC = Elem<T>::VEC;
    const int nch = D / VEC;
    const long total = rows * nch;
    for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
        const long r = id / nch;
        const int c = (int)(id % nch);
        st16(dst + r * ldd + c * VEC, ld16(src + r * lds_ + c * VEC));
    }
}

// out[r, :] = a[r, :] + b[r, :]  (contiguous [rows, D]; sums gradient blocks of views that feed several pairs)
template <typename T>
__global__ __launch_bounds__(256) void add_rows_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, long n16) {
    constexpr int VEC = Elem<T>::VEC;
    for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n16; id += (long)gridDim.x * blockDim.x) {
        float x[VEC], y[VEC];
        load_vec<T>(a + id * VEC, x);
        load_vec<T>(b + id * VEC, y);
#pragma unroll
        for (int k = 0; k < VEC; ++k) x[k] += y[k];
        store_vec<T>(out + id * VEC, x);
    }
}

// ---- row scatter-add / column sum (patch-embed bias and position-table gradients, x_clip.py:358,382-383) ----
// table[idx[r], :] += src[r, :]  (idx == nullptr: skipped)    colsum[:] += sum_r src[r, :]  (nullptr: skipped)
// Waves stride over rows; the column sum stays in registers until the end (one atomic per column per wave).
template <typename T, int MAXC>
__global__ __launch_bounds__(256) void rows_scatter_add_kernel(const T* __restrict__ src, long lds_, const int* __restrict__ idx,
                                                               float* __restrict__ table, float* __restrict__ colsum, long rows,
                                                               int D, float* __restrict__ colsum_partial) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = lane_id();
    const int nch = D / VEC;
    float acc[MAXC][VEC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[i][k] = 0.f;
    for (long r = (long)blockIdx.x * 4 + wave_id(); r < rows; r += (long)gridDim.x * 4) {
        float* trow = (table != nullptr && idx != nullptr) ? table + (long)idx[r] * D : nullptr;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                float v[VEC];
                load_vec<T>(src + r * lds_ + c * VEC, v);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    acc[i][k] += v[k];
                    if (trow != nullptr) atomic_add(trow + c * VEC + k, v[k]);
                }
            }
        }
    }
    if (colsum_partial != nullptr) {                          // one partial row per WAVE; colsum_fold_kernel adds them up
        float* out = colsum_partial + ((long)blockIdx.x * 4 + wave_id()) * D;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch)
#pragma unroll
                for (int k = 0; k < VEC; ++k) out[c * VEC + k] = acc[i][k];
        }
    } else if (colsum != nullptr) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch)
#pragma unroll
                for (int k = 0; k < VEC; ++k) atomic_add(colsum + c * VEC + k, acc[i][k]);
        }
    }
}

// ---- sorted segmented scatter-add (token-embedding / position-table gradients) --------------------------------
// table[ids[e], :] += src[row(perm[e]), :] for e in [0, count), where `ids` is SORTED ascending and perm[e] is the
// original position of entry e; row(p) = (p / n_in) * n_out + p % n_in + row_off maps a flat token index to its row of
// src (text: p = b * n + j -> row b * (n+1) + j + 1 behind the CLS slot; plain row scatter: n_in = n_out = 1).
// One wave per `chunk` consecutive sorted entries: equal ids are summed in registers and flushed with one fp32
// atomic per column when the id changes -- a vocabulary row hit k times costs ~1 flush instead of k, which is what made
// the unsorted scatter (one atomic per element) the slowest non-GEMM kernel of the step.
// The flush goes through the wave's D floats of LDS (`staged`: the launcher grants 4 x D x 4 bytes) so that one atomic instruction covers 64
// CONSECUTIVE floats of the table row: a lane accumulates the 16-byte chunks it loaded (VEC consecutive columns), and flushing those directly
// makes every atomic instruction touch 64 different 32-byte sectors -- with ids drawn uniformly (a run is ~26 entries at 262 k tokens over
// 10 k ids) the kernel was bound by that: 13 M sector atomics per launch.
template <typename T, int MAXC>
__global__ __launch_bounds__(256) void scatter_add_sorted_kernel(const T* __restrict__ src, long lds_, const long long* __restrict__ ids,
                                                                 const long long* __restrict__ perm, float* __restrict__ table,
                                                                 long count, int D, int n_in, int n_out, int row_off, int chunk,
                                                                 long long table_rows, int staged) {
    constexpr int VEC = Elem<T>::VEC;
    XC_LDS_DYNAMIC(lds);
    float* const stage = reinterpret_cast<float*>(lds) + (long)wave_id() * D;
    const int lane = lane_id();
    const int nch = D / VEC;
    const long e0 = ((long)blockIdx.x * 4 + wave_id()) * chunk;
    if (e0 >= count) return;
    const long e1 = e0 + chunk < count ? e0 + chunk : count;
    float acc[MAXC][VEC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[i][k] = 0.f;
    long long cur = ids[e0];
    for (long e = e0; e <= e1; ++e) {
        const long long id = e < e1 ? ids[e] : -1;
        if (id != cur) {                                   // wave-uniform: flush the finished run
            float* trow = table + (long)cur * D;
            const bool in_table = cur >= 0 && cur < table_rows;  // an id without a row is dropped, never written out of bounds
            if (staged) {
#pragma unroll
                for (int i = 0; i < MAXC; ++i) {
                    const int c = lane + 64 * i;
                    if (c < nch)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) { stage[c * VEC + k] = acc[i][k]; acc[i][k] = 0.f; }
                }
                lds_fence();                               // (wave-private hand-off: the lanes read what other lanes stored)
                if (in_table)
                    for (int j = lane; j < D; j += 64) atomic_add(trow + j, stage[j]);
                lds_fence();                               // (... before the next flush overwrites it)
            } else {
#pragma unroll
                for (int i = 0; i < MAXC; ++i) {
                    const int c = lane + 64 * i;
                    if (c < nch)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) {
                            if (in_table) atomic_add(trow + c * VEC + k, acc[i][k]);
                            acc[i][k] = 0.f;
                        }
                }
            }
            cur = id;
        }
        if (e < e1) {
            const long p = (long)perm[e];
            const long row = (p / n_in) * n_out + p % n_in + row_off;
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = lane + 64 * i;
                if (c < nch) {
                    float v[VEC];
                    load_vec<T>(src + row * lds_ + c * VEC, v);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[i][k] += v[k];
                }
            }
        }
    }
}

// ---- fp32 accumulator -> storage type (gain / embedding gradients) ---------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cast_from_f32_kernel(const float* __restrict__ src, T* __restrict__ dst, long n,
                                                            float scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = from_f32<T>(src[i] * scale);
}

// ---- rotary position embedding on the packed qkv rows (x_clip.py:155-176, 221-223) --------------------------------------------
// X [rows, slots * 64] (q | k | v, heads contiguous; slots = 3 * heads), token position = row % n.  In every 64-wide head slot
// the first ROT = 32 features are rotated pairwise (j, j + 16) by the angle pos * inv_freq[j] (the module's buffer, 16 fp32 values:
// bit-identical angles to the reference's table); the other 32 pass through
// (the reference applies it to q, k AND v).  sign = +1: forward; -1: the transposed rotation = its backward.  In place; one
// lane owns the two 16-byte chunks that hold a set of pairs, so no exchange is needed.
template <typename T>
__global__ __launch_bounds__(256) void rotary_kernel(T* __restrict__ X, long ld, long rows, int n, int slots, int slot_width,
                                                     const float* __restrict__ inv_freq, float sign) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int HALF = 16;                                   // ROT / 2
    constexpr int CPH = HALF / VEC;                            // chunks per half: 2 (bf16) / 4 (fp32)
    const long items = rows * slots * CPH;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
        const int c = (int)(it % CPH);
        const long rs = it / CPH;
        const int slot = (int)(rs % slots);
        const long row = rs / slots;
        const float pos = (float)(row % n);
        T* p = X + row * ld + slot * slot_width + c * VEC;
        float a[VEC], b[VEC];
        load_vec<T>(p, a);                                     // features j      = c VEC + (0 .. VEC-1)
        load_vec<T>(p + HALF, b);                              // features j + 16
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float sn, cs;
            sincosf(pos * inv_freq[c * VEC + e], &sn, &cs);
            sn *= sign;
            const float x1 = a[e], x2 = b[e];
            a[e] = x1 * cs - x2 * sn;
            b[e] = x2 * cs + x1 * sn;
        }
        store_vec<T>(p, a);
        store_vec<T>(p + HALF, b);
    }
}

// the same rotation for rot < 32 features per head (heads narrower than 32: the reference rotates min(dim_head, 32), x_clip.py:311): pairs
// (j, j + rot / 2), one lane per pair, element accesses -- rot / 2 is not a whole 16-byte chunk in general (dim_head 24: 12 pairs)
template <typename T>
__global__ __launch_bounds__(256) void rotary_pairs_kernel(T* __restrict__ X, long ld, long rows, int n, int slots, int slot_width, int half,
                                                           const float* __restrict__ inv_freq, float sign) {
    const long items = rows * slots * half;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
        const int j = (int)(it % half);
        const long rs = it / half;
        const int slot = (int)(rs % slots);
        const long row = rs / slots;
        T* p = X + row * ld + slot * slot_width + j;
        float sn, cs;
        sincosf((float)(row % n) * inv_freq[j], &sn, &cs);
        sn *= sign;
        const float x1 = to_f32(p[0]), x2 = to_f32(p[half]);
        p[0] = from_f32<T>(x1 * cs - x2 * sn);
        p[half] = from_f32<T>(x2 * cs + x1 * sn);
    }
}

// ---- feed-forward dropout (reference nn.Dropout between the inner LayerNorm and the second Linear, x_clip.py:193-194) -------------
// y[i] = x[i] * keep(i) / (1 - p) over a contiguous [n] tensor, keep from drop_hash(seed, i) (common.h).  The same launch on the
// gradient is the backward.  In place is fine.  One 16-byte chunk per lane.
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, long n, uint32_t thresh, float scale,
                                                      uint64_t seed) {
    constexpr int VEC = Elem<T>::VEC;
    const long chunks = n / VEC;
    for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < chunks; c += (long)gridDim.x * blockDim.x) {
        float v[VEC];
        load_vec<T>(x + c * VEC, v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = drop_hash(seed, (uint64_t)(c * VEC + e)) >= thresh ? v[e] * scale : 0.f;
        store_vec<T>(y + c * VEC, v);
    }
}

// ---- depthwise 4 x 4 / stride 2 / pad 1 convolution over a square token grid (`downsample_image_embeds`, x_clip.py:560-568) ----
// x [batch, h * h, C] token-major (channels contiguous), w [C, 16] (the Conv2d weight [C, 1, 4, 4]), y [batch, (h/2)^2, C]:
//   y[b, (i, j), c] = sum_{u, v} w[c, 4 u + v] x[b, (2 i - 1 + u, 2 j - 1 + v), c]      (out-of-range taps are zero padding)
// A lane owns one 16-byte channel chunk: every x / y access is a coalesced row segment, and the chunk's 16 x VEC weights are one
// contiguous run of w.
template <typename T>
XC_DEV void dwconv_load_w(const T* __restrict__ w, int c0, float (&wf)[Elem<T>::VEC][16]) {
    constexpr int VEC = Elem<T>::VEC;
#pragma unroll
    for (int q = 0; q < 16; ++q) {                             // flat element e = k * 16 + tap, VEC elements per load
        float t[VEC];
        load_vec<T>(w + (long)c0 * 16 + q * VEC, t);
#pragma unroll
        for (int e = 0; e < VEC; ++e) wf[(q * VEC + e) / 16][(q * VEC + e) % 16] = t[e];
    }
}
template <typename T>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const T* __restrict__ x, const T* __restrict__ w, T* __restrict__ y, int batch,
                                                         int h, int C) {
    constexpr int VEC = Elem<T>::VEC;
    const int ho = h / 2, nch = C / VEC;
    const long items = (long)batch * ho * ho * nch;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
        const int c = (int)(it % nch);
        const long t = it / nch;
        const int j = (int)(t % ho), i = (int)((t / ho) % ho);
        const long b = t / ((long)ho * ho);
        float wf[VEC][16];
        dwconv_load_w<T>(w, c * VEC, wf);
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = 2 * i - 1 + u;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int q = 2 * j - 1 + v;
                if (p >= 0 && p < h && q >= 0 && q < h) {
                    float xv[VEC];
                    load_vec<T>(x + ((b * h + p) * h + q) * (long)C + c * VEC, xv);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] += wf[e][u * 4 + v] * xv[e];
                }
            }
        }
        store_vec<T>(y + t * (long)C + c * VEC, acc);
    }
}
// backward: dx (gather form: an input token feeds at most 2 x 2 outputs) and per-wave partial rows of dw [C * 16] (fp32) that the
// caller folds with colsum_fold_kernel.  partial: [gridDim.x * 4, C * 16], zero-initialised by the caller.
template <typename T>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ w,
                                                         T* __restrict__ dx, float* __restrict__ partial, int batch, int h, int C) {
    constexpr int VEC = Elem<T>::VEC;
    const int ho = h / 2, nch = C / VEC;
    // ---- dx ----
    const long in_items = (long)batch * h * h * nch;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < in_items; it += (long)gridDim.x * blockDim.x) {
        const int c = (int)(it % nch);
        const long t = it / nch;
        const int q = (int)(t % h), p = (int)((t / h) % h);
        const long b = t / ((long)h * h);
        float wf[VEC][16];
        dwconv_load_w<T>(w, c * VEC, wf);
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
        for (int di = 0; di < 2; ++di) {
            const int i = (p + 1) / 2 - di, u = p + 1 - 2 * i;                 // 2 i - 1 + u = p
#pragma unroll
            for (int dj = 0; dj < 2; ++dj) {
                const int j = (q + 1) / 2 - dj, v = q + 1 - 2 * j;
                if (i >= 0 && i < ho && u >= 0 && u < 4 && j >= 0 && j < ho && v >= 0 && v < 4) {
                    float g[VEC];
                    load_vec<T>(dy + ((b * ho + i) * ho + j) * (long)C + c * VEC, g);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        float wsel = 0.f;                                      // w[e][4 u + v] with a run-time tap: select over the 16 registers
#pragma unroll
                        for (int tap = 0; tap < 16; ++tap) wsel = (tap == u * 4 + v) ? wf[e][tap] : wsel;
                        acc[e] += wsel * g[e];
                    }
                }
            }
        }
        store_vec<T>(dx + t * (long)C + c * VEC, acc);
    }
    // ---- dw partials: lane <-> channel chunk, (work-group, wave) <-> a strided share of the (b, i, j) outputs ----
    const int lane = lane_id(), wave = wave_id();
    const long outs = (long)batch * ho * ho;
    float* prow = partial + ((long)blockIdx.x * 4 + wave) * C * 16;
    for (int c = lane; c < nch; c += 64) {
        float wacc[VEC][16];
#pragma unroll
        for (int e = 0; e < VEC; ++e)
#pragma unroll
            for (int tap = 0; tap < 16; ++tap) wacc[e][tap] = 0.f;
        for (long t = (long)blockIdx.x * 4 + wave; t < outs; t += (long)gridDim.x * 4) {
            const int j = (int)(t % ho), i = (int)((t / ho) % ho);
            const long b = t / ((long)ho * ho);
            float g[VEC];
            load_vec<T>(dy + t * (long)C + c * VEC, g);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = 2 * i - 1 + u;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int q = 2 * j - 1 + v;
                    if (p >= 0 && p < h && q >= 0 && q < h) {
                        float xv[VEC];
                        load_vec<T>(x + ((b * h + p) * h + q) * (long)C + c * VEC, xv);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) wacc[e][u * 4 + v] += g[e] * xv[e];
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e)
#pragma unroll
            for (int tap = 0; tap < 16; ++tap) prow[(long)(c * VEC + e) * 16 + tap] = wacc[e][tap];
    }
}

// ---- masked-language-model head (mlm.py:96-109) ---------------------------------------------------------------------------------
// gather: out[r, :] = src[idx[r], :]   (only the masked positions of the encoder output go through the vocabulary projection)
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ src, long lds_, const int* __restrict__ idx,
                                                          T* __restrict__ out, long rows, int D) {
    constexpr int VEC = Elem<T>::VEC;
    const int nch = D / VEC;
    const long total = rows * nch;
    for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
        const long r = id / nch;
        const int c = (int)(id % nch);
        st16(out + r * (long)D + c * VEC, ld16(src + (long)idx[r] * lds_ + c * VEC));
    }
}
// cross-entropy over the first `cols` columns of logits [rows, ld] (any padding columns up to ld are ignored) with one int64 label
// per row: lse[r] = log sum_c exp(x[r, c]);  *loss_accum += sum_r (lse[r] - x[r, label[r]]).  One wave per row, a wave walks rows
// blockIdx.x * 4 + wave, + 4 gridDim.x, ...; the grid is capped (ROWLOSS_MAX_BLOCKS) and every work-group adds ONE partial sum: tens of
// thousands of atomics on the same address serialise in L2 at ~10 ns each (measured on the SimSiam loss: 110 us for 8448 work-groups).
constexpr int ROWLOSS_MAX_BLOCKS = 1024;
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* __restrict__ x, long ld, const long long* __restrict__ label, int rows,
                                                     int cols, float* __restrict__ lse, float* __restrict__ loss) {
    constexpr int VEC = Elem<T>::VEC;
    XC_LDS_DYNAMIC(lds_raw);
    float* red = reinterpret_cast<float*>(lds_raw);        // [4]
    const int lane = lane_id();
    const int nch = (cols + VEC - 1) / VEC;
    float part = 0.f;
    for (long r = (long)blockIdx.x * 4 + wave_id(); r < rows; r += (long)gridDim.x * 4) {
        const T* row = x + r * ld;
        float m = -3.0e38f;
        for (int c = lane; c < nch; c += 64) {
            float v[VEC];
            load_vec<T>(row + c * VEC, v);
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                if (c * VEC + e < cols) m = fmaxf(m, v[e]);
        }
        m = wave_max(m);
        float l = 0.f;
        for (int c = lane; c < nch; c += 64) {
            float v[VEC];
            load_vec<T>(row + c * VEC, v);
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                if (c * VEC + e < cols) l += fast_exp(v[e] - m);
        }
        l = wave_sum(l);
        const float v = m + logf(l);
        if (lane == 0) lse[r] = v;
        part += v - to_f32(row[label[r]]);
    }
    if (lane == 0) red[wave_id()] = part;
    sync();
    if (threadIdx.x == 0) atomic_add(loss, red[0] + red[1] + red[2] + red[3]);
}
// in place: x[r, c] <- scale * (exp(x[r, c] - lse[r]) - [c == label[r]]) for c < cols, 0 for the padding columns;  scale = *gmul / rows
// (mean over the selected rows times the upstream gradient, a device scalar)
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(T* __restrict__ x, long ld, const long long* __restrict__ label,
                                                     const float* __restrict__ lse, const float* __restrict__ gmul, int rows, int cols) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = lane_id();
    const long r = (long)blockIdx.x * 4 + wave_id();
    if (r >= rows) return;
    T* row = x + r * ld;
    const float scale = *gmul / (float)rows, L = lse[r];
    const int lab = (int)label[r];
    for (int c = lane; c < (int)(ld / VEC); c += 64) {
        float v[VEC];
        load_vec<T>(row + c * VEC, v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int col = c * VEC + e;
            v[e] = col < cols ? scale * (fast_exp(v[e] - L) - (col == lab ? 1.f : 0.f)) : 0.f;
        }
        store_vec<T>(row + c * VEC, v);
    }
}

// ---- diagnostics: the shader clock as the hardware sees it ---------------------------------------------------------------------------
// One wave reads the shader-cycle counter (s_memtime) against the constant 100 MHz counter (s_memrealtime) over `ticks` ticks of 10 ns:
// out[0] = shader cycles, out[1] = 10 ns ticks elapsed.  Launched between the kernels of a step it reads the clock the power management
// holds the part at under that load (DVFS reacts in milliseconds, the sample takes microseconds): bench.py reports it so that a slow
// box and a slow kernel can be told apart (DESIGN_APPENDIX.md section 5: 1.55 - 1.68 GHz under the GEMMs, 2.0 - 2.15 GHz with the MFMA loop alone).
__global__ void clock_sample_kernel(uint64_t* out, long ticks) {
    if (threadIdx.x != 0) return;
    const uint64_t c0 = shader_cycles(), r0 = realtime_10ns();
    uint64_t r1 = r0;
    while ((long)(r1 - r0) < ticks) {
        nap();
        r1 = realtime_10ns();
    }
    out[0] = shader_cycles() - c0;
    out[1] = r1 - r0;
}

}  // namespace xc
