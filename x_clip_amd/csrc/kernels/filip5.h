// filip5.h -- the fine-grained (FILIP) head's forward with its reductions INSIDE the token-similarity GEMM: the block
//     s[(x, t), (y, k)] = <T[x, t], I[y, k]>          (reference x_clip.py:797-803, the 'x t d, y i d -> x y t i' einsum)
// is never written.  The production GEMM loop (gemm4.h g5_run: 256 x 256 tiles, ring of three A stages, LDS-DMA) runs over
// rows = text tokens (x, t) and columns = image tokens (y, k); the epilogue of a tile reduces its accumulators in both directions
//     row direction   : max_k over the columns of one image y      -> t2i[x, y] = sum_t w[x,t] max_k s / cnt[x]   (x_clip.py:805-807)
//     column direction: max_{t live} over the rows of one text x   -> i2t[x, y] = mean_k max_t s                  (x_clip.py:809-811)
// down to 4-byte partials {bf16 value | int16 arg-max}: per (row, 64-column wave block, image: such a block overlaps at most three)
// and per (column, 128-row wave block, text: at most five).  Two small merge kernels combine the partials of the
// blocks a segment spans into t2i / i2t and the int16 arg-max maps the backward routes through (filip.h filip_route_kernel: unchanged).
// What the chunked form (filip.h filip_reduce_rows_kernel over a [b nt, yc ni] workspace of at most 1 GiB) paid per step at the
// configs[3] shape -- the write and re-read of 4 GB of similarities and 4.1 ms of reduction passes -- becomes ~1000 vector
// instructions per wave and tile behind the tile's 256 MFMAs.
//
// Requirements (the host falls back to the chunked form otherwise): bf16, d a whole number of K steps, ni >= 32 (a 64-column wave
// block then overlaps at most F5_SIDES = 3 images) and nt >= 32 (a 128-row wave block overlaps at most F5_SLOTS = 5 texts) -- e.g. the
// README's FILIP model under patch dropout (32 of 64 patches kept) as well as BASELINE configs[3] (77 x 98).
//
// Row direction, in the accumulator layout (MFMA operands swapped: a lane owns a row (lane & 31) and 32 of its wave block's 64 columns,
// j * 32 + (e & 3) + 8 (e >> 2) + 4 (lane >> 5)): the lane scans its columns in increasing order keeping a (max, arg) per image of
// the block -- which image an 8-column group belongs to is wave-uniform except for the groups an image boundary cuts --, meets its
// partner lane (lane ^ 32) once per 32-row block, and leaves up to three entries.  fp32 compares; first index wins ties
// like torch.max.
// Column direction, through the wave's private 4 KiB slice of the freed A stage (what the plain GEMM's whole-line epilogue uses): the
// 32 x 64 block goes down as bf16 in the swizzled line layout, then lane = column walks the 32 rows, skipping padding tokens (their
// rows take no part, x_clip.py:809) and flushing an entry whenever the text changes.
#pragma once
#include "gemm4.h"

namespace xc {

constexpr uint32_t F5_EMPTY = 0xff800000u;                      // value -inf, arg 0
constexpr int F5_SIDES = 3;                                      // images a 64-column wave block can overlap (ni >= 32)
constexpr int F5_SLOTS = 5;                                      // texts a 128-row wave block can overlap (nt >= 32)

struct Filip5Params {
    const unsigned char* mask;       // [M] one byte per text-token row (x, t): 1 = real token
    uint32_t* rowpart;               // [M][nblk64][F5_SIDES]   row-direction partials
    uint32_t* colpart;               // [nrblk][F5_SLOTS][N]    column-direction partials
    int M, N, nt, ni, nblk64;
};

XC_DEV uint32_t f5_entry(float v, int arg) { return ((uint32_t)f2bf(v) << 16) | ((uint32_t)arg & 0xffffu); }

struct Filip5Epilogue {
    const Filip5Params& f;

    XC_DEV void finish() {}
    XC_DEV bool packs_lines(int, int) const { return false; }
    XC_DEV void pack_lines(f32x16 (&)[4][2], unsigned char*, u32x4 (&)[4][4], int, int) const {}
    template <bool NT = false> XC_DEV void store_lines(const u32x4 (&)[4][4], int, int) const {}

    // one element of the row scan: compile-time column c (without the lane's 4 h) into side accumulator (v, a)
    static XC_DEV void take(float s, int c, float& v, int& a) {
        if (s > v) { v = s; a = c; }
    }

    XC_DEV int with_scratch(f32x16 (&acc)[4][2], int m0, int n0, unsigned char* scratch) {
        const int lane = threadIdx.x & 63, h = lane >> 5, r31 = lane & 31;
        const int wave = uniform(threadIdx.x >> 6), wm = wave >> 2, wn = wave & 3;
        const float NEG = -3.0e38f;
        const int gc0 = n0 + wn * 64;                                   // first global column of the wave block
        const int gr0 = m0 + wm * 128;                                  // first global row
        // ---- geometry of the row direction: the (at most two) image boundaries inside the 64 columns, the number of real columns ----
        const int yL = gc0 / f.ni;
        int cend = f.N - gc0;                                           // columns >= cend are padding of the operand (or past it)
        cend = cend < 64 ? cend : 64;
        int cb1 = (yL + 1) * f.ni - gc0;                                // block-relative first column of image yL + 1
        int cb2 = cb1 + f.ni;                                           // ... of image yL + 2
        if (cb1 > cend) cb1 = cend;
        if (cb2 > cend) cb2 = cend;
        // ---- geometry of the column direction: texts x0, x0 + 1, x0 + 2 overlap the 128 rows ----
        const int x0 = gr0 / f.nt;
        int slot = 0;                                                   // text x0 + slot is being accumulated
        int next_change = (x0 + 1) * f.nt - gr0;                        // block-relative row at which the text changes
        float cval = NEG;                                               // column-direction state (lane = column of the wave block),
        int carg = 0;                                                   // carried over the four 32-row blocks
        const int rblk = gr0 >> 7;
        const bool col_ok = gc0 + lane < f.N;
        uint32_t* const cdst = f.colpart + ((long)rblk * F5_SLOTS) * f.N + gc0 + lane;

        unsigned char* const wr = scratch + r31 * 128 + 8 * h;          // + chunk position * 16   (pack_lines_t's layout)
        // the four row groups' token-mask bytes of this lane's rows, requested before the row direction's arithmetic (one dependent
        // global round trip per group sat in front of every column scan otherwise)
        // (read through a clamped index, unconditionally: behind `g < f.M &&` the compiler made each of the four a branch with its own
        //  load and full drain of the memory counter -- four serialized round trips at the head of every tile's epilogue)
        unsigned char mbyte[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g = gr0 + i * 32 + r31;
            mbyte[i] = f.mask[g < f.M ? g : f.M - 1];
        }
        bool rowlive[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) rowlive[i] = (gr0 + i * 32 + r31 < f.M) && mbyte[i] != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int grow = gr0 + i * 32 + r31;
            // ================= row direction =================
            if (cend > 0) {
                float v0 = NEG, v1 = NEG, v2 = NEG;                     // (max, arg) over the lane's columns of image yL, yL + 1, yL + 2
                int a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int g0 = j * 32 + 8 * q;                  // the 8-column group both half-waves' quads lie in
                        if (g0 + 8 <= cb1) {                            // (uniform) wholly inside the first image
#pragma unroll
                            for (int k = 0; k < 4; ++k) take(acc[i][j][4 * q + k], g0 + k, v0, a0);
                        } else if (g0 >= cb1 && g0 + 8 <= cb2) {        // (uniform) wholly inside the second
#pragma unroll
                            for (int k = 0; k < 4; ++k) take(acc[i][j][4 * q + k], g0 + k, v1, a1);
                        } else if (g0 >= cb2 && g0 + 8 <= cend) {       // (uniform) wholly inside the third, all real columns
#pragma unroll
                            for (int k = 0; k < 4; ++k) take(acc[i][j][4 * q + k], g0 + k, v2, a2);
                        } else if (g0 < cend) {                         // a group that a boundary (or the operand's end) cuts
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int c = g0 + k + 4 * h;
                                const float s = acc[i][j][4 * q + k];
                                if (c < cb1) take(s, g0 + k, v0, a0);
                                else if (c < cb2) take(s, g0 + k, v1, a1);
                                else if (c < cend) take(s, g0 + k, v2, a2);
                            }
                        }
                    }
                a0 += 4 * h; a1 += 4 * h; a2 += 4 * h;                  // (the lane's constant column offset)
                // partner lane: the same row, the interleaved other 32 columns; smaller column wins ties
                const float p0 = shfl_xor(v0, 32), p1 = shfl_xor(v1, 32), p2 = shfl_xor(v2, 32);
                const int q0 = shfl_xor(a0, 32), q1 = shfl_xor(a1, 32), q2 = shfl_xor(a2, 32);
                if (p0 > v0 || (p0 == v0 && q0 < a0)) { v0 = p0; a0 = q0; }
                if (p1 > v1 || (p1 == v1 && q1 < a1)) { v1 = p1; a1 = q1; }
                if (p2 > v2 || (p2 == v2 && q2 < a2)) { v2 = p2; a2 = q2; }
                if (grow < f.M) {
                    // lane h = 0 leaves the entries of the first and third image, h = 1 that of the second: arg = token index inside the image
                    uint32_t* const dst = f.rowpart + ((long)grow * f.nblk64 + (gc0 >> 6)) * F5_SIDES;
                    if (h == 0) {
                        dst[0] = (v0 > 0.5f * NEG) ? f5_entry(v0, gc0 + a0 - yL * f.ni) : F5_EMPTY;
                        dst[2] = (v2 > 0.5f * NEG) ? f5_entry(v2, gc0 + a2 - (yL + 2) * f.ni) : F5_EMPTY;
                    } else {
                        dst[1] = (v1 > 0.5f * NEG) ? f5_entry(v1, gc0 + a1 - (yL + 1) * f.ni) : F5_EMPTY;
                    }
                }
            }
            // ================= column direction =================
            // the 32 x 64 block as bf16 lines in the wave's scratch: row r31, 16-byte chunk (4 j + q) ^ (r31 & 7), 8 bytes at 8 h
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float* a = reinterpret_cast<const float*>(&acc[i][j]) + 4 * q;
                    const u32x2 v = {f2bf_pk(a[0], a[1]), f2bf_pk(a[2], a[3])};
                    *reinterpret_cast<u32x2*>(wr + (((4 * j + q) ^ (r31 & 7)) << 4)) = v;
                }
            const uint32_t live = wave_ballot32(rowlive[i]);            // bit r: row r of the group is a real token (uniform)
            lds_fence();
            const unsigned char* const col = scratch + (lane & 7) * 2;  // + r * 128 + ((chunk ^ (r & 7)) << 4): column `lane`, chunk lane >> 3
            const int chunk = lane >> 3;
            auto flush = [&]() {
                // (a text whose rows inside this block all lie past M does not exist: the last row tile's padding)
                const int first = next_change - f.nt > 0 ? next_change - f.nt : 0;
                if (col_ok && gr0 + first < f.M) cdst[(long)slot * f.N] = (cval > 0.5f * NEG) ? f5_entry(cval, carg) : F5_EMPTY;
                cval = NEG; carg = 0;
                ++slot;
                next_change += f.nt;
            };
            if (next_change == i * 32) flush();                         // (uniform) the text changes at the group's first row
            if (live == 0xffffffffu && (next_change <= i * 32 || next_change >= i * 32 + 32)) {
                // the common group: 32 real tokens of one text.  All 32 reads are issued before the first compare (one dependent
                // LDS round trip per row was the whole cost of this direction: ~100 cycles x 128 rows per tile)
                const int t0 = gr0 + i * 32 - (x0 + slot) * f.nt;       // token index of the group's first row inside its text
                bf16_t raw[32];
#pragma unroll
                for (int r = 0; r < 32; ++r) raw[r] = *reinterpret_cast<const bf16_t*>(col + r * 128 + ((chunk ^ (r & 7)) << 4));
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    const float s = bf2f(raw[r]);
                    if (s > cval) { cval = s; carg = t0 + r; }
                }
            } else {
#pragma unroll 4
                for (int r = 0; r < 32; ++r) {
                    const int rb = i * 32 + r;                          // block-relative row
                    if (r != 0 && rb == next_change) flush();           // (uniform) the text changes here: leave the finished entry
                    if ((live >> r) & 1u) {                             // (uniform) padding tokens take no part
                        const float s = bf2f(*reinterpret_cast<const bf16_t*>(col + r * 128 + ((chunk ^ (r & 7)) << 4)));
                        const int t = gr0 + rb - (x0 + slot) * f.nt;    // token index inside the text
                        if (s > cval) { cval = s; carg = t; }
                    }
                }
            }
            lds_fence();                                                // (the next 32 rows overwrite the slice)
        }
        // the last text of the block (only if a row of it exists)
        {
            const int first = next_change - f.nt > 0 ? next_change - f.nt : 0;
            if (slot < F5_SLOTS && gr0 + first < f.M && col_ok) cdst[(long)slot * f.N] = (cval > 0.5f * NEG) ? f5_entry(cval, carg) : F5_EMPTY;
        }
        return 0;
    }
};

XC_DEV Gemm2Params filip5_gemm_params(const bf16_t* X, const bf16_t* Y, int M, int N, int d) {
    Gemm2Params g;
    g.A = X; g.B = Y; g.C = nullptr;
    g.lda = d; g.ldb = d; g.ldc = 0;
    g.M = M; g.N = N; g.K = d; g.alpha = 1.f;
    g.bias = nullptr; g.residual = nullptr; g.ldr = 0; g.addrows = nullptr; g.rowidx = nullptr; g.ld_add = 0;
    g.partial = nullptr; g.k_per_split = d;
    g.tiles_m = (M + G2_BM - 1) / G2_BM; g.tiles_n = (N + G2_BN - 1) / G2_BN;
    g.stream_out = 0; g.band_n = 0;
    return g;
}

// X [M = bx * nt, d] text-token latents, Y [N = by * ni (rows past it readable as zero padding up to the chunk), d] image-token latents
__global__ __launch_bounds__(G2_THREADS, 2) void filip5_kernel(const bf16_t* X, const bf16_t* Y, int d, Filip5Params f) {
    XC_LDS_DYNAMIC(lds);
    const Gemm2Params g = filip5_gemm_params(X, Y, f.M, f.N, d);
    g5_run<false, false, Filip5Epilogue>(g, lds, Filip5Epilogue{f});
}

// ---- merges ----------------------------------------------------------------------------------------------------------------------
// t2i[x, y0 + y] = temp * sum_t w[x, t] max_k s / max(cnt[x], 1e-6),  kmax[x, t, y0 + y] = arg max_k:  one work-group per (text x,
// 256 images); a thread owns an image and walks the text's tokens, combining the wave blocks the image's columns span
__global__ __launch_bounds__(256) void filip5_merge_rows_kernel(const uint32_t* __restrict__ rowpart, const unsigned char* __restrict__ mask,
                                                                const float* __restrict__ log_temp, float* __restrict__ t2i, long ldo,
                                                                short* __restrict__ kmax, float* __restrict__ cnt, int nt, int ni, int by,
                                                                int nblk64, int y0, int ytotal) {
    const int x = blockIdx.x;
    const int y = blockIdx.y * 256 + threadIdx.x;
    const float temp = expf(*log_temp);
    float wsum = 0.f, acc = 0.f;
    const int c_lo = y * ni, c_hi = (y + 1) * ni - 1;
    const int b_lo = c_lo >> 6, b_hi = c_hi >> 6;
    for (int t = 0; t < nt; ++t) {
        const long row = (long)x * nt + t;
        const bool w = mask[row] != 0;
        if (w) wsum += 1.f;
        if (y < by) {
            float best = -3.0e38f;
            int bk = 0;
            for (int b = b_lo; b <= b_hi; ++b) {
                const int side = y - (b << 6) / ni;               // images are numbered from the one the block's first column lies in
                const uint32_t e = rowpart[(row * nblk64 + b) * F5_SIDES + side];
                const float v = u2f(e & 0xffff0000u);
                if (v > best) { best = v; bk = (int)(e & 0xffffu); }
            }
            kmax[row * ytotal + y0 + y] = (short)bk;
            if (w) acc += best;
        }
    }
    if (y < by) t2i[(long)x * ldo + y0 + y] = temp * acc / fmaxf(wsum, 1e-6f);
    if (y == 0 && y0 == 0) cnt[x] = wsum;
}

// i2t[x, y0 + y] = temp * mean_k max_{t live} s,  tmax[x, y0 + y, k] = arg max_t:  one work-group per (text x, group of whole images
// of at most 256 columns); a thread owns a column and combines the <= 2 row blocks the text's rows span; the images' means are
// formed in LDS
__global__ __launch_bounds__(256) void filip5_merge_cols_kernel(const uint32_t* __restrict__ colpart, const float* __restrict__ log_temp,
                                                                float* __restrict__ i2t, long ldo, short* __restrict__ tmax, int nt, int ni,
                                                                int by, int N, int ypb, int y0, int ytotal) {
    XC_LDS_DYNAMIC(lds);
    float* vals = reinterpret_cast<float*>(lds);               // [ypb * ni]
    const int x = blockIdx.x;
    const int yfirst = blockIdx.y * ypb;
    const float temp = expf(*log_temp);
    const int r_lo = x * nt, r_hi = (x + 1) * nt - 1;
    const int rb_lo = r_lo >> 7, rb_hi = r_hi >> 7;
    for (int c = threadIdx.x; c < ypb * ni; c += 256) {
        const int y = yfirst + c / ni;
        float best = -3.0e38f;
        int bt = 0;
        if (y < by) {
            const int col = yfirst * ni + c;
            for (int rb = rb_lo; rb <= rb_hi; ++rb) {
                const int slot = x - (rb << 7) / nt;              // texts are numbered from the block's first row's text
                const uint32_t e = colpart[((long)rb * F5_SLOTS + slot) * N + col];
                const float v = u2f(e & 0xffff0000u);
                if (v > best) { best = v; bt = (int)(e & 0xffffu); }
            }
            tmax[((long)x * ytotal + y0) * ni + col] = (short)bt;
        }
        vals[c] = best;
    }
    sync();
    for (int yy = threadIdx.x; yy < ypb; yy += 256) {
        if (yfirst + yy < by) {
            float s = 0.f;
            for (int k = 0; k < ni; ++k) s += vals[yy * ni + k];
            i2t[(long)x * ldo + y0 + yfirst + yy] = temp * s / (float)ni;
        }
    }
}

}  // namespace xc
