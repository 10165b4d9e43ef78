// gemm9.h -- the FF2 input gradient with the GEGLU-LayerNorm backward in its epilogue: dh = dOut W2 never reaches memory.
//
// Reference chain (x_clip.py:180-199, FeedForward = Linear -> GEGLU -> LayerNorm -> Dropout -> Linear): h = LayerNorm(a) gamma with
// a = u gelu(t), (u | t) = FF1's output; y = h W2^T.  The unfused backward (rows.h ln_geglu_bwd_kernel) reads dh (bf16), u | t and writes
// d(u | t): 5 row-widths of the widest tensor of the model per token, a quarter of the step's LayerNorm-family bytes.  Its two row
// statistics do not need dh:
//     s1 = sum_j dh_j gamma_j       = dOut . (W2 gamma)                     (a weight-only vector, ffn_wgamma_kernel)
//     s2 = sum_j dh_j gamma_j ahat_j = sum_j dh_j h_j = dOut . (h W2^T) = dOut . y,   y = x2 - x1  (the block's output minus its input)
// so they are computed from [rows, dim] tensors BEFORE the product (ffn_rowstats_kernel), and the product's epilogue turns each accumulator
// straight into the two gradients:  da = rstd (dh gamma - s1 / F - ahat s2 / F);  du = da gelu(t);  dt = da u gelu'(t).  What is saved:
// the write and the read of dh (2 of the 5 row-widths).  What it costs: ~30 vector instructions per element in a GEMM epilogue, where
// the matrix cores idle meanwhile -- measured in profiles/r05_*_ln_fusion_ab.log.
//
// Layout of the epilogue (per wave 128 x 64 of the 256 x 256 tile, accumulators in gemm4.h's transposed form: a lane owns a row): u | t
// come in and the gradients go out as whole 128-byte lines through the wave's 4 KiB LDS slice (as gemm4.h store_full_res_lds does for a
// skip term); the per-column products dh ahat (the LayerNorm gain's gradient) are summed over the tile's rows in registers, across the
// 32 lanes of a half-wave at the end of the tile, and written to a [2 tiles_m, F] fp32 slab (row = row tile x wave row: no atomics;
// rows.h colsum_fold_kernel sums it).  Interior tiles only: the host sends other shapes down the unfused path.
#pragma once
#include "gemm4.h"

namespace xc {

struct GegluBwdArgs {
    const bf16_t* x; long ldx;        // [M, 2 F]: value | gate, FF1's output
    bf16_t* dx; long lddx;            // [M, 2 F]: its gradient
    const bf16_t* gamma;              // [F] LayerNorm gain
    const float* rowc;                // [M, 4] per row {rstd, -mean rstd, s1 / F rstd, s2 / F rstd} (ffn_rowstats_kernel; mean / rstd = the
                                      // forward LayerNorm's statistics over a = u gelu(t))
    float* dg_partial;                // [2 tiles_m, F] per (row tile, wave row) column sums of dh ahat
    int F;
};

struct G4GegluBwdEpilogue {
    const Gemm2Params& p;
    const GegluBwdArgs& e;
    static constexpr bool DEFER_FRAGS = true;                  // (the epilogue needs the registers of the next tile's first fragments)
    XC_DEV void finish() const {}
    XC_DEV bool packs_lines(int, int) const { return false; }
    XC_DEV void pack_lines(f32x16 (&)[4][2], unsigned char*, u32x4 (&)[4][4], int = 0, int = 0) const {}
    template <bool NT = false> XC_DEV void store_lines(const u32x4 (&)[4][4], int, int) const {}
    template <bool NT, int NG> XC_DEV void store_line_groups(const u32x4 (&)[NG][4], int, int, int) const {}

    XC_DEV int with_scratch(f32x16 (&acc)[4][2], int m0, int n0, unsigned char* scratch) const {
        const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
        const int wave = uniform(threadIdx.x >> 6), wm = wave >> 2, wn = wave & 3;
        // whole-line descriptors of the tile's u, t, du, dt blocks
        const BufRsrc ru = make_rsrc(e.x + (long)m0 * e.ldx + n0, 255u * (uint32_t)e.ldx * 2u + 512u);
        const BufRsrc rt = make_rsrc(e.x + (long)m0 * e.ldx + e.F + n0, 255u * (uint32_t)e.ldx * 2u + 512u);
        const BufRsrc rdu = make_rsrc(e.dx + (long)m0 * e.lddx + n0, 255u * (uint32_t)e.lddx * 2u + 512u);
        const BufRsrc rdt = make_rsrc(e.dx + (long)m0 * e.lddx + e.F + n0, 255u * (uint32_t)e.lddx * 2u + 512u);
        const uint32_t vx = ((uint32_t)(wm * 128 + (lane >> 3)) * (uint32_t)e.ldx + (uint32_t)(wn * 64 + 8 * (lane & 7))) * 2u;
        const uint32_t vd = ((uint32_t)(wm * 128 + (lane >> 3)) * (uint32_t)e.lddx + (uint32_t)(wn * 64 + 8 * (lane & 7))) * 2u;
        const uint32_t x8 = (uint32_t)e.ldx * 16u, d8 = (uint32_t)e.lddx * 16u;        // 8 rows
        unsigned char* const quad = scratch + r * 128 + 8 * h;                        // accumulator layout: + chunk position * 16
        unsigned char* const line = scratch + (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);   // line layout: + 1024 per 8 rows
        float dg[2][16];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < 16; ++c) dg[j][c] = 0.f;

        // Register budget (128 accumulators + 32 column sums live throughout): per 32-row group the gate quads are held (16 registers), the
        // value quads are read from the LDS slice one at a time and overwritten IN PLACE by du (a lane's own 8 bytes), dt replaces the
        // gate quads; the next group's lines are requested once this group's have been consumed.
        u32x4 ul[4], tl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { tl[k] = buf_ld16_nt<0>(rt, vx, x8 * (uint32_t)k); ul[k] = buf_ld16_nt<0>(ru, vx, x8 * (uint32_t)k); }
        // a row's four constants {rstd, -mean rstd, s1 / F rstd, s2 / F rstd} (ffn_rowstats_kernel) as one 16-byte load, one group ahead
        const float* const rs = e.rowc + ((long)m0 + wm * 128 + r) * 4;
        u32x4 rc_next = ld16(rs);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4 rc = rc_next;
            if (i < 3) rc_next = ld16(rs + (long)(i + 1) * 32 * 4);
            const float rstd = u2f(rc[0]), shift = u2f(rc[1]), k1 = u2f(rc[2]), k2 = u2f(rc[3]);
            u32x2 tq[2][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(line + k * 1024) = tl[k];
            lds_fence();
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) tq[j][q] = *reinterpret_cast<const u32x2*>(quad + (((4 * j + q) ^ (r & 7)) << 4));
            lds_fence();
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(line + k * 1024) = ul[k];
            lds_fence();
            // this lane's 32 columns of gamma, requested BEFORE the next group's lines: the memory counter retires in order, so a load issued
            // behind the prefetch could only be used once the prefetch had landed -- every group waited for its successor's lines (the first
            // version).  (Once per tile, in front of everything, would also keep it ahead of the previous group's stores: 82 spilled registers.)
            u32x2 gq[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) gq[j][q] = *reinterpret_cast<const u32x2*>(e.gamma + n0 + wn * 64 + 32 * j + 8 * q + 4 * h);
            if (i < 3) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    tl[k] = buf_ld16_nt<0>(rt, vx, x8 * (uint32_t)(4 * (i + 1) + k));
                    ul[k] = buf_ld16_nt<0>(ru, vx, x8 * (uint32_t)(4 * (i + 1) + k));
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned char* const at = quad + (((4 * j + q) ^ (r & 7)) << 4);
                    const u32x2 uv = *reinterpret_cast<const u32x2*>(at);
                    const u32x2 gv = gq[j][q];
                    const float uu[4] = {u2f(uv[0] << 16), u2f(uv[0] & 0xffff0000u), u2f(uv[1] << 16), u2f(uv[1] & 0xffff0000u)};
                    const float tt[4] = {u2f(tq[j][q][0] << 16), u2f(tq[j][q][0] & 0xffff0000u), u2f(tq[j][q][1] << 16), u2f(tq[j][q][1] & 0xffff0000u)};
                    const float gg[4] = {u2f(gv[0] << 16), u2f(gv[0] & 0xffff0000u), u2f(gv[1] << 16), u2f(gv[1] & 0xffff0000u)};
                    float du[4], dt[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float cdf, pdf;
                        gelu_parts(tt[c], cdf, pdf);
                        const float ge = tt[c] * cdf;                                  // gelu(t)
                        const float udge = uu[c] * (cdf + tt[c] * pdf);                // u gelu'(t)
                        const float ah = uu[c] * ge * rstd + shift;                    // normalised a
                        const float dh = acc[i][j][4 * q + c];
                        dg[j][4 * q + c] += dh * ah;
                        const float da = dh * gg[c] * rstd - k1 - ah * k2;
                        du[c] = da * ge;
                        dt[c] = da * udge;
                    }
                    *reinterpret_cast<u32x2*>(at) = u32x2{f2bf_pk(du[0], du[1]), f2bf_pk(du[2], du[3])};     // (this lane's own 8 bytes: read, then overwritten)
                    tq[j][q] = u32x2{f2bf_pk(dt[0], dt[1]), f2bf_pk(dt[2], dt[3])};
                }
            }
            lds_fence();
            // du: lines -> memory; then dt: quads -> lines -> memory
            u32x4 o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = *reinterpret_cast<const u32x4*>(line + k * 1024);
            lds_fence();
#pragma unroll
            for (int k = 0; k < 4; ++k) buf_st16_nt<0>(rdu, vd, d8 * (uint32_t)(4 * i + k), o[k]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x2*>(quad + (((4 * j + q) ^ (r & 7)) << 4)) = tq[j][q];
            lds_fence();
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = *reinterpret_cast<const u32x4*>(line + k * 1024);
            lds_fence();
#pragma unroll
            for (int k = 0; k < 4; ++k) buf_st16_nt<0>(rdt, vd, d8 * (uint32_t)(4 * i + k), o[k]);
        }
        // the gain gradient's partial sums: over the 32 rows (lanes) of each half-wave, then one slab row per (row tile, wave row)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float v = dg[j][c];
                v += shfl_xor(v, 16); v += shfl_xor(v, 8); v += shfl_xor(v, 4); v += shfl_xor(v, 2); v += shfl_xor(v, 1);
                dg[j][c] = v;
            }
        if (r == 0) {
            float* out = e.dg_partial + ((long)(m0 / G2_BM) * 2 + wm) * e.F + n0 + wn * 64 + 4 * h;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x4 v = {f2u(dg[j][4 * q]), f2u(dg[j][4 * q + 1]), f2u(dg[j][4 * q + 2]), f2u(dg[j][4 * q + 3])};
                    st16(out + 32 * j + 8 * q, v);
                }
        }
        return 0;                                              // (loads and stores mixed: the next wait drains them)
    }
};

__global__ __launch_bounds__(G2_THREADS, 2) void gemm9_geglu_bwd_kernel(Gemm2Params p, GegluBwdArgs e) {
    XC_LDS_DYNAMIC(lds);
    g5_run<false, true, G4GegluBwdEpilogue>(p, lds, G4GegluBwdEpilogue{p, e});
}

// wg[c] = sum_j W[c, j] gamma[j] (fp32): the weight-only vector of s1.  One wave per row of W.
__global__ __launch_bounds__(256) void ffn_wgamma_kernel(const bf16_t* __restrict__ W, long ldw, const bf16_t* __restrict__ gamma, float* __restrict__ wg,
                                                         int rows, int F) {
    const int lane = lane_id();
    const long row = (long)blockIdx.x * 4 + wave_id();
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane * 8; c < F; c += 64 * 8) {
        float a[8], b[8];
        load_vec<bf16_t>(W + row * ldw + c, a);
        load_vec<bf16_t>(gamma + c, b);
#pragma unroll
        for (int k = 0; k < 8; ++k) s += a[k] * b[k];
    }
    s = wave_sum(s);
    if (lane == 0) wg[row] = s;
}

// the four per-row constants of the fused epilogue: with s1 = dOut[r, :] . wg and s2 = dOut[r, :] . (x2[r, :] - x1[r, :]),
// rowc[r] = {rstd, -mean rstd, s1 / F rstd, s2 / F rstd}.  One wave per row (D <= 4096 features)
__global__ __launch_bounds__(256) void ffn_rowstats_kernel(const bf16_t* __restrict__ dout, long ldd, const bf16_t* __restrict__ x2, long ld2,
                                                           const bf16_t* __restrict__ x1, long ld1, const float* __restrict__ wg,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           float* __restrict__ rowc, int rows, int D, float invF) {
    const int lane = lane_id();
    const long row = (long)blockIdx.x * 4 + wave_id();
    if (row >= rows) return;
    float a1 = 0.f, a2 = 0.f;
    for (int c = lane * 8; c < D; c += 64 * 8) {
        float d[8], p[8], q[8];
        load_vec<bf16_t>(dout + row * ldd + c, d);
        load_vec<bf16_t>(x2 + row * ld2 + c, p);
        load_vec<bf16_t>(x1 + row * ld1 + c, q);
        const u32x4 w0 = ld16(wg + c), w1 = ld16(wg + c + 4);
        const float w[8] = {u2f(w0[0]), u2f(w0[1]), u2f(w0[2]), u2f(w0[3]), u2f(w1[0]), u2f(w1[1]), u2f(w1[2]), u2f(w1[3])};
#pragma unroll
        for (int k = 0; k < 8; ++k) { a1 += d[k] * w[k]; a2 += d[k] * (p[k] - q[k]); }
    }
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
    if (lane == 0) {
        const float rs = rstd[row];
        const u32x4 v = {f2u(rs), f2u(-mean[row] * rs), f2u(a1 * invF * rs), f2u(a2 * invF * rs)};
        st16(rowc + row * 4, v);
    }
}

}  // namespace xc
