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

// The epilogue below is bound by the vector ALU, not by memory: ~45 instruction issues per element (gelu_parts: 16 + a reciprocal + an
// exponential at quarter rate; the LayerNorm / GEGLU chain rule: ~20) x 65,536 elements per tile over a CU's 64 lanes per clock = 23 us of
// the tile's 25 us of epilogue, with the tile's 512 KB of lines moving underneath (profiles/r05_t_*: leaving the line stores in flight
// changed nothing).  So the arithmetic runs on PAIRS of elements -- a lane's two neighbours of a bf16 dword -- in ext_vector float2, which
// hipcc selects as v_pk_mul / v_pk_add / v_pk_fma_f32 (two fp32 operations per issue on gfx950); only the reciprocal, the exponential, the
// absolute value and the sign transfer stay per element.
typedef float f32x2 __attribute__((ext_vector_type(2)));
XC_DEV f32x2 splat2(float v) { return f32x2{v, v}; }
XC_DEV f32x2 bf16x2_unpack(uint32_t w) { return f32x2{u2f(w << 16), u2f(w & 0xffff0000u)}; }
// common.h's gelu_parts on a pair (the same Abramowitz-Stegun 7.1.26 polynomial, its coefficients halved so that
// Phi(x) = 0.5 + sign(x) (0.5 - (0.5 poly) exp(-x^2 / 2)) comes out of one more fma)
XC_DEV void gelu_parts2(f32x2 x, f32x2& cdf, f32x2& pdf) {
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2 den = ax * splat2(0.3275911f * 0.70710678118654752f) + splat2(1.0f);
    const f32x2 t = {fast_rcp(den[0]), fast_rcp(den[1])};
    const f32x2 ea = (x * x) * splat2(-0.5f * 1.4426950408889634f);
    const f32x2 e = {fast_exp2(ea[0]), fast_exp2(ea[1])};                             // exp(-x^2 / 2)
    f32x2 hp = t * splat2(0.5f * 1.061405429f) + splat2(0.5f * -1.453152027f);
    hp = hp * t + splat2(0.5f * 1.421413741f);
    hp = hp * t + splat2(0.5f * -0.284496736f);
    hp = hp * t + splat2(0.5f * 0.254829592f);
    hp = hp * t;
    const f32x2 he = splat2(0.5f) - hp * e;                                           // 0.5 erf(|x| / sqrt 2)
    cdf = splat2(0.5f) + f32x2{copysignf(he[0], x[0]), copysignf(he[1], x[1])};
    pdf = e * splat2(0.39894228040143268f);
}

// ABL (measurement build only, XCLIP_GEMM9_ABL; results are garbage except for 1): 1 the line stores as the untracked asm form (what the first
// version did), 2 no GELU arithmetic (constants for cdf / pdf), 4 no line loads behind the first group's, 8 no line stores
// WN: waves along N of the work-group's tile (4: g5_run's 2 x 4 waves on 256 x 256; 2: two waves along N (a 256 x 128 tile; round 5 experiment, removed))
template <int ABL = 0, int WN = 4>
struct G4GegluBwdEpilogue {
    const Gemm2Params& p;
    const GegluBwdArgs& e;
    static constexpr bool DEFER_FRAGS = true;                  // (the epilogue needs the registers of the next tile's first fragments)
    XC_DEV void finish() const {}
    XC_DEV bool packs_lines(int, int) const { return false; }
    XC_DEV void pack_lines(f32x16 (&)[4][2], unsigned char*, u32x4 (&)[4][4], int = 0, int = 0) const {}
    template <bool NT = false> XC_DEV void store_lines(const u32x4 (&)[4][4], int, int) const {}
    template <bool NT, int NG> XC_DEV void store_line_groups(const u32x4 (&)[NG][4], int, int, int) const {}

    // A line store the compiler's wait-count pass can see (xc_device.h buf_st16_nt_tracked): the epilogue's loads are compiler-tracked, and
    // behind untracked asm stores every use of the prefetched lines waited for the previous group's eight stores to be acknowledged --
    // one store round trip per 32-row group.  The address is recomputed per store (opaque: not hoisted into 16 registers the epilogue
    // does not have).
    XC_DEV void store_line(BufRsrc r, uint32_t voff, uint32_t soff, u32x4 v) const {
        if (ABL & 8) return;
        if (ABL & 1) { buf_st16_nt<0>(r, voff, soff, v); return; }
        buf_st16_nt_tracked(r, opaque(voff) + soff, v);
    }

    XC_DEV int with_scratch(f32x16 (&acc)[4][2], int m0, int n0, unsigned char* scratch) const {
        const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
        const int wave = uniform(threadIdx.x >> 6), wm = wave >> (WN == 4 ? 2 : 1), wn = wave & (WN - 1);
        // whole-line descriptors of the tile's u, t, du, dt blocks
        const BufRsrc ru = make_rsrc(e.x + (long)m0 * e.ldx + n0, 255u * (uint32_t)e.ldx * 2u + 512u);
        const BufRsrc rt = make_rsrc(e.x + (long)m0 * e.ldx + e.F + n0, 255u * (uint32_t)e.ldx * 2u + 512u);
        const BufRsrc rdu = make_rsrc(e.dx + (long)m0 * e.lddx + n0, 255u * (uint32_t)e.lddx * 2u + 512u);
        const BufRsrc rdt = make_rsrc(e.dx + (long)m0 * e.lddx + e.F + n0, 255u * (uint32_t)e.lddx * 2u + 512u);
        const uint32_t vx = ((uint32_t)(wm * 128 + (lane >> 3)) * (uint32_t)e.ldx + (uint32_t)(wn * 64 + 8 * (lane & 7))) * 2u;
        const uint32_t vd = ((uint32_t)(wm * 128 + (lane >> 3)) * (uint32_t)e.lddx + (uint32_t)(wn * 64 + 8 * (lane & 7))) * 2u;
        const uint32_t x8 = (uint32_t)e.ldx * 16u, d8 = (uint32_t)e.lddx * 16u;        // 8 rows
        // (round 6, measured and not kept: the two 8-byte halves of a chunk swapped in rows whose bit 3 is set -- the quad accesses of rows r
        //  and r + 8 then use different bank pairs, 4-way -> 2-way conflicts (22.7 % of the kernel's LDS cycles, profiles/r05_w_sq_gemm9.txt):
        //  1280 - 1296 us against 1266 - 1284 without, profiles/r06_k_gemm9_exchange_half_swap_ab.log: the epilogue is bound by its vector ALU)
        unsigned char* const quad = scratch + r * 128 + 8 * h;                        // accumulator layout: + chunk position * 16
        unsigned char* const line = scratch + (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);   // line layout: + 1024 per 8 rows
        f32x2 dg[2][8];                                        // column sums of dh ahat: pairs (2 c, 2 c + 1) of the lane's 16 columns per j
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < 8; ++c) dg[j][c] = splat2(0.f);

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
            if (i < 3 && !(ABL & 4)) {
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
                    const f32x2 rstd2 = splat2(rstd), shift2 = splat2(shift), k12 = splat2(k1), k22 = splat2(k2);
                    u32x2 duw, dtw;
#pragma unroll
                    for (int cp = 0; cp < 2; ++cp) {                                  // columns (2 cp, 2 cp + 1) of the quad: one bf16 dword
                        const f32x2 uu = bf16x2_unpack(uv[cp]), tt = bf16x2_unpack(tq[j][q][cp]), gg = bf16x2_unpack(gv[cp]);
                        f32x2 cdf, pdf;
                        if (ABL & 2) { cdf = splat2(0.5f); pdf = splat2(0.4f); } else gelu_parts2(tt, cdf, pdf);
                        const f32x2 ge = tt * cdf;                                     // gelu(t)
                        const f32x2 udge = uu * (cdf + tt * pdf);                      // u gelu'(t)
                        const f32x2 ah = (uu * ge) * rstd2 + shift2;                   // normalised a
                        const f32x2 dh = {acc[i][j][4 * q + 2 * cp], acc[i][j][4 * q + 2 * cp + 1]};
                        dg[j][2 * q + cp] += dh * ah;
                        const f32x2 da = (dh * gg) * rstd2 - k12 - ah * k22;
                        const f32x2 du = da * ge, dt = da * udge;
                        duw[cp] = f2bf_pk(du[0], du[1]);
                        dtw[cp] = f2bf_pk(dt[0], dt[1]);
                    }
                    *reinterpret_cast<u32x2*>(at) = duw;                              // (this lane's own 8 bytes: read, then overwritten)
                    tq[j][q] = dtw;
                }
            }
            lds_fence();
            // du: lines -> memory; then dt: quads -> lines -> memory
            u32x4 o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = *reinterpret_cast<const u32x4*>(line + k * 1024);
            lds_fence();
#pragma unroll
            for (int k = 0; k < 4; ++k) store_line(rdu, vd, d8 * (uint32_t)(4 * i + k), o[k]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x2*>(quad + (((4 * j + q) ^ (r & 7)) << 4)) = tq[j][q];
            lds_fence();
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = *reinterpret_cast<const u32x4*>(line + k * 1024);
            lds_fence();
#pragma unroll
            for (int k = 0; k < 4; ++k) store_line(rdt, vd, d8 * (uint32_t)(4 * i + k), o[k]);
        }
        // the gain gradient's partial sums: over the 32 rows (lanes) of each half-wave, then one slab row per (row tile, wave row)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float v = dg[j][c >> 1][c & 1];
                v += shfl_xor(v, 16); v += shfl_xor(v, 8); v += shfl_xor(v, 4); v += shfl_xor(v, 2); v += shfl_xor(v, 1);
                dg[j][c >> 1][c & 1] = v;
            }
        if (r == 0) {
            float* out = e.dg_partial + ((long)(m0 / G2_BM) * 2 + wm) * e.F + n0 + wn * 64 + 4 * h;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x4 v = {f2u(dg[j][2 * q][0]), f2u(dg[j][2 * q][1]), f2u(dg[j][2 * q + 1][0]), f2u(dg[j][2 * q + 1][1])};
                    st16(out + 32 * j + 8 * q, v);
                }
        }
        // the last group's 4 + 4 line stores and the 8 stores of the column sums (issued by every wave: the lane mask only empties them) are
        // the wave's 16 youngest memory operations; the next tile's first counted wait may leave them in flight (g5_run: in_flight == 16)
        return (ABL & (1 | 8)) ? 0 : 16;
    }
};

template <int ABL = 0>
__global__ __launch_bounds__(G2_THREADS, 2) void gemm9_geglu_bwd_kernel(Gemm2Params p, GegluBwdArgs e) {
    XC_LDS_DYNAMIC(lds);
    g5_run<false, true, G4GegluBwdEpilogue<ABL>>(p, lds, G4GegluBwdEpilogue<ABL>{p, e});
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
