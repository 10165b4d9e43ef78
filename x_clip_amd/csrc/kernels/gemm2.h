// gemm2.h -- the bf16 production GEMM (every nn.Linear of the encoders: forward, dgrad, wgrad; reference
// x_clip.py:191-195,209-210,358): 256 x 256 output tile, 8 waves (2 along M x 4 along N, 128 x 64 per wave =
// 4 x 2 v_mfma_f32_32x32x16_bf16 accumulators), K step 64, operands staged by LDS DMA (global_load_lds_dwordx4: no
// VGPR round trip, no ds_write), two LDS stages of 64 KiB.
//
//   C[M,N] = alpha * op(A)[M,K] * op(B)[K,N]  (+ bias[n] + addrows[rowidx[m]][n] + residual[m,n])
//   normal operand : X[outer * ld + k]   (activations [tokens, features] as A; Linear weights [out, in] as B)
//   k-major operand: X[k * ld + outer]   (dgrad: W as B;  wgrad: dy as A and x as B, contraction over tokens)
//
// LDS images (per operand and stage, 32 KiB), chosen so that the DMA destination is lane-linear and both kinds of
// fragment read are bank-conflict free (the swizzle lives in the per-lane SOURCE address and in the read address,
// never in the DMA destination -- cdna_hip_programming.md rule 21):
//   normal : [256 rows][64 k] bf16, 128-byte rows; 16-byte chunk c of row r sits at slot c ^ ((r >> 1) & 7).  A
//            fragment (row r, 8 consecutive k) is one ds_read_b128; the 16 rows of a ds_read_b128 lane group land on
//            16 distinct 16-byte slots of the 256-byte bank row.
//   k-major: 4 panels of [64 k][64 outer] bf16 (128-byte rows); the two 64-byte halves of a row are swapped when
//            bit 1 of k is set.  A fragment (outer index i, 8 consecutive k) is two ds_read_b64_tr_b16 (hardware
//            transpose, 4 k each); the 4 rows x 64 bytes a half-wave reads cover the 256-byte bank row exactly once.
// The loop is the two-phase form: DMA of tile t+1 is issued before the MFMAs of tile t, one vmcnt(0) + barrier per
// K step.  The accumulators leave through LDS as fp32 in four 64-row passes so that every global store (and the
// bias / residual / row-gather reads) is a coalesced 16-byte access with one rounding to bf16.
//
// Requirements (checked by the host, which otherwise falls back to gemm.h): bf16, K % 64 == 0 per split,
// contiguous dims multiples of 8, 16-byte aligned bases.  M and N may be ragged: out-of-range rows are clamped on
// load and masked on store.
#pragma once
#include "gemm.h"

namespace xc {

constexpr int G2_BM = 256, G2_BN = 256, G2_BK = 64, G2_THREADS = 512;
constexpr int G2_OPER_BYTES = 256 * 64 * 2;                 // one operand tile
constexpr int G2_STAGE_BYTES = 2 * G2_OPER_BYTES;
constexpr int G2_LDC = 256 + 4;                             // fp32 epilogue staging stride (64 rows per pass)
constexpr int G2_LDS_BYTES = 2 * G2_STAGE_BYTES;            // 128 KiB (>= 64 * G2_LDC * 4)

// ---- DMA of one operand tile: 32 wave-instructions of 1 KiB, 4 per wave ---------------------------------------
template <bool KMAJOR>
XC_DEV void g2_stage(const bf16_t* X, long ld, int outer0, int nouter, int k0, unsigned char* tile, int wave, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int id = wave * 4 + q;                         // which 1 KiB piece of the 32 KiB image
        const bf16_t* src;
        if (!KMAJOR) {
            const int row = id * 8 + (lane >> 3);            // tile row (outer index)
            const int chunk = (lane & 7) ^ ((row >> 1) & 7); // logical 16-byte chunk stored at slot lane & 7
            int g = outer0 + row;
            g = g < nouter ? g : nouter - 1;                 // ragged edge: clamp (masked on store)
            src = X + (long)g * ld + k0 + chunk * 8;
        } else {
            const int panel = id >> 3;
            const int row = (id & 7) * 8 + (lane >> 3);      // k within the tile
            const int chunk = (lane & 7) ^ (((row >> 1) & 1) << 2);
            int g = outer0 + panel * 64 + chunk * 8;
            g = g < nouter ? g : nouter - 8;                 // nouter is a multiple of 8 (host check)
            src = X + (long)(k0 + row) * ld + g;
        }
        glds16(src, tile + id * 1024);
    }
}

// one 1 KiB piece (q = 0..3) of an operand tile: lets the caller spread the DMA issue between MFMAs
template <bool KMAJOR>
XC_DEV void g2_stage_piece(const bf16_t* X, long ld, int outer0, int nouter, int k0, unsigned char* tile, int wave, int lane, int q) {
    const int id = wave * 4 + q;
    const bf16_t* src;
    if (!KMAJOR) {
        const int row = id * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        int g = outer0 + row;
        g = g < nouter ? g : nouter - 1;
        src = X + (long)g * ld + k0 + chunk * 8;
    } else {
        const int panel = id >> 3;
        const int row = (id & 7) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (((row >> 1) & 1) << 2);
        int g = outer0 + panel * 64 + chunk * 8;
        g = g < nouter ? g : nouter - 8;
        src = X + (long)(k0 + row) * ld + g;
    }
    glds16(src, tile + id * 1024);
}

// ---- fragment reads ----------------------------------------------------------------------------------------------
// normal image: rows [r0, r0 + 32) of the tile, k-block kk (16 k): lane (i = lane & 31, h = lane >> 5) -> 8 k
XC_DEV u32x4 g2_frag_normal(const unsigned char* tile, int r0, int kk, int lane) {
    const int row = r0 + (lane & 31);
    const int chunk = (kk * 2 + (lane >> 5)) ^ ((row >> 1) & 7);
    return ld16(tile + row * 128 + chunk * 16);
}
// k-major image: outer columns [c0, c0 + 32) (c0 a multiple of 32), k-block kk
XC_DEV u32x4 g2_frag_kmajor(const unsigned char* tile, int c0, int kk, int lane) {
    const int g = lane >> 4, tt = lane & 15;
    const int col = c0 + 16 * (g & 1) + (tt & 3) * 4;
    const int krow = kk * 16 + 8 * (g >> 1) + (tt >> 2);
    const int panel = col >> 6, colp = col & 63;
    const int chunk = (colp >> 3) ^ (((krow >> 1) & 1) << 2);          // rows krow and krow + 4 share bit 1
    const unsigned char* p = tile + panel * 8192 + krow * 128 + chunk * 16 + (colp & 7) * 2;
    const s16x4 lo = lds_read_tr16(p);                                   // k = 8 * (lane >> 5) + 0..3
    const s16x4 hi = lds_read_tr16(p + 4 * 128);                         // k = 8 * (lane >> 5) + 4..7
    u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
    u32x4 f = {a[0], a[1], b[0], b[1]};
    return f;
}

struct Gemm2Params {
    const bf16_t* A; const bf16_t* B; bf16_t* C;
    long lda, ldb, ldc;
    int M, N, K;
    float alpha;
    const bf16_t* bias; const bf16_t* residual; long ldr;
    const bf16_t* addrows; const int* rowidx; long ld_add;
    float* partial;            // split-K slabs [splits][M][N] fp32, or null
    int k_per_split;
    int tiles_m, tiles_n;
    int band_n;                // ring kernel: N tiles per band of the tile order (0 = plain n-fastest order; gemm4.h g5_run)
    int stream_out;            // interior bf16 tiles leave with non-temporal stores (an output larger than the L2s: gemm4.h store_lines)
    int split_lin = 0;         // > 0: split-K launch on a 1-D grid, = the number of K slices (g2_where); 0: blockIdx.x = tile, blockIdx.y = slice
};

// Which (tile, K slice) a work-group of a split-K launch computes.  On the 2-D grid (tile, slice) the hardware's round-robin placement
// (linear id mod 8 = XCD) scatters the tiles of ONE slice -- the work-groups that read the same K range of both operands -- over all eight
// L2s: a weight-gradient launch (32 tiles x 8 slices: 16 + 2 operand panels per slice) found half of its panel reads in L2 where 72 % are
// shared.  The 1-D form lists the (slice, tile) pairs slice-major and gives every XCD one contiguous eighth of the list: work-group `lin`
// sits on XCD lin & 7 and is that XCD's (lin >> 3)-th pair.  With 8 | slices an XCD holds whole slices; otherwise (the QKV gradient's 21
// slices of 12 tiles) at most two XCDs share a slice.  The grid is 8 * ceil(pairs / 8) work-groups; the few without a pair leave at once.
XC_DEV void g2_pair(const Gemm2Params& p, int ntiles, int& tile, int& slice, bool& valid) {
    const int total = ntiles * p.split_lin, chunk = (total + 7) >> 3;
    const int lin = blockIdx.x, j = lin >> 3, w = (lin & 7) * chunk + j;
    valid = j < chunk && w < total;
    slice = w / ntiles;
    tile = w - slice * ntiles;
}
XC_DEV bool g2_where(const Gemm2Params& p, int ntiles, int& tile0, int& slice, int& stride) {
    if (p.split_lin > 0) {
        bool valid;
        g2_pair(p, ntiles, tile0, slice, valid);
        stride = ntiles;                                       // (one tile per work-group)
        return valid;
    }
    tile0 = blockIdx.x;
    slice = blockIdx.y;
    stride = gridDim.x;
    return (int)blockIdx.x < ntiles;
}
XC_DEV int g2_slice(const Gemm2Params& p, int ntiles) {
    if (p.split_lin > 0) {
        int tile, slice;
        bool valid;
        g2_pair(p, ntiles, tile, slice, valid);
        return slice;
    }
    return (int)blockIdx.y;
}

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(G2_THREADS, 2) void gemm2_kernel(Gemm2Params p) {
    XC_LDS_DYNAMIC(lds);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = uniform(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                 // 2 x 4 waves, 128 x 64 outputs each
    const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int m0 = (tile / p.tiles_n) * G2_BM, n0 = (tile % p.tiles_n) * G2_BN;
    const int kbeg = blockIdx.y * p.k_per_split;
    const int kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
    const int nt = (kend - kbeg) / G2_BK;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto stage = [&](int buf, int k0) {
        unsigned char* base = lds + buf * G2_STAGE_BYTES;
        g2_stage<A_KMAJOR>(p.A, p.lda, m0, p.M, k0, base, wave, lane);
        g2_stage<B_KMAJOR>(p.B, p.ldb, n0, p.N, k0, base + G2_OPER_BYTES, wave, lane);
    };

    if (nt > 0) stage(0, kbeg);
    wait_vmem();
    sync();
    for (int t = 0; t < nt; ++t) {
        const unsigned char* As = lds + (t & 1) * G2_STAGE_BYTES;
        const unsigned char* Bs = As + G2_OPER_BYTES;
        if (t + 1 < nt) stage((t + 1) & 1, kbeg + (t + 1) * G2_BK);          // in flight under this tile's MFMAs
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            u32x4 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = A_KMAJOR ? g2_frag_kmajor(As, wm * 128 + i * 32, kk, lane) : g2_frag_normal(As, wm * 128 + i * 32, kk, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                b[j] = B_KMAJOR ? g2_frag_kmajor(Bs, wn * 64 + j * 32, kk, lane) : g2_frag_normal(Bs, wn * 64 + j * 32, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mma_kblock(a[i], b[j], acc[i][j], (bf16_t*)nullptr);
        }
        wait_vmem();
        sync();
    }

    // ---- epilogue: four passes of 64 rows through LDS (fp32), coalesced 16-byte global accesses --------------------
    float* Cs = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        if (wm == (pass >> 1)) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = (pass & 1) * 2 + ii;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        Cs[(ii * 32 + mfma_row(r, lane)) * G2_LDC + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
            }
        }
        sync();
        const int mrow0 = m0 + pass * 64;
        for (int id = tid; id < 64 * 32; id += G2_THREADS) {
            const int row = id >> 5, col = (id & 31) * 8;
            const int gm = mrow0 + row, gn = n0 + col;
            if (gm < p.M && gn < p.N) {
                if (p.partial != nullptr) {
                    float* slab = p.partial + ((long)blockIdx.y * p.M + gm) * p.N + gn;
                    st16(slab, ld16(Cs + row * G2_LDC + col));
                    st16(slab + 4, ld16(Cs + row * G2_LDC + col + 4));
                } else {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = Cs[row * G2_LDC + col + k] * p.alpha;
                    if (p.bias != nullptr) {
                        float t[8];
                        load_vec<bf16_t>(p.bias + gn, t);
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] += t[k];
                    }
                    if (p.addrows != nullptr) {
                        float t[8];
                        load_vec<bf16_t>(p.addrows + (long)p.rowidx[gm] * p.ld_add + gn, t);
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] += t[k];
                    }
                    if (p.residual != nullptr) {
                        float t[8];
                        load_vec<bf16_t>(p.residual + (long)gm * p.ldr + gn, t);
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] += t[k];
                    }
                    store_vec<bf16_t>(p.C + (long)gm * p.ldc + gn, v);
                }
            }
        }
        sync();
    }
}

}  // namespace xc
