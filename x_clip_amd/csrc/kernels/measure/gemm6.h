// gemm6.h -- EXPERIMENTAL (XCLIP_GEMM=6): the short-K forward product with TWO co-resident work-groups per CU.
//
// The ring kernel of gemm4.h holds a CU with one 8-wave work-group (160 KiB of LDS): its two waves per SIMD reach a tile boundary
// together, and for a K = 512 product the boundary (epilogue + the K steps it disturbs) is a quarter of the tile while the matrix
// pipe idles.  Here a work-group is FOUR waves on a 256 x 128 tile (each wave 128 x 64 as before), a K step is 32 wide and both
// operands ring through three 24 KiB stages: 72 KiB of LDS, so two work-groups share a CU, one wave of each per SIMD, out of phase --
// one group's boundary runs under the other's MFMAs.  The price: a third more LDS-DMA pieces per MFMA (the A panel is staged by both
// groups) and a barrier every 16 MFMAs instead of 32.
//
// Scope of the experiment: C[M, N] = alpha * A[M, K] B[N, K]^T in bf16, both operands row-major (the forward "NT" product), M a
// multiple of 256, N of 128, K of 32 with at least 4 K steps -- the host (xclip_api.hip) sends everything else to gemm4.h.
//
// LDS image of an operand stage: rows of 32 bf16 = 64 bytes; the 16-byte chunk c of row r sits at slot c ^ ((r >> 2) & 3).  A DMA piece
// (1 KiB, one buffer_load ... lds per wave) is 16 rows; lane l of the piece lands at row (l >> 2), slot (l & 3), so it FETCHES chunk
// (l & 3) ^ ((l >> 4) & 3).  An MFMA operand fragment (lane = row, half h: k = 16 kk + 8 h ...) is the ds_read_b128 of chunk 2 kk + h; the
// 16 lanes of one LDS cycle (rows r .. r + 15, one chunk) then cover the sixteen 16-byte slots of a 256-byte bank row exactly once.
#pragma once
#include "../gemm4.h"

namespace xc {

constexpr int G6_BM = 256, G6_BN = 128, G6_BK = 32, G6_THREADS = 256;
constexpr int G6_A_BYTES = G6_BM * G6_BK * 2;                 // 16 KiB
constexpr int G6_B_BYTES = G6_BN * G6_BK * 2;                 //  8 KiB
constexpr int G6_STAGE_BYTES = G6_A_BYTES + G6_B_BYTES;       // 24 KiB
constexpr int G6_LDS_BYTES = 3 * G6_STAGE_BYTES;              // 72 KiB: two work-groups per CU

// fragment of the 32-row block `blk` (compile time) of an operand image, k-block kk: the wave's image base + this lane's offset for kk
template <int BLK>
XC_DEV u32x4 g6_frag(const unsigned char* base_plus_lane) { return lds_read16_async<BLK * 32 * 64>(base_plus_lane); }

XC_DEV void g6_read_frags(const unsigned char* As_lane, const unsigned char* Bs_lane, u32x4 (&a)[4], u32x4 (&b)[2]) {
    b[0] = g6_frag<0>(Bs_lane);
    b[1] = g6_frag<1>(Bs_lane);
    a[0] = g6_frag<0>(As_lane);
    a[1] = g6_frag<1>(As_lane);
    a[2] = g6_frag<2>(As_lane);
    a[3] = g6_frag<3>(As_lane);
}

__global__ __launch_bounds__(G6_THREADS, 2) void gemm6_kernel(Gemm2Params p) {
    XC_LDS_DYNAMIC(lds);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = uniform(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const int tiles_n = p.N / G6_BN, ntiles = (p.M / G6_BM) * tiles_n;
    const int nt = p.K / G6_BK, stride = gridDim.x;
    if ((int)blockIdx.x >= ntiles) return;

    auto tile_origin = [&](int id, int& m0, int& n0) {        // n fastest inside an XCD's run; wide outputs in bands (gemm4.h)
        const int tile = xcd_remap(id, ntiles);
        const int band = 2 * p.band_n;                        // (bands are given in 256-wide tiles)
        if (band > 0) {
            const int per_band = (p.M / G6_BM) * band;
            const int bi = tile / per_band, rem = tile - bi * per_band;
            m0 = (rem / band) * G6_BM;
            n0 = (bi * band + rem % band) * G6_BN;
        } else {
            m0 = (tile / tiles_n) * G6_BM;
            n0 = (tile % tiles_n) * G6_BN;
        }
    };

    // ---- DMA: this wave's 4 A pieces (rows 64 w + 16 q ...) and 2 B pieces (rows 32 w + 16 q ...) of a stage ----
    const uint32_t chunk = (uint32_t)((lane & 3) ^ ((lane >> 4) & 3));
    const uint32_t va = ((uint32_t)(64 * wave + (lane >> 2)) * (uint32_t)p.lda + chunk * 8u) * 2u;
    const uint32_t vb = ((uint32_t)(32 * wave + (lane >> 2)) * (uint32_t)p.ldb + chunk * 8u) * 2u;
    const uint32_t sa = (uint32_t)p.lda * 32u, sb = (uint32_t)p.ldb * 32u;      // 16 rows * ld * 2 bytes
    int d_id = blockIdx.x, d_t = 0;                           // the iterator's (tile, K step); past the end it repeats its last position
    const bf16_t* pa;
    const bf16_t* pb;
    {
        int m0, n0;
        tile_origin(d_id, m0, n0);
        pa = p.A + (long)m0 * p.lda;
        pb = p.B + (long)n0 * p.ldb;
    }
    const uint32_t abytes = 255u * (uint32_t)p.lda * 2u + 64u, bbytes = 127u * (uint32_t)p.ldb * 2u + 64u;
    auto issue = [&](unsigned char* stage) {                  // six pieces of the iterator's position into `stage`, then advance
        const BufRsrc ra = make_rsrc(pa, abytes), rb = make_rsrc(pb, bbytes);
#pragma unroll
        for (int q = 0; q < 4; ++q) buf_glds16(ra, va, sa * (uint32_t)q, stage + (4 * wave + q) * 1024);
#pragma unroll
        for (int q = 0; q < 2; ++q) buf_glds16(rb, vb, sb * (uint32_t)q, stage + G6_A_BYTES + (2 * wave + q) * 1024);
        if (++d_t == nt) {
            if (d_id + stride < ntiles) {
                d_t = 0;
                d_id += stride;
                int m0, n0;
                tile_origin(d_id, m0, n0);
                pa = p.A + (long)m0 * p.lda;
                pb = p.B + (long)n0 * p.ldb;
            } else {
                d_t = nt - 1;                                 // (dead stage: nobody reads it)
            }
        } else {
            pa += G6_BK;
            pb += G6_BK;
        }
    };

    // ---- fragment addresses: image base of the wave's rows + this lane's (row, chunk) offset for the two k-blocks of a step ----
    const int c31 = lane & 31, h = lane >> 5;
    const int foff0 = c31 * 64 + (((0 + h) ^ ((c31 >> 2) & 3)) << 4), foff1 = c31 * 64 + (((2 + h) ^ ((c31 >> 2) & 3)) << 4);
    const int a_rows = wm * 128 * 64, b_rows = G6_A_BYTES + wn * 64 * 64;

    // ---- epilogue geometry (interior tiles only): whole-line stores through this wave's own 4 KiB of a free A stage ----
    const G4GemmEpilogue<G4_PLAIN> packer{p};
    const uint32_t vc = ((uint32_t)(wm * 128 + (lane >> 3)) * (uint32_t)p.ldc + (uint32_t)(wn * 64 + 8 * (lane & 7))) * 2u;
    const uint32_t s8 = (uint32_t)p.ldc * 16u;

    // prologue: steps 0, 1, 2 into stages 0, 1, 2; step 0 must have landed
    issue(lds);
    issue(lds + G6_STAGE_BYTES);
    issue(lds + 2 * G6_STAGE_BYTES);
    XC_WAIT_VMEM_LE(12);
    barrier_nodrain();
    u32x4 a[2][4], b[2][2];
    g6_read_frags(lds + a_rows + foff0, lds + b_rows + foff0, a[0], b[0]);
    lds_wait<0>(a[0], b[0]);

    int st = 0;                                               // stage of the current step (step % 3)
    bool stores_behind = false;                               // the previous tile's 16 stores per lane may still be in flight
    for (int id = blockIdx.x; id < ntiles; id += stride) {
        int m0, n0;
        tile_origin(id, m0, n0);
        f32x16 acc[4][2];
        for (int t = 0; t < nt; ++t) {
            unsigned char* const cur = lds + st * G6_STAGE_BYTES;
            const int st1 = st == 2 ? 0 : st + 1;
            unsigned char* const nxt = lds + st1 * G6_STAGE_BYTES;
            const bool last = t == nt - 1;
            // ---- k-block 0: fragments of k-block 1 on their way ----
            g6_read_frags(cur + a_rows + foff1, cur + b_rows + foff1, a[1], b[1]);
            sched_fence();
            if (t == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = mfma_32x32x16_bf16_zero(__builtin_bit_cast(s16x8, b[0][j]), __builtin_bit_cast(s16x8, a[0][i]));
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mma_kblock(b[0][j], a[0][i], acc[i][j], (bf16_t*)nullptr);
            }
            sched_fence();
            lds_wait<0>(a[1], b[1]);
            sched_fence();
            // ---- k-block 1: step s + 1 has landed for every wave; nobody reads this stage any more ----
            // outstanding, oldest first: [step s + 1] [step s + 2] (+ at the first two steps of a tile: [s + 2 or s + 3 issued at the
            // boundary] [16 stores]) -- everything younger than step s + 1 may stay in flight
            if (stores_behind && t < 2) XC_WAIT_VMEM_LE(22);
            else XC_WAIT_VMEM_LE(6);
            barrier_nodrain();
            g6_read_frags(nxt + a_rows + foff0, nxt + b_rows + foff0, a[0], b[0]);
            sched_fence();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mma_kblock(b[1][j], a[1][i], acc[i][j], (bf16_t*)nullptr);
            }
            sched_fence();
            if (!last) issue(cur);                            // step s + 3 into the stage the barrier has just freed
            sched_fence();
            lds_wait<0>(a[0], b[0]);
            sched_fence();
            st = st1;
        }
        // ---- tile boundary: the last step left its stage empty -- exchange through this wave's own slice of it, then the six pieces that
        //      step skipped, then the stores (gemm4.h: pieces queued behind 128 KiB of stores reach the L2 late) ----
        {
            unsigned char* const freed = lds + (st == 0 ? 2 : st - 1) * G6_STAGE_BYTES;
            u32x4 o[4][4];
            packer.pack_lines(acc, freed + wave * 4096, o);
            lds_drain();
            issue(freed);
            const BufRsrc rc = make_rsrc(p.C + (long)m0 * p.ldc + n0, 255u * (uint32_t)p.ldc * 2u + 256u);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (p.stream_out) buf_st16_nt<0>(rc, vc, s8 * (uint32_t)(4 * i + k), o[i][k]);
                    else buf_st16<0>(rc, vc, s8 * (uint32_t)(4 * i + k), o[i][k]);
                }
            stores_behind = true;
        }
    }
    XC_WAIT_VMEM_LE(0);                                       // trailing pieces must land before the LDS is released
}

}  // namespace xc
