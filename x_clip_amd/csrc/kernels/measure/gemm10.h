// gemm10.h -- gemm9.h's kernel (net.4's input gradient with the GEGLU-LayerNorm backward in its epilogue) with TWO four-wave work-groups per CU.
//
// What gemm9.h measures (profiles/r05_u_gemm9_ablation.log, r05_w_sq_gemm9.txt): a 256 x 256 tile is 13 us of K loop and 25 us of epilogue, one
// after the other on a CU that holds ONE 8-wave work-group (160 KiB of LDS): matrix pipe busy 27 %, a SIMD's vector ALU 44 %, no unit
// saturated -- the kernel is the sum of its phases.  measure/gemm6.h already has the loop that lets two work-groups share a CU (four waves on
// 256 x 128 tiles, K step 32, both operands in rings of three 24 KiB stages = 72 KiB); as a plain GEMM it lost, because its K loop pays a third
// more LDS-DMA pieces and twice the barriers per MFMA for hiding a boundary that is a quarter of a K = 512 tile.  Here the "boundary" is two
// thirds of the tile: one group's epilogue (vector ALU, LDS exchange, 512 KB of lines) runs under the other group's MFMAs.  The second
// work-group of a CU starts half a tile late so that the two do not walk in step.
//
// RESULT (profiles/r05_aa_gemm10_two_groups_ab.log, MI355X, XCLIP_GEMM9_GROUPS=2 on the measurement build): SLOWER -- 1457 us against 1277 for the
// text tower's layer (44.2 against 38.7 us per 256 x 256 of output), whatever the start delay of the second group (0 / 17 / 30 us): the two
// groups' phases do overlap, but the 32-deep K step's doubled barriers and LDS-DMA pieces and two epilogues contending for one vector ALU / LDS
// cost more than the overlap returns.  AND NOT CORRECT on the hardware when a work-group walks more than one tile
// (test_ffn_dgrad_geglu_fused[33792-2048-512] fails; one tile per work-group and the emulator pass): a counted wait around the epilogue is
// short -- not chased, the form is not worth it.  MEASUREMENT BUILD ONLY; the product never launches this kernel.
//
// Same contract, same results as gemm9.h (the epilogue IS gemm9.h's, on the 2 x 2 wave layout): da = dOut W2 with W2 [D, F] k-major, interior
// tiles only (M % 256 == 0, F % 128 == 0, D % 32 == 0, at least 4 K steps).
#pragma once
#include "../gemm9.h"

namespace xc {

constexpr int G10_BM = 256, G10_BN = 128, G10_BK = 32, G10_THREADS = 256;
constexpr int G10_A_BYTES = G10_BM * G10_BK * 2;              // 16 KiB: rows of 32 bf16 = 64 bytes, chunk c of row r at slot c ^ ((r >> 2) & 3) (gemm6.h)
constexpr int G10_B_BYTES = G10_BN * G10_BK * 2;              //  8 KiB: two panels of [32 k][64 n] (gemm2.h's k-major image, half as deep)
constexpr int G10_STAGE_BYTES = G10_A_BYTES + G10_B_BYTES;    // 24 KiB
constexpr int G10_LDS_BYTES = 3 * G10_STAGE_BYTES;            // 72 KiB: two work-groups per CU
constexpr int G9_GROUPS_DEFAULT = 1;                          // which of the two forms xclip_ffn_dgrad_geglu launches (measurement build: XCLIP_GEMM9_GROUPS)
constexpr int G10_STAGGER_10NS = 1700;                        // start delay of a CU's second work-group: half a tile

// the wave's A fragments (normal image, 32-row block BLK) and B fragments (k-major image, 32-column block J of the wave's panel) of one k-block
template <int BLK>
XC_DEV u32x4 g10_frag_a(const unsigned char* base_plus_lane) { return lds_read16_async<BLK * 32 * 64>(base_plus_lane); }
XC_DEV u32x4 g10_frag_b(const unsigned char* p) {
    const u32x2 lo = lds_read_tr16_async(p);                  // k = 8 * (lane >> 5) + 0..3 of the k-block
    const u32x2 hi = lds_read_tr16_async(p + 4 * 128);        // ... + 4..7
    u32x4 f = {lo[0], lo[1], hi[0], hi[1]};
    return f;
}
XC_DEV void g10_read_frags(const unsigned char* As_lane, const unsigned char* B0, const unsigned char* B1, u32x4 (&a)[4], u32x4 (&b)[2]) {
    b[0] = g10_frag_b(B0);
    b[1] = g10_frag_b(B1);
    a[0] = g10_frag_a<0>(As_lane);
    a[1] = g10_frag_a<1>(As_lane);
    a[2] = g10_frag_a<2>(As_lane);
    a[3] = g10_frag_a<3>(As_lane);
}

// a lane's loop-invariant byte offsets: DMA sources (va, vb) and fragment reads (fa0 / fa1: A, k-blocks 0 / 1 of a step; fb[j]: B, column
// block j, k-block 0).  They are RECOMPUTED behind every tile's epilogue (from a lane index the optimiser cannot see through) instead of
// living through it: the epilogue has no register to spare (gemm9.h), and six carried offsets were 39 spilled registers around its line loads.
struct G10Offsets {
    uint32_t va, vb;
    int fa0, fa1, fb[2];
};
XC_DEV G10Offsets g10_offsets(const Gemm2Params& p, int wave, int lane) {
    G10Offsets o;
    // DMA: this wave's 4 A pieces (rows 64 w + 16 q ...) and 2 B pieces (panel w >> 1, k rows 16 (w & 1) + 8 q ...) of a stage
    const uint32_t chunk_a = (uint32_t)((lane & 3) ^ ((lane >> 4) & 3));
    o.va = ((uint32_t)(64 * wave + (lane >> 2)) * (uint32_t)p.lda + chunk_a * 8u) * 2u;
    const uint32_t krow_b = (uint32_t)(16 * (wave & 1) + (lane >> 3));
    const uint32_t chunk_b = (uint32_t)((lane & 7) ^ (((krow_b >> 1) & 1) << 2));
    o.vb = (krow_b * (uint32_t)p.ldb + (uint32_t)(64 * (wave >> 1)) + chunk_b * 8u) * 2u;
    // fragments: A as measure/gemm6.h (row c31 of a 32-row block, chunk 2 kk + h); B as gemm2.h g2_frag_kmajor on a 32-deep panel: lane -> (k
    // row, column) of the transposing read; k-block 1 adds 16 k rows
    const int wm = wave >> 1, wn = wave & 1;
    const int c31 = lane & 31, h = lane >> 5;
    o.fa0 = wm * 128 * 64 + c31 * 64 + (((0 + h) ^ ((c31 >> 2) & 3)) << 4);
    o.fa1 = wm * 128 * 64 + c31 * 64 + (((2 + h) ^ ((c31 >> 2) & 3)) << 4);
    const int g = lane >> 4, tt = lane & 15;
    const int krow0 = 8 * (g >> 1) + (tt >> 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int colp = 32 * j + 16 * (g & 1) + (tt & 3) * 4;
        const int chunk = (colp >> 3) ^ (((krow0 >> 1) & 1) << 2);
        o.fb[j] = G10_A_BYTES + wn * 4096 + krow0 * 128 + chunk * 16 + (colp & 7) * 2;
    }
    return o;
}

template <int ABL = 0>
__global__ __launch_bounds__(G10_THREADS, 2) void gemm10_geglu_bwd_kernel(Gemm2Params p, GegluBwdArgs e, int stagger_10ns) {
    XC_LDS_DYNAMIC(lds);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = uniform(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const int tiles_n = p.N / G10_BN, ntiles = (p.M / G10_BM) * tiles_n;
    const int nt = p.K / G10_BK, stride = gridDim.x;
    if ((int)blockIdx.x >= ntiles) return;
    // the CU's second work-group (its LDS does not start at 0) begins half a tile late: bounded by an iteration count as well as by the clock
    if (stagger_10ns > 0 && lds_base_granule() != 0) {
        const uint64_t until = realtime_10ns() + (uint64_t)stagger_10ns;
        for (int spin = 0; spin < 4 * stagger_10ns && realtime_10ns() < until; ++spin) nap();
    }
    auto tile_origin = [&](int id, int& m0, int& n0) {        // n fastest inside an XCD's run: the column tiles of a row panel share dOut's rows in one L2
        const int tile = xcd_remap(id, ntiles);
        m0 = (tile / tiles_n) * G10_BM;
        n0 = (tile % tiles_n) * G10_BN;
    };

    G10Offsets o = g10_offsets(p, wave, (int)opaque((uint32_t)lane));
    const uint32_t sa = (uint32_t)p.lda * 32u, sb = (uint32_t)p.ldb * 16u;      // 16 rows of A / 8 k rows of B, in bytes
    int d_id = blockIdx.x, d_t = 0;                           // the iterator's (tile, K step); past the end it repeats its last position
    const bf16_t* pa;
    const bf16_t* pb;
    {
        int m0, n0;
        tile_origin(d_id, m0, n0);
        pa = p.A + (long)m0 * p.lda;
        pb = p.B + n0;
    }
    const uint32_t abytes = 255u * (uint32_t)p.lda * 2u + 64u, bbytes = 31u * (uint32_t)p.ldb * 2u + 256u;
    auto issue = [&](unsigned char* stage) {                  // six pieces of the iterator's position into `stage`, then advance
        const BufRsrc ra = make_rsrc(pa, abytes), rb = make_rsrc(pb, bbytes);
#pragma unroll
        for (int q = 0; q < 4; ++q) buf_glds16(ra, o.va, sa * (uint32_t)q, stage + (4 * wave + q) * 1024);
#pragma unroll
        for (int q = 0; q < 2; ++q) buf_glds16(rb, o.vb, sb * (uint32_t)q, stage + G10_A_BYTES + (2 * wave + q) * 1024);
        if (++d_t == nt) {
            if (d_id + stride < ntiles) {
                d_t = 0;
                d_id += stride;
                int m0, n0;
                tile_origin(d_id, m0, n0);
                pa = p.A + (long)m0 * p.lda;
                pb = p.B + n0;
            } else {
                d_t = nt - 1;                                 // (dead stage: nobody reads it)
            }
        } else {
            pa += G10_BK;
            pb += (long)G10_BK * p.ldb;
        }
    };

    constexpr int BKK = 16 * 128;                             // k-block 1 of a step: 16 k rows further

    const G4GegluBwdEpilogue<ABL, 2> epi{p, e};

    // prologue: steps 0, 1, 2 into stages 0, 1, 2; step 0 must have landed
    issue(lds);
    issue(lds + G10_STAGE_BYTES);
    issue(lds + 2 * G10_STAGE_BYTES);
    XC_WAIT_VMEM_LE(12);
    barrier_nodrain();
    u32x4 a[2][4], b[2][2];
    g10_read_frags(lds + o.fa0, lds + o.fb[0], lds + o.fb[1], a[0], b[0]);
    lds_wait<0>(a[0], b[0]);

    int st = 0;                                               // stage of the current step (step % 3)
    bool stores_behind = false;                               // the previous tile's 16 youngest stores per lane may still be in flight
    for (int id = blockIdx.x; id < ntiles; id += stride) {
        int m0, n0;
        tile_origin(id, m0, n0);
        f32x16 acc[4][2];
        for (int t = 0; t < nt; ++t) {
            unsigned char* const cur = lds + st * G10_STAGE_BYTES;
            const int st1 = st == 2 ? 0 : st + 1;
            unsigned char* const nxt = lds + st1 * G10_STAGE_BYTES;
            const bool last = t == nt - 1;
            // ---- k-block 0: fragments of k-block 1 on their way ----
            g10_read_frags(cur + o.fa1, cur + o.fb[0] + BKK, cur + o.fb[1] + BKK, a[1], b[1]);
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
            // outstanding, oldest first: [step s + 1] [step s + 2]; at the first two steps of a tile [the epilogue's 16 youngest stores] [the
            // six pieces issued behind the epilogue] in their place (steps s + 1, s + 2 landed under the epilogue's own waits: the counter
            // retires in order) -- everything younger than step s + 1 may stay in flight
            if (stores_behind && t < 2) XC_WAIT_VMEM_LE(22);
            else XC_WAIT_VMEM_LE(6);
            barrier_nodrain();
            // (the epilogue needs the 24 registers of the next tile's first fragments: behind it, not here -- gemm4.h DEFER_FRAGS)
            if (!last) g10_read_frags(nxt + o.fa0, nxt + o.fb[0], nxt + o.fb[1], a[0], b[0]);
            sched_fence();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mma_kblock(b[1][j], a[1][i], acc[i][j], (bf16_t*)nullptr);
            }
            sched_fence();
            if (!last) issue(cur);                            // step s + 3 into the stage the barrier has just freed
            sched_fence();
            if (!last) lds_wait<0>(a[0], b[0]);
            sched_fence();
            st = st1;
        }
        // ---- tile boundary: the last step left its stage empty -- the epilogue works through this wave's own 4 KiB of it (the A part: 16 KiB),
        //      THEN the six pieces that step skipped go into it, then the next tile's first fragments ----
        {
            unsigned char* const freed = lds + (st == 0 ? 2 : st - 1) * G10_STAGE_BYTES;
            stores_behind = epi.with_scratch(acc, m0, n0, freed + wave * 4096) == 16;
            lds_drain();                                      // this wave's last exchange reads have returned before the pieces overwrite the slice
            o = g10_offsets(p, wave, (int)opaque((uint32_t)lane));   // (fresh: nothing of the loop's offsets lives through the epilogue)
            issue(freed);
            unsigned char* const first = lds + st * G10_STAGE_BYTES;
            g10_read_frags(first + o.fa0, first + o.fb[0], first + o.fb[1], a[0], b[0]);
            lds_wait<0>(a[0], b[0]);
        }
    }
    XC_WAIT_VMEM_LE(0);                                       // trailing pieces must land before the LDS is released
}

}  // namespace xc
