// gemm3.h -- the scheduled form of the bf16 production GEMM (same contract, LDS images, tile shape and two-stage K loop
// as gemm2.h: 256 x 256 output tile, 8 waves of 128 x 64, K step 64, LDS DMA staging, transpose reads for k-major
// operands).  Two things change, both driven by measurements of gemm2.h on MI355X (about 8 us fixed + 2.1 us per K step
// per tile against 0.85 us of MFMA work per K step):
//   * the fragment reads are software pipelined by hand: the six ds_reads of k-block kk+1 are issued BEFORE the eight
//     MFMAs of k-block kk, as volatile asm the compiler's wait-count pass does not see, and waited for (lds_wait) AFTER
//     those MFMAs.  (With builtin / plain loads hipcc put s_waitcnt lgkmcnt(0) between every read batch and the MFMAs it
//     was meant to overlap, whatever the source order and sched_barriers said: LDS reads and MFMAs simply added up,
//     0.69 + 1.15 us per K step.  A ring of four 32-deep stages with counted vmcnt was measured SLOWER than the two-stage
//     loop.)
//   * the epilogue goes straight from the accumulators to global memory: the MFMAs are issued with swapped operands
//     (D^T = B-fragment x A-fragment), so a lane owns ONE output row and 4 consecutive columns per register quad;
//     v_permlane32_swap pairs the quads into 16-byte stores (cdna_hip_programming.md T21).  No LDS staging and no
//     epilogue barriers; bias / row-gather / residual terms are added in fp32 before the single rounding to bf16.
#pragma once
#include "gemm2.h"

namespace xc {

// Fragment I (0..3) of a wave's strip starting at tile row / column o0, k-block kk, issued as an untracked LDS read
// (xc_device.h: lds_read16_async); same addressing as g2_frag_normal / g2_frag_kmajor.
template <bool KMAJOR, int I>
XC_DEV u32x4 g3_frag(const unsigned char* tile, int o0, int kk, int lane) {
    if (!KMAJOR) {
        const int row = o0 + (lane & 31);                                     // + 32 I rows = + 4096 I bytes, same swizzle
        const int chunk = (kk * 2 + (lane >> 5)) ^ ((row >> 1) & 7);
        return lds_read16_async<I * 4096>(tile + row * 128 + chunk * 16);
    } else {
        const int g = lane >> 4, tt = lane & 15;
        const int col = o0 + I * 32 + 16 * (g & 1) + (tt & 3) * 4;
        const int krow = kk * 16 + 8 * (g >> 1) + (tt >> 2);
        const int panel = col >> 6, colp = col & 63;
        const int chunk = (colp >> 3) ^ (((krow >> 1) & 1) << 2);
        const unsigned char* p = tile + panel * 8192 + krow * 128 + chunk * 16 + (colp & 7) * 2;
        const u32x2 lo = lds_read_tr16_async(p);                              // k = 8 * (lane >> 5) + 0..3
        const u32x2 hi = lds_read_tr16_async(p + 4 * 128);                    // k = 8 * (lane >> 5) + 4..7
        u32x4 f = {lo[0], lo[1], hi[0], hi[1]};
        return f;
    }
}
template <bool A_KMAJOR, bool B_KMAJOR>
XC_DEV void g3_read_frags(const unsigned char* As, const unsigned char* Bs, int am, int bn, int kk, int lane, u32x4 (&a)[4], u32x4 (&b)[2]) {
    b[0] = g3_frag<B_KMAJOR, 0>(Bs, bn, kk, lane);
    b[1] = g3_frag<B_KMAJOR, 1>(Bs, bn, kk, lane);
    a[0] = g3_frag<A_KMAJOR, 0>(As, am, kk, lane);
    a[1] = g3_frag<A_KMAJOR, 1>(As, am, kk, lane);
    a[2] = g3_frag<A_KMAJOR, 2>(As, am, kk, lane);
    a[3] = g3_frag<A_KMAJOR, 3>(As, am, kk, lane);
}
XC_DEV void g3_add4(float (&v)[4], const bf16_t* p) {
    const u32x2 t = *reinterpret_cast<const u32x2*>(p);
    v[0] += u2f(t[0] << 16); v[1] += u2f(t[0] & 0xffff0000u); v[2] += u2f(t[1] << 16); v[3] += u2f(t[1] & 0xffff0000u);
}

// (An L2 prefetch stream -- one 4-byte LDS-DMA touch per future 128-byte line, three K steps ahead -- measured SLOWER: qkv fwd
// 644 -> 603 TF/s, wgrad 900 -> 742; vmcnt retires in order, so every K step then waits on a one-step-old HBM access.)
constexpr int G3_LDS_BYTES = G2_LDS_BYTES;                    // the two 64 KiB stages; the epilogue goes straight from registers

// The persistent tile loop shared by the GEMM and by the contrastive-head kernels (simloss3.h): `epi(acc, m0, n0, full)` is
// called once per finished 256 x 256 tile with the TRANSPOSED accumulators (see below) and must report how many global
// stores per lane it issued when `full` (interior tile) so the next tile's first wait can leave exactly those in flight.
// (The return value is no longer needed by the loop below -- its one wait per K step drains everything -- and is ignored.)
//
// Schedule of one K step (4 k-blocks C0..C3 of 8 MFMAs per wave; fragments of k-block kk+1 are read under the MFMAs of kk):
//
//     [read kk1] C0 + B pieces of DMA(s+1)   [read kk2] C1   [read kk3] C2   wait DMA(s+1); BARRIER   [read kk0 of s+1] C3 + A
//                                                                                                      pieces of DMA(s+2)
//
// The barrier that publishes the next stage sits BEFORE the last k-block, not after it: when a wave leaves the barrier it
// still has eight MFMAs whose operands are already in registers, and those cover the LDS latency of the first fragment reads
// from the new stage.  (With the barrier at the K-step boundary every wave came out of it with nothing to compute until twelve
// ds_reads returned -- all eight waves at once, 96 KiB through the 128 B/clk LDS port, ~0.4 us of idle matrix cores per K
// step.)  Stage s&1 is free for DMA(s+2) from that same barrier on (every wave has finished its kk3 reads), which also gives
// the A operand -- the one that streams from HBM -- a full K step of lookahead with only two 64 KiB stages.
// DMA issue schedule: the piece (0-3 = A, 4-7 = B) that follows MFMA pair i of k-block kk, -1 = none: A0-3 in C3 (the k-block
// right after the barrier that frees the stage), B0-3 in the next step's C0.  Measured alternatives, all slower: 3 + 3 + 2 over
// C3 / C0 / C1 (+1.5 %), 2 per k-block (+3 %), all eight in one k-block with the two wave halves alternating (+12 %).
XC_DEV constexpr int g3_dma_piece(int kk, int i) { return kk == 3 ? i : (kk == 0 ? 4 + i : -1); }

// which tiles a launch walks: every tile of the tiles_m x tiles_n grid in the XCD-aware order (the default), or a list of its own
// (simloss5.h: only the tiles on the diagonal / at the ragged edges).  A list may name a tile with m0 >= p.M: it computes nothing useful
// and its epilogue stores nothing (a hole in a list with a fixed number of slots per row tile).
struct G3AllTiles {
    XC_DEV int count(const Gemm2Params& p) const { return p.tiles_m * p.tiles_n; }
    XC_DEV void origin(const Gemm2Params& p, int id, int& m0, int& n0) const {
        const int tile = xcd_remap(id, p.tiles_m * p.tiles_n);
        m0 = (tile / p.tiles_n) * G2_BM;
        n0 = (tile % p.tiles_n) * G2_BN;
    }
};

template <bool A_KMAJOR, bool B_KMAJOR, int ABL, class Epilogue, class Tiles = G3AllTiles>
XC_DEV void g3_run(const Gemm2Params& p, unsigned char* lds, Epilogue epi, Tiles tiles = Tiles{}) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = uniform(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int ntiles = tiles.count(p);
    const int kbeg = blockIdx.y * p.k_per_split;
    const int kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
    const int nt = (kend - kbeg) / G2_BK;
    const int stride = gridDim.x;

    // Persistent: this work-group walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... (tile ids are XCD-remapped so that
    // concurrently running work-groups of one XCD share A row-panels in that XCD's L2).  The K steps of consecutive tiles
    // form ONE stream through the two LDS stages; the DMA iterator (tile did, K step dt) runs up to two steps ahead of the
    // MFMAs, across tile boundaries, and the epilogue's stores drain under the next tile's first k-blocks.
    auto tile_origin = [&](int id, int& m0, int& n0) { tiles.origin(p, id, m0, n0); };
    int did = blockIdx.x, dt = 0, dm = 0, dn = 0;
    bool dvalid = did < ntiles && nt > 0;
    if (!dvalid) return;                                      // (uniform over the work-group)
    tile_origin(did, dm, dn);
    auto dma_next = [&]() {
        if (++dt == nt) {
            dt = 0;
            did += stride;
            dvalid = did < ntiles;
            if (dvalid) tile_origin(did, dm, dn);
        }
        if (ABL & 2) dvalid = false;
    };
    g2_stage<A_KMAJOR>(p.A, p.lda, dm, p.M, kbeg, lds, wave, lane);
    g2_stage<B_KMAJOR>(p.B, p.ldb, dn, p.N, kbeg, lds + G2_OPER_BYTES, wave, lane);
    dma_next();
    XC_WAIT_VMEM_LE(0);
    barrier_nodrain();                                        // step 0 has landed for every wave
    if (dvalid) {                                             // the pieces of DMA(1) that the loop would have issued in "C3 of step -1"
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = g3_dma_piece(3, i);
            if (piece < 0) continue;
            if (piece < 4) { if (!(ABL & 16)) g2_stage_piece<A_KMAJOR>(p.A, p.lda, dm, p.M, kbeg + dt * G2_BK, lds + G2_STAGE_BYTES, wave, lane, piece); }
            else if (!(ABL & 32)) g2_stage_piece<B_KMAJOR>(p.B, p.ldb, dn, p.N, kbeg + dt * G2_BK, lds + G2_STAGE_BYTES + G2_OPER_BYTES, wave, lane, piece - 4);
        }
    }

    u32x4 a[2][4], b[2][2];
    if (ABL & 8) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[x][i] = zero16();
#pragma unroll
            for (int j = 0; j < 2; ++j) b[x][j] = zero16();
        }
    } else {
        g3_read_frags<A_KMAJOR, B_KMAJOR>(lds, lds + G2_OPER_BYTES, wm * 128, wn * 64, 0, lane, a[0], b[0]);
        lds_wait<0>(a[0], b[0]);
    }

    int step = 0;                                             // running K-step counter: LDS stage = step & 1
    for (int id = blockIdx.x; id < ntiles; id += stride) {
    int m0, n0;
    tile_origin(id, m0, n0);
    // acc[i][j] holds the TRANSPOSED 32 x 32 block: register r of lane l is
    // C[m = i-block row (l & 31)][n = j-block column (r & 3) + 8 (r >> 2) + 4 (l >> 5)]
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int t = 0; t < nt; ++t, ++step) {
        unsigned char* cur_stage = lds + (step & 1) * G2_STAGE_BYTES;
        unsigned char* nxt_stage = lds + ((step + 1) & 1) * G2_STAGE_BYTES;
        const unsigned char* As = cur_stage;
        const unsigned char* Bs = cur_stage + G2_OPER_BYTES;
        const bool more = (t + 1 < nt) || (id + stride < ntiles);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk < 3) {
                if (!(ABL & 8))                                              // fragments of the NEXT k-block first ...
                    g3_read_frags<A_KMAJOR, B_KMAJOR>(As, Bs, wm * 128, wn * 64, kk + 1, lane, a[nxt], b[nxt]);
            } else {
                XC_WAIT_VMEM_LE(0);                                          // this wave's share of DMA(step + 1) (and any epilogue stores)
                barrier_nodrain();                                           // ... everybody's; and nobody reads stage step & 1 any more
                if (more && !(ABL & 8))
                    g3_read_frags<A_KMAJOR, B_KMAJOR>(nxt_stage, nxt_stage + G2_OPER_BYTES, wm * 128, wn * 64, 0, lane, a[nxt], b[nxt]);
            }
            sched_fence();                                                   // ... then this k-block's MFMAs, order pinned
            // this wave's 8 DMA pieces of the stage freed by the barrier above (A0-3, B0-3) trickle out behind MFMA pairs of
            // C3 and of the next step's first k-blocks: g3_dma_piece(kk, i) says which one goes after pair i of k-block kk
            unsigned char* dst = (kk == 3) ? cur_stage : nxt_stage;
            const int dk = kbeg + dt * G2_BK;
            const int dma_m = (ABL & 64) ? (dm & 256) : dm;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (ABL & 1) { asm volatile("" :: "v"(a[cur][i]), "v"(b[cur][j])); }
                    else acc[i][j] = mma_kblock(b[cur][j], a[cur][i], acc[i][j], (bf16_t*)nullptr);   // D^T
                }
                const int piece = g3_dma_piece(kk, i);
                if (piece >= 0 && dvalid) {
                    sched_fence();
                    if (piece < 4) {
                        if (!(ABL & 16)) g2_stage_piece<A_KMAJOR>(p.A, p.lda, dma_m, p.M, dk, dst, wave, lane, piece);
                    } else {
                        if (!(ABL & 32)) g2_stage_piece<B_KMAJOR>(p.B, p.ldb, dn, p.N, dk, dst + G2_OPER_BYTES, wave, lane, piece - 4);
                    }
                    sched_fence();
                }
            }
            if (kk == 0 && dvalid) dma_next();                  // this wave's share of the stage is on its way
            sched_fence();
            if (!(ABL & 8)) lds_wait<0>(a[nxt], b[nxt]);                     // landed long ago: the 8 MFMAs above covered the latency
            sched_fence();
        }
    }

    (void)epi(acc, m0, n0);
    }   // tile loop
    epi.finish();                                             // per-work-group leftovers of the epilogue (e.g. one atomic per wave)
}

// ---- the GEMM epilogue: registers -> global, one output row per lane ------------------------------------------------------------
template <int ABL>
struct G3GemmEpilogue {
    const Gemm2Params& p;
    XC_DEV void finish() const {}
    XC_DEV int operator()(f32x16 (&acc)[4][2], int m0, int n0) const {
        const int lane = threadIdx.x & 63, h = lane >> 5;
        const int wave = uniform(threadIdx.x >> 6), wm = wave >> 2, wn = wave & 3;
        const bool full = (m0 + G2_BM <= p.M) && (n0 + G2_BN <= p.N);       // interior tile: no per-element range checks
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gm = m0 + wm * 128 + i * 32 + (lane & 31);
            const bool row_ok = full || gm < p.M;
            const int gmc = row_ok ? gm : p.M - 1;
            const long add_row = p.addrows != nullptr ? (long)p.rowidx[gmc] * p.ld_add : 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nb = n0 + wn * 64 + j * 32;
                if (p.partial != nullptr) {
                    float* slab = p.partial + ((long)g2_slice(p, p.tiles_m * p.tiles_n) * p.M + gmc) * p.N;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int gn = nb + 4 * h + 8 * q;
                        if (row_ok && (full || gn < p.N)) {
                            u32x4 v = {f2u(acc[i][j][4 * q]), f2u(acc[i][j][4 * q + 1]), f2u(acc[i][j][4 * q + 2]), f2u(acc[i][j][4 * q + 3])};
                            st16(slab + gn, v);
                        }
                    }
                    continue;
                }
                float v[4][4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[q][e] = acc[i][j][4 * q + e] * p.alpha;
                if (p.bias != nullptr || p.addrows != nullptr || p.residual != nullptr) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        int gn = nb + 4 * h + 8 * q;                         // 4 consecutive columns; clamped reads, masked stores
                        gn = (full || gn < p.N) ? gn : p.N - 4;
                        if (p.bias != nullptr) g3_add4(v[q], p.bias + gn);
                        if (p.addrows != nullptr) g3_add4(v[q], p.addrows + add_row + gn);
                        if (p.residual != nullptr) g3_add4(v[q], p.residual + (long)gmc * p.ldr + gn);
                    }
                }
                uint32_t pk[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pk[q][0] = (uint32_t)f2bf(v[q][0]) | ((uint32_t)f2bf(v[q][1]) << 16);
                    pk[q][1] = (uint32_t)f2bf(v[q][2]) | ((uint32_t)f2bf(v[q][3]) << 16);
                }
                // quads (0,1) and (2,3): lower lanes end up with columns [0,8) / [16,24), upper lanes with [8,16) / [24,32)
#pragma unroll
                for (int qq = 0; qq < 4; qq += 2) {
                    permlane32_swap(pk[qq][0], pk[qq + 1][0]);
                    permlane32_swap(pk[qq][1], pk[qq + 1][1]);
                    const int gn = nb + qq * 8 + 8 * h;
                    if (row_ok && (full || gn < p.N)) {
                        u32x4 o = {pk[qq][0], pk[qq][1], pk[qq + 1][0], pk[qq + 1][1]};
                        if (ABL & 4) { asm volatile("" :: "v"(o)); }
                        else st16(p.C + (long)gm * p.ldc + gn, o);
                    }
                }
            }
        }
        // interior tiles issue exactly 16 (bf16) / 32 (fp32 split-K slab) stores per lane; ragged ones are drained fully
        return full ? (p.partial != nullptr ? 32 : 16) : 0;
    }
};

// ABL (measurement only, XCLIP_GEMM_ABL) is a bit mask: 1 = MFMAs removed, 2 = DMA only for the first K step, 4 = epilogue
// stores removed, 8 = LDS fragment reads removed, 16 / 32 = the A / B operand's DMA removed, 64 = A's DMA source wrapped to 512
// rows (L2-resident); 0 = the product kernel.  Per K step per CU at M=263168 N=512 K=2048, warm clocks
// (profiles/r01_step9_gemm_ablation.log):  full 2.11 us;  MFMAs alone 1.14;  MFMA + LDS reads 1.57;  MFMA + DMA 1.79;  DMA alone
// 1.65 (1.15 with A from L2: the L2 -> LDS path gives ~60 GB/s per CU whatever the source pattern -- 128-byte row pieces and
// contiguous 32 KiB blocks time the same, padded row strides too);  skeleton 0.21;  the epilogue's stores cost 0.1 us per K
// step at K=2048 and 0.56 at K=512.  Before the fragment reads went to untracked asm the full loop was 2.39 and MFMA + reads
// 1.78.  Negative results kept for the record: a ring of four 32-deep stages, an A-ring of three + B-ring of two 64-deep stages
// (all 160 KiB), s_setprio around the MFMA groups, L2 prefetch touches, staggered work-group starts, other DMA issue schedules,
// accumulators pinned to AGPRs (the compiler then splits 128 / 128 and spills), an epilogue transposed through wave-private LDS
// so that every store is a full 128-byte line (K=512: 508 vs 510 us -- the 86 us the stores cost there are not a coalescing
// problem), the tile's last B pieces issued before the epilogue so its stores may stay in flight one more K step (no change),
// one fragment read in front of each MFMA instead of six up front (+1 %), non-temporal epilogue stores (K=512: 731 vs 521 us),
// four waves of 128 x 128 instead of eight of 128 x 64 (a third less fragment traffic, but hipcc spills ~100 of the 256 + 64
// live registers and a lone wave per SIMD has nobody to hide behind: 626 vs 537 us at K=2048), a 2-D blocked tile order (8 A panels
// x 4 weight tiles per XCD round instead of 2 x 16, to keep the weights in L2: FF1 +2 %, QKV -4 %, others within noise).
// Where the K=512 epilogue goes (M=263168 N=1536): 529 us as is, 452 with every tile storing into the same L2-resident 128 KiB, 410
// with no stores -- 42 us of store issue / L2 acceptance, 77 us of the 808 MB write stream reaching HBM next to the A reads.
template <bool A_KMAJOR, bool B_KMAJOR, int ABL = 0>
__global__ __launch_bounds__(G2_THREADS, 2) void gemm3_kernel(Gemm2Params p) {
    XC_LDS_DYNAMIC(lds);
    g3_run<A_KMAJOR, B_KMAJOR, ABL>(p, lds, G3GemmEpilogue<ABL>{p});
}

}  // namespace xc
