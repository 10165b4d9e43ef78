// gemm4.h -- the production bf16 GEMM: gemm3.h's tile, LDS images and K-step schedule (256 x 256 output tile, 8 waves of 128 x 64,
// K step 64, two 64 KiB stages filled by LDS DMA, fragment reads pipelined under the MFMAs, barrier before the last k-block,
// persistent work-groups), with the instruction overhead AROUND the matrix work removed.  What the ISA of gemm3.h showed
// (hipcc -S, the K-step loop of gemm3_kernel<false, false>):
//   * every one of the 8 DMA pieces a wave issues per K step rebuilt its 64-bit source address from the tile origin -- 11 VALU
//     instructions each, two v_mul_lo_u32 and a v_mad_u64_u32 among them -- behind a vector-compare branch on "is the DMA
//     iterator still valid": ~90 VALU instructions and 8 branches per K step wedged between the MFMA pairs of an in-order wave;
//   * the epilogue tested "interior tile or column in range" per store with exec-mask branches and evaluated the optional bias /
//     row-gather / residual / split-K pointers at run time for each of the 8 accumulator blocks.
// Here the DMA goes through buffer descriptors (xc_device.h: BufRsrc): the descriptor base carries the tile origin and the K
// position and is advanced with scalar instructions once per K step; a lane's byte offset inside the tile is ONE loop-invariant VGPR
// per piece parity; ragged edges need no clamps (a row or byte past the descriptor's extent reads as zero).  The DMA iterator never
// becomes invalid -- past the last tile it re-fetches its last position into an LDS stage nobody reads again -- so the K loop has no
// branches but its own back-edge.  The epilogue is compiled per MODE (what the caller's optional terms are) with a straight-line
// interior-tile path: 64 conversions, 32 v_permlane32_swap, 16 descriptor stores with immediate offsets.
#pragma once
#include <type_traits>
#include "gemm3.h"

namespace xc {

// what the epilogue of a launch has to do, fixed at compile time (the host picks the instantiation)
enum : int {
    G4_PLAIN = 0,        // C = alpha * acc                         (bf16)
    G4_SLAB = 1,         // fp32 split-K slab, no alpha
    G4_TERMS = 2,        // + bias / gathered rows / residual, any subset (the general form of gemm3.h)
    G4_RES = 3           // C = alpha * acc + residual (bf16): the block's second skip connection in the FF2 forward, accumulating GEMMs
};

// per-lane byte offset of DMA piece q (0..3) of this wave inside an operand tile whose descriptor base is the tile's first element
// at the current K position; pieces q and q + 2 differ by 16 rows (normal) / 16 k-rows (k-major): that part travels in soffset
template <bool KMAJOR>
XC_DEV uint32_t g4_voff(long ld, int wave, int lane, int q) {
    const int id = wave * 4 + q;
    if (!KMAJOR) {
        const int row = id * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        return ((uint32_t)row * (uint32_t)ld + (uint32_t)chunk * 8u) * 2u;
    } else {
        const int panel = id >> 3;
        const int row = (id & 7) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (((row >> 1) & 1) << 2);
        return ((uint32_t)row * (uint32_t)ld + (uint32_t)(panel * 64 + chunk * 8)) * 2u;
    }
}
// One operand of the DMA iterator: the (wave-uniform) address of the tile's first element at the iterator's K position and the
// extent of the descriptor that covers one K step from there.  Reads past the tile's last valid row (normal operand) or past the
// operand's last valid column in the tile's last k-row (k-major: earlier rows run on into the next row -- valid memory whose values
// only reach output columns that are never stored) return zero instead of faulting.  The address moves by a constant per K step
// (two scalar adds); only a tile change recomputes it.
template <bool KMAJOR>
struct G4Operand {
    const bf16_t* pos;
    uint32_t bytes;
    XC_DEV void tile(const bf16_t* X, long ld, int outer0, int nouter, int k0) {
        int valid = nouter - outer0;
        valid = valid < 256 ? valid : 256;
        if (!KMAJOR) {
            pos = X + (long)outer0 * ld + k0;
            bytes = (uint32_t)(valid - 1) * (uint32_t)ld * 2u + 128u;
        } else {
            pos = X + (long)k0 * ld + outer0;
            bytes = 63u * (uint32_t)ld * 2u + (uint32_t)valid * 2u;
        }
    }
    XC_DEV void advance(long ld) { pos += KMAJOR ? (long)G2_BK * ld : (long)G2_BK; }
    XC_DEV BufRsrc rsrc() const { return make_rsrc(pos, bytes); }
};

struct G4Dma {
    BufRsrc a, b;
};

template <bool A_KMAJOR, bool B_KMAJOR, class Epilogue>
XC_DEV void g4_run(const Gemm2Params& p, unsigned char* lds, Epilogue epi) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = uniform(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int ntiles = p.tiles_m * p.tiles_n;
    int tile0, slice, stride;
    const bool mine = g2_where(p, ntiles, tile0, slice, stride);
    const int kbeg = slice * p.k_per_split;
    const int kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
    const int nt = (kend - kbeg) / G2_BK;
    if (!mine || nt <= 0) return;                                // (uniform over the work-group)

    auto tile_origin = [&](int id, int& m0, int& n0) {
        const int tile = p.split_lin > 0 ? id : xcd_remap(id, ntiles);      // (1-D split launches place their work-groups themselves)
        m0 = (tile / p.tiles_n) * G2_BM;
        n0 = (tile % p.tiles_n) * G2_BN;
    };
    // loop-invariant per-lane offsets of the DMA pieces (parity of the piece index) and the scalar step between pieces q, q + 2
    const uint32_t va[2] = {g4_voff<A_KMAJOR>(p.lda, wave, lane, 0), g4_voff<A_KMAJOR>(p.lda, wave, lane, 1)};
    const uint32_t vb[2] = {g4_voff<B_KMAJOR>(p.ldb, wave, lane, 0), g4_voff<B_KMAJOR>(p.ldb, wave, lane, 1)};
    const uint32_t sa = (uint32_t)p.lda * 32u, sb = (uint32_t)p.ldb * 32u;     // 16 rows * ld * 2 bytes
    unsigned char* const my0 = lds + wave * 4096;                               // this wave's first piece inside an operand image

    // the DMA iterator: tile `did`, K step `dt`; runs up to two steps ahead of the MFMAs, across tile boundaries
    int did = tile0, dt = 0, dm = 0, dn = 0;
    tile_origin(did, dm, dn);
    G4Operand<A_KMAJOR> oa;
    G4Operand<B_KMAJOR> ob;
    oa.tile(p.A, p.lda, dm, p.M, kbeg);
    ob.tile(p.B, p.ldb, dn, p.N, kbeg);
    G4Dma d;
    auto dma_next = [&]() {
        if (++dt == nt) {
            if (did + stride < ntiles) {
                dt = 0;
                did += stride;
                tile_origin(did, dm, dn);
                oa.tile(p.A, p.lda, dm, p.M, kbeg);
                ob.tile(p.B, p.ldb, dn, p.N, kbeg);
            } else {
                dt = nt - 1;                                  // past the end: the same position again, into a stage nobody reads
            }
        } else {
            oa.advance(p.lda);
            ob.advance(p.ldb);
        }
        d.a = oa.rsrc();
        d.b = ob.rsrc();
    };
    d.a = oa.rsrc();
    d.b = ob.rsrc();
    // (round 6, measured and not kept: these pieces as asm units the compiler's wait-count pass does not see (xc_device.h buf_glds16_raw) --
    //  the weight gradients of configs[1] 2737 / 2720 us with the builtin against 2739 / 2751 us, profiles/r06_j_wgrad_raw_ab.log: the slab
    //  epilogue's compiler-inserted waits sit in the ragged-tile path only)
    auto piece_a = [&](int q, unsigned char* stage) { buf_glds16(d.a, va[q & 1], (q >> 1) ? sa : 0u, stage + (my0 - lds) + q * 1024); };
    auto piece_b = [&](int q, unsigned char* stage) { buf_glds16(d.b, vb[q & 1], (q >> 1) ? sb : 0u, stage + G2_OPER_BYTES + (my0 - lds) + q * 1024); };

#pragma unroll
    for (int q = 0; q < 4; ++q) piece_a(q, lds);
#pragma unroll
    for (int q = 0; q < 4; ++q) piece_b(q, lds);
    dma_next();
    XC_WAIT_VMEM_LE(0);
    barrier_nodrain();                                        // step 0 has landed for every wave
#pragma unroll
    for (int q = 0; q < 4; ++q) piece_a(q, lds + G2_STAGE_BYTES);    // what "C3 of step -1" would have issued: A of step 1

    u32x4 a[2][4], b[2][2];
    g3_read_frags<A_KMAJOR, B_KMAJOR>(lds, lds + G2_OPER_BYTES, wm * 128, wn * 64, 0, lane, a[0], b[0]);
    lds_wait<0>(a[0], b[0]);

    int step = 0;                                             // running K-step counter: LDS stage = step & 1
    for (int id = tile0; id < ntiles; id += stride) {
        int m0, n0;
        tile_origin(id, m0, n0);
        // acc[i][j] holds the TRANSPOSED 32 x 32 block (MFMA operands swapped): register r of lane l is
        // C[m = i-block row (l & 31)][n = j-block column (r & 3) + 8 (r >> 2) + 4 (l >> 5)]
        // (not initialised: the first k-block of the tile runs its MFMAs with C = 0 instead of 128 register moves per wave)
        f32x16 acc[4][2];

        for (int t = 0; t < nt; ++t, ++step) {
            unsigned char* cur_stage = lds + (step & 1) * G2_STAGE_BYTES;
            unsigned char* nxt_stage = lds + ((step + 1) & 1) * G2_STAGE_BYTES;
            const unsigned char* As = cur_stage;
            const unsigned char* Bs = cur_stage + G2_OPER_BYTES;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk < 3) {
                    g3_read_frags<A_KMAJOR, B_KMAJOR>(As, Bs, wm * 128, wn * 64, kk + 1, lane, a[nxt], b[nxt]);
                } else {
                    XC_WAIT_VMEM_LE(0);                          // this wave's share of DMA(step + 1) (and any epilogue stores)
                    barrier_nodrain();                           // ... everybody's; and nobody reads stage step & 1 any more
                    // (after the work-group's very last step these fragments are never used: reading them keeps the loop branch-free)
                    g3_read_frags<A_KMAJOR, B_KMAJOR>(nxt_stage, nxt_stage + G2_OPER_BYTES, wm * 128, wn * 64, 0, lane, a[nxt], b[nxt]);
                }
                sched_fence();
                if (kk == 0 && t == 0) {                         // first k-block of the tile: C = 0
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = mfma_32x32x16_bf16_zero(__builtin_bit_cast(s16x8, b[cur][j]), __builtin_bit_cast(s16x8, a[cur][i]));
                        sched_fence(); piece_b(i, nxt_stage); sched_fence();
                    }
                } else
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mma_kblock(b[cur][j], a[cur][i], acc[i][j], (bf16_t*)nullptr);   // D^T
                    // DMA issue schedule of gemm3.h: B0-3 behind the MFMA pairs of C0 (into the next stage), A0-3 behind those of C3
                    // (into the stage the barrier above has just freed, for the step after next)
                    if (kk == 0) { sched_fence(); piece_b(i, nxt_stage); sched_fence(); }
                    if (kk == 3) { sched_fence(); piece_a(i, cur_stage); sched_fence(); }
                }
                if (kk == 0) dma_next();                         // this wave's share of that stage is on its way: next position
                sched_fence();
                lds_wait<0>(a[nxt], b[nxt]);
                sched_fence();
            }
        }
        // (the 32 KiB above the two stages: 4 KiB per wave for the whole-line form of the fp32 slab epilogue)
        (void)epi.with_scratch(acc, m0, n0, lds + 2 * G2_STAGE_BYTES + wave * 4096);
    }
    XC_WAIT_VMEM_LE(0);                                       // the trailing (redundant) DMA pieces must land before the LDS is released
    epi.finish();
}

// ---- the same K step with the HBM-streamed operand three stages deep -------------------------------------------------------------
// SQ counters of the loop above on the text-tower shapes (profiles/r02_run4_gemm4_sq_pmc.txt): the MFMA pipe is busy 45 % (K = 512)
// to 65 % (K = 2048) of the time and the waves spend 35-38 % of their cycles PARKED -- at the one s_waitcnt vmcnt(0) + barrier per K
// step, i.e. waiting for the stage that was requested one step (~1.8 us) earlier: with every CU streaming, an LDS-DMA piece that has
// to come from HBM is not back within one K step.  More lookahead needs a third stage, and 3 x 64 KiB do not fit the 160 KiB of LDS;
// but only A (the activations) streams from HBM -- B (a weight panel) is re-read from L2 by every tile of its column.  So: A in a
// ring of THREE 32 KiB stages, B in two (3 x 32 + 2 x 32 = 160 KiB, all of the CU's LDS).  Per step s:
//     k-block 0:  issue A(s + 2) into A stage (s + 2) % 3   (free since the barrier of step s - 1)
//     k-block 3:  s_waitcnt vmcnt(4) -- everything but the four A pieces just issued, i.e. A(s + 1) and B(s + 1) have landed --,
//                 barrier, fragments of step s + 1, then issue B(s + 2) into B stage s & 1 (free from that barrier on)
// A piece is requested 1.75 K steps before the wait that needs it (was 1.0), B 1.0 (was 0.75).  The counted wait is safe because a
// wave's vector-memory operations -- LDS-DMA loads and stores alike -- retire in issue order on gfx9-class hardware (one vmcnt; the
// compiler's own counted waits after mixed loads and stores rely on it).
// (Measured and not kept: letting the first wait of a tile leave the previous tile's 16 / 32 stores in flight as well -- vmcnt(4 + 16)
// -- did not change the K = 512 shapes (814 vs 810 TFLOP/s, profiles/r02_run8_gemm5_counted_boundary_wait_probe.log).  Every counted
// wait here only ever leaves LDS-DMA LOADS outstanding.)
constexpr int G5_LDS_BYTES = 5 * G2_OPER_BYTES;              // 160 KiB

template <bool KMAJOR>
struct G5Iter {                                               // one operand's DMA iterator (tile, K step) with its descriptor
    int did, dt, outer0;
    G4Operand<KMAJOR> op;
};

// ABL (measurement only, XCLIP_GEMM_ABL; results are garbage): 1 no MFMA, 2 no DMA after the prologue, 4 no epilogue, 8 no fragment
// reads, 16 no barrier, 32 no counted DMA wait, 64 coalesced stores of misplaced values, 128 the row-per-lane epilogue, 256 every other CU half a tile late,
// 512 C[0..15] <- shader cycles and 10 ns ticks of work-group 0 (the effective clock).
// (an epilogue whose with_scratch may return 8 -- that many small stores left in flight -- says so with `static constexpr bool LOOSE8`)
template <class E, class = void> struct g5_loose8 { static constexpr bool value = false; };
template <class E> struct g5_loose8<E, decltype((void)E::LOOSE8)> { static constexpr bool value = E::LOOSE8; };
// (an epilogue that needs the 24 registers the loop otherwise keeps live across a tile boundary -- the first fragments of the NEXT tile, read
//  under the last k-block's MFMAs -- says so with `static constexpr bool DEFER_FRAGS`: those fragments are then read after the epilogue,
//  at the price of one exposed LDS round trip per tile)
template <class E, class = void> struct g5_defer_frags { static constexpr bool value = false; };
template <class E> struct g5_defer_frags<E, decltype((void)E::DEFER_FRAGS)> { static constexpr bool value = E::DEFER_FRAGS; };

// the held groups of a split tile boundary (g5_run): a function template of its own so that epilogues without line stores are never asked
// for store_line_groups
template <bool ON, int NG, class Epilogue>
XC_DEV void g5_store_held(const Epilogue& epi, const u32x4 (&held)[NG][4], int i0, int m0, int n0, bool stream) {
    if constexpr (ON) {
        if (stream) epi.template store_line_groups<true, NG>(held, i0, m0, n0);
        else epi.template store_line_groups<false, NG>(held, i0, m0, n0);
    }
}

template <bool A_KMAJOR, bool B_KMAJOR, class Epilogue, int ABL = 0>
XC_DEV void g5_run(const Gemm2Params& p, unsigned char* lds, Epilogue epi) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = uniform(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int kbeg = blockIdx.y * p.k_per_split;
    const int kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
    const int nt = (kend - kbeg) / G2_BK;
    const int stride = gridDim.x;
    if ((int)blockIdx.x >= ntiles || nt <= 0) return;            // (uniform over the work-group)
    const uint64_t clk0 = (ABL & (512 | 2048)) ? shader_cycles() : 0, rt0 = (ABL & 512) ? realtime_10ns() : 0;
    if ((ABL & 256) && ((blockIdx.x >> 3) & 1)) {                // measurement: every other CU of an XCD starts half a tile late
        const uint64_t until = realtime_10ns() + (uint64_t)(nt * 90 + 250);
        while (realtime_10ns() < until) nap();
    }

    // Tile order inside an XCD's contiguous run: n fastest, so that the N tiles sharing an A panel run together on one L2.  With more
    // than G4_BAND N tiles (FF1: 16) that makes every round of an XCD touch the whole B operand (4 MiB = its entire L2); the order is
    // then banded -- all M panels against N tiles [0, G4_BAND), then against the next band -- so that an XCD keeps ONE band of B
    // resident and reads each A panel once per band instead (p.band_n = 0: plain order).
    auto tile_origin = [&](int id, int& m0, int& n0) {
        const int tile = xcd_remap(id, ntiles);
        if (p.band_n > 0) {
            const int per_band = p.tiles_m * p.band_n;
            const int band = tile / per_band, rem = tile - band * per_band;
            m0 = (rem / p.band_n) * G2_BM;
            n0 = (band * p.band_n + rem % p.band_n) * G2_BN;
        } else {
            m0 = (tile / p.tiles_n) * G2_BM;
            n0 = (tile % p.tiles_n) * G2_BN;
        }
    };
    const uint32_t va[2] = {g4_voff<A_KMAJOR>(p.lda, wave, lane, 0), g4_voff<A_KMAJOR>(p.lda, wave, lane, 1)};
    const uint32_t vb[2] = {g4_voff<B_KMAJOR>(p.ldb, wave, lane, 0), g4_voff<B_KMAJOR>(p.ldb, wave, lane, 1)};
    const uint32_t sa = (uint32_t)p.lda * 32u, sb = (uint32_t)p.ldb * 32u;
    unsigned char* const ldsA = lds;                          // three A stages
    unsigned char* const ldsB = lds + 3 * G2_OPER_BYTES;      // two B stages
    const int mine = wave * 4096;                             // this wave's four 1 KiB pieces inside an operand image

    // both iterators walk the same sequence of (tile, K step) positions; past the last one they repeat it (dead stage, no branch)
    int a_id = blockIdx.x, a_t = 0, b_id = blockIdx.x, b_t = 0;
    G4Operand<A_KMAJOR> oa;
    G4Operand<B_KMAJOR> ob;
    {
        int m0, n0;
        tile_origin(a_id, m0, n0);
        oa.tile(p.A, p.lda, m0, p.M, kbeg);
        ob.tile(p.B, p.ldb, n0, p.N, kbeg);
    }
    BufRsrc ra = oa.rsrc(), rb = ob.rsrc();
    auto next_a = [&]() {
        if (++a_t == nt) {
            if (a_id + stride < ntiles) {
                a_t = 0;
                a_id += stride;
                int m0, n0;
                tile_origin(a_id, m0, n0);
                oa.tile(p.A, p.lda, m0, p.M, kbeg);
            } else {
                a_t = nt - 1;
            }
        } else {
            oa.advance(p.lda);
        }
        ra = oa.rsrc();
    };
    auto next_b = [&]() {
        if (++b_t == nt) {
            if (b_id + stride < ntiles) {
                b_t = 0;
                b_id += stride;
                int m0, n0;
                tile_origin(b_id, m0, n0);
                ob.tile(p.B, p.ldb, n0, p.N, kbeg);
            } else {
                b_t = nt - 1;
            }
        } else {
            ob.advance(p.ldb);
        }
        rb = ob.rsrc();
    };
    auto piece_a0 = [&](int q, unsigned char* stage) { buf_glds16(ra, va[q & 1], (q >> 1) ? sa : 0u, stage + mine + q * 1024); };
    auto piece_b0 = [&](int q, unsigned char* stage) { buf_glds16(rb, vb[q & 1], (q >> 1) ? sb : 0u, stage + mine + q * 1024); };
    auto piece_a = [&](int q, unsigned char* stage) { if (!(ABL & 2)) piece_a0(q, stage); };
    auto piece_b = [&](int q, unsigned char* stage) { if (!(ABL & 2)) piece_b0(q, stage); };

    // prologue: A(0), B(0), A(1), B(1) in this order; the first two must have landed before step 0
#pragma unroll
    for (int q = 0; q < 4; ++q) piece_a0(q, ldsA);
#pragma unroll
    for (int q = 0; q < 4; ++q) piece_b0(q, ldsB);
    next_a();
    next_b();
#pragma unroll
    for (int q = 0; q < 4; ++q) piece_a0(q, ldsA + G2_OPER_BYTES);
#pragma unroll
    for (int q = 0; q < 4; ++q) piece_b0(q, ldsB + G2_OPER_BYTES);
    next_a();
    next_b();
    XC_WAIT_VMEM_LE(8);
    barrier_nodrain();

    u32x4 a[2][4], b[2][2];
    g3_read_frags<A_KMAJOR, B_KMAJOR>(ldsA, ldsB, wm * 128, wn * 64, 0, lane, a[0], b[0]);
    lds_wait<0>(a[0], b[0]);

    uint64_t stamp[12] = {};                                  // (ABL & 2048: this work-group's third tile, shader cycles)
    int tile_no = 0;
    bool a_early = false;                                     // the next step's A pieces have been issued at the tile boundary
    int in_flight = 0;                                        // stores per lane the previous tile's epilogue left behind (16 or 0)
    // SPLIT (measurement: XCLIP_GEMM5_ABL & 4096).  A tile's 16 line stores are older than the B pieces the next tile's first step requests
    // and the memory counter retires in order: the second step's wait for those pieces is a wait for all 128 KiB of stores, 1.75 K steps
    // after every CU of the chip released its tile at the same moment.  Here only the first 8 stores leave at the boundary; the other 8
    // (32 registers, carried through the next tile's first K step) are issued BEHIND that step's B pieces, so the second step's wait leaves
    // them in flight and the third step's takes them: the burst is halved and has one more K step to drain.
    constexpr int HG = (ABL & 4096) ? 2 : ((ABL & 8192) ? 1 : 0);   // 32-row groups held back: rows 64 .. 127 (spills 31 registers) / rows 96 .. 127
    constexpr bool SPLIT = HG != 0;
    constexpr int HS = 4 * HG;                                // ... = that many stores per lane
    u32x4 held[HG ? HG : 1][4];
    bool pending = false, late8 = false;
    int hm0 = 0, hn0 = 0;
    int step = 0, sa3 = 0;                                    // running K-step counter (B stage = step & 1) and A stage = step % 3
    for (int id = blockIdx.x; id < ntiles; id += stride) {
        int m0, n0;
        tile_origin(id, m0, n0);
        f32x16 acc[4][2];                                     // (first k-block of the tile runs with C = 0)

        if ((ABL & 2048) && (tile_no == 2 || tile_no == 3)) stamp[tile_no == 2 ? 0 : 10] = shader_cycles();
        // one K step; FIRST = the tile's first (compiled on its own: C = 0 in its first k-block, the boundary's counted waits, the held
        // half of the previous tile's stores -- none of that is in the loop over the other steps)
        auto kstep = [&](const int t, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const int sa_next = sa3 == 2 ? 0 : sa3 + 1;       // A stage of step s + 1
            const int sa_free = sa3 == 0 ? 2 : sa3 - 1;       // A stage of step s + 2 == the one step s - 1 used
            const unsigned char* As = ldsA + sa3 * G2_OPER_BYTES;
            const unsigned char* Bs = ldsB + (step & 1) * G2_OPER_BYTES;
            unsigned char* const a_dst = ldsA + sa_free * G2_OPER_BYTES;
            unsigned char* const b_dst = ldsB + (step & 1) * G2_OPER_BYTES;
            const bool early = FIRST && a_early;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk < 3) {
                    if (!(ABL & 8)) g3_read_frags<A_KMAJOR, B_KMAJOR>(As, Bs, wm * 128, wn * 64, kk + 1, lane, a[nxt], b[nxt]);
                } else {
                    // all but this step's four A pieces: A(s + 1), B(s + 1) are in LDS (and a finished tile's stores have been taken)
                    // (first step of a tile behind an interior bf16 tile: its 16 whole-line stores are YOUNGER than A(s + 1), B(s + 1)
                    //  -- the counter retires in issue order -- and may stay in flight for one more K step)
                    if (!(ABL & 32)) {
                        if (!(ABL & 1024) && FIRST && in_flight == 16) XC_WAIT_VMEM_LE(20);
                        else if (g5_loose8<Epilogue>::value && FIRST && in_flight == 8) XC_WAIT_VMEM_LE(12);
                        else if (SPLIT && FIRST && in_flight == 16 - HS) { if (HG == 2) XC_WAIT_VMEM_LE(12); else XC_WAIT_VMEM_LE(16); }
                        else if (SPLIT && !FIRST && late8) { if (HG == 2) XC_WAIT_VMEM_LE(12); else XC_WAIT_VMEM_LE(8); }   // the held stores were issued behind B(s + 1): still in flight
                        else XC_WAIT_VMEM_LE(4);
                        if (SPLIT && !FIRST) late8 = false;
                    }
                    if (!(ABL & 16)) barrier_nodrain();          // ... for every wave; and nobody reads A stage sa3 / B stage step & 1 any more
                    if (!(ABL & 8) && !(g5_defer_frags<Epilogue>::value && t == nt - 1))
                    g3_read_frags<A_KMAJOR, B_KMAJOR>(ldsA + sa_next * G2_OPER_BYTES, ldsB + ((step + 1) & 1) * G2_OPER_BYTES, wm * 128, wn * 64,
                                                      0, lane, a[nxt], b[nxt]);
                }
                sched_fence();
                if (ABL & 1) {
                    if (kk == 0 && FIRST) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
#pragma unroll
                                for (int e = 0; e < 16; ++e) acc[i][j][e] = __builtin_bit_cast(float, a[cur][i][e & 3] ^ b[cur][j][e & 3]);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        reg_keep(a[cur][i]);
                        if (kk == 0 && !early) { sched_fence(); piece_a(i, a_dst); sched_fence(); }
                        if (kk == 3) { sched_fence(); piece_b(i, b_dst); sched_fence(); }
                    }
                } else
                if (kk == 0 && FIRST) {                          // first k-block of the tile: C = 0
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = mfma_32x32x16_bf16_zero(__builtin_bit_cast(s16x8, b[cur][j]), __builtin_bit_cast(s16x8, a[cur][i]));
                        sched_fence(); if (!early) piece_a(i, a_dst); sched_fence();
                    }
                } else
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mma_kblock(b[cur][j], a[cur][i], acc[i][j], (bf16_t*)nullptr);   // D^T
                    if (kk == 0) { sched_fence(); piece_a(i, a_dst); sched_fence(); }
                    if (kk == 3) { sched_fence(); piece_b(i, b_dst); sched_fence(); }
                }
                if (SPLIT && FIRST && kk == 3 && pending) {          // (uniform) the previous tile's held line stores, behind this step's B pieces
                    g5_store_held<SPLIT, HG ? HG : 1>(epi, held, 4 - HG, hm0, hn0, p.stream_out != 0);
                    pending = false;
                    late8 = true;
                }
                if (kk == 0 && !early) next_a();
                if (kk == 3) next_b();
                sched_fence();
                if (!(g5_defer_frags<Epilogue>::value && kk == 3 && t == nt - 1)) lds_wait<0>(a[nxt], b[nxt]);
                sched_fence();
            }
            sa3 = sa_next;
            if ((ABL & 2048) && tile_no == 2 && t < 8) stamp[1 + t] = shader_cycles();
            if ((ABL & 2048) && tile_no == 3 && FIRST) stamp[11] = shader_cycles();
            ++step;
        };
        kstep(0, std::true_type{});
        for (int t = 1; t < nt; ++t) kstep(t, std::false_type{});
        if ((ABL & 2048) && tile_no == 2) reg_keep(acc[3][1]);
        ++tile_no;
        if (ABL & 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) reg_keep(acc[i][j]);
        } else if (!(ABL & 128) && epi.packs_lines(m0, n0)) {
            // interior bf16 tile: exchange through the A stage the last step read (free until this wave's own next pieces), then
            // the four A pieces the next K step would issue in its k-block 0, THEN the stores
            unsigned char* const freed = ldsA + (sa3 == 0 ? 2 : sa3 - 1) * G2_OPER_BYTES;
            u32x4 o[4][4];
            epi.pack_lines(acc, freed + mine, o, m0, n0);
            lds_drain();
#pragma unroll
            for (int q = 0; q < 4; ++q) piece_a(q, freed);
            next_a();
            a_early = true;
            bool split_now = false;
            if constexpr (SPLIT) split_now = nt >= 3 && id + stride < ntiles;   // (a next tile exists: its first step issues the second half)
            if constexpr (SPLIT) if (split_now) {
                u32x4 (&first)[4 - HG][4] = reinterpret_cast<u32x4 (&)[4 - HG][4]>(o[0]);
                g5_store_held<SPLIT, 4 - HG>(epi, first, 0, m0, n0, p.stream_out != 0);
#pragma unroll
                for (int i = 0; i < HG; ++i)
#pragma unroll
                    for (int k = 0; k < 4; ++k) held[i][k] = o[4 - HG + i][k];
                pending = true;
                hm0 = m0;
                hn0 = n0;
                in_flight = 16 - HS;
            }
            if (!split_now) {
                if (p.stream_out) epi.template store_lines<true>(o, m0, n0);
                else epi.template store_lines<false>(o, m0, n0);
                in_flight = 16;
            }
        } else {
            // (residual / slab tiles take the same 4 KiB slice for their whole-line forms; ragged tiles and the general terms ignore it)
            in_flight = epi.with_scratch(acc, m0, n0, ldsA + (sa3 == 0 ? 2 : sa3 - 1) * G2_OPER_BYTES + mine);
            a_early = false;
        }
        if (g5_defer_frags<Epilogue>::value) {                  // the next tile's first fragments (after the work-group's last tile: never used)
            g3_read_frags<A_KMAJOR, B_KMAJOR>(ldsA + sa3 * G2_OPER_BYTES, ldsB + (step & 1) * G2_OPER_BYTES, wm * 128, wn * 64, 0, lane, a[0], b[0]);
            lds_wait<0>(a[0], b[0]);
        }
    }
    if (SPLIT && pending) g5_store_held<SPLIT, HG ? HG : 1>(epi, held, 4 - HG, hm0, hn0, p.stream_out != 0);   // (cannot happen: the split needs a next tile)
    XC_WAIT_VMEM_LE(0);                                       // trailing (redundant) pieces must land before the LDS is released
    epi.finish();
    if ((ABL & 2048) && blockIdx.x == 0 && (threadIdx.x & 63) == 0) {   // one row of 12 stamps per wave
        uint64_t* out = reinterpret_cast<uint64_t*>(p.C) + 16 + 12 * (threadIdx.x >> 6);
        for (int q = 0; q < 12; ++q) out[q] = stamp[q] - clk0;
    }
    if ((ABL & 512) && blockIdx.x == 0 && threadIdx.x == 0) {   // measurement: shader cycles and 10 ns ticks this work-group lived
        uint64_t* out = reinterpret_cast<uint64_t*>(p.C);
        out[0] = shader_cycles() - clk0;
        out[1] = realtime_10ns() - rt0;
    }
}

// ---- epilogue: registers -> global ------------------------------------------------------------------------------------------------
// Interior tiles leave as whole 128-byte lines through a wave-private LDS transposition (store_lines / store_full_res_lds /
// store_full_slab_lds below); ragged tiles keep the row-per-lane form (one output row per lane, store_bf16<false> / the general epilogue).
//
// How the tile boundary was understood (MI355X, K = 512 shapes: a tile is 8 K steps and its boundary cost ~6 us of ~20).  First the
// things that were measured and did NOT pay with the row-per-lane stores (profiles/r02_run2/3/7/8/9): holding half of the packed tile
// in 32 registers and issuing its 8 stores one at a time behind MFMA pairs of the next tile's first K step (789 vs 801 TF/s on the QKV
// forward); static s_setprio 1 for waves 4-7; sc1 / sc0 sc1 write-through stores (543 vs 801: every wait then sits on an HBM
// acknowledgement); nt stores (+4 % at N = 4096, -15 % at N = 512); a counted first wait that leaves the stores in flight; one store
// per K step of the next tile (772 vs 810); cutting the epilogue's VALU work from ~550 to ~230 instructions per wave.  Then the
// ablation harness (XCLIP_GEMM5_ABL, profiles/r02_run16): the SAME 16 store instructions per lane cost 4.1 us instead of 5.8 when each
// covers 8 rows x 128 contiguous bytes instead of 32 rows x 32 bytes -- a store instruction occupies the CU's one address path for
// about as many cycles as it touches cache lines -- and the cycle stamps (profiles/r02_run17): the K steps after a boundary ran at
// 3000-4100 cycles instead of 2440 because the next tile's A pieces queued behind 128 KiB of stores.  Hence the whole-line forms, the
// A pieces issued between pack_lines and store_lines (g5_run), and -- now that a store is cheap to issue -- the non-temporal hint for
// outputs that cannot stay in the L2s anyway (stream_out).
template <int MODE>
struct G4GemmEpilogue {
    const Gemm2Params& p;
    XC_DEV void finish() const {}

    XC_DEV uint32_t lane_off_bf16() const {
        const int lane = threadIdx.x & 63, h = lane >> 5;
        const int wave = uniform(threadIdx.x >> 6), wm = wave >> 2, wn = wave & 3;
        return ((uint32_t)(wm * 128 + (lane & 31)) * (uint32_t)p.ldc + (uint32_t)(wn * 64 + 8 * h)) * 2u;
    }

    // bf16 output, straight-line.  FULL = interior tile; otherwise rows past M fall outside the descriptor (dropped by the hardware)
    // and each store tests its 8 columns against N
    template <bool FULL>
    XC_DEV void store_bf16(f32x16 (&acc)[4][2], int m0, int n0) const {
        int rows = p.M - m0, cols = p.N - n0;
        rows = (FULL || rows > 256) ? 256 : rows;
        cols = (FULL || cols > 256) ? 256 : cols;
        const BufRsrc rc = make_rsrc(p.C + (long)m0 * p.ldc + n0, (uint32_t)(rows - 1) * (uint32_t)p.ldc * 2u + (uint32_t)cols * 2u);
        const uint32_t vc = lane_off_bf16();
        const uint32_t si = (uint32_t)p.ldc * 64u;                                  // 32 rows * ldc * 2 bytes
        const int col0 = (uniform(threadIdx.x >> 6) & 3) * 64 + 8 * ((threadIdx.x & 63) >> 5);   // this lane's first column in the tile
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint32_t pk[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pk[q][0] = f2bf_pk(acc[i][j][4 * q] * p.alpha, acc[i][j][4 * q + 1] * p.alpha);
                    pk[q][1] = f2bf_pk(acc[i][j][4 * q + 2] * p.alpha, acc[i][j][4 * q + 3] * p.alpha);
                }
                // quads (0,1) and (2,3): lower lanes end up with columns [0,8) / [16,24), upper lanes with [8,16) / [24,32)
                permlane32_swap(pk[0][0], pk[1][0]);
                permlane32_swap(pk[0][1], pk[1][1]);
                permlane32_swap(pk[2][0], pk[3][0]);
                permlane32_swap(pk[2][1], pk[3][1]);
                const u32x4 o0 = {pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
                const u32x4 o1 = {pk[2][0], pk[2][1], pk[3][0], pk[3][1]};
                if (j == 0) {
                    if (FULL || col0 < cols) buf_st16<0>(rc, vc, si * i, o0);
                    if (FULL || col0 + 16 < cols) buf_st16<32>(rc, vc, si * i, o1);
                } else {
                    if (FULL || col0 + 32 < cols) buf_st16<64>(rc, vc, si * i, o0);
                    if (FULL || col0 + 48 < cols) buf_st16<96>(rc, vc, si * i, o1);
                }
            }
        }
    }
    // interior tile, bf16 output, every store a set of whole 128-byte lines.  A store instruction of the row-per-lane form above
    // touches 32 different rows (32 bytes in each) and costs the CU's one address path ~80 cycles; 128 of them per tile were the
    // tile boundary's 5.8 us.  The same 16 stores per lane with 8 lanes side by side on a row (8 rows x 128 bytes per instruction)
    // take 4.1 us (profiles/r02_run16_gemm5_ablation.log, mask 64).  The exchange between the two lane arrangements goes through
    // 4 KiB of LDS per wave, 32 rows at a time: `scratch` is this wave's own slice of the A stage the K loop has just finished with
    // -- the only writes there are this wave's own DMA pieces, the next of which the caller issues once pack_lines is through --
    // so no work-group barrier is involved.  16-byte chunk c of row r sits at chunk position c ^ (r & 7): the 8-byte writes of the
    // 32 rows of a half-wave then spread over all banks, and so do the 16-byte reads of 8 lanes per row.
    // Two phases, because the caller puts its next four DMA pieces BETWEEN them: the CU's vector-memory path is a queue, and pieces
    // issued behind 128 KiB of stores reached the L2 ~2000 cycles late -- the second and third K step of every tile then waited for
    // them (3000-4100 cycles instead of 2440: profiles/r02_run17_gemm5_step_stamps.log).
    // one 32-row group: the wave's accumulator blocks acc_i[j] -> the four line pieces o_i[k] (rows 8 k + lane / 8 of the group)
    template <bool UNIT_ALPHA>
    XC_DEV void pack_lines_i(f32x16 (&acc_i)[2], unsigned char* scratch, u32x4 (&o_i)[4]) const {
        pack_lines_i<UNIT_ALPHA>(acc_i, scratch, o_i, (int)(threadIdx.x & 63));
    }
    // (`lane` as an argument: an epilogue that has no registers to spare passes an OPAQUE copy of it, so that the nine per-lane exchange
    //  addresses below are recomputed per tile -- a handful of VALU -- instead of being hoisted out of the tile loop and spilled: the G
    //  kernel of simloss5.h reloaded them from scratch in front of every tile's exchange, 13 spilled registers -> 1; the launch time did
    //  not move, profiles/r06_j_sim_g_*.log)
    template <bool UNIT_ALPHA>
    XC_DEV void pack_lines_i(f32x16 (&acc_i)[2], unsigned char* scratch, u32x4 (&o_i)[4], int lane) const {
        const int r = lane & 31, h = lane >> 5;
        unsigned char* const wr = scratch + r * 128 + 8 * h;                        // + chunk position * 16
        const unsigned char* const rd = scratch + (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);   // + 1024 per 8 rows
        const float al = p.alpha;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                           // columns 32 j + 8 q + 4 h + (0..3) of row r
                const float* a = reinterpret_cast<const float*>(&acc_i[j]) + 4 * q;
                const u32x2 v = UNIT_ALPHA ? u32x2{f2bf_pk(a[0], a[1]), f2bf_pk(a[2], a[3])}
                                           : u32x2{f2bf_pk(a[0] * al, a[1] * al), f2bf_pk(a[2] * al, a[3] * al)};
                *reinterpret_cast<u32x2*>(wr + (((4 * j + q) ^ (r & 7)) << 4)) = v;
            }
        }
        lds_fence();
#pragma unroll
        for (int k = 0; k < 4; ++k) o_i[k] = *reinterpret_cast<const u32x4*>(rd + k * 1024);
        lds_fence();                                                                // (the next 32 rows overwrite the slice)
    }
    template <bool UNIT_ALPHA>
    XC_DEV void pack_lines_t(f32x16 (&acc)[4][2], unsigned char* scratch, u32x4 (&o)[4][4]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) pack_lines_i<UNIT_ALPHA>(acc[i], scratch, o[i]);
    }
    // may this tile go through pack_lines / store_lines?  (uniform)
    XC_DEV bool packs_lines(int m0, int n0) const { return MODE == G4_PLAIN && (m0 + G2_BM <= p.M) && (n0 + G2_BN <= p.N); }
    XC_DEV void pack_lines(f32x16 (&acc)[4][2], unsigned char* scratch, u32x4 (&o)[4][4], int = 0, int = 0) const {
        if (p.alpha == 1.f) pack_lines_t<true>(acc, scratch, o);                    // (most products: no multiplies)
        else pack_lines_t<false>(acc, scratch, o);
    }
    template <bool NT = false>
    XC_DEV void store_lines(const u32x4 (&o)[4][4], int m0, int n0) const {
        const int lane = threadIdx.x & 63;
        const int wave = uniform(threadIdx.x >> 6), wm = wave >> 2, wn = wave & 3;
        const BufRsrc rc = make_rsrc(p.C + (long)m0 * p.ldc + n0, 255u * (uint32_t)p.ldc * 2u + 512u);
        const uint32_t vc = ((uint32_t)(wm * 128 + (lane >> 3)) * (uint32_t)p.ldc + (uint32_t)(wn * 64 + 8 * (lane & 7))) * 2u;
        const uint32_t s8 = (uint32_t)p.ldc * 16u;                                  // 8 rows * ldc * 2 bytes
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (NT) buf_st16_nt<0>(rc, vc, s8 * (uint32_t)(4 * i + k), o[i][k]);
                else buf_st16<0>(rc, vc, s8 * (uint32_t)(4 * i + k), o[i][k]);
            }
    }
    // NG of the four 32-row groups (groups i0 ... i0 + NG - 1) of store_lines: g5_run's split boundary
    template <bool NT, int NG>
    XC_DEV void store_line_groups(const u32x4 (&o)[NG][4], int i0, int m0, int n0) const {
        const int lane = threadIdx.x & 63;
        const int wave = uniform(threadIdx.x >> 6), wm = wave >> 2, wn = wave & 3;
        const BufRsrc rc = make_rsrc(p.C + (long)m0 * p.ldc + n0, 255u * (uint32_t)p.ldc * 2u + 512u);
        const uint32_t vc = ((uint32_t)(wm * 128 + (lane >> 3)) * (uint32_t)p.ldc + (uint32_t)(wn * 64 + 8 * (lane & 7))) * 2u;
        const uint32_t s8 = (uint32_t)p.ldc * 16u;
#pragma unroll
        for (int i = 0; i < NG; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (NT) buf_st16_nt<0>(rc, vc, s8 * (uint32_t)(4 * (i0 + i) + k), o[i][k]);
                else buf_st16<0>(rc, vc, s8 * (uint32_t)(4 * (i0 + i) + k), o[i][k]);
            }
    }
    // interior tile, bf16 output with a residual term, residual loads AND stores as whole 128-byte lines.  (History: the general
    // epilogue reads the residual 8 bytes at a time where it uses it, and with LDS-DMA pieces in flight every such load drained vmcnt --
    // 32 dependent round trips per tile, the FF2 forward at 785 TFLOP/s against 1127 without the skip term; a first straight-line
    // version brought the accumulators into the row-per-lane store layout with v_permlane32_swap on fp32 quads: 1029.)  Per 32-row group: the
    // residual lines (requested one group ahead) go into the wave's LDS slice in line order and come back in the accumulator layout
    // (8 bytes per 32 x 32 quad), the sum is formed in fp32 and rounded once, and the packed result takes the way of store_lines.
    // In place (residual == C) is fine: a group's lines are loaded before they are stored, by the same lanes.
    template <bool NT = false>
    XC_DEV void store_full_res_lds(f32x16 (&acc)[4][2], int m0, int n0, unsigned char* scratch) const {
        const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
        const int wave = uniform(threadIdx.x >> 6), wm = wave >> 2, wn = wave & 3;
        const BufRsrc rc = make_rsrc(p.C + (long)m0 * p.ldc + n0, 255u * (uint32_t)p.ldc * 2u + 512u);
        const BufRsrc rr = make_rsrc(p.residual + (long)m0 * p.ldr + n0, 255u * (uint32_t)p.ldr * 2u + 512u);
        const uint32_t vc = ((uint32_t)(wm * 128 + (lane >> 3)) * (uint32_t)p.ldc + (uint32_t)(wn * 64 + 8 * (lane & 7))) * 2u;
        const uint32_t vr = ((uint32_t)(wm * 128 + (lane >> 3)) * (uint32_t)p.ldr + (uint32_t)(wn * 64 + 8 * (lane & 7))) * 2u;
        const uint32_t c8 = (uint32_t)p.ldc * 16u, r8 = (uint32_t)p.ldr * 16u;      // 8 rows * ld * 2 bytes
        unsigned char* const quad = scratch + r * 128 + 8 * h;                      // accumulator layout: + chunk position * 16
        unsigned char* const line = scratch + (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);   // line layout: + 1024 per 8 rows
        const float al = p.alpha;
        u32x4 res[2][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) res[0][k] = buf_ld16<0>(rr, vr, r8 * (uint32_t)k);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < 3) {
#pragma unroll
                for (int k = 0; k < 4; ++k) res[(i + 1) & 1][k] = buf_ld16<0>(rr, vr, r8 * (uint32_t)(4 * (i + 1) + k));
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(line + k * 1024) = res[i & 1][k];
            lds_fence();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {                                       // columns 32 j + 8 q + 4 h + (0..3) of row r
                    unsigned char* const at = quad + (((4 * j + q) ^ (r & 7)) << 4);
                    const u32x2 rv = *reinterpret_cast<const u32x2*>(at);
                    const float* a = reinterpret_cast<const float*>(&acc[i][j]) + 4 * q;
                    const u32x2 v = {f2bf_pk(a[0] * al + u2f(rv[0] << 16), a[1] * al + u2f(rv[0] & 0xffff0000u)),
                                     f2bf_pk(a[2] * al + u2f(rv[1] << 16), a[3] * al + u2f(rv[1] & 0xffff0000u))};
                    *reinterpret_cast<u32x2*>(at) = v;                              // (this lane's own 8 bytes: read, then overwritten)
                }
            }
            lds_fence();
            u32x4 o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = *reinterpret_cast<const u32x4*>(line + k * 1024);
            lds_fence();                                                            // (the next group's residual overwrites the slice)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (NT) buf_st16_nt<0>(rc, vc, c8 * (uint32_t)(4 * i + k), o[k]);     // (an output the L2s cannot hold anyway: g5_run stream_out)
                else buf_st16<0>(rc, vc, c8 * (uint32_t)(4 * i + k), o[k]);
            }
        }
    }
    // interior tile, fp32 split-K slab, every store 8 rows x 128 contiguous bytes (as store_lines; the weight-gradient GEMMs are one
    // tile per work-group, and 256 row-per-lane store instructions were ~12 us of each launch): eight passes of 32 rows x 32 columns
    // through the wave's 4 KiB scratch
    XC_DEV void store_full_slab_lds(f32x16 (&acc)[4][2], int m0, int n0, unsigned char* scratch) const {
        const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
        const int wave = uniform(threadIdx.x >> 6), wm = wave >> 2, wn = wave & 3;
        const float* slab = p.partial + ((long)g2_slice(p, p.tiles_m * p.tiles_n) * p.M + m0) * p.N + n0;
        const BufRsrc rc = make_rsrc(slab, 255u * (uint32_t)p.N * 4u + 1024u);
        const uint32_t vc = ((uint32_t)(wm * 128 + (lane >> 3)) * (uint32_t)p.N + (uint32_t)(wn * 64 + 4 * (lane & 7))) * 4u;
        const uint32_t s8 = (uint32_t)p.N * 32u;                                    // 8 rows * N * 4 bytes
        unsigned char* const wr = scratch + r * 128;                                // + chunk position * 16
        const unsigned char* const rd = scratch + (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);   // + 1024 per 8 rows
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {                                       // columns 32 j + 8 q + 4 h + (0..3) of row r: chunk 2 q + h
                    const u32x4 v = {f2u(acc[i][j][4 * q]), f2u(acc[i][j][4 * q + 1]), f2u(acc[i][j][4 * q + 2]), f2u(acc[i][j][4 * q + 3])};
                    *reinterpret_cast<u32x4*>(wr + (((2 * q + h) ^ (r & 7)) << 4)) = v;
                }
                lds_fence();
                u32x4 o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = *reinterpret_cast<const u32x4*>(rd + k * 1024);
                lds_fence();
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (j == 0) buf_st16<0>(rc, vc, s8 * (uint32_t)(4 * i + k), o[k]);
                    else buf_st16<128>(rc, vc, s8 * (uint32_t)(4 * i + k), o[k]);
                }
            }
        }
    }
    // the caller owns 4 KiB of LDS per wave that nothing else touches while the epilogue runs (g4_run)
    XC_DEV int with_scratch(f32x16 (&acc)[4][2], int m0, int n0, unsigned char* scratch) const {
        const bool full = (m0 + G2_BM <= p.M) && (n0 + G2_BN <= p.N);
        if (MODE == G4_SLAB && full) {
            store_full_slab_lds(acc, m0, n0, scratch);
            return 32;
        }
        if (MODE == G4_RES && full) {
            // (round 4, measured and not kept: non-temporal stores here as in the plain epilogue -- FF2 + skip 496.5 -> 503.2 us alone,
            //  262 -> 270 us average in the step, and the LayerNorm that reads the output next did not gain: 63.6 -> 64.0 us;
            //  profiles/r04_f_ab_gemm_nt0.log, r04_f_kernel_stats.txt against r04_c_kernel_stats.txt)
            store_full_res_lds<false>(acc, m0, n0, scratch);
            return 0;                                            // (loads and stores mixed: the next wait drains them)
        }
        return (*this)(acc, m0, n0);
    }

    // -> how many vector-memory operations per lane the epilogue issued when that number is fixed (interior tiles: 16 / 32 stores),
    //    0 when it is not (ragged tiles, optional terms with their loads): the caller then drains everything at its next wait
    XC_DEV int operator()(f32x16 (&acc)[4][2], int m0, int n0) const {
        const bool full = (m0 + G2_BM <= p.M) && (n0 + G2_BN <= p.N);       // interior tile (uniform)
        if (MODE == G4_PLAIN) {                                  // (never looks at the optional-term pointers: fewer live scalars)
            if (full) { store_bf16<true>(acc, m0, n0); return 16; }
            store_bf16<false>(acc, m0, n0);
            return 0;
        }
        // (interior slab / residual tiles leave through with_scratch's whole-line forms)
        // ragged slab tiles and the optional epilogue terms: the general form (per-element range checks, clamped reads)
        (void)G3GemmEpilogue<0>{p}(acc, m0, n0);
        return 0;
    }
};

// the three-deep A ring (g5_run); epilogues as above, no deferral
// measurement only (XCLIP_GEMM5_ABL & 64): the plain epilogue's conversions and its 16 stores per lane, but every store covers
// 8 rows x 128 contiguous bytes instead of 32 rows x 32 bytes (the VALUES land in the wrong places)
struct G4ProbeEpilogue {
    const Gemm2Params& p;
    XC_DEV void finish() const {}
    XC_DEV int operator()(f32x16 (&acc)[4][2], int m0, int n0) const {
        const int lane = threadIdx.x & 63;
        const int wave = uniform(threadIdx.x >> 6), wm = wave >> 2, wn = wave & 3;
        const BufRsrc rc = make_rsrc(p.C + (long)m0 * p.ldc + n0, 255u * (uint32_t)p.ldc * 2u + 512u);
        const uint32_t vc = ((uint32_t)(wm * 128 + (lane >> 3)) * (uint32_t)p.ldc + (uint32_t)(wn * 64 + 8 * (lane & 7))) * 2u;
        const uint32_t s8 = (uint32_t)p.ldc * 16u;                                  // 8 rows * ldc * 2 bytes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint32_t pk[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pk[q][0] = f2bf_pk(acc[i][j][4 * q] * p.alpha, acc[i][j][4 * q + 1] * p.alpha);
                    pk[q][1] = f2bf_pk(acc[i][j][4 * q + 2] * p.alpha, acc[i][j][4 * q + 3] * p.alpha);
                }
                permlane32_swap(pk[0][0], pk[1][0]);
                permlane32_swap(pk[0][1], pk[1][1]);
                permlane32_swap(pk[2][0], pk[3][0]);
                permlane32_swap(pk[2][1], pk[3][1]);
                const u32x4 o0 = {pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
                const u32x4 o1 = {pk[2][0], pk[2][1], pk[3][0], pk[3][1]};
                buf_st16<0>(rc, vc, s8 * (uint32_t)(4 * i + 2 * j), o0);
                buf_st16<0>(rc, vc, s8 * (uint32_t)(4 * i + 2 * j + 1), o1);
            }
        }
        return 16;
    }
    XC_DEV bool packs_lines(int, int) const { return false; }
    XC_DEV int with_scratch(f32x16 (&acc)[4][2], int m0, int n0, unsigned char*) const { return (*this)(acc, m0, n0); }
    XC_DEV void pack_lines(f32x16 (&)[4][2], unsigned char*, u32x4 (&)[4][4], int = 0, int = 0) const {}
    template <bool NT = false> XC_DEV void store_lines(const u32x4 (&)[4][4], int, int) const {}
    template <bool NT, int NG> XC_DEV void store_line_groups(const u32x4 (&)[NG][4], int, int, int) const {}
};

template <bool A_KMAJOR, bool B_KMAJOR, int MODE, int ABL = 0>
__global__ __launch_bounds__(G2_THREADS, 2) void gemm5_kernel(Gemm2Params p) {
    XC_LDS_DYNAMIC(lds);
    if constexpr ((ABL & 64) != 0) g5_run<A_KMAJOR, B_KMAJOR, G4ProbeEpilogue, ABL>(p, lds, G4ProbeEpilogue{p});
    else g5_run<A_KMAJOR, B_KMAJOR, G4GemmEpilogue<MODE>, ABL>(p, lds, G4GemmEpilogue<MODE>{p});
}

template <bool A_KMAJOR, bool B_KMAJOR, int MODE>
__global__ __launch_bounds__(G2_THREADS, 2) void gemm4_kernel(Gemm2Params p) {
    XC_LDS_DYNAMIC(lds);
    g4_run<A_KMAJOR, B_KMAJOR>(p, lds, G4GemmEpilogue<MODE>{p});
}

}  // namespace xc
