// gemm3.h -- the scheduled form of the bf16 production GEMM (same contract, LDS images, tile shape and two-stage K loop
// as gemm2.h: 256 x 256 output tile, 8 waves of 128 x 64, K step 64, LDS DMA staging, transpose reads for k-major
// operands).  Two things change, both driven by measurements of gemm2.h on MI355X (about 8 us fixed + 2.1 us per K step
// per tile against 0.85 us of MFMA work per K step):
//   * the fragment reads are software pipelined by hand: the six ds_reads of k-block kk+1 are issued BEFORE the eight
//     MFMAs of k-block kk and the order is pinned (sched_barrier), so the compiler's counted lgkmcnt leaves them in flight
//     under the MFMAs.  (Left alone, hipcc emits read -> lgkmcnt(0) -> 2-4 MFMAs -> read ... and exposes the LDS latency
//     eight times per K step; a ring of four 32-deep stages with counted vmcnt was measured SLOWER than the two-stage
//     loop -- the barriers, not the DMA latency, were the cost.)
//   * the epilogue goes straight from the accumulators to global memory: the MFMAs are issued with swapped operands
//     (D^T = B-fragment x A-fragment), so a lane owns ONE output row and 4 consecutive columns per register quad;
//     v_permlane32_swap pairs the quads into 16-byte stores (cdna_hip_programming.md T21).  No LDS staging and no
//     epilogue barriers; bias / row-gather / residual terms are added in fp32 before the single rounding to bf16.
#pragma once
#include "gemm2.h"

namespace xc {

template <bool KMAJOR>
XC_DEV u32x4 g3_frag(const unsigned char* tile, int o0, int kk, int lane) {
    return KMAJOR ? g2_frag_kmajor(tile, o0, kk, lane) : g2_frag_normal(tile, o0, kk, lane);
}
XC_DEV void g3_add4(float (&v)[4], const bf16_t* p) {
    const u32x2 t = *reinterpret_cast<const u32x2*>(p);
    v[0] += u2f(t[0] << 16); v[1] += u2f(t[0] & 0xffff0000u); v[2] += u2f(t[1] << 16); v[3] += u2f(t[1] & 0xffff0000u);
}

// L2 prefetch stream (one 4-byte LDS-DMA touch per future 128-byte line, G3_PD K steps ahead).  MEASURED SLOWER on MI355X
// (qkv fwd 644 -> 603 TF/s, wgrad 900 -> 742): vmcnt retires in order, so every K step then waits on a one-step-old HBM
// access.  Kept behind this switch as a documented negative result.
constexpr bool G3_L2_PREFETCH = false;
constexpr int G3_PD = 3;                                      // L2 prefetch distance in K steps
constexpr int G3_LDS_BYTES = G2_LDS_BYTES + 8 * 256;          // + one 256-byte prefetch sink per wave

// one 4-byte touch per lane = one 128-byte line of a future operand tile (256 lines per tile): pulls the line into this
// XCD's L2 ~G3_PD K steps before the 16-byte DMA asks for it (first touches otherwise pay HBM latency EVERY K step, in all
// the N-tiles that share the A panel at once)
template <bool KMAJOR>
XC_DEV void g3_prefetch(const bf16_t* X, long ld, int outer0, int nouter, int k0, int line, unsigned char* sink) {
    const bf16_t* src;
    if (!KMAJOR) {
        int g = outer0 + line;
        g = g < nouter ? g : nouter - 1;
        src = X + (long)g * ld + k0;
    } else {
        int g = outer0 + (line & 3) * 64;
        g = g < nouter ? g : nouter - 8;
        src = X + (long)(k0 + (line >> 2)) * ld + g;
    }
    glds4(src, sink);
}

// The persistent tile loop shared by the GEMM and by the contrastive-head kernels (simloss3.h): `epi(acc, m0, n0, full)` is
// called once per finished 256 x 256 tile with the TRANSPOSED accumulators (see below) and must report how many global
// stores per lane it issued when `full` (interior tile) so the next tile's first wait can leave exactly those in flight.
template <bool A_KMAJOR, bool B_KMAJOR, int ABL, class Epilogue>
XC_DEV void g3_run(const Gemm2Params& p, unsigned char* lds, Epilogue epi) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = uniform(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int kbeg = blockIdx.y * p.k_per_split;
    const int kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
    const int nt = (kend - kbeg) / G2_BK;
    const int h = lane >> 5;

    // Persistent: this work-group walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... (tile ids are XCD-remapped so that
    // concurrently running work-groups of one XCD share A row-panels in that XCD's L2).  The K steps of consecutive tiles
    // form ONE stream through the two LDS stages: the DMA of the next tile's first K step is issued before this tile's
    // epilogue, and the epilogue's stores drain under the next tile's MFMAs (counted vmcnt, bare barriers).
    auto tile_origin = [&](int id, int& m0, int& n0) {
        const int tile = xcd_remap(id, ntiles);
        m0 = (tile / p.tiles_n) * G2_BM;
        n0 = (tile % p.tiles_n) * G2_BN;
    };
    auto stage = [&](int buf, int m0, int n0, int k0) {
        unsigned char* base = lds + buf * G2_STAGE_BYTES;
        g2_stage<A_KMAJOR>(p.A, p.lda, m0, p.M, k0, base, wave, lane);
        g2_stage<B_KMAJOR>(p.B, p.ldb, n0, p.N, k0, base + G2_OPER_BYTES, wave, lane);
    };
    int step = 0;                                             // running K-step counter: LDS stage = step & 1
    int m0, n0;
    if ((int)blockIdx.x < ntiles && nt > 0) {
        tile_origin(blockIdx.x, m0, n0);
        stage(0, m0, n0, kbeg);
    }
    int stores_pending = 0;                                  // epilogue stores per lane of the previous tile still in flight
    for (int id = blockIdx.x; id < ntiles; id += gridDim.x) {
    tile_origin(id, m0, n0);
    // acc[i][j] holds the TRANSPOSED 32 x 32 block: register r of lane l is
    // C[m = m0 + wm*128 + i*32 + (l & 31)][n = n0 + wn*64 + j*32 + (r & 3) + 8 (r >> 2) + 4 (l >> 5)]
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int t = 0; t < nt; ++t, ++step) {
        // this K step's DMA (issued one step ago) has landed for every wave; the previous tile's epilogue stores
        // (16 or 32 per lane, younger than that DMA) may stay in flight
        // (in issue order: 8 DMA pieces, 1 prefetch touch, then possibly the epilogue stores; vmcnt retires in order)
        if (G3_L2_PREFETCH) {
            if (stores_pending == 16) XC_WAIT_VMEM_LE(17);
            else if (stores_pending == 32) XC_WAIT_VMEM_LE(33);
            else if (step > 0) XC_WAIT_VMEM_LE(1);
            else XC_WAIT_VMEM_LE(0);
        } else {
            if (stores_pending == 16) XC_WAIT_VMEM_LE(16);
            else if (stores_pending == 32) XC_WAIT_VMEM_LE(32);
            else XC_WAIT_VMEM_LE(0);
        }
        stores_pending = 0;
        barrier_nodrain();
        const unsigned char* As = lds + (step & 1) * G2_STAGE_BYTES;
        const unsigned char* Bs = As + G2_OPER_BYTES;
        // next K step of this tile -- or the first K step of the NEXT tile -- goes into the other stage (its last readers
        // passed the barrier above).  Its 8 DMA pieces per wave are NOT issued here in a burst (measured: 47 % of the wave
        // cycles were issue stalls with all 8 waves queueing 8 pieces each right after the barrier) but one at a time
        // between the MFMAs of the first two k-blocks below.
        bool dma = false;
        int dm = m0, dn = n0, dk = kbeg;
        if (!(ABL & 2)) {
            if (t + 1 < nt) {
                dma = true; dk = kbeg + (t + 1) * G2_BK;
            } else if (id + (int)gridDim.x < ntiles) {
                dma = true; tile_origin(id + gridDim.x, dm, dn);
            }
        }
        unsigned char* dbase = lds + ((step + 1) & 1) * G2_STAGE_BYTES;
        // L2 prefetch of the tiles G3_PD steps ahead (possibly in the next tile): waves 0-3 touch A's 256 lines, waves 4-7
        // B's when B streams from HBM too (wgrad), else A's once more; exactly ONE instruction per wave per K step
        if (G3_L2_PREFETCH) {
            int tp = t + G3_PD, mp = m0, np = n0;
            bool ok = true;
            if (tp >= nt) {
                tp -= nt;
                const int idn = id + (int)gridDim.x;
                ok = idn < ntiles && tp < nt;
                if (ok) tile_origin(idn, mp, np);
            }
            const int kp = kbeg + (ok ? tp : t) * G2_BK;       // nothing ahead: re-touch the current step (count stays uniform)
            unsigned char* sink = lds + G2_LDS_BYTES + wave * 256;
            if (wave < 4 || !(A_KMAJOR && B_KMAJOR)) g3_prefetch<A_KMAJOR>(p.A, p.lda, mp, p.M, kp, tid & 255, sink);
            else g3_prefetch<B_KMAJOR>(p.B, p.ldb, np, p.N, kp, tid & 255, sink);
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
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) if (!(ABL & 8)) b[0][j] = g3_frag<B_KMAJOR>(Bs, wn * 64 + j * 32, 0, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) if (!(ABL & 8)) a[0][i] = g3_frag<A_KMAJOR>(As, wm * 128 + i * 32, 0, lane);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk < 3 && !(ABL & 8)) {                                        // fragments of the NEXT k-block first ...
#pragma unroll
                for (int j = 0; j < 2; ++j) b[nxt][j] = g3_frag<B_KMAJOR>(Bs, wn * 64 + j * 32, kk + 1, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i) a[nxt][i] = g3_frag<A_KMAJOR>(As, wm * 128 + i * 32, kk + 1, lane);
            }
            sched_fence();                                                   // ... then this k-block's MFMAs, order pinned
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (ABL & 1) { asm volatile("" :: "v"(a[cur][i]), "v"(b[cur][j])); }
                    else acc[i][j] = mma_kblock(b[cur][j], a[cur][i], acc[i][j], (bf16_t*)nullptr);   // D^T
                }
                if (kk < 2 && dma) {                                         // one DMA piece behind every MFMA pair
                    sched_fence();
                    if (kk == 0) { if (!(ABL & 16)) g2_stage_piece<A_KMAJOR>(p.A, p.lda, dm, p.M, dk, dbase, wave, lane, i); }
                    else if (!(ABL & 32)) g2_stage_piece<B_KMAJOR>(p.B, p.ldb, dn, p.N, dk, dbase + G2_OPER_BYTES, wave, lane, i);
                    sched_fence();
                }
            }
            sched_fence();
        }
    }

    stores_pending = epi(acc, m0, n0);
    }   // tile loop
}

// ---- the GEMM epilogue: registers -> global, one output row per lane ------------------------------------------------------------
template <int ABL>
struct G3GemmEpilogue {
    const Gemm2Params& p;
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
                    float* slab = p.partial + ((long)blockIdx.y * p.M + gmc) * p.N;
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
// stores removed, 8 = LDS fragment reads removed, 16 / 32 = the A / B operand's DMA removed; 0 = the product kernel.
// Per K step per CU at M=263168 N=512 K=2048 (profiles/r01_step7_gemm_ablation_bitmask.log): full 2.39 us; MFMAs alone 1.15;
// LDS reads alone 0.69; DMA alone 1.83 (A only 1.41, B only 0.91); skeleton 0.14.  Deeper lookahead for the DMA (a ring of four
// 32-deep stages; an A ring of three + B ring of two 64-deep stages filling all 160 KiB, counted vmcnt) measured 3 - 8 % SLOWER
// than this two-stage loop, s_setprio around the MFMA groups null: the residual is the fine-grained MFMA / LDS / DMA interleave.
template <bool A_KMAJOR, bool B_KMAJOR, int ABL = 0>
__global__ __launch_bounds__(G2_THREADS, 2) void gemm3_kernel(Gemm2Params p) {
    XC_LDS_DYNAMIC(lds);
    g3_run<A_KMAJOR, B_KMAJOR, ABL>(p, lds, G3GemmEpilogue<ABL>{p});
}

}  // namespace xc
