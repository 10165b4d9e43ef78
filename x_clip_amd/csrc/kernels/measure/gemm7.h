// gemm7.h -- EXPERIMENTAL (XCLIP_GEMM=7): the ring kernel of gemm4.h (g5_run) with FOUR waves of 128 x 128 per 256 x 256 tile instead of
// eight of 128 x 64: one wave per SIMD, 256 accumulator registers per lane (the unified 512-entry file), 8 fragment reads per 16 MFMAs
// instead of 6 per 8 -- a third fewer LDS bytes per MFMA on a part that is power-limited under this kernel (DESIGN_APPENDIX.md section 3).  Same LDS
// images, same A-ring-of-three / B-ring-of-two, same whole-line epilogue; every wave stages 8 + 8 DMA pieces per K step instead of 4 + 4.
//
// Scope of the experiment: bf16 C = alpha * A[M, K] B[N, K]^T, both operands row-major, M and N multiples of 256, K a multiple of 64 with
// at least two K steps -- the host (xclip_api.hip) sends everything else to gemm4.h.
#pragma once
#include "../gemm4.h"

namespace xc {

constexpr int G7_THREADS = 256;

// per-lane byte offset of this wave's DMA piece of parity `par` (pieces q and q + 2 are 16 rows apart: that part travels in soffset)
XC_DEV uint32_t g7_voff(long ld, int wave, int lane, int par) {
    const int row = (wave * 8 + par) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    return ((uint32_t)row * (uint32_t)ld + (uint32_t)chunk * 8u) * 2u;
}

XC_DEV void g7_read_frags(const unsigned char* As, const unsigned char* Bs, int am, int bn, int kk, int lane, u32x4 (&a)[4], u32x4 (&b)[4]) {
    b[0] = g3_frag<false, 0>(Bs, bn, kk, lane);
    b[1] = g3_frag<false, 1>(Bs, bn, kk, lane);
    a[0] = g3_frag<false, 0>(As, am, kk, lane);
    a[1] = g3_frag<false, 1>(As, am, kk, lane);
    b[2] = g3_frag<false, 2>(Bs, bn, kk, lane);
    b[3] = g3_frag<false, 3>(Bs, bn, kk, lane);
    a[2] = g3_frag<false, 2>(As, am, kk, lane);
    a[3] = g3_frag<false, 3>(As, am, kk, lane);
}

__global__ __launch_bounds__(G7_THREADS, 1) void gemm7_kernel(Gemm2Params p) {
    XC_LDS_DYNAMIC(lds);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = uniform(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const int ntiles = p.tiles_m * p.tiles_n, nt = p.K / G2_BK, stride = gridDim.x;
    if ((int)blockIdx.x >= ntiles) return;

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
    const uint32_t va[2] = {g7_voff(p.lda, wave, lane, 0), g7_voff(p.lda, wave, lane, 1)};
    const uint32_t vb[2] = {g7_voff(p.ldb, wave, lane, 0), g7_voff(p.ldb, wave, lane, 1)};
    const uint32_t sa = (uint32_t)p.lda * 32u, sb = (uint32_t)p.ldb * 32u;      // 16 rows * ld * 2 bytes
    unsigned char* const ldsA = lds;                          // three A stages
    unsigned char* const ldsB = lds + 3 * G2_OPER_BYTES;      // two B stages
    const int mine = wave * 8192;                             // this wave's eight 1 KiB pieces inside an operand image

    int a_id = blockIdx.x, a_t = 0, b_id = blockIdx.x, b_t = 0;
    G4Operand<false> oa, ob;
    {
        int m0, n0;
        tile_origin(a_id, m0, n0);
        oa.tile(p.A, p.lda, m0, p.M, 0);
        ob.tile(p.B, p.ldb, n0, p.N, 0);
    }
    BufRsrc ra = oa.rsrc(), rb = ob.rsrc();
    auto next_a = [&]() {
        if (++a_t == nt) {
            if (a_id + stride < ntiles) {
                a_t = 0;
                a_id += stride;
                int m0, n0;
                tile_origin(a_id, m0, n0);
                oa.tile(p.A, p.lda, m0, p.M, 0);
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
                ob.tile(p.B, p.ldb, n0, p.N, 0);
            } else {
                b_t = nt - 1;
            }
        } else {
            ob.advance(p.ldb);
        }
        rb = ob.rsrc();
    };
    auto piece_a = [&](int q, unsigned char* stage) { buf_glds16(ra, va[q & 1], sa * (uint32_t)(q >> 1), stage + mine + q * 1024); };
    auto piece_b = [&](int q, unsigned char* stage) { buf_glds16(rb, vb[q & 1], sb * (uint32_t)(q >> 1), stage + mine + q * 1024); };

    // prologue: A(0), B(0), A(1), B(1); the first two must have landed before step 0
#pragma unroll
    for (int q = 0; q < 8; ++q) piece_a(q, ldsA);
#pragma unroll
    for (int q = 0; q < 8; ++q) piece_b(q, ldsB);
    next_a();
    next_b();
#pragma unroll
    for (int q = 0; q < 8; ++q) piece_a(q, ldsA + G2_OPER_BYTES);
#pragma unroll
    for (int q = 0; q < 8; ++q) piece_b(q, ldsB + G2_OPER_BYTES);
    next_a();
    next_b();
    XC_WAIT_VMEM_LE(16);
    barrier_nodrain();

    u32x4 a[2][4], b[2][4];
    g7_read_frags(ldsA, ldsB, wm * 128, wn * 128, 0, lane, a[0], b[0]);
    lds_wait4<0>(a[0], b[0]);

    const G4GemmEpilogue<G4_PLAIN> packer{p};
    const uint32_t vc = ((uint32_t)(wm * 128 + (lane >> 3)) * (uint32_t)p.ldc + (uint32_t)(wn * 128 + 8 * (lane & 7))) * 2u;
    const uint32_t s8 = (uint32_t)p.ldc * 16u;

    bool a_early = false, stores_behind = false;
    int step = 0, sa3 = 0;
    for (int id = blockIdx.x; id < ntiles; id += stride) {
        int m0, n0;
        tile_origin(id, m0, n0);
        f32x16 acc[2][4][2];                                  // [64-column half][32-row block][32-column block of the half]
        for (int t = 0; t < nt; ++t, ++step) {
            const int sa_next = sa3 == 2 ? 0 : sa3 + 1;
            const int sa_free = sa3 == 0 ? 2 : sa3 - 1;
            const unsigned char* As = ldsA + sa3 * G2_OPER_BYTES;
            const unsigned char* Bs = ldsB + (step & 1) * G2_OPER_BYTES;
            unsigned char* const a_dst = ldsA + sa_free * G2_OPER_BYTES;
            unsigned char* const b_dst = ldsB + (step & 1) * G2_OPER_BYTES;
            const bool early = a_early && t == 0;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk < 3) {
                    g7_read_frags(As, Bs, wm * 128, wn * 128, kk + 1, lane, a[nxt], b[nxt]);
                } else {
                    // everything but this step's eight A pieces (and, behind a tile boundary, the 32 stores + 8 early pieces that are younger
                    // than A(s + 1), B(s + 1)): the next stage is in LDS
                    if (t == 0 && stores_behind) XC_WAIT_VMEM_LE(40);
                    else XC_WAIT_VMEM_LE(8);
                    barrier_nodrain();
                    g7_read_frags(ldsA + sa_next * G2_OPER_BYTES, ldsB + ((step + 1) & 1) * G2_OPER_BYTES, wm * 128, wn * 128, 0, lane, a[nxt], b[nxt]);
                }
                sched_fence();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x16& c = acc[j >> 1][i][j & 1];
                        if (kk == 0 && t == 0) c = mfma_32x32x16_bf16_zero(__builtin_bit_cast(s16x8, b[cur][j]), __builtin_bit_cast(s16x8, a[cur][i]));
                        else c = mma_kblock(b[cur][j], a[cur][i], c, (bf16_t*)nullptr);                    // D^T
                        if (j & 1) {                                                                       // one piece behind every MFMA pair
                            const int q = 2 * i + (j >> 1);
                            if (kk == 0 && !early) { sched_fence(); piece_a(q, a_dst); sched_fence(); }
                            if (kk == 3) { sched_fence(); piece_b(q, b_dst); sched_fence(); }
                        }
                    }
                }
                if (kk == 0 && !early) next_a();
                if (kk == 3) next_b();
                sched_fence();
                lds_wait4<0>(a[nxt], b[nxt]);
                sched_fence();
            }
            sa3 = sa_next;
        }
        // tile boundary: both 64-column halves through this wave's 4 KiB of the freed A stage, the next step's A pieces, then the stores
        {
            unsigned char* const freed = ldsA + (sa3 == 0 ? 2 : sa3 - 1) * G2_OPER_BYTES;
            u32x4 o[2][4][4];
            packer.pack_lines(acc[0], freed + mine, o[0]);
            packer.pack_lines(acc[1], freed + mine, o[1]);
            lds_drain();
#pragma unroll
            for (int q = 0; q < 8; ++q) piece_a(q, freed);
            next_a();
            a_early = true;
            const BufRsrc rc = make_rsrc(p.C + (long)m0 * p.ldc + n0, 255u * (uint32_t)p.ldc * 2u + 512u);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (p.stream_out) {
                        buf_st16_nt<0>(rc, vc, s8 * (uint32_t)(4 * i + k), o[0][i][k]);
                        buf_st16_nt<128>(rc, vc, s8 * (uint32_t)(4 * i + k), o[1][i][k]);
                    } else {
                        buf_st16<0>(rc, vc, s8 * (uint32_t)(4 * i + k), o[0][i][k]);
                        buf_st16<128>(rc, vc, s8 * (uint32_t)(4 * i + k), o[1][i][k]);
                    }
                }
            stores_behind = true;
        }
    }
    XC_WAIT_VMEM_LE(0);
}

}  // namespace xc
