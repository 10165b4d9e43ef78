// gemm_small.h -- bf16 products whose OUTPUT is a few 256 x 256 tiles: the B-row products of a pooled last layer (only the CLS row of
// every text goes through the out-projection / feed-forward and their gradients: M = batch = 1024 against N, K in {512 .. 4096}), the
// latent projections, their weight gradients (reference x_clip.py:191-195,209-210,245,713-715 on [b, d] rows).  gemm4.h's persistent
// 256 x 256 kernels put such a product on 8 .. 64 of the 256 CUs and take one K step (~1.8 us) per 64 of contraction behind a two-launch
// split-K: 18 .. 54 us for 0.5 .. 4 GFLOP (profiles/r05_q_bench_shapes.log: 27 .. 170 TF/s, 0.51 ms per step over 23 launches).  These
// launches are bound by latency, not by any pipe, so the kernel here is built for latency:
//   * 64 x 64 output tiles, one work-group of 4 waves (2 x 2, one 32 x 32 MFMA block each): M = 1024, N = 512 is 128 work-groups;
//   * the operands of a K step (64 deep: 8 KiB per operand, gemm2.h's images -- a 64-row slice of the normal image, ONE panel of the
//     k-major one -- so g2_frag_normal / g2_frag_kmajor read them) arrive by LDS DMA into a ring of up to EIGHT 16 KiB stages: with
//     K = 512 the whole contraction is requested before the first wait, and the waits are counted (the memory counter retires in order);
//   * one barrier per K step (a stage is refilled behind the barrier of the step after the one that read it), no split-K, no second
//     launch; the epilogue goes from the accumulators to global memory, one output row per lane (8-byte stores; alpha and the optional
//     skip term in fp32 before the one rounding).
// Requirements (host: gs_takes): bf16, M, N, K multiples of 64, no bias / gathered rows.  Everything else stays on gemm4.h / gemm.h.
#pragma once
#include "gemm2.h"

namespace xc {

constexpr int GS_BM = 64, GS_BN = 64, GS_BK = 64, GS_THREADS = 256;
constexpr int GS_OPER_BYTES = 64 * 64 * 2;                    // one operand of one K step
constexpr int GS_STAGE_BYTES = 2 * GS_OPER_BYTES;
constexpr int GS_MAX_STAGES = 8;

struct GemmSmallParams {
    const bf16_t* A; const bf16_t* B; bf16_t* C;
    long lda, ldb, ldc;
    int M, N, K;
    float alpha;
    const bf16_t* residual; long ldr;
    int tiles_m, tiles_n;
    int stages;                                                // 2 .. GS_MAX_STAGES (dynamic LDS = stages * GS_STAGE_BYTES)
};

// per-lane byte offset of DMA piece q (0, 1) of this wave inside a 64-row operand tile whose descriptor base is the tile's first
// element at the current K position (the 64-row form of gemm4.h's g4_voff: 8 pieces of 1 KiB, two per wave)
template <bool KMAJOR>
XC_DEV uint32_t gs_voff(long ld, int wave, int lane, int q) {
    const int row = (wave * 2 + q) * 8 + (lane >> 3);          // outer row (normal) / k row (k-major) of the tile
    const int chunk = KMAJOR ? ((lane & 7) ^ (((row >> 1) & 1) << 2)) : ((lane & 7) ^ ((row >> 1) & 7));
    return ((uint32_t)row * (uint32_t)ld + (uint32_t)chunk * 8u) * 2u;
}

#define GS_WAIT_CASE(R) case R: XC_WAIT_VMEM_LE(4 * R); break;

template <bool A_KMAJOR, bool B_KMAJOR, bool RES>
__global__ __launch_bounds__(GS_THREADS) void gemm_small_kernel(GemmSmallParams p) {
    XC_LDS_DYNAMIC(lds);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = uniform(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);     // n fastest: the tiles of an A panel run on one L2
    const int m0 = (tile / p.tiles_n) * GS_BM, n0 = (tile % p.tiles_n) * GS_BN;
    const int nt = p.K / GS_BK, S = p.stages;

    // descriptors: the tile's first element of each operand at K step 0; a K step moves the base (scalar), the lanes' offsets never change
    const bf16_t* pa = A_KMAJOR ? p.A + m0 : p.A + (long)m0 * p.lda;
    const bf16_t* pb = B_KMAJOR ? p.B + n0 : p.B + (long)n0 * p.ldb;
    const long step_a = A_KMAJOR ? (long)GS_BK * p.lda : (long)GS_BK, step_b = B_KMAJOR ? (long)GS_BK * p.ldb : (long)GS_BK;
    const uint32_t ext_a = (uint32_t)(63 * p.lda + 64) * 2u, ext_b = (uint32_t)(63 * p.ldb + 64) * 2u;   // 64 rows of 64 elements, rows ld apart
    const uint32_t va[2] = {gs_voff<A_KMAJOR>(p.lda, wave, lane, 0), gs_voff<A_KMAJOR>(p.lda, wave, lane, 1)};
    const uint32_t vb[2] = {gs_voff<B_KMAJOR>(p.ldb, wave, lane, 0), gs_voff<B_KMAJOR>(p.ldb, wave, lane, 1)};
    auto issue = [&](int s) {                                  // this wave's four pieces of K step s into stage s % S
        unsigned char* st = lds + (s % S) * GS_STAGE_BYTES + wave * 2048;
        const BufRsrc ra = make_rsrc(pa + (long)s * step_a, ext_a), rb = make_rsrc(pb + (long)s * step_b, ext_b);
        buf_glds16(ra, va[0], 0u, st);
        buf_glds16(ra, va[1], 0u, st + 1024);
        buf_glds16(rb, vb[0], 0u, st + GS_OPER_BYTES);
        buf_glds16(rb, vb[1], 0u, st + GS_OPER_BYTES + 1024);
    };
    const int ahead = (S - 1 < nt) ? S - 1 : nt;
    for (int s = 0; s < ahead; ++s) issue(s);

    // acc: the TRANSPOSED 32 x 32 block (MFMA operands swapped): register r of lane l is C[m = l & 31][n = (r & 3) + 8 (r >> 2) + 4 (l >> 5)]
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int s = 0; s < nt; ++s) {
        // K steps younger than s that have been requested: min(s + S - 2, nt - 1) - s of them, four pieces each
        int young = nt - 1 - s;
        young = young < S - 2 ? young : S - 2;
        switch (young) {
            GS_WAIT_CASE(0) GS_WAIT_CASE(1) GS_WAIT_CASE(2) GS_WAIT_CASE(3) GS_WAIT_CASE(4) GS_WAIT_CASE(5)
            default: XC_WAIT_VMEM_LE(24); break;
        }
        barrier_nodrain();                                     // step s is in LDS for every wave; and everybody has read step s - 1
        if (s + S - 1 < nt) issue(s + S - 1);                  // ... whose stage takes step s + S - 1
        const unsigned char* As = lds + (s % S) * GS_STAGE_BYTES;
        const unsigned char* Bs = As + GS_OPER_BYTES;
        u32x4 a[4], b[4];                                      // (all eight reads of the step in flight before its first MFMA)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            a[kk] = A_KMAJOR ? g2_frag_kmajor(As, wm * 32, kk, lane) : g2_frag_normal(As, wm * 32, kk, lane);
            b[kk] = B_KMAJOR ? g2_frag_kmajor(Bs, wn * 32, kk, lane) : g2_frag_normal(Bs, wn * 32, kk, lane);
        }
        sched_fence();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = mma_kblock(b[kk], a[kk], acc, (bf16_t*)nullptr);     // D^T
    }

    // epilogue: lane -> output row m0 + 32 wm + (lane & 31), four 4-column groups at n0 + 32 wn + 4 (lane >> 5) + 8 q
    const int row = m0 + wm * 32 + (lane & 31);
    const int col = n0 + wn * 32 + 4 * (lane >> 5);
    bf16_t* crow = p.C + (long)row * p.ldc + col;
    u32x2 res[4];
    if (RES) {
        const bf16_t* rrow = p.residual + (long)row * p.ldr + col;
#pragma unroll
        for (int q = 0; q < 4; ++q) res[q] = *reinterpret_cast<const u32x2*>(rrow + 8 * q);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[4 * q + e] * p.alpha;
        if (RES) {
            v[0] += u2f(res[q][0] << 16); v[1] += u2f(res[q][0] & 0xffff0000u);
            v[2] += u2f(res[q][1] << 16); v[3] += u2f(res[q][1] & 0xffff0000u);
        }
        const u32x2 o = {f2bf_pk(v[0], v[1]), f2bf_pk(v[2], v[3])};
        *reinterpret_cast<u32x2*>(crow + 8 * q) = o;
    }
}
#undef GS_WAIT_CASE

// which products go this way: an output of at most GS_MAX_TILES256 tiles of the 256 x 256 kernels (a quarter of the part's CUs)
// and at most GS_MAX_FLOP of work -- a long contraction behind few tiles (a vision-tower weight gradient: 512 x 512 over 32768 rows)
// wants split-K on the big tiles, not 512 K steps at this kernel's 16 KiB per step
// and at most GS_MAX_KSTEPS K steps: the 64 rows of an operand tile sit `ld` elements apart, and at ld = 4096 (8 KiB) every row of
// every work-group's tile falls on the same few L2 channels -- 0.66 us per K step (M = 1024, N = 512, K = 4096: 42 us against the
// split-K launch's 21, whose slices start at different K positions; profiles/r05_r_gemm_small.log)
constexpr int GS_MAX_KSTEPS = 32;
constexpr int GS_MAX_TILES256 = 64;
constexpr int64_t GS_MAX_FLOP = 6000000000LL;                  // (the default of xclip_gemm_small_limit)
inline bool gs_takes(int64_t M, int64_t N, int64_t K, bool bias_or_rows, int64_t lda, int64_t ldb, int64_t max_flop) {
    if (bias_or_rows || M % GS_BM || N % GS_BN || K % GS_BK || K < GS_BK || K / GS_BK > GS_MAX_KSTEPS) return false;
    if (((M + 255) / 256) * ((N + 255) / 256) > GS_MAX_TILES256 || 2 * M * N * K > max_flop) return false;
    // 32-bit byte offsets inside a K step's descriptor
    return 64 * lda * 2 < (int64_t(1) << 31) && 64 * ldb * 2 < (int64_t(1) << 31);
}
inline int gs_stages(int64_t tiles, int64_t nt) {
    // up to eight stages (128 KiB: one work-group per CU) while the tiles fit one round; four (two work-groups per CU) beyond
    const int64_t cap = tiles > 256 ? 4 : GS_MAX_STAGES;
    const int64_t s = nt + 1 < cap ? nt + 1 : cap;
    return (int)(s < 2 ? 2 : s);
}

}  // namespace xc
