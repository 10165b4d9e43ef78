// gemm.h -- MFMA GEMM for every Linear on the path (QKV / out-proj / FF1 / FF2 / patch-embed / latent
// projections, reference x_clip.py:191-195,209-210,358,368,556,570) and their dgrad / wgrad.
//
//   C[M,N] = alpha * op(A)[M,K] * op(B)[K,N]   (+ bias[n] + residual[m,n] + addrows[rowidx[m]][n])
//
// Operand storage per operand ("k-major" = the contraction index is the slow one in memory):
//   A normal : A[m*lda + k]          A k-major : A[k*lda + m]
//   B normal : B[n*ldb + k]  (an nn.Linear weight [out, in])     B k-major : B[k*ldb + n]
//   forward  y = x W^T        : A normal , B normal            ("NT")
//   dgrad    dx = dy W        : A normal , B k-major           ("NN")
//   wgrad    dW = dy^T x      : A k-major, B k-major           ("TN", split-K over the token dimension)
//
// Tiling (gfx950): 128x128 output tile per 256-thread work-group, 2x2 waves of 64x64, each wave 2x2
// v_mfma_f32_32x32x16_bf16 (bf16) or v_mfma_f32_32x32x2_f32 (fp32, exact fma chain) accumulators (64 acc
// VGPRs).  K step = 128 bytes of the contraction (64 bf16 / 32 fp32) staged through LDS in two buffers:
// global -> registers for tile t+1 is issued before the MFMAs of tile t and written to LDS after them
// (one barrier per K step).  LDS rows are k-contiguous and padded by one 16-byte chunk (stride 144 B), which
// makes the 16-byte fragment reads conflict-free for the 16-lane service groups of ds_read_b128.
// k-major operands are transposed in registers (4 x 16-byte loads -> VEC x 4-element LDS writes).
// The accumulators are staged through LDS as fp32 for the epilogue, so every global store (and bias /
// residual / row-gather read) is a coalesced 16-byte access and the rounding to bf16 happens once, after
// the fp32 epilogue.  Work-groups are mapped to tiles XCD-aware (n fastest inside one XCD's run of tiles)
// so an A row-panel is fetched from HBM once per XCD and re-used from that XCD's L2.
//
// Split-K (wgrad): blockIdx.y selects a K range; partial tiles are written as fp32 slabs and reduced by
// splitk_reduce_kernel (deterministic, no atomics).
//
// Requirements checked by the host: contiguous dims multiples of the 16-byte chunk, 16-byte aligned bases.
#pragma once
#include "common.h"

namespace xc {

struct GemmParams {
    const void* A; const void* B; void* C;
    long lda, ldb, ldc;
    int M, N, K;
    float alpha;
    const void* bias;          // [N] or null
    const void* residual;      // [M, N] (ldr) or null
    long ldr;
    const void* addrows;       // [P, N] (ld_add) gathered by rowidx[m], or null
    const int* rowidx;
    long ld_add;
    float* partial;            // split-K slabs [splits][M][N] fp32, or null
    int k_per_split;
    int tiles_m, tiles_n;
    long batch_a = 0, batch_b = 0, batch_c = 0;   // element strides between the problems of a batched launch (blockIdx.z; xclip_gemm_batched)
};

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_THREADS = 256;

template <typename T>
struct GemmCfg {
    static constexpr int VEC = Elem<T>::VEC;
    static constexpr int BK = 8 * VEC;                  // 128 bytes of contraction per step
    static constexpr int LDT = BK + VEC;                // padded LDS row (elements)
    static constexpr int TILE = 128 * LDT;              // elements per operand buffer
    static constexpr int LDC = 128 + 4;                 // fp32 epilogue staging stride
    static constexpr int LDS_BYTES = (4 * TILE * (int)sizeof(T)) > (128 * LDC * 4) ? (4 * TILE * (int)sizeof(T)) : (128 * LDC * 4);
};

// ---- tile loaders ----------------------------------------------------------------------------------------
// normal operand: tile = 128 rows (outer index) x BK contraction elements; thread t owns chunk (t%8) of rows
// t/8 + 32*i.
template <typename T>
XC_DEV void load_normal(const T* base, long ld, int row0, int nrows, int k0, int kend, int tid, u32x4 (&r)[4]) {
    constexpr int VEC = Elem<T>::VEC;
    const int kc = k0 + (tid & 7) * VEC;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + (tid >> 3) + 32 * i;
        r[i] = (row < nrows && kc < kend) ? ld16(base + (long)row * ld + kc) : zero16();
    }
}
template <typename T>
XC_DEV void store_normal(T* tile, int tid, const u32x4 (&r)[4]) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int LDT = GemmCfg<T>::LDT;
#pragma unroll
    for (int i = 0; i < 4; ++i) st16(tile + ((tid >> 3) + 32 * i) * LDT + (tid & 7) * VEC, r[i]);
}
// k-major operand: tile = BK contraction rows x 128 outer elements; thread owns the 16-byte chunk
// cc = t % (128/VEC) of contraction rows 4*rg .. 4*rg+3, rg = t / (128/VEC).
template <typename T>
XC_DEV void load_kmajor(const T* base, long ld, int col0, int ncols, int k0, int kend, int tid, u32x4 (&r)[4]) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int CPR = 128 / VEC;
    const int col = col0 + (tid % CPR) * VEC;
    const int kb = k0 + (tid / CPR) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (col < ncols && kb + i < kend) ? ld16(base + (long)(kb + i) * ld + col) : zero16();
}
template <typename T>
XC_DEV void store_kmajor(T* tile, int tid, const u32x4 (&r)[4]) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int CPR = 128 / VEC;
    tr4_store(tile, GemmCfg<T>::LDT, (tid % CPR) * VEC, (tid / CPR) * 4, r);
}

// ---- one K step of MFMAs on a staged tile pair ------------------------------------------------------------
template <typename T>
XC_DEV void mma_tile(const T* As, const T* Bs, int wm, int wn, int lane, f32x16 (&acc)[2][2]) {
    constexpr int VEC = Elem<T>::VEC, LDT = GemmCfg<T>::LDT;
    const int r = lane & 31, kh = (lane >> 5) * VEC;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {                       // 4 k-blocks of 2*VEC per 128-byte K step
        u32x4 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[i] = ld16(As + (wm * 64 + i * 32 + r) * LDT + kk * 2 * VEC + kh);
            b[i] = ld16(Bs + (wn * 64 + i * 32 + r) * LDT + kk * 2 * VEC + kh);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mma_kblock(a[i], b[j], acc[i][j], (T*)nullptr);
    }
}

// The shared main loop: on return the 128x128 fp32 tile (un-scaled products) sits in LDS at Cs[row*LDC + col]
// and all threads have passed a barrier.
template <typename T, bool A_KMAJOR, bool B_KMAJOR>
XC_DEV void gemm_mainloop(const T* A, long lda, const T* B, long ldb, int M, int N, int m0, int n0, int kbeg, int kend,
                          unsigned char* lds) {
    typedef GemmCfg<T> Cfg;
    constexpr int BK = Cfg::BK, TILE = Cfg::TILE, LDC = Cfg::LDC;
    T* As = reinterpret_cast<T*>(lds);              // [2][128][LDT]
    T* Bs = As + 2 * TILE;                          // [2][128][LDT]
    float* Cs = reinterpret_cast<float*>(lds);      // [128][LDC] (re-uses the operand buffers after the K loop)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nt = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 ra[4], rb[4];
    auto fetch = [&](int t) {
        const int k0 = kbeg + t * BK;
        if (A_KMAJOR) load_kmajor<T>(A, lda, m0, M, k0, kend, tid, ra);
        else load_normal<T>(A, lda, m0, M, k0, kend, tid, ra);
        if (B_KMAJOR) load_kmajor<T>(B, ldb, n0, N, k0, kend, tid, rb);
        else load_normal<T>(B, ldb, n0, N, k0, kend, tid, rb);
    };
    auto stage = [&](int buf) {
        if (A_KMAJOR) store_kmajor<T>(As + buf * TILE, tid, ra);
        else store_normal<T>(As + buf * TILE, tid, ra);
        if (B_KMAJOR) store_kmajor<T>(Bs + buf * TILE, tid, rb);
        else store_normal<T>(Bs + buf * TILE, tid, rb);
    };

    if (nt > 0) {
        fetch(0);
        stage(0);
    }
    sync();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) fetch(t + 1);                       // HBM/L2 latency hides under this tile's MFMAs
        mma_tile<T>(As + buf * TILE, Bs + buf * TILE, wm, wn, lane, acc);
        if (t + 1 < nt) stage(buf ^ 1);
        sync();
    }
    // accumulators -> LDS (fp32); the trailing barrier of the K loop guarantees nobody still reads the
    // operand buffers.  Consecutive lanes write consecutive columns: conflict-free.
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Cs[(wm * 64 + i * 32 + mfma_row(r, lane)) * LDC + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
    sync();
}

template <typename T, bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
    typedef GemmCfg<T> Cfg;
    constexpr int VEC = Cfg::VEC, LDC = Cfg::LDC;
    XC_LDS_DYNAMIC(lds);
    const float* Cs = reinterpret_cast<const float*>(lds);
    const int tid = threadIdx.x;
    const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int m0 = (tile / p.tiles_n) * GEMM_BM, n0 = (tile % p.tiles_n) * GEMM_BN;
    const int kbeg = blockIdx.y * p.k_per_split;
    const int kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
    gemm_mainloop<T, A_KMAJOR, B_KMAJOR>(reinterpret_cast<const T*>(p.A) + (long)blockIdx.z * p.batch_a, p.lda,
                                         reinterpret_cast<const T*>(p.B) + (long)blockIdx.z * p.batch_b, p.ldb, p.M, p.N, m0, n0, kbeg, kend, lds);

    // ---- epilogue: fp32 tile in LDS -> coalesced 16-byte stores ------------------------------------------
    constexpr int CPR = 128 / VEC;                          // output chunks per tile row
    if (p.partial != nullptr) {
        float* slab = p.partial + (long)blockIdx.y * p.M * p.N;
        for (int id = tid; id < 128 * 32; id += GEMM_THREADS) {
            const int row = id >> 5, col = (id & 31) * 4;
            const int gm = m0 + row, gn = n0 + col;
            if (gm < p.M && gn < p.N) st16(slab + (long)gm * p.N + gn, ld16(Cs + row * LDC + col));
        }
        return;
    }
    T* C = reinterpret_cast<T*>(p.C) + (long)blockIdx.z * p.batch_c;
    const T* bias = reinterpret_cast<const T*>(p.bias);
    const T* resid = reinterpret_cast<const T*>(p.residual);
    const T* addr = reinterpret_cast<const T*>(p.addrows);
    for (int id = tid; id < 128 * CPR; id += GEMM_THREADS) {
        const int row = id / CPR, col = (id % CPR) * VEC;
        const int gm = m0 + row, gn = n0 + col;
        if (gm >= p.M || gn >= p.N) continue;
        float v[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] = Cs[row * LDC + col + k] * p.alpha;
        if (bias != nullptr) {
            float t[VEC];
            load_vec<T>(bias + gn, t);
#pragma unroll
            for (int k = 0; k < VEC; ++k) v[k] += t[k];
        }
        if (addr != nullptr) {
            float t[VEC];
            load_vec<T>(addr + (long)p.rowidx[gm] * p.ld_add + gn, t);
#pragma unroll
            for (int k = 0; k < VEC; ++k) v[k] += t[k];
        }
        if (resid != nullptr) {
            float t[VEC];
            load_vec<T>(resid + (long)gm * p.ldr + gn, t);
#pragma unroll
            for (int k = 0; k < VEC; ++k) v[k] += t[k];
        }
        store_vec<T>(C + (long)gm * p.ldc + gn, v);
    }
}

// Sum split-K slabs, scale, convert:  C[m, n] = alpha * sum_s partial[s][m][n] (+ residual[m, n]: the skip term of a product whose row
// tail was split over K, xclip_api.hip gemm2_tail; residual may alias C)
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, T* C, long ldc,
                                                            int M, int N, int splits, float alpha, const T* residual = nullptr, long ldr = 0) {
    const long total = (long)M * (N / 4);
    for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
        const long m = id / (N / 4);
        const int n = (int)(id % (N / 4)) * 4;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < splits; ++k) {
            float t[4];
            load_vec<float>(partial + ((long)k * M + m) * N + n, t);
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] += t[q];
        }
        if (residual != nullptr) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] = s[q] * alpha + to_f32(residual[m * ldr + n + q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) C[m * ldc + n + q] = from_f32<T>(s[q]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) C[m * ldc + n + q] = from_f32<T>(s[q] * alpha);
        }
    }
}

}  // namespace xc
