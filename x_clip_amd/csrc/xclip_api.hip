// xclip_api.hip -- the extern "C" boundary (include/xclip.h): argument checks, dtype / shape dispatch and kernel
// launches.  No allocation, no synchronisation, no global mutable state besides the thread-local error string.
#include "xc_device.h"

#include "api_common.h"
#include "kernels/colnorm.h"
#include "kernels/filip.h"
#include "kernels/filip5.h"
#include "kernels/gemm.h"
#include "kernels/gemm2.h"
#include "kernels/gemm3.h"
#include "kernels/gemm4.h"
#include "kernels/gemm8.h"
#include "kernels/gemm9.h"
#include "kernels/gemm_small.h"
#ifdef XCLIP_MEASURE                                             // negative-result experiments, measurement build only (DESIGN_APPENDIX.md 6b)
#include "kernels/measure/gemm6.h"
#include "kernels/measure/gemm7.h"
#endif
#include "kernels/rows.h"
#include "kernels/simloss.h"
#include "kernels/simloss5.h"
#include "kernels/sort.h"
#include "kernels/tokens.h"

using namespace xc;
using namespace xcapi;

namespace xcapi {

thread_local char g_err[512] = "";

int fail(const char* fn, const char* what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", fn, what);
    return 1;
}
int check_launch(const char* fn) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: HIP error: %s", fn, hipGetErrorString(e));
        return 2;
    }
    return 0;
}

}  // namespace xcapi

namespace {

// smallest power of two >= ceil(dim / (64 * VEC)), i.e. 16-byte chunks per lane when one wave holds a row
inline int chunks_per_lane(int64_t dim, int vec) {
    const int64_t need = (dim / vec + 63) / 64;
    int c = 1;
    while (c < need) c <<= 1;
    return c;
}

// Non-temporal row accesses, per kernel (bit 1 GEGLU-LayerNorm backward, 2 GEGLU-LayerNorm forward, 4 LayerNorm forward, 8 LayerNorm backward,
// 16 chained forward, 32 chained backward).  Inside the step (profiles/r04_m_kernel_stats_nt{0,63}.txt): GEGLU forward -8 %, GEGLU backward -2 %,
// plain forward / backward / chained backward -1 ... -2 %, chained forward +1 % -- the hint stays off there.  XCLIP_ROWS_NT (measurement build)
// overrides the mask.
constexpr int ROWS_NT = 1 | 2 | 4 | 8 | 32;
inline bool rows_nt_flipped(int bit) { return ((measure_env("XCLIP_ROWS_NT", ROWS_NT) ^ ROWS_NT) & bit) != 0; }

template <typename T, int MAXC>
void launch_ln_fwd(const void* x, int64_t ldx, const void* g, const void* res, void* y, int64_t ldy, int y_grp, float* mean,
                   float* rstd, int rows, int dim, float eps, int geglu, hipStream_t st) {
    dim3 grid((rows + 3) / 4), block(256);
#ifdef XCLIP_MEASURE
    if constexpr (MAXC <= 2) {
        // measured and not kept (profiles/r03_k_ln_fwd_rows_per_wave.log): several rows per wave with all loads ahead of the reductions
        // (rows.h ln_fwd_rows_kernel; XCLIP_LN_FWD = 1 / 2 / 4 rows per wave).  263 k x 512: 108.7 us with the one-row kernel, 105 / 99 /
        // 117 us; with the residual 137 against 141 / 147 / 168 us; 1.18 M x 1024: 796 (6.1 TB/s) against 904 / 1000 / 1514 us.  The
        // one-row kernel is at the copy ceiling when it runs alone; its lower in-step figure is not a property of the kernel.
        static const int rpw = measure_env("XCLIP_LN_FWD", 0);
        if (!geglu && rpw > 0) {
#define XC_LNR(R) { dim3 g2((rows + 4 * R - 1) / (4 * R)); hipLaunchKernelGGL((ln_fwd_rows_kernel<T, MAXC, R>), g2, block, 0, st, (const T*)x, (long)ldx, (const T*)g, (const T*)res, (T*)y, mean, rstd, rows, dim, eps, (long)ldy, y_grp); return; }
            if (rpw == 1) XC_LNR(1)
            if (rpw == 4) XC_LNR(4)
            XC_LNR(2)
#undef XC_LNR
        }
    }
#endif
    constexpr bool NTG = (ROWS_NT & 2) != 0, NTP = (ROWS_NT & 4) != 0;
#ifdef XCLIP_MEASURE
    if constexpr (sizeof(T) == 2 && MAXC <= 4) {
        if (geglu && rows_nt_flipped(2)) {
            hipLaunchKernelGGL((ln_fwd_kernel<T, MAXC, true, !NTG>), grid, block, 0, st, (const T*)x, (long)ldx, (const T*)g, (const T*)res,
                               (T*)y, mean, rstd, rows, dim, eps, (long)ldy, y_grp);
            return;
        }
        if (!geglu && rows_nt_flipped(4)) {
            hipLaunchKernelGGL((ln_fwd_kernel<T, MAXC, false, !NTP>), grid, block, 0, st, (const T*)x, (long)ldx, (const T*)g, (const T*)res,
                               (T*)y, mean, rstd, rows, dim, eps, (long)ldy, y_grp);
            return;
        }
    }
#endif
    if (geglu)
        hipLaunchKernelGGL((ln_fwd_kernel<T, MAXC, true, NTG>), grid, block, 0, st, (const T*)x, (long)ldx, (const T*)g, (const T*)res,
                           (T*)y, mean, rstd, rows, dim, eps, (long)ldy, y_grp);
    else
        hipLaunchKernelGGL((ln_fwd_kernel<T, MAXC, false, NTP>), grid, block, 0, st, (const T*)x, (long)ldx, (const T*)g,
                           (const T*)res, (T*)y, mean, rstd, rows, dim, eps, (long)ldy, y_grp);
}
constexpr int LN_BWD_MAX_BLOCKS = 4096;          // MI355X, 263k x 512 rows: 1024 work-groups 180 us, 2048: 189, 4096: 163, 8192: 165
inline int ln_bwd_blocks(int64_t rows) {
    const int64_t b = (rows + 3) / 4;
    return (int)(b > LN_BWD_MAX_BLOCKS ? LN_BWD_MAX_BLOCKS : b);
}
// the GEGLU variant: 80 VGPRs and 20 KB of LDS per work-group = six resident work-groups per CU; the grid is a whole number of such
// rounds (never more than ln_bwd_blocks(rows): the workspace is sized by that)
inline int ln_geglu_bwd_blocks(int64_t rows) {
    const int64_t b = (rows + 1) / 2, cap = ln_bwd_blocks(rows);
    const int per_round = 6 * xc_policy_cus();
    int64_t want = (int64_t)measure_env("XCLIP_LNG_BLOCKS", 2 * per_round);
    if (want > cap) want = cap;
    return (int)(b < want ? (b < 1 ? 1 : b) : want);
}

template <typename T, int MAXC>
int launch_ln_bwd(const void* dy, const void* x, int64_t ldx, const void* g, const float* mean, const float* rstd,
                  const void* dres, void* dx, int64_t lddx, float* dg, int rows, int dim, int geglu, hipStream_t st) {
    const int blocks = ln_bwd_blocks(rows);
    dim3 grid(blocks), block(256);
    const size_t lds = (size_t)3 * dim * sizeof(float);
    constexpr bool NTG = (ROWS_NT & 1) != 0, NTP = (ROWS_NT & 8) != 0;
#define XC_LNB(KERNEL, GRID, LDS) do { XC_ALLOW_LDS((KERNEL), LDS); hipLaunchKernelGGL((KERNEL), GRID, block, LDS, st, (const T*)dy, (const T*)x, (long)ldx, (const T*)g, mean, rstd, (T*)dx, (long)lddx, dg, rows, dim); } while (0)
#define XC_LNB_RES(KERNEL, GRID, LDS) do { XC_ALLOW_LDS((KERNEL), LDS); hipLaunchKernelGGL((KERNEL), GRID, block, LDS, st, (const T*)dy, (const T*)x, (long)ldx, (const T*)g, mean, rstd, (const T*)dres, (T*)dx, (long)lddx, dg, rows, dim); } while (0)
    if (geglu) {
        // rows wider than 3 chunks per lane are shared by two waves (register budget: three waves per SIMD)
        constexpr int SPLIT = (MAXC >= 4 && MAXC % 2 == 0) ? 2 : 1;
        constexpr int C = MAXC / SPLIT;
        const bool split = SPLIT == 2 && (dim / Elem<T>::VEC) % 2 == 0;
        // four waves' dg partials of their column part + the row sums + gamma
        const size_t lds2 = ((size_t)4 * (dim / (split ? 2 : 1)) + 16) * sizeof(float) + (size_t)dim * sizeof(T)
                            + (measure_env("XCLIP_LNG_LDSPAD", 0) && split ? (size_t)2 * dim * sizeof(float) : 0);   // what rounds 1-3 requested
        dim3 ggrid(ln_geglu_bwd_blocks(rows));
        if (split) {
#ifdef XCLIP_MEASURE
            if (rows_nt_flipped(1)) {
                XC_LNB((ln_geglu_bwd_kernel<T, C, SPLIT, !NTG>), ggrid, lds2);
                return (int)ggrid.x;
            }
#endif
            XC_LNB((ln_geglu_bwd_kernel<T, C, SPLIT, NTG>), ggrid, lds2);
            return (int)ggrid.x;
        } else if (MAXC <= 4) {
            constexpr int C1 = MAXC <= 4 ? MAXC : 1;
            XC_LNB((ln_geglu_bwd_kernel<T, C1, 1, NTG>), ggrid, lds2);
            return (int)ggrid.x;
        } else {                                               // very wide rows with an odd chunk count: the generic two-pass kernel
            XC_LNB_RES((ln_bwd_kernel<T, MAXC, true, NTG>), grid, lds);
        }
    } else {
#ifdef XCLIP_MEASURE
        if constexpr (sizeof(T) == 2 && MAXC <= 2) {
            if (rows_nt_flipped(8)) {
                XC_LNB_RES((ln_bwd_kernel<T, MAXC, false, !NTP>), grid, lds);
                return blocks;
            }
        }
#endif
        XC_LNB_RES((ln_bwd_kernel<T, MAXC, false, NTP>), grid, lds);
    }
#undef XC_LNB
#undef XC_LNB_RES
    return blocks;
}

// dispatch on (dtype, chunks per lane); F is a macro taking (T, MAXC)
#define XC_DISPATCH_ROW(dtype, cpl, F)                                   \
    do {                                                                 \
        if ((dtype) == XCLIP_BF16) {                                     \
            switch (cpl) {                                               \
                case 1: F(bf16_t, 1); break;                             \
                case 2: F(bf16_t, 2); break;                             \
                case 4: F(bf16_t, 4); break;                             \
                case 8: F(bf16_t, 8); break;                             \
                default: return fail(__func__, "row too wide (bf16 rows up to 4096 elements)"); \
            }                                                            \
        } else {                                                         \
            switch (cpl) {                                               \
                case 1: F(float, 1); break;                              \
                case 2: F(float, 2); break;                              \
                case 4: F(float, 4); break;                              \
                case 8: F(float, 8); break;                              \
                case 16: F(float, 16); break;                            \
                default: return fail(__func__, "row too wide (fp32 rows up to 4096 elements)"); \
            }                                                            \
        }                                                                \
    } while (0)

template <typename T, bool AK, bool BK_>
void launch_gemm(const GemmParams& p, int splits, hipStream_t st) {
    XC_ALLOW_LDS((gemm_kernel<T, AK, BK_>), GemmCfg<T>::LDS_BYTES);
    dim3 grid(p.tiles_m * p.tiles_n, splits), block(GEMM_THREADS);
    hipLaunchKernelGGL((gemm_kernel<T, AK, BK_>), grid, block, GemmCfg<T>::LDS_BYTES, st, p);
}

// Which bf16 GEMM runs.  Product (0): g5_run (A in a ring of three LDS stages) for the layouts whose B operand is a weight panel that
// lives in L2 -- forward (NT) and dgrad (NN): +2 ... +6 % there -- and g4_run for wgrad (TN), where BOTH operands stream from HBM and
// the deeper A ring measured 3-11 % SLOWER (profiles/r02_run5_gemm5_ring_probe.log).  Measurement build only (XCLIP_GEMM, read once;
// A/B runs): 2 = gemm2.h (two-phase), 3 = gemm3.h (first scheduled kernel), 4 = g4_run everywhere, 5 = g5_run everywhere, 6 / 7 = the
// experiments under kernels/measure/.
inline int gemm_generation() {
    static const int v = [] { const int e = measure_env("XCLIP_GEMM", 0); return (e >= 2 && e <= 7) ? e : 0; }();
    return v;
}
// gemm8.h on / off.  Product: G8_DEFAULT.  Measurement build: XCLIP_GEMM8=0 / 1 at load, xclip_measure_gemm8() at run time (same-process A/B).
constexpr int G8_DEFAULT = 1;
static int g_gemm8 = -1;
inline bool gemm8_on() {
    if (g_gemm8 < 0) g_gemm8 = measure_env("XCLIP_GEMM8", G8_DEFAULT);
    return XC_ASM_UNITS && g_gemm8 != 0;
}
template <bool AK, bool BK_, int MODE>
void launch_gemm4(const Gemm2Params& p, dim3 pgrid, bool ring3, hipStream_t st) {
#ifdef XCLIP_MEASURE
    if constexpr (!AK && !BK_ && MODE == G4_PLAIN) {             // measurement: XCLIP_GEMM5_ABL=<mask> (gemm4.h g5_run), NT plain only
        static const int abl = measure_env("XCLIP_GEMM5_ABL", 0);
        if (ring3 && abl) {
#define XC_ABL5(N) case N: XC_ALLOW_LDS((gemm5_kernel<false, false, G4_PLAIN, N>), G5_LDS_BYTES); hipLaunchKernelGGL((gemm5_kernel<false, false, G4_PLAIN, N>), pgrid, dim3(G2_THREADS), G5_LDS_BYTES, st, p); return;
            switch (abl) { XC_ABL5(1) XC_ABL5(2) XC_ABL5(4) XC_ABL5(8) XC_ABL5(9) XC_ABL5(10) XC_ABL5(11) XC_ABL5(14) XC_ABL5(16) XC_ABL5(32) XC_ABL5(48) XC_ABL5(12) XC_ABL5(6) XC_ABL5(26) XC_ABL5(58) XC_ABL5(64) XC_ABL5(74) XC_ABL5(128) XC_ABL5(256) XC_ABL5(512) XC_ABL5(1024) XC_ABL5(2560) XC_ABL5(4096) XC_ABL5(8192) XC_ABL5(526) XC_ABL5(522) default: break; }
#undef XC_ABL5
        }
    }
#endif
#ifdef XCLIP_MEASURE
    constexpr bool both = true;                                   // every layout on either loop (XCLIP_GEMM=4 / 5)
#else
    constexpr bool both = false;                                  // the product carries one loop per layout: ring for NT / NN, two-stage for TN
    (void)ring3;
#endif
    if constexpr (both || !AK) {
        if (both ? ring3 : true) {
            XC_ALLOW_LDS((gemm5_kernel<AK, BK_, MODE>), G5_LDS_BYTES);
            hipLaunchKernelGGL((gemm5_kernel<AK, BK_, MODE>), pgrid, dim3(G2_THREADS), G5_LDS_BYTES, st, p);
            return;
        }
    }
    if constexpr (both || AK) {
        XC_ALLOW_LDS((gemm4_kernel<AK, BK_, MODE>), G5_LDS_BYTES);   // two stages + 32 KiB of epilogue scratch (gemm4.h g4_run)
        hipLaunchKernelGGL((gemm4_kernel<AK, BK_, MODE>), pgrid, dim3(G2_THREADS), G5_LDS_BYTES, st, p);
    }
}
template <bool AK, bool BK_>
void launch_gemm2(const Gemm2Params& p, int splits, hipStream_t st) {
    dim3 grid(p.tiles_m * p.tiles_n, splits), block(G2_THREADS);
    const int gen = gemm_generation();
#ifdef XCLIP_MEASURE
    if (gen == 2) {
        XC_ALLOW_LDS((gemm2_kernel<AK, BK_>), G2_LDS_BYTES);
        hipLaunchKernelGGL((gemm2_kernel<AK, BK_>), grid, block, G2_LDS_BYTES, st, p);
        return;
    }
#endif
    // persistent: one work-group per CU walks the tiles (split-K problems are sized to ~one tile per work-group already)
    int gx = p.tiles_m * p.tiles_n;
    const int cus = xc_num_cus();
    if (gx * splits > cus && splits == 1) gx = gx < cus ? gx : cus;
    dim3 pgrid(gx, splits);
    // 32-bit in-tile byte offsets: leading dimensions below 2^22 elements (anything else is not a Linear of this model)
    const bool small_ld = p.lda < (1L << 22) && p.ldb < (1L << 22) && p.ldc < (1L << 22) && (long)p.N < (1L << 21);
    if (gen != 3 && small_ld) {
        const bool ring3 = gen == 5 || ((gen == 0 || gen >= 6) && !AK);   // (6, 7: gemm6.h / gemm7.h for the shapes they take, the default otherwise)
        // split-K on the two-stage loop (the weight gradients): a 1-D grid whose work-groups place themselves so that an XCD holds whole
        // K slices (gemm2.h g2_where); XCLIP_GEMM_SPLIT2D=1 (measurement build) keeps the (tile, slice) grid for the A/B
        static const int split2d = measure_env("XCLIP_GEMM_SPLIT2D", 0);
        if (splits > 1 && !ring3 && !split2d) {
            Gemm2Params q = p;
            q.split_lin = splits;
            const dim3 lgrid((unsigned)(8 * ((gx * splits + 7) / 8)), 1);
            if (q.partial != nullptr) launch_gemm4<AK, BK_, G4_SLAB>(q, lgrid, ring3, st);
            else launch_gemm4<AK, BK_, G4_PLAIN>(q, lgrid, ring3, st);
            return;
        }
        const bool terms = p.bias != nullptr || p.residual != nullptr || p.addrows != nullptr;
#ifdef XCLIP_MEASURE
        // experiment (XCLIP_GEMM=7): four waves of 128 x 128 per tile, one per SIMD (measure/gemm7.h)
        if (gen == 7 && !AK && !BK_ && !terms && p.partial == nullptr && splits == 1 && p.M % G2_BM == 0 && p.N % G2_BN == 0 && p.K / G2_BK >= 2) {
            XC_ALLOW_LDS(gemm7_kernel, G5_LDS_BYTES);
            hipLaunchKernelGGL(gemm7_kernel, pgrid, dim3(G7_THREADS), G5_LDS_BYTES, st, p);
            return;
        }
        // experiment (XCLIP_GEMM=6, XCLIP_GEMM6_MAXK=<K>): two 4-wave work-groups per CU on 256 x 128 tiles for the short-K forward products
        // whose tiles are all interior (gemm6.h)
        if (gen == 6 && !AK && !BK_ && !terms && p.partial == nullptr && splits == 1 && p.M % G6_BM == 0 && p.N % G6_BN == 0 &&
            p.K % G6_BK == 0 && p.K / G6_BK >= 4) {
            static const int maxk = measure_env("XCLIP_GEMM6_MAXK", 1024);
            if (p.K <= maxk) {
                const int tiles6 = (p.M / G6_BM) * (p.N / G6_BN);
                const int g6 = tiles6 < 2 * cus ? tiles6 : 2 * cus;
                XC_ALLOW_LDS(gemm6_kernel, G6_LDS_BYTES);
                hipLaunchKernelGGL(gemm6_kernel, dim3((unsigned)g6), dim3(G6_THREADS), G6_LDS_BYTES, st, p);
                return;
            }
        }
#endif
        // the hand-scheduled ring kernel (gemm8.h) for the plain interior products it takes
        if constexpr (!AK) {
            if (gemm8_on() && g8_takes(p, splits, gx)) {
#define XC_G8(V) do { XC_ALLOW_LDS((gemm8_kernel<BK_, V>), G5_LDS_BYTES); hipLaunchKernelGGL((gemm8_kernel<BK_, V>), dim3((unsigned)gx), dim3(G2_THREADS), G5_LDS_BYTES, st, p); } while (0)
#ifdef XCLIP_MEASURE
                switch (g_gemm8) { case 2: XC_G8(2); return; case 3: XC_G8(3); return; case 4: XC_G8(4); return; case 5: XC_G8(5); return; case 6: XC_G8(6); return; case 10: XC_G8(0); return; case 11: XC_G8(1); return; default: break; }
#endif
                if (p.stream_out) XC_G8(1); else XC_G8(0);
#undef XC_G8
                return;
            }
        }
        const bool res_only = p.residual != nullptr && p.bias == nullptr && p.addrows == nullptr && p.ldr < (1L << 22);
        if (p.partial != nullptr) launch_gemm4<AK, BK_, G4_SLAB>(p, pgrid, ring3, st);
        else if (res_only) launch_gemm4<AK, BK_, G4_RES>(p, pgrid, ring3, st);
        else if (terms) launch_gemm4<AK, BK_, G4_TERMS>(p, pgrid, ring3, st);
        else launch_gemm4<AK, BK_, G4_PLAIN>(p, pgrid, ring3, st);
        return;
    }
#ifdef XCLIP_MEASURE
    static const int abl = measure_env("XCLIP_GEMM_ABL", 0);
    if (abl != 0 && !AK && !BK_) {                      // measurement-only variants of the NT kernel
#define XC_ABL(N) case N: XC_ALLOW_LDS((gemm3_kernel<false, false, N>), G3_LDS_BYTES); hipLaunchKernelGGL((gemm3_kernel<false, false, N>), pgrid, block, G3_LDS_BYTES, st, p); return;
        switch (abl) { XC_ABL(1) XC_ABL(2) XC_ABL(4) XC_ABL(8) XC_ABL(3) XC_ABL(9) XC_ABL(10) XC_ABL(11) XC_ABL(14) XC_ABL(15) XC_ABL(25) XC_ABL(41) XC_ABL(64) XC_ABL(73) XC_ABL(105) default: break; }
#undef XC_ABL
    }
#endif
    XC_ALLOW_LDS((gemm3_kernel<AK, BK_>), G3_LDS_BYTES);
    hipLaunchKernelGGL((gemm3_kernel<AK, BK_>), pgrid, block, G3_LDS_BYTES, st, p);
}

// the 256x256 DMA-staged kernel (gemm2.h) takes every bf16 problem whose contraction is a multiple of its K step and
// whose output is at least a tile wide; everything else (fp32, ragged K, tiny outputs) goes to the 128x128 kernel
inline bool use_gemm2(int64_t M, int64_t N, int64_t K, int dtype) {
    return dtype == XCLIP_BF16 && K % G2_BK == 0 && M >= 128 && N >= 128 && M % 8 == 0;
}
int gemm2_splits(int64_t M, int64_t N, int64_t K) {
    const int64_t tiles = ((M + G2_BM - 1) / G2_BM) * ((N + G2_BN - 1) / G2_BN);
    int64_t s = xc_policy_cus() / tiles;                      // one 8-wave work-group per CU: enough K slices to fill the part that runs it
    static const int force = measure_env("XCLIP_GEMM_SPLITS", 0);     // (measurement build: a fixed slice count, for the sweep)
    if (force > 0) s = force;
    const int64_t maxs = K / (4 * G2_BK);
    if (s > maxs) s = maxs;
    if (s > 64) s = 64;
    return s < 2 ? 1 : (int)s;
}

// ---- the row tail of a persistent GEMM ------------------------------------------------------------------------------------------------
// One work-group per CU walks tiles_m x tiles_n tiles: the launch takes ceil(tiles / CUs) rounds of one tile time each, and a last round
// that fills a few CUs costs as much as a full one.  The text tower's rows are batch x 257 (the CLS token): 263,168 rows = 1028 row tiles,
// x 2 column tiles (every N = 512 product: out-projection, FF2, the QKV / FF1 input gradients) = 2056 tiles = 8 rounds + 8 tiles -- a NINTH
// round for 3 % of a round's work, 12 % of the launch (115 us of the FF1 input gradient's 940); the vision tower's 33,792 rows are
// 132 row tiles: 264 tiles = TWO rounds for 1.03.  So the rows are cut at the last whole round: the main launch takes `main_rows`, and the
// few row tiles behind it become a split-K problem of their own that fills the part for a few K steps -- fp32 slabs + the reduction
// (which also adds the skip term of an FF2 forward).  Worth it when a tile is long (>= 16 K steps) and the tail small (<= half a round).
// -> rows of the main launch (a multiple of the tile), 0 = no cut
int64_t gemm2_tail_cut(int64_t M, int64_t N, int64_t K) {
    static const int on = measure_env("XCLIP_GEMM_TAIL", 1);   // (measurement build: 0 = the uncut launch, for the A/B)
    if (!on) return 0;
    const int64_t cus = xc_policy_cus();
    const int64_t tm = (M + G2_BM - 1) / G2_BM, tn = (N + G2_BN - 1) / G2_BN, tiles = tm * tn;
    const int64_t rounds = tiles / cus;
    if (rounds < 1 || tiles % cus == 0 || K / G2_BK < 16 || M % 8 != 0) return 0;
    const int64_t tm_main = rounds * cus / tn;                 // the most row tiles `rounds` rounds can hold
    const int64_t tail_tiles = tiles - tm_main * tn;
    if (tm_main < 1 || tail_tiles * 2 > cus || M - tm_main * G2_BM < 128) return 0;      // (a tail the 256 x 256 kernels would not take)
    return tm_main * G2_BM;
}

// split-K policy shared by xclip_gemm and xclip_gemm_workspace_bytes: fill ~512 work-groups, keep >= 4 K steps each
int gemm_splits(int64_t M, int64_t N, int64_t K, int dtype) {
    const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
    const int bk = 8 * vec_of(dtype);
    int64_t s = 2 * xc_policy_cus() / tiles;                    // two 4-wave work-groups per CU
    const int64_t maxs = K / (4 * bk);
    if (s > maxs) s = maxs;
    if (s > 64) s = 64;
    return s < 2 ? 1 : (int)s;
}

}  // namespace

extern "C" {

int xclip_abi_version(void) { return XCLIP_ABI_VERSION; }
const char* xclip_last_error(void) { return g_err; }
#ifndef XCLIP_BUILD_TOOLCHAIN
#define XCLIP_BUILD_TOOLCHAIN "unknown toolchain"
#endif
const char* xclip_build_info(void) { return XCLIP_BUILD_TOOLCHAIN; }

int xclip_layernorm_fwd(const void* x, int64_t ldx, const void* g, const void* res, void* y, int64_t ldy, int64_t y_grp,
                        float* mean, float* rstd, int64_t rows, int64_t dim, float eps, int geglu, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    const int vec = vec_of(dtype);
    XC_REQUIRE(rows >= 0 && dim > 0 && dim % vec == 0 && ldx % vec == 0, "dim / ldx must be multiples of the 16-byte chunk");
    XC_REQUIRE(ldx >= (geglu ? 2 * dim : dim), "ldx too small");
    XC_REQUIRE(ldy >= dim && ldy % vec == 0 && y_grp >= 0, "ldy must cover a row, chunk aligned; y_grp >= 0");
    XC_REQUIRE(aligned16(x) && aligned16(g) && aligned16(y) && aligned16(res), "pointers must be 16-byte aligned");
    if (rows == 0) return 0;
    const int cpl = chunks_per_lane(dim, vec);
#define F(T, C) launch_ln_fwd<T, C>(x, ldx, g, res, y, ldy, (int)y_grp, mean, rstd, (int)rows, (int)dim, eps, geglu, (hipStream_t)stream)
    XC_DISPATCH_ROW(dtype, cpl, F);
#undef F
    return check_launch(__func__);
}

int64_t xclip_layernorm_bwd_workspace_bytes(int64_t rows, int64_t dim) { return (int64_t)ln_bwd_blocks(rows) * dim * 4; }

int xclip_layernorm_bwd(const void* dy, const void* x, int64_t ldx, const void* g, const float* mean, const float* rstd,
                        const void* dres, void* dx, int64_t lddx, float* dg_accum, void* workspace, int64_t workspace_bytes,
                        int64_t rows, int64_t dim, int geglu, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    const int vec = vec_of(dtype);
    XC_REQUIRE(rows >= 0 && dim > 0 && dim % vec == 0 && ldx % vec == 0 && lddx % vec == 0, "dims must be multiples of the 16-byte chunk");
    XC_REQUIRE(ldx >= (geglu ? 2 * dim : dim) && lddx >= (geglu ? 2 * dim : dim), "leading dimension too small");
    XC_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(g) && aligned16(dx) && aligned16(dres), "pointers must be 16-byte aligned");
    XC_REQUIRE(!(geglu && dres != nullptr), "dres is not defined for the GEGLU variant");
    XC_REQUIRE(workspace != nullptr && workspace_bytes >= xclip_layernorm_bwd_workspace_bytes(rows, dim) && aligned16(workspace),
               "workspace of xclip_layernorm_bwd_workspace_bytes(rows, dim) bytes required");
    if (rows == 0) return 0;
    const int cpl = chunks_per_lane(dim, vec);
    float* partial = (float*)workspace;
    int nblk = 0;                                              // rows of dg partials the kernel wrote
#define F(T, C) nblk = launch_ln_bwd<T, C>(dy, x, ldx, g, mean, rstd, dres, dx, lddx, partial, (int)rows, (int)dim, geglu, (hipStream_t)stream)
    XC_DISPATCH_ROW(dtype, cpl, F);
#undef F
    int slices = nblk / 64;
    if (slices < 1) slices = 1;
    hipLaunchKernelGGL(colsum_fold_kernel, dim3((unsigned)((dim + 63) / 64), (unsigned)slices), dim3(256), 1024, (hipStream_t)stream,
                       (const float*)partial, (long)dim, dg_accum, nblk, (int)dim);
    return check_launch(__func__);
}

int xclip_layernorm_bwd_ffnstats(const void* dy, const void* x, int64_t ldx, const void* g, const float* mean, const float* rstd,
                                 const void* dres, void* dx, int64_t lddx, float* dg_accum, void* workspace, int64_t workspace_bytes,
                                 int64_t rows, int64_t dim, const void* x1_below, int64_t ld1, const float* wg, const float* mean4,
                                 const float* rstd4, float inv_f, float* rowc, int dtype, void* stream) {
    XC_REQUIRE(dtype == XCLIP_BF16, "bf16 only (the fused feed-forward backward the row constants feed)");
    XC_REQUIRE(rows >= 0 && dim > 0 && dim % 8 == 0 && ldx % 8 == 0 && lddx % 8 == 0 && ld1 % 8 == 0, "dims must be multiples of the 16-byte chunk");
    XC_REQUIRE(ldx >= dim && lddx >= dim && ld1 >= dim, "leading dimension too small");
    XC_REQUIRE(dy && x && g && mean && rstd && dx && dg_accum && x1_below && wg && mean4 && rstd4 && rowc, "null pointer");
    XC_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(g) && aligned16(dx) && aligned16(dres) && aligned16(x1_below) && aligned16(wg) && aligned16(rowc),
               "pointers must be 16-byte aligned");
    XC_REQUIRE(workspace != nullptr && workspace_bytes >= xclip_layernorm_bwd_workspace_bytes(rows, dim) && aligned16(workspace),
               "workspace of xclip_layernorm_bwd_workspace_bytes(rows, dim) bytes required");
    if (rows == 0) return 0;
    const int cpl = chunks_per_lane(dim, 8);
    XC_REQUIRE(cpl == 1 || cpl == 2 || cpl == 4 || cpl == 8, "row too wide (bf16 rows up to 4096 elements)");
    float* partial = (float*)workspace;
    const int blocks = ln_bwd_blocks(rows);
    const size_t lds = (size_t)3 * dim * sizeof(float);
    constexpr bool NTP = (ROWS_NT & 8) != 0;
    const LnFfnStats fs{x1_below, (long)ld1, wg, mean4, rstd4, inv_f, rowc};
    hipStream_t st = (hipStream_t)stream;
#define XC_LNF(C) do { XC_ALLOW_LDS((ln_bwd_kernel<bf16_t, C, false, NTP, true>), lds); \
        hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, C, false, NTP, true>), dim3((unsigned)blocks), dim3(256), lds, st, (const bf16_t*)dy, (const bf16_t*)x, (long)ldx, \
                           (const bf16_t*)g, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, (long)lddx, partial, (int)rows, (int)dim, fs); } while (0)
    switch (cpl) { case 1: XC_LNF(1); break; case 2: XC_LNF(2); break; case 4: XC_LNF(4); break; default: XC_LNF(8); break; }
#undef XC_LNF
    int slices = blocks / 64;
    if (slices < 1) slices = 1;
    hipLaunchKernelGGL(colsum_fold_kernel, dim3((unsigned)((dim + 63) / 64), (unsigned)slices), dim3(256), 1024, st, (const float*)partial, (long)dim, dg_accum,
                       blocks, (int)dim);
    return check_launch(__func__);
}

int xclip_layernorm_chain_fwd(const void* p, const void* g1, const void* res, void* x1, float* mean1, float* rstd1, const void* g2,
                              void* h2, float* mean2, float* rstd2, int64_t rows, int64_t dim, float eps, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    const int vec = vec_of(dtype);
    XC_REQUIRE(rows >= 0 && dim > 0 && dim % vec == 0, "dim must be a multiple of the 16-byte chunk");
    XC_REQUIRE(p && g1 && res && x1 && g2 && h2 && mean1 && rstd1 && mean2 && rstd2, "null pointer");
    XC_REQUIRE(aligned16(p) && aligned16(g1) && aligned16(res) && aligned16(x1) && aligned16(g2) && aligned16(h2), "pointers must be 16-byte aligned");
    if (rows == 0) return 0;
    const int cpl = chunks_per_lane(dim, vec);
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    constexpr bool NTC = (ROWS_NT & 16) != 0;
#ifdef XCLIP_MEASURE
    if (dtype == XCLIP_BF16 && cpl == 1 && rows_nt_flipped(16)) {
        hipLaunchKernelGGL((ln_chain_fwd_kernel<bf16_t, 1, !NTC>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)p, (const bf16_t*)g1, (const bf16_t*)res, (bf16_t*)x1, mean1, rstd1, (const bf16_t*)g2, (bf16_t*)h2, mean2, rstd2, (int)rows, (int)dim, eps);
        return check_launch(__func__);
    }
#endif
#define F(T, C) hipLaunchKernelGGL((ln_chain_fwd_kernel<T, C, NTC>), grid, block, 0, (hipStream_t)stream, (const T*)p, (const T*)g1, (const T*)res, (T*)x1, mean1, rstd1, (const T*)g2, (T*)h2, mean2, rstd2, (int)rows, (int)dim, eps)
    XC_DISPATCH_ROW(dtype, cpl, F);
#undef F
    return check_launch(__func__);
}

int64_t xclip_layernorm_chain_bwd_workspace_bytes(int64_t rows, int64_t dim) { return (int64_t)ln_bwd_blocks(rows) * 2 * dim * 4; }

int xclip_layernorm_chain_bwd(const void* dh2, const void* x1, const void* g2, const float* mean2, const float* rstd2, const void* dres,
                              void* dx1, const void* p, const void* g1, const float* mean1, const float* rstd1, void* dp, float* dg2_accum,
                              float* dg1_accum, void* workspace, int64_t workspace_bytes, int64_t rows, int64_t dim, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    const int vec = vec_of(dtype);
    XC_REQUIRE(rows >= 0 && dim > 0 && dim % vec == 0, "dim must be a multiple of the 16-byte chunk");
    XC_REQUIRE(dh2 && x1 && g2 && mean2 && rstd2 && dres && dx1 && p && g1 && mean1 && rstd1 && dp && dg2_accum && dg1_accum, "null pointer");
    XC_REQUIRE(aligned16(dh2) && aligned16(x1) && aligned16(dres) && aligned16(dx1) && aligned16(p) && aligned16(dp) && aligned16(g1) && aligned16(g2),
               "pointers must be 16-byte aligned");
    XC_REQUIRE(workspace != nullptr && workspace_bytes >= xclip_layernorm_chain_bwd_workspace_bytes(rows, dim), "workspace too small");
    if (rows == 0) return 0;
    const int cpl = chunks_per_lane(dim, vec);
    const int nblk = ln_bwd_blocks(rows);
    dim3 grid((unsigned)nblk), block(256);
    float* partial = (float*)workspace;
    const size_t lds = (size_t)6 * dim * sizeof(float);
    constexpr bool NTC = (ROWS_NT & 32) != 0;
#ifdef XCLIP_MEASURE
    if (dtype == XCLIP_BF16 && cpl == 1 && rows_nt_flipped(32)) {
        XC_ALLOW_LDS((ln_chain_bwd_kernel<bf16_t, 1, !NTC>), lds);
        hipLaunchKernelGGL((ln_chain_bwd_kernel<bf16_t, 1, !NTC>), grid, block, lds, (hipStream_t)stream, (const bf16_t*)dh2, (const bf16_t*)x1, (const bf16_t*)g2, mean2, rstd2, (const bf16_t*)dres, (bf16_t*)dx1, (const bf16_t*)p, (const bf16_t*)g1, mean1, rstd1, (bf16_t*)dp, partial, (int)rows, (int)dim);
    } else
#endif
#define F(T, C) do { XC_ALLOW_LDS((ln_chain_bwd_kernel<T, C, NTC>), lds); hipLaunchKernelGGL((ln_chain_bwd_kernel<T, C, NTC>), grid, block, lds, (hipStream_t)stream, (const T*)dh2, (const T*)x1, (const T*)g2, mean2, rstd2, (const T*)dres, (T*)dx1, (const T*)p, (const T*)g1, mean1, rstd1, (T*)dp, partial, (int)rows, (int)dim); } while (0)
    XC_DISPATCH_ROW(dtype, cpl, F);
#undef F
    int slices = nblk / 64;
    if (slices < 1) slices = 1;
    hipLaunchKernelGGL(colsum_fold2_kernel, dim3((unsigned)((dim + 63) / 64), (unsigned)slices, 2), dim3(256), 1024, (hipStream_t)stream,
                       (const float*)partial, dg2_accum, dg1_accum, nblk, (int)dim);      // both gains' partials in one launch
    return check_launch(__func__);
}

int xclip_l2norm_fwd(const void* x, void* y, float* rnorm, int64_t rows, int64_t dim, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    const int vec = vec_of(dtype);
    XC_REQUIRE(dim > 0 && dim % vec == 0, "dim must be a multiple of the 16-byte chunk");
    XC_REQUIRE(aligned16(x) && aligned16(y), "pointers must be 16-byte aligned");
    if (rows == 0) return 0;
    const int cpl = chunks_per_lane(dim, vec);
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define F(T, C) hipLaunchKernelGGL((l2norm_fwd_kernel<T, C>), grid, block, 0, (hipStream_t)stream, (const T*)x, (T*)y, rnorm, (int)rows, (int)dim)
    XC_DISPATCH_ROW(dtype, cpl, F);
#undef F
    return check_launch(__func__);
}

int xclip_l2norm_bwd(const void* dy, const void* y, const float* rnorm, void* dx, int64_t rows, int64_t dim, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    const int vec = vec_of(dtype);
    XC_REQUIRE(dim > 0 && dim % vec == 0, "dim must be a multiple of the 16-byte chunk");
    XC_REQUIRE(aligned16(dy) && aligned16(y) && aligned16(dx), "pointers must be 16-byte aligned");
    if (rows == 0) return 0;
    const int cpl = chunks_per_lane(dim, vec);
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define F(T, C) hipLaunchKernelGGL((l2norm_bwd_kernel<T, C>), grid, block, 0, (hipStream_t)stream, (const T*)dy, (const T*)y, rnorm, (T*)dx, (int)rows, (int)dim)
    XC_DISPATCH_ROW(dtype, cpl, F);
#undef F
    return check_launch(__func__);
}

int xclip_text_embed_fwd(const int64_t* tokens, const void* E, const void* P, const void* cls, void* out, int64_t batch,
                         int64_t n, int64_t dim, int64_t vocab, int32_t* bad_token_flag, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(vocab > 0, "vocab (rows of the embedding table) must be positive");
    XC_REQUIRE(dim > 0 && dim % vec_of(dtype) == 0, "dim must be a multiple of the 16-byte chunk");
    XC_REQUIRE(aligned16(E) && aligned16(P) && aligned16(cls) && aligned16(out), "pointers must be 16-byte aligned");
    const int64_t rows = batch * (n + (cls ? 1 : 0));
    if (rows == 0) return 0;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dtype == XCLIP_BF16)
        hipLaunchKernelGGL((text_embed_fwd_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const long long*)tokens,
                           (const bf16_t*)E, (const bf16_t*)P, (const bf16_t*)cls, (bf16_t*)out, (int)batch, (int)n, (int)dim,
                           (long long)vocab, bad_token_flag);
    else
        hipLaunchKernelGGL((text_embed_fwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const long long*)tokens,
                           (const float*)E, (const float*)P, (const float*)cls, (float*)out, (int)batch, (int)n, (int)dim,
                           (long long)vocab, bad_token_flag);
    return check_launch(__func__);
}

int xclip_text_embed_bwd(const void* dout, const int64_t* tokens, float* dE_accum, float* dP_accum, float* dcls_accum,
                         int64_t batch, int64_t n, int64_t dim, int64_t vocab, int has_cls, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    const int vec = vec_of(dtype);
    XC_REQUIRE(dim > 0 && dim % vec == 0, "dim must be a multiple of the 16-byte chunk");
    XC_REQUIRE(aligned16(dout), "pointers must be 16-byte aligned");
    XC_REQUIRE(has_cls == 0 || dcls_accum != nullptr, "dcls_accum required with a CLS row");
    if (batch == 0) return 0;
    const int cpl = chunks_per_lane(dim, vec);
    int ysplit = (int)((batch + 31) / 32);
    if (ysplit > 16) ysplit = 16;
    dim3 grid((unsigned)(n + (has_cls ? 1 : 0)), ysplit), block(256);
    const size_t red_bytes = (size_t)3 * dim * sizeof(float);
#define F(T, C) hipLaunchKernelGGL((text_embed_bwd_kernel<T, C>), grid, block, red_bytes, (hipStream_t)stream, (const T*)dout, (const long long*)tokens, dE_accum, dP_accum, dcls_accum, (int)batch, (int)n, (int)dim, has_cls ? 1 : 0, (long long)vocab)
    XC_DISPATCH_ROW(dtype, cpl, F);
#undef F
    return check_launch(__func__);
}

int xclip_patchify(const void* image, const int32_t* keep, void* out, int64_t ldo, int64_t batch, int64_t channels,
                   int64_t height, int64_t width, int64_t patch, int64_t nkeep, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(patch > 0 && height % patch == 0 && width % patch == 0, "image must be divisible by the patch size");
    XC_REQUIRE(ldo % vec_of(dtype) == 0 && ldo >= patch * patch * channels, "ldo must cover a patch row, chunk aligned");
    XC_REQUIRE(aligned16(out), "pointers must be 16-byte aligned");
    const int64_t total = batch * nkeep * (ldo / vec_of(dtype));
    if (total == 0) return 0;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    dim3 grid((unsigned)blocks), block(256);
    // three bf16 channels, patch edge and image width multiples of 8, no padding behind a row: whole 16-byte pieces in and out
    static const int rgb8 = measure_env("XCLIP_PATCHIFY_RGB8", 1);     // (measurement build: 0 = the element-wise kernel, for the A/B)
    if (rgb8 && dtype == XCLIP_BF16 && channels == 3 && patch % 8 == 0 && width % 8 == 0 && ldo == patch * patch * 3 && aligned16(image)) {
        int64_t b8 = (batch * nkeep * patch * (patch / 8) + 255) / 256;
        if (b8 > 16384) b8 = 16384;
        hipLaunchKernelGGL(patchify_rgb8_kernel, dim3((unsigned)b8), block, 0, (hipStream_t)stream, (const bf16_t*)image, (const int*)keep,
                           (bf16_t*)out, (long)ldo, (int)batch, (int)height, (int)width, (int)patch, (int)nkeep);
        return check_launch(__func__);
    }
    if (dtype == XCLIP_BF16)
        hipLaunchKernelGGL((patchify_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)image, (const int*)keep,
                           (bf16_t*)out, (long)ldo, (int)batch, (int)channels, (int)height, (int)width, (int)patch, (int)nkeep);
    else
        hipLaunchKernelGGL((patchify_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)image, (const int*)keep,
                           (float*)out, (long)ldo, (int)batch, (int)channels, (int)height, (int)width, (int)patch, (int)nkeep);
    return check_launch(__func__);
}

int xclip_token_mean_fwd(const void* x, int64_t x_batch_stride, void* out, int64_t batch, int64_t n, int64_t dim, int dtype,
                         void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    const int vec = vec_of(dtype);
    XC_REQUIRE(dim > 0 && dim % vec == 0 && n > 0, "dim must be a multiple of the 16-byte chunk");
    XC_REQUIRE(x_batch_stride >= n * dim && x_batch_stride % vec == 0, "batch stride must cover n rows, chunk aligned");
    XC_REQUIRE(aligned16(x) && aligned16(out), "pointers must be 16-byte aligned");
    if (batch == 0) return 0;
    dim3 grid((unsigned)batch, (unsigned)((dim / vec + 63) / 64)), block(64);
    if (dtype == XCLIP_BF16)
        hipLaunchKernelGGL((token_mean_fwd_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)x, (long)x_batch_stride, (bf16_t*)out, (int)n, (int)dim);
    else
        hipLaunchKernelGGL((token_mean_fwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)x, (long)x_batch_stride, (float*)out, (int)n, (int)dim);
    return check_launch(__func__);
}

int xclip_token_mean_bwd(const void* dout, const void* dsrc, int64_t src_batch_stride, void* dx, int64_t batch, int64_t n,
                         int64_t dim, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    const int vec = vec_of(dtype);
    XC_REQUIRE(dim > 0 && dim % vec == 0 && n > 0, "dim must be a multiple of the 16-byte chunk");
    XC_REQUIRE(dsrc == nullptr || (src_batch_stride >= n * dim && src_batch_stride % vec == 0), "bad source batch stride");
    XC_REQUIRE(aligned16(dout) && aligned16(dx) && aligned16(dsrc), "pointers must be 16-byte aligned");
    if (batch == 0) return 0;
    dim3 grid((unsigned)((batch * n + 3) / 4)), block(256);
    if (dtype == XCLIP_BF16)
        hipLaunchKernelGGL((token_mean_bwd_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)dout, (const bf16_t*)dsrc, (long)src_batch_stride, (bf16_t*)dx, (int)batch, (int)n, (int)dim);
    else
        hipLaunchKernelGGL((token_mean_bwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)dout, (const float*)dsrc, (long)src_batch_stride, (float*)dx, (int)batch, (int)n, (int)dim);
    return check_launch(__func__);
}

int xclip_copy_rows(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t dim, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    const int vec = vec_of(dtype);
    XC_REQUIRE(dim > 0 && dim % vec == 0 && lds % vec == 0 && ldd % vec == 0 && lds >= dim && ldd >= dim, "dim / strides must be chunk multiples covering a row");
    XC_REQUIRE(aligned16(src) && aligned16(dst), "pointers must be 16-byte aligned");
    if (rows == 0) return 0;
    int64_t blocks = (rows * (dim / vec) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    dim3 grid((unsigned)blocks), block(256);
    if (dtype == XCLIP_BF16)
        hipLaunchKernelGGL((copy_rows_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)src, (long)lds, (bf16_t*)dst, (long)ldd, (long)rows, (int)dim);
    else
        hipLaunchKernelGGL((copy_rows_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)src, (long)lds, (float*)dst, (long)ldd, (long)rows, (int)dim);
    return check_launch(__func__);
}

int xclip_add(const void* a, const void* b, void* out, int64_t count, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(count % vec_of(dtype) == 0 && aligned16(a) && aligned16(b) && aligned16(out), "count must be a chunk multiple, 16-byte aligned");
    if (count == 0) return 0;
    const int64_t n16 = count / vec_of(dtype);
    int64_t blocks = (n16 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == XCLIP_BF16)
        hipLaunchKernelGGL((add_rows_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, (long)n16);
    else
        hipLaunchKernelGGL((add_rows_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)a, (const float*)b, (float*)out, (long)n16);
    return check_launch(__func__);
}

int64_t xclip_rows_scatter_add_workspace_bytes(int64_t rows, int64_t dim) {
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 256) blocks = 256;
    return blocks * 4 * dim * 4;
}

int xclip_rows_scatter_add(const void* src, int64_t lds, const int32_t* idx, float* table_accum, float* colsum_accum, int64_t rows,
                           int64_t dim, void* workspace, int64_t workspace_bytes, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    const int vec = vec_of(dtype);
    XC_REQUIRE(dim > 0 && dim % vec == 0 && lds % vec == 0 && lds >= dim, "dim / lds must be chunk multiples covering a row");
    XC_REQUIRE(aligned16(src), "pointers must be 16-byte aligned");
    XC_REQUIRE(table_accum == nullptr || idx != nullptr, "table_accum needs idx");
    if (rows == 0) return 0;
    const int cpl = chunks_per_lane(dim, vec);
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 256) blocks = 256;
    // column sums go through per-wave partial rows + a fold (thousands of waves adding into the same `dim` floats serialise)
    float* partial = (colsum_accum != nullptr && workspace != nullptr && workspace_bytes >= blocks * 4 * dim * 4) ? (float*)workspace : nullptr;
    dim3 grid((unsigned)blocks), block(256);
#define F(T, C) hipLaunchKernelGGL((rows_scatter_add_kernel<T, C>), grid, block, 0, (hipStream_t)stream, (const T*)src, (long)lds, (const int*)idx, table_accum, colsum_accum, (long)rows, (int)dim, partial)
    XC_DISPATCH_ROW(dtype, cpl, F);
#undef F
    if (partial != nullptr) {
        // every wave writes its row (waves without rows write zeros), so all blocks * 4 rows are defined
        hipLaunchKernelGGL(colsum_fold_kernel, dim3((unsigned)((dim + 63) / 64), 4), dim3(256), 1024, (hipStream_t)stream,
                           (const float*)partial, (long)dim, colsum_accum, (int)(blocks * 4), (int)dim);
    }
    return check_launch(__func__);
}

int xclip_scatter_add_sorted(const void* src, int64_t lds, const int64_t* sorted_ids, const int64_t* perm, float* table_accum,
                             int64_t table_rows, int64_t count, int64_t dim, int64_t n_in, int64_t n_out, int64_t row_off, int dtype,
                             void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    const int vec = vec_of(dtype);
    XC_REQUIRE(dim > 0 && dim % vec == 0 && lds % vec == 0 && lds >= dim, "dim / lds must be chunk multiples covering a row");
    XC_REQUIRE(aligned16(src) && sorted_ids != nullptr && perm != nullptr && table_accum != nullptr, "bad pointers");
    XC_REQUIRE(n_in > 0 && n_out > 0 && table_rows > 0, "bad row map");
    if (count == 0) return 0;
    const int cpl = chunks_per_lane(dim, vec);
    int64_t chunk = (count + 16383) / 16384;               // ~16k waves; at least 16 entries per wave
    if (chunk < 16) chunk = 16;
    const int64_t waves = (count + chunk - 1) / chunk;
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    // the flush is staged through LDS (one wave's row of fp32: coalesced atomics) where four rows fit the default 64 KiB
    const size_t stage_bytes = (size_t)4 * dim * sizeof(float) <= 64 * 1024 ? (size_t)4 * dim * sizeof(float) : 0;
    const int staged = measure_env("XCLIP_SCATTER_STAGED", 1) && stage_bytes > 0;
#define F(T, C) hipLaunchKernelGGL((scatter_add_sorted_kernel<T, C>), grid, block, staged ? stage_bytes : 0, (hipStream_t)stream, (const T*)src, (long)lds, (const long long*)sorted_ids, (const long long*)perm, table_accum, (long)count, (int)dim, (int)n_in, (int)n_out, (int)row_off, (int)chunk, (long long)table_rows, staged)
    XC_DISPATCH_ROW(dtype, cpl, F);
#undef F
    return check_launch(__func__);
}

int64_t xclip_sort_ids_workspace_bytes(int64_t count) {
    if (count <= 0) return 0;
    const int64_t nblk = (count + SORT_BLOCK - 1) / SORT_BLOCK;
    return 2 * ((count * 8 + 15) & ~(int64_t)15) + 2 * nblk * SORT_BINS * 4;
}

int xclip_sort_ids(const int64_t* ids, int64_t count, int64_t id_limit, int64_t* sorted_ids, int64_t* perm, void* workspace,
                   int64_t workspace_bytes, void* stream) {
    XC_REQUIRE(count >= 0 && count < ((int64_t)1 << 32) && id_limit >= 1 && id_limit <= ((int64_t)1 << 32), "count / id_limit out of range");
    if (count == 0) return 0;
    XC_REQUIRE(ids != nullptr && sorted_ids != nullptr && perm != nullptr, "bad pointers");
    XC_REQUIRE(workspace != nullptr && aligned16(workspace) && workspace_bytes >= xclip_sort_ids_workspace_bytes(count), "workspace too small or misaligned");
    const int64_t nblk = (count + SORT_BLOCK - 1) / SORT_BLOCK, half = (count * 8 + 15) & ~(int64_t)15;
    uint64_t* bufs[2] = {reinterpret_cast<uint64_t*>(workspace), reinterpret_cast<uint64_t*>(static_cast<unsigned char*>(workspace) + half)};
    int* const hist = reinterpret_cast<int*>(static_cast<unsigned char*>(workspace) + 2 * half);
    int* const offs = hist + nblk * SORT_BINS;
    int bits = 0;
    while (((int64_t)1 << bits) < id_limit) ++bits;
    const int passes = bits <= 8 ? 1 : (bits + 7) / 8;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)nblk), block(256);
    const size_t lds = SORT_BINS * sizeof(int);
    const long long* src_ids = reinterpret_cast<const long long*>(ids);
    for (int ps = 0; ps < passes; ++ps) {
        const bool first = ps == 0, last = ps == passes - 1;
        const uint64_t* in = first ? nullptr : bufs[(ps - 1) & 1];
        uint64_t* out = last ? nullptr : bufs[ps & 1];
        const int shift = 8 * ps;
        if (first) hipLaunchKernelGGL((sort_hist_kernel<true>), grid, block, lds, st, src_ids, in, (long)count, shift, hist);
        else hipLaunchKernelGGL((sort_hist_kernel<false>), grid, block, lds, st, src_ids, in, (long)count, shift, hist);
        hipLaunchKernelGGL(sort_scan_kernel, dim3(1), dim3(1024), 5 * lds, st, (const int*)hist, offs, (int)nblk);
#define S(F, L) hipLaunchKernelGGL((sort_scatter_kernel<F, L>), grid, block, lds, st, src_ids, in, out, (long long*)sorted_ids, (long long*)perm, (long)count, shift, (const int*)offs)
        if (first && last) S(true, true); else if (first) S(true, false); else if (last) S(false, true); else S(false, false);
#undef S
    }
    return check_launch(__func__);
}

int xclip_cast_from_f32(const float* src, void* dst, int64_t count, float scale, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    if (count == 0) return 0;
    int64_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    dim3 grid((unsigned)blocks), block(256);
    if (dtype == XCLIP_BF16)
        hipLaunchKernelGGL((cast_from_f32_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, src, (bf16_t*)dst, (long)count, scale);
    else
        hipLaunchKernelGGL((cast_from_f32_kernel<float>), grid, block, 0, (hipStream_t)stream, src, (float*)dst, (long)count, scale);
    return check_launch(__func__);
}

int xclip_clock_sample(uint64_t* out2, int64_t ticks_10ns, void* stream) {
    XC_REQUIRE(out2 != nullptr && ticks_10ns > 0 && ticks_10ns <= 100000, "out2 must hold two uint64; 0 < ticks <= 100000 (1 ms)");
    hipLaunchKernelGGL(clock_sample_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out2, (long)ticks_10ns);
    return check_launch(__func__);
}

namespace {
inline int dwconv_bwd_blocks(int64_t batch, int64_t h, int64_t C, int dtype) {
    const int64_t items = batch * h * h * (C / vec_of(dtype));
    int64_t b = (items + 255) / 256;
    return (int)(b > 512 ? 512 : (b < 1 ? 1 : b));
}
}  // namespace
int64_t xclip_dwconv4s2_workspace_bytes(int64_t batch, int64_t h, int64_t C, int dtype) {
    return (int64_t)dwconv_bwd_blocks(batch, h, C, dtype) * 4 * C * 16 * 4;
}
int xclip_dwconv4s2_fwd(const void* x, const void* w, void* y, int64_t batch, int64_t h, int64_t C, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(batch >= 0 && h >= 2 && h % 2 == 0 && C > 0 && C % vec_of(dtype) == 0, "token grid must be even-sided, channels whole 16-byte chunks");
    XC_REQUIRE(x && w && y && aligned16(x) && aligned16(w) && aligned16(y), "null or misaligned pointer");
    if (batch == 0) return 0;
    const int64_t items = batch * (h / 2) * (h / 2) * (C / vec_of(dtype));
    int64_t blocks = (items + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    dim3 grid((unsigned)blocks), block(256);
    if (dtype == XCLIP_BF16)
        hipLaunchKernelGGL((dwconv_fwd_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, (int)batch, (int)h, (int)C);
    else
        hipLaunchKernelGGL((dwconv_fwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)x, (const float*)w, (float*)y, (int)batch, (int)h, (int)C);
    return check_launch(__func__);
}
int xclip_dwconv4s2_bwd(const void* dy, const void* x, const void* w, void* dx, float* dw_accum, void* workspace, int64_t workspace_bytes,
                        int64_t batch, int64_t h, int64_t C, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(batch >= 0 && h >= 2 && h % 2 == 0 && C > 0 && C % vec_of(dtype) == 0, "token grid must be even-sided, channels whole 16-byte chunks");
    XC_REQUIRE(dy && x && w && dx && dw_accum && aligned16(dy) && aligned16(x) && aligned16(w) && aligned16(dx), "null or misaligned pointer");
    XC_REQUIRE(workspace != nullptr && workspace_bytes >= xclip_dwconv4s2_workspace_bytes(batch, h, C, dtype), "workspace too small");
    if (batch == 0) return 0;
    const int blocks = dwconv_bwd_blocks(batch, h, C, dtype);
    dim3 grid((unsigned)blocks), block(256);
    float* partial = (float*)workspace;
    if (dtype == XCLIP_BF16)
        hipLaunchKernelGGL((dwconv_bwd_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)dx, partial, (int)batch, (int)h, (int)C);
    else
        hipLaunchKernelGGL((dwconv_bwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)dy, (const float*)x, (const float*)w, (float*)dx, partial, (int)batch, (int)h, (int)C);
    hipLaunchKernelGGL(colsum_fold_kernel, dim3((unsigned)((C * 16 + 63) / 64), 4), dim3(256), 1024, (hipStream_t)stream,
                       (const float*)partial, (long)(C * 16), dw_accum, blocks * 4, (int)(C * 16));
    return check_launch(__func__);
}

int xclip_gather_rows(const void* src, int64_t lds, const int32_t* idx, void* out, int64_t rows, int64_t dim, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(rows >= 0 && dim > 0 && dim % vec_of(dtype) == 0 && lds % vec_of(dtype) == 0, "dim / row stride must be whole 16-byte chunks");
    if (rows == 0) return 0;                                  // (an empty tensor has a null data pointer)
    XC_REQUIRE(src && idx && out && aligned16(src) && aligned16(out), "null or misaligned pointer");
    int64_t blocks = (rows * (dim / vec_of(dtype)) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    dim3 grid((unsigned)blocks), block(256);
    if (dtype == XCLIP_BF16)
        hipLaunchKernelGGL((gather_rows_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)src, (long)lds, idx, (bf16_t*)out, (long)rows, (int)dim);
    else
        hipLaunchKernelGGL((gather_rows_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)src, (long)lds, idx, (float*)out, (long)rows, (int)dim);
    return check_launch(__func__);
}
int xclip_cross_entropy_fwd(const void* logits, int64_t ld, const int64_t* labels, int64_t rows, int64_t cols, float* lse, float* loss_accum,
                            int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(rows >= 0 && cols > 0 && ld >= cols && ld % vec_of(dtype) == 0, "row stride must cover the columns in whole 16-byte chunks");
    if (rows == 0) return 0;                                  // (an empty tensor has a null data pointer)
    XC_REQUIRE(logits && labels && lse && loss_accum && aligned16(logits), "null or misaligned pointer");
    int64_t blocks = (rows + 3) / 4;
    if (blocks > ROWLOSS_MAX_BLOCKS) blocks = ROWLOSS_MAX_BLOCKS;
    dim3 grid((unsigned)blocks), block(256);
    if (dtype == XCLIP_BF16)
        hipLaunchKernelGGL((ce_fwd_kernel<bf16_t>), grid, block, 16, (hipStream_t)stream, (const bf16_t*)logits, (long)ld, (const long long*)labels, (int)rows, (int)cols, lse, loss_accum);
    else
        hipLaunchKernelGGL((ce_fwd_kernel<float>), grid, block, 16, (hipStream_t)stream, (const float*)logits, (long)ld, (const long long*)labels, (int)rows, (int)cols, lse, loss_accum);
    return check_launch(__func__);
}
int xclip_cross_entropy_bwd(void* logits, int64_t ld, const int64_t* labels, const float* lse, const float* gmul, int64_t rows, int64_t cols,
                            int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(rows >= 0 && cols > 0 && ld >= cols && ld % vec_of(dtype) == 0, "row stride must cover the columns in whole 16-byte chunks");
    if (rows == 0) return 0;                                  // (an empty tensor has a null data pointer)
    XC_REQUIRE(logits && labels && lse && gmul && aligned16(logits), "null or misaligned pointer");
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dtype == XCLIP_BF16)
        hipLaunchKernelGGL((ce_bwd_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (bf16_t*)logits, (long)ld, (const long long*)labels, lse, gmul, (int)rows, (int)cols);
    else
        hipLaunchKernelGGL((ce_bwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (float*)logits, (long)ld, (const long long*)labels, lse, gmul, (int)rows, (int)cols);
    return check_launch(__func__);
}

int xclip_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(n >= 0 && n % vec_of(dtype) == 0 && p >= 0.f && p < 1.f, "n must be a whole number of 16-byte chunks, p in [0, 1)");
    XC_REQUIRE(x && y && aligned16(x) && aligned16(y), "null or misaligned pointer");
    if (n == 0) return 0;
    int64_t blocks = (n / vec_of(dtype) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    const uint32_t th = drop_thresh(p);
    const float sc = 1.0f / (1.0f - p);
    if (dtype == XCLIP_BF16)
        hipLaunchKernelGGL((dropout_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, (long)n, th, sc, seed);
    else
        hipLaunchKernelGGL((dropout_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, (long)n, th, sc, seed);
    return check_launch(__func__);
}

int xclip_rotary(void* x, int64_t ld, int64_t rows, int64_t n, int64_t slots, int64_t slot_width, int64_t rot, const float* inv_freq, int inverse,
                 int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(slot_width == 64 || slot_width == 128, "head slots are 64 or 128 wide");
    XC_REQUIRE(rot >= 2 && rot <= 32 && rot % 2 == 0, "rot = min(dim_head, 32) rotated features per head: even, 2 .. 32");
    XC_REQUIRE(rows >= 0 && n > 0 && slots > 0 && ld >= slots * slot_width && ld % vec_of(dtype) == 0, "bad shape");
    XC_REQUIRE(x && aligned16(x) && inv_freq, "null or misaligned pointer");
    if (rows == 0) return 0;
    const int64_t items = rows * slots * (16 / vec_of(dtype));
    int64_t blocks = (items + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    dim3 grid((unsigned)blocks), block(256);
    const float sign = inverse ? -1.0f : 1.0f;
    if (rot != 32) {                                             // narrow heads: element pairs
        int64_t pb = (rows * slots * (rot / 2) + 255) / 256;
        if (pb > 8192) pb = 8192;
        if (dtype == XCLIP_BF16)
            hipLaunchKernelGGL((rotary_pairs_kernel<bf16_t>), dim3((unsigned)pb), block, 0, (hipStream_t)stream, (bf16_t*)x, (long)ld, (long)rows, (int)n, (int)slots, (int)slot_width, (int)(rot / 2), inv_freq, sign);
        else
            hipLaunchKernelGGL((rotary_pairs_kernel<float>), dim3((unsigned)pb), block, 0, (hipStream_t)stream, (float*)x, (long)ld, (long)rows, (int)n, (int)slots, (int)slot_width, (int)(rot / 2), inv_freq, sign);
        return check_launch(__func__);
    }
    if (dtype == XCLIP_BF16)
        hipLaunchKernelGGL((rotary_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (bf16_t*)x, (long)ld, (long)rows, (int)n, (int)slots, (int)slot_width, inv_freq, sign);
    else
        hipLaunchKernelGGL((rotary_kernel<float>), grid, block, 0, (hipStream_t)stream, (float*)x, (long)ld, (long)rows, (int)n, (int)slots, (int)slot_width, inv_freq, sign);
    return check_launch(__func__);
}

// the 256 x 256 bf16 kernels behind xclip_gemm.  reduce = false: a split-K problem leaves its fp32 slabs in `workspace` for the caller
// (*splits_out of them; alpha not applied).
static int gemm2_run(int a_kmajor, int b_kmajor, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
                     int64_t K, float alpha, const void* bias, const void* residual, int64_t ldr, const void* addrows, const int32_t* rowidx,
                     int64_t ld_add, void* workspace, int64_t workspace_bytes, hipStream_t st, bool reduce, int* splits_out) {
    const bool plain = (bias == nullptr && residual == nullptr && addrows == nullptr);
    Gemm2Params q;
    q.A = (const bf16_t*)A; q.B = (const bf16_t*)B; q.C = (bf16_t*)C; q.lda = lda; q.ldb = ldb; q.ldc = ldc;
    q.M = (int)M; q.N = (int)N; q.K = (int)K; q.alpha = alpha;
    q.bias = (const bf16_t*)bias; q.residual = (const bf16_t*)residual; q.ldr = ldr;
    q.addrows = (const bf16_t*)addrows; q.rowidx = rowidx; q.ld_add = ld_add;
    q.tiles_m = (int)((M + G2_BM - 1) / G2_BM); q.tiles_n = (int)((N + G2_BN - 1) / G2_BN);
    // an output that cannot stay in the 8 x 4 MiB of L2 anyway is streamed past it: the A / B panels the sibling tiles share then
    // survive a round's 32 MiB of output (XCLIP_GEMM_NT=0 / 1 forces the policy, for measurement)
    static const int nt_env = measure_env("XCLIP_GEMM_NT", -1);
    q.stream_out = nt_env >= 0 ? nt_env : (M * N * 2 > (int64_t)(48 << 20) ? 1 : 0);
    // more than 8 N tiles (FF1: 16): banded tile order for the ring kernel (XCLIP_GEMM_BAND=<tiles per band>, 0 = off, for measurement)
    static const int band_env = measure_env("XCLIP_GEMM_BAND", -1);
    // FF1 forward in the step: 1227 -> 1155 us (profiles/r02_run22_gemm_banded_order.log); the widest band of 4..8 tiles that divides
    // the N tiles, none if there is none (9 tiles) or the operand fits anyway (<= 8 tiles)
    q.band_n = 0;
    if (band_env != 0 && q.tiles_n > 8)
        for (int b = band_env > 0 ? band_env : 8; b >= 4 && q.band_n == 0; --b)
            if (q.tiles_n % b == 0) q.band_n = b;
    int splits = gemm2_splits(M, N, K);
    if (splits > 1 && (!plain || workspace == nullptr || workspace_bytes < (int64_t)splits * M * N * 4)) splits = 1;
    q.k_per_split = (int)((((K / G2_BK) + splits - 1) / splits) * G2_BK);
    // rounding the slice up can leave the last slices EMPTY (96 K steps over 17 slices -> 6 per slice, 16 slices cover them): a slice
    // without work returns without touching its slab and the reduction would add whatever the workspace held before
    splits = (int)((K + q.k_per_split - 1) / q.k_per_split);
    q.partial = splits > 1 ? (float*)workspace : nullptr;
    if (!a_kmajor && !b_kmajor) launch_gemm2<false, false>(q, splits, st);
    else if (!a_kmajor && b_kmajor) launch_gemm2<false, true>(q, splits, st);
    else launch_gemm2<true, true>(q, splits, st);
    if (splits_out != nullptr) *splits_out = splits;
    if (splits > 1 && reduce) {
        int64_t blocks = (M * (N / 4) + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, (const float*)workspace,
                           (bf16_t*)C, (long)ldc, (int)M, (int)N, splits, alpha);
    }
    return check_launch(__func__);
}

// the latency-built 64 x 64 kernel (gemm_small.h) for outputs of a few tiles and a few GFLOP
static int64_t g_small_flop = GS_MAX_FLOP;
int64_t xclip_gemm_small_limit(int64_t max_flop) {
    const int64_t was = g_small_flop;
    if (max_flop >= 0) g_small_flop = max_flop;
    return was;
}
static int gemm_small_run(int a_kmajor, int b_kmajor, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                          int64_t N, int64_t K, float alpha, const void* residual, int64_t ldr, hipStream_t st) {
    GemmSmallParams q;
    q.A = (const bf16_t*)A; q.B = (const bf16_t*)B; q.C = (bf16_t*)C; q.lda = lda; q.ldb = ldb; q.ldc = ldc;
    q.M = (int)M; q.N = (int)N; q.K = (int)K; q.alpha = alpha; q.residual = (const bf16_t*)residual; q.ldr = ldr;
    q.tiles_m = (int)(M / GS_BM); q.tiles_n = (int)(N / GS_BN);
    const int64_t tiles = (int64_t)q.tiles_m * q.tiles_n;
    q.stages = gs_stages(tiles, K / GS_BK);
    const int lds = q.stages * GS_STAGE_BYTES;
#define XC_GS(AK, BK_, RES) do { XC_ALLOW_LDS((gemm_small_kernel<AK, BK_, RES>), GS_MAX_STAGES * GS_STAGE_BYTES); \
        hipLaunchKernelGGL((gemm_small_kernel<AK, BK_, RES>), dim3((unsigned)tiles), dim3(GS_THREADS), lds, st, q); } while (0)
    if (residual != nullptr) {
        if (!a_kmajor && !b_kmajor) XC_GS(false, false, true); else if (!a_kmajor) XC_GS(false, true, true); else XC_GS(true, true, true);
    } else {
        if (!a_kmajor && !b_kmajor) XC_GS(false, false, false); else if (!a_kmajor) XC_GS(false, true, false); else XC_GS(true, true, false);
    }
#undef XC_GS
    return check_launch(__func__);
}

// (Measured and not kept: the row tail of a SHORT-K persistent product -- fewer than 16 K steps, which gemm2_tail_cut leaves whole -- as a
//  second launch of the 64 x 64 kernel behind the whole rounds.  Bit-identical results (the same MFMA k-blocks in the same order), and the
//  round it saves is worth about what the dependent launch costs: text out-projection 153.8 -> 152.0 us, QKV 406 -> 403, FF1 1101 -> 1102
//  (profiles/r05_s_bench_shapes_short_k_tail_on_small_kernel.log against r05_r_bench_shapes_small_kernel.log).)

#ifdef XCLIP_MEASURE
// measurement build: gemm8.h on / off at run time (same-process A/B); -> the previous setting
extern "C" int xclip_measure_gemm8(int on) { const int was = gemm8_on() ? 1 : 0; g_gemm8 = on; return was; }
#endif

int64_t xclip_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int dtype) {
    if (use_gemm2(M, N, K, dtype)) {
        const int64_t cut = gemm2_tail_cut(M, N, K);           // (normal-A layouts only; the k-major-A caller just gets a little more than it needs)
        if (cut > 0) return (int64_t)gemm2_splits(M - cut, N, K) * (M - cut) * N * 4;
    }
    const int s = use_gemm2(M, N, K, dtype) ? gemm2_splits(M, N, K) : gemm_splits(M, N, K, dtype);
    return s > 1 ? (int64_t)s * M * N * 4 : 0;
}

int xclip_gemm(int a_kmajor, int b_kmajor, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
               int64_t M, int64_t N, int64_t K, float alpha, const void* bias, const void* residual, int64_t ldr,
               const void* addrows, const int32_t* rowidx, int64_t ld_add, void* workspace, int64_t workspace_bytes,
               int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    const int vec = vec_of(dtype);
    XC_REQUIRE(M >= 0 && N > 0 && K > 0, "bad shape");
    XC_REQUIRE(N % vec == 0 && ldc % vec == 0, "N and ldc must be multiples of the 16-byte chunk");
    XC_REQUIRE(lda % vec == 0 && ldb % vec == 0, "lda / ldb must be multiples of the 16-byte chunk");
    // operands are read in whole 16-byte chunks: a row must be allocated (and, for a normal operand whose K is not a
    // chunk multiple, zero padded by the caller) up to the next chunk boundary
    const int64_t Kp = (K + vec - 1) / vec * vec, Mp = (M + vec - 1) / vec * vec;
    XC_REQUIRE(a_kmajor ? (lda >= Mp) : (lda >= Kp), "lda must cover A's contiguous dim rounded up to the 16-byte chunk");
    XC_REQUIRE(b_kmajor ? (ldb >= N) : (ldb >= Kp), "ldb must cover B's contiguous dim rounded up to the 16-byte chunk");
    XC_REQUIRE(!(a_kmajor && !b_kmajor), "layout (A k-major, B normal) is not used on this path");
    XC_REQUIRE(aligned16(A) && aligned16(B) && aligned16(C) && aligned16(bias) && aligned16(residual) && aligned16(addrows) && aligned16(workspace),
               "pointers must be 16-byte aligned");
    XC_REQUIRE(residual == nullptr || ldr % vec == 0, "ldr must be a multiple of the 16-byte chunk");
    XC_REQUIRE(addrows == nullptr || (rowidx != nullptr && ld_add % vec == 0), "addrows needs rowidx and an aligned ld");
    if (M == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const bool plain = (bias == nullptr && residual == nullptr && addrows == nullptr);
    if (dtype == XCLIP_BF16 && gs_takes(M, N, K, bias != nullptr || addrows != nullptr, lda, ldb, g_small_flop))
        return gemm_small_run(a_kmajor, b_kmajor, A, lda, B, ldb, C, ldc, M, N, K, alpha, residual, ldr, st);
    if (use_gemm2(M, N, K, dtype)) {
        // the row tail as a split-K problem of its own (gemm2_tail_cut): the main rows, then the tail's slabs, then the reduction that also
        // applies alpha and the skip term
        const int64_t cut = (!a_kmajor && bias == nullptr && addrows == nullptr && workspace != nullptr) ? gemm2_tail_cut(M, N, K) : 0;
        const int64_t mt = M - cut;
        if (cut > 0 && gemm2_splits(mt, N, K) > 1 && workspace_bytes >= (int64_t)gemm2_splits(mt, N, K) * mt * N * 4) {
            int rc = gemm2_run(a_kmajor, b_kmajor, A, lda, B, ldb, C, ldc, cut, N, K, alpha, nullptr, residual, ldr, nullptr, nullptr, 0, nullptr, 0, st,
                               true, nullptr);
            if (rc != 0) return rc;
            const char* At = (const char*)A + cut * lda * 2;     // (bf16: use_gemm2)
            char* Ct = (char*)C + cut * ldc * 2;
            const char* Rt = residual != nullptr ? (const char*)residual + cut * ldr * 2 : nullptr;
            int tsplits = 1;
            rc = gemm2_run(a_kmajor, b_kmajor, At, lda, B, ldb, Ct, ldc, mt, N, K, alpha, nullptr, nullptr, 0, nullptr, nullptr, 0, workspace,
                           workspace_bytes, st, false, &tsplits);
            if (rc != 0) return rc;
            if (tsplits > 1) {
                int64_t blocks = (mt * (N / 4) + 255) / 256;
                if (blocks > 4096) blocks = 4096;
                hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, (const float*)workspace, (bf16_t*)Ct,
                                   (long)ldc, (int)mt, (int)N, tsplits, alpha, (const bf16_t*)Rt, (long)ldr);
            } else if (Rt != nullptr) {
                return xcapi::fail(__func__, "internal: tail GEMM did not split");
            }
            return check_launch(__func__);
        }
        return gemm2_run(a_kmajor, b_kmajor, A, lda, B, ldb, C, ldc, M, N, K, alpha, bias, residual, ldr, addrows, rowidx, ld_add, workspace,
                         workspace_bytes, st, true, nullptr);
    }
    GemmParams p;
    p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.alpha = alpha;
    p.bias = bias; p.residual = residual; p.ldr = ldr; p.addrows = addrows; p.rowidx = rowidx; p.ld_add = ld_add;
    p.tiles_m = (int)((M + 127) / 128); p.tiles_n = (int)((N + 127) / 128);
    int splits = gemm_splits(M, N, K, dtype);
    if (splits > 1 && (!plain || workspace == nullptr || workspace_bytes < (int64_t)splits * M * N * 4)) splits = 1;
    const int bk = 8 * vec;
    p.k_per_split = (int)((((K + splits - 1) / splits) + bk - 1) / bk * bk);
    splits = (int)((K + p.k_per_split - 1) / p.k_per_split);      // no empty slices (see above)
    p.partial = splits > 1 ? (float*)workspace : nullptr;
    if (dtype == XCLIP_BF16) {
        if (!a_kmajor && !b_kmajor) launch_gemm<bf16_t, false, false>(p, splits, st);
        else if (!a_kmajor && b_kmajor) launch_gemm<bf16_t, false, true>(p, splits, st);
        else launch_gemm<bf16_t, true, true>(p, splits, st);
    } else {
        if (!a_kmajor && !b_kmajor) launch_gemm<float, false, false>(p, splits, st);
        else if (!a_kmajor && b_kmajor) launch_gemm<float, false, true>(p, splits, st);
        else launch_gemm<float, true, true>(p, splits, st);
    }
    if (splits > 1) {
        int64_t blocks = (M * (N / 4) + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        if (dtype == XCLIP_BF16)
            hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, (const float*)workspace,
                               (bf16_t*)C, (long)ldc, (int)M, (int)N, splits, alpha);
        else
            hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, (const float*)workspace,
                               (float*)C, (long)ldc, (int)M, (int)N, splits, alpha);
    }
    return check_launch(__func__);
}

int xclip_ffn_dgrad_geglu_ok(int64_t M, int64_t F, int64_t D, int dtype) {
    return dtype == XCLIP_BF16 && M > 0 && M % G2_BM == 0 && F % G2_BN == 0 && D % G2_BK == 0 && D >= 2 * G2_BK && D <= 4096 && F < (1L << 20) &&
           (M / G2_BM) * (F / G2_BN) < (1L << 30);
}
int64_t xclip_ffn_dgrad_geglu_workspace_bytes(int64_t M, int64_t F, int64_t D) {
    return ((D + 3) / 4 * 4 + 4 * M + 2 * (M / G2_BM) * F) * 4;
}
// rowc_in != nullptr: the rows' constants are already there (written by xclip_layernorm_bwd_ffnstats in the pass that produced dout);
// otherwise the weight vector and the row pass run here (x2 / x1 required)
static int ffn_dgrad_geglu_run(const char* fn, const void* dout, int64_t ldd, const void* w2, int64_t ldw, const void* x, int64_t ldx, const void* gamma,
                               const float* mean, const float* rstd, const void* x2, int64_t ld2, const void* x1, int64_t ld1, const float* rowc_in,
                               void* dx, int64_t lddx, float* dg_accum, void* workspace, int64_t workspace_bytes, int64_t M, int64_t F, int64_t D,
                               int dtype, void* stream) {
#define XC_REQ(cond, msg) do { if (!(cond)) return xcapi::fail(fn, msg); } while (0)
    XC_REQ(xclip_ffn_dgrad_geglu_ok(M, F, D, dtype), "shape / dtype not taken by the fused kernel (xclip_ffn_dgrad_geglu_ok)");
    XC_REQ(dout && w2 && x && gamma && dx && dg_accum, "null pointer");
    XC_REQ(rowc_in != nullptr || (mean && rstd && x2 && x1), "null pointer");
    XC_REQ(aligned16(dout) && aligned16(w2) && aligned16(x) && aligned16(gamma) && aligned16(x2) && aligned16(x1) && aligned16(dx) && aligned16(workspace) && aligned16(rowc_in),
           "pointers must be 16-byte aligned");
    XC_REQ(ldd % 8 == 0 && ldw % 8 == 0 && ldx % 8 == 0 && ld2 % 8 == 0 && ld1 % 8 == 0 && lddx % 8 == 0, "leading dimensions must be multiples of the 16-byte chunk");
    XC_REQ(ldd >= D && (rowc_in != nullptr || (ld2 >= D && ld1 >= D)) && ldw >= F && ldx >= 2 * F && lddx >= 2 * F, "leading dimension too small");
    XC_REQ(ldd < (1L << 22) && ldw < (1L << 22) && ldx < (1L << 22) && lddx < (1L << 22), "leading dimensions beyond the 32-bit tile offsets");
    XC_REQ(workspace != nullptr && workspace_bytes >= xclip_ffn_dgrad_geglu_workspace_bytes(M, F, D), "workspace too small");
#undef XC_REQ
    hipStream_t st = (hipStream_t)stream;
    float* wg = (float*)workspace;
    float* rowc = wg + (D + 3) / 4 * 4;
    float* slab = rowc + 4 * M;
    if (rowc_in == nullptr) {
        hipLaunchKernelGGL(ffn_wgamma_kernel, dim3((unsigned)((D + 3) / 4)), dim3(256), 0, st, (const bf16_t*)w2, (long)ldw, (const bf16_t*)gamma, wg, (int)D, (int)F);
        hipLaunchKernelGGL(ffn_rowstats_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, (const bf16_t*)dout, (long)ldd, (const bf16_t*)x2, (long)ld2,
                           (const bf16_t*)x1, (long)ld1, (const float*)wg, mean, rstd, rowc, (int)M, (int)D, 1.0f / (float)F);
    } else {
        rowc = const_cast<float*>(rowc_in);
    }
    Gemm2Params q{};
    q.A = (const bf16_t*)dout; q.B = (const bf16_t*)w2; q.C = nullptr; q.lda = ldd; q.ldb = ldw; q.ldc = F;
    q.M = (int)M; q.N = (int)F; q.K = (int)D; q.alpha = 1.f;
    q.bias = nullptr; q.residual = nullptr; q.ldr = 0; q.addrows = nullptr; q.rowidx = nullptr; q.ld_add = 0; q.partial = nullptr;
    q.k_per_split = (int)D;
    q.tiles_m = (int)(M / G2_BM); q.tiles_n = (int)(F / G2_BN);
    q.band_n = 0;
    if (q.tiles_n > 8)
        for (int b = 8; b >= 4 && q.band_n == 0; --b)
            if (q.tiles_n % b == 0) q.band_n = b;
    q.stream_out = 1;
    GegluBwdArgs e;
    e.x = (const bf16_t*)x; e.ldx = ldx; e.dx = (bf16_t*)dx; e.lddx = lddx; e.gamma = (const bf16_t*)gamma;
    e.rowc = rowc; e.dg_partial = slab; e.F = (int)F;

    int gx = q.tiles_m * q.tiles_n;
    const int cus = xc_num_cus();
    if (gx > cus) gx = cus;
    // (round 5's two-work-groups-per-CU form of this kernel -- 256 x 128 tiles, four waves each -- measured slower, 1457 against 1277 us, and
    //  was not correct beyond one tile per work-group on the hardware; removed in round 6, log: profiles/r05_aa_gemm10_two_groups_ab.log)
    {
#define XC_G9(N) do { XC_ALLOW_LDS(gemm9_geglu_bwd_kernel<N>, G5_LDS_BYTES); \
        hipLaunchKernelGGL(gemm9_geglu_bwd_kernel<N>, dim3((unsigned)gx, 1), dim3(G2_THREADS), G5_LDS_BYTES, st, q, e); } while (0)
#ifdef XCLIP_MEASURE
    static const int abl = measure_env("XCLIP_GEMM9_ABL", 0);
    switch (abl) {
        case 1: XC_G9(1); break;
        case 2: XC_G9(2); break;
        case 4: XC_G9(4); break;
        case 8: XC_G9(8); break;
        case 14: XC_G9(14); break;
        default: XC_G9(0); break;
    }
#else
    XC_G9(0);
#endif
#undef XC_G9
    }
    const int nrows = 2 * q.tiles_m;
    int slices = nrows / 64;
    if (slices < 1) slices = 1;
    hipLaunchKernelGGL(colsum_fold_kernel, dim3((unsigned)((F + 63) / 64), (unsigned)slices), dim3(256), 1024, st, (const float*)slab, (long)F, dg_accum,
                       nrows, (int)F);
    return check_launch(fn);
}
int xclip_ffn_dgrad_geglu(const void* dout, int64_t ldd, const void* w2, int64_t ldw, const void* x, int64_t ldx, const void* gamma,
                          const float* mean, const float* rstd, const void* x2, int64_t ld2, const void* x1, int64_t ld1, void* dx,
                          int64_t lddx, float* dg_accum, void* workspace, int64_t workspace_bytes, int64_t M, int64_t F, int64_t D,
                          int dtype, void* stream) {
    XC_REQUIRE(mean && rstd && x2 && x1, "null pointer");
    return ffn_dgrad_geglu_run(__func__, dout, ldd, w2, ldw, x, ldx, gamma, mean, rstd, x2, ld2, x1, ld1, nullptr, dx, lddx, dg_accum, workspace,
                               workspace_bytes, M, F, D, dtype, stream);
}
int xclip_ffn_dgrad_geglu_rowc(const void* dout, int64_t ldd, const void* w2, int64_t ldw, const void* x, int64_t ldx, const void* gamma,
                               const float* rowc, void* dx, int64_t lddx, float* dg_accum, void* workspace, int64_t workspace_bytes,
                               int64_t M, int64_t F, int64_t D, int dtype, void* stream) {
    XC_REQUIRE(rowc != nullptr, "null pointer");
    return ffn_dgrad_geglu_run(__func__, dout, ldd, w2, ldw, x, ldx, gamma, nullptr, nullptr, nullptr, 0, nullptr, 0, rowc, dx, lddx, dg_accum, workspace,
                               workspace_bytes, M, F, D, dtype, stream);
}
int xclip_ffn_wgamma(const void* w2, int64_t ldw, const void* gamma, float* wg, int64_t D, int64_t F, int dtype, void* stream) {
    XC_REQUIRE(dtype == XCLIP_BF16, "bf16 only (the fused feed-forward backward)");
    XC_REQUIRE(w2 && gamma && wg && aligned16(w2) && aligned16(gamma) && aligned16(wg), "null or misaligned pointer");
    XC_REQUIRE(D > 0 && F > 0 && F % 8 == 0 && ldw % 8 == 0 && ldw >= F, "bad shape");
    hipLaunchKernelGGL(ffn_wgamma_kernel, dim3((unsigned)((D + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w2, (long)ldw, (const bf16_t*)gamma, wg, (int)D, (int)F);
    return check_launch(__func__);
}

int xclip_gemm_batched(int a_kmajor, int b_kmajor, const void* A, int64_t lda, int64_t stride_a, const void* B, int64_t ldb, int64_t stride_b,
                       void* C, int64_t ldc, int64_t stride_c, int64_t batch, int64_t M, int64_t N, int64_t K, float alpha, int dtype,
                       void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    const int vec = vec_of(dtype);
    XC_REQUIRE(batch >= 0 && batch <= 65535 && M >= 0 && N > 0 && K > 0, "bad shape (up to 65535 problems per launch)");
    XC_REQUIRE(N % vec == 0 && ldc % vec == 0 && lda % vec == 0 && ldb % vec == 0 && stride_a % vec == 0 && stride_b % vec == 0 && stride_c % vec == 0,
               "N, the leading dimensions and the batch strides must be multiples of the 16-byte chunk");
    const int64_t Kp = (K + vec - 1) / vec * vec, Mp = (M + vec - 1) / vec * vec;
    XC_REQUIRE(a_kmajor ? (lda >= Mp) : (lda >= Kp), "lda must cover A's contiguous dim rounded up to the 16-byte chunk");
    XC_REQUIRE(b_kmajor ? (ldb >= N) : (ldb >= Kp), "ldb must cover B's contiguous dim rounded up to the 16-byte chunk");
    XC_REQUIRE(!(a_kmajor && !b_kmajor), "layout (A k-major, B normal) is not used on this path");
    XC_REQUIRE(aligned16(A) && aligned16(B) && aligned16(C), "pointers must be 16-byte aligned");
    if (M == 0 || batch == 0) return 0;
    GemmParams p;
    p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.alpha = alpha;
    p.bias = nullptr; p.residual = nullptr; p.ldr = 0; p.addrows = nullptr; p.rowidx = nullptr; p.ld_add = 0; p.partial = nullptr;
    p.tiles_m = (int)((M + 127) / 128); p.tiles_n = (int)((N + 127) / 128);
    p.k_per_split = (int)((K + 8 * vec - 1) / (8 * vec) * (8 * vec));
    p.batch_a = stride_a; p.batch_b = stride_b; p.batch_c = stride_c;
    dim3 grid(p.tiles_m * p.tiles_n, 1, (unsigned)batch), block(GEMM_THREADS);
    hipStream_t st = (hipStream_t)stream;
#define XC_GB(T, AK, BK_) do { XC_ALLOW_LDS((gemm_kernel<T, AK, BK_>), GemmCfg<T>::LDS_BYTES); hipLaunchKernelGGL((gemm_kernel<T, AK, BK_>), grid, block, GemmCfg<T>::LDS_BYTES, st, p); } while (0)
    if (dtype == XCLIP_BF16) {
        if (!a_kmajor && !b_kmajor) XC_GB(bf16_t, false, false); else if (!a_kmajor) XC_GB(bf16_t, false, true); else XC_GB(bf16_t, true, true);
    } else {
        if (!a_kmajor && !b_kmajor) XC_GB(float, false, false); else if (!a_kmajor) XC_GB(float, false, true); else XC_GB(float, true, true);
    }
#undef XC_GB
    return check_launch(__func__);
}

int xclip_rowdot(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t rows, int64_t dim, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    const int vec = vec_of(dtype);
    XC_REQUIRE(dim > 0 && dim % vec == 0 && lda % vec == 0 && ldb % vec == 0 && lda >= dim && ldb >= dim, "dim / leading dimensions must be multiples of the 16-byte chunk");
    XC_REQUIRE(aligned16(a) && aligned16(b) && out, "null or misaligned pointer");
    if (rows == 0) return 0;
    const int cpl = chunks_per_lane(dim, vec);
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define F(T, C) hipLaunchKernelGGL((rowdot_kernel<T, C>), grid, block, 0, (hipStream_t)stream, (const T*)a, (long)lda, (const T*)b, (long)ldb, (T*)out, (int)rows, (int)dim)
    XC_DISPATCH_ROW(dtype, cpl, F);
#undef F
    return check_launch(__func__);
}

int xclip_filip_reduce(const void* S, int64_t lds, const uint8_t* mask, const float* log_temp, float* t2i, float* i2t, int64_t ldo,
                       int16_t* kmax, int16_t* tmax, float* cnt, int64_t bx, int64_t nt, int64_t yc, int64_t ni, int64_t y0,
                       int64_t ytotal, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(bx > 0 && nt > 0 && yc > 0 && ni > 0 && ni <= 256, "FILIP reductions support up to 256 image tokens");
    XC_REQUIRE(lds >= yc * ni && y0 >= 0 && y0 + yc <= ytotal && ldo >= ytotal, "bad chunk geometry");
    XC_REQUIRE(S && mask && log_temp && t2i && i2t && kmax && tmax && cnt, "null pointer");
    // chunk rows that fit the register budget of the row-coalesced kernel (the host sizes its chunks for it) take that one
    if (yc * ni <= 256 * vec_of(dtype) * FILIP_MAXCH && yc <= 2048 && ni >= vec_of(dtype) && lds % vec_of(dtype) == 0 && aligned16(S)) {
        // split the chunk's images over blockIdx.y only when there are fewer text samples than CUs; sub-ranges start on 16-byte chunks
        const int vec = vec_of(dtype);
        int64_t g = ni, h = vec;
        while (h) { const int64_t r = g % h; g = h; h = r; }          // gcd(ni, vec)
        const int64_t step = vec / g;
        int64_t nsplit = ((int64_t)xc_num_cus() + bx - 1) / bx;          // (more, shorter rows per work-group measured slower: the per-row cost is fixed)
        if (nsplit < 1) nsplit = 1;
        int64_t ysplit = ((yc + nsplit - 1) / nsplit + step - 1) / step * step;
        if (ysplit < step) ysplit = step;
        nsplit = (yc + ysplit - 1) / ysplit;
        const int64_t nchunks = (ysplit * ni + vec - 1) / vec;
        const size_t shm = (size_t)nchunks * 32 + (size_t)ysplit * 4;
        XC_ALLOW_LDS((filip_reduce_rows_kernel<bf16_t>), 96 * 1024);
        XC_ALLOW_LDS((filip_reduce_rows_kernel<float>), 96 * 1024);
        dim3 g2((unsigned)bx, (unsigned)nsplit), b2(256);
        if (dtype == XCLIP_BF16)
            hipLaunchKernelGGL((filip_reduce_rows_kernel<bf16_t>), g2, b2, shm, (hipStream_t)stream, (const bf16_t*)S, (long)lds, mask, log_temp, t2i, i2t, (long)ldo, kmax, tmax, cnt, (int)bx, (int)nt, (int)yc, (int)ni, (int)y0, (int)ytotal, (int)ysplit);
        else
            hipLaunchKernelGGL((filip_reduce_rows_kernel<float>), g2, b2, shm, (hipStream_t)stream, (const float*)S, (long)lds, mask, log_temp, t2i, i2t, (long)ldo, kmax, tmax, cnt, (int)bx, (int)nt, (int)yc, (int)ni, (int)y0, (int)ytotal, (int)ysplit);
        return check_launch(__func__);
    }
    dim3 grid((unsigned)((bx * yc + 3) / 4)), block(256);
    if (dtype == XCLIP_BF16)
        hipLaunchKernelGGL((filip_reduce_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)S, (long)lds, mask, log_temp, t2i, i2t, (long)ldo, kmax, tmax, cnt, (int)bx, (int)nt, (int)yc, (int)ni, (int)y0, (int)ytotal);
    else
        hipLaunchKernelGGL((filip_reduce_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)S, (long)lds, mask, log_temp, t2i, i2t, (long)ldo, kmax, tmax, cnt, (int)bx, (int)nt, (int)yc, (int)ni, (int)y0, (int)ytotal);
    return check_launch(__func__);
}

int xclip_filip_fused_ok(int64_t nt, int64_t ni, int64_t d, int dtype) {
    return dtype == XCLIP_BF16 && d > 0 && d % G2_BK == 0 && nt >= 32 && ni >= 32 && nt <= 32767 && ni <= 32767;
}
int64_t xclip_filip_fused_workspace_bytes(int64_t bx, int64_t nt, int64_t yc, int64_t ni) {
    const int64_t M = bx * nt, N = yc * ni;
    return (M * ((N + 63) / 64) * F5_SIDES + ((M + 127) / 128) * F5_SLOTS * N) * 4;
}
int xclip_filip_fused_fwd(const void* X, const uint8_t* mask, const void* Y, const float* log_temp, float* t2i, float* i2t, int64_t ldo,
                          int16_t* kmax, int16_t* tmax, float* cnt, void* workspace, int64_t workspace_bytes, int64_t bx, int64_t nt,
                          int64_t yc, int64_t ni, int64_t d, int64_t y0, int64_t ytotal, int dtype, void* stream) {
    XC_REQUIRE(xclip_filip_fused_ok(nt, ni, d, dtype), "the fused FILIP forward needs bf16, d % 64 == 0, nt >= 32, ni >= 32");
    XC_REQUIRE(bx > 0 && yc > 0 && y0 >= 0 && y0 + yc <= ytotal && ldo >= ytotal, "bad chunk geometry");
    XC_REQUIRE(bx * nt < (1LL << 31) && yc * ni < (1LL << 31), "problem too large for 32-bit tile indices");
    XC_REQUIRE(X && Y && mask && log_temp && t2i && i2t && kmax && tmax && cnt && aligned16(X) && aligned16(Y), "null or misaligned pointer");
    XC_REQUIRE(workspace && workspace_bytes >= xclip_filip_fused_workspace_bytes(bx, nt, yc, ni), "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    Filip5Params f;
    f.mask = mask;
    f.M = (int)(bx * nt); f.N = (int)(yc * ni); f.nt = (int)nt; f.ni = (int)ni; f.nblk64 = (int)((f.N + 63) / 64);
    f.rowpart = (uint32_t*)workspace;
    f.colpart = f.rowpart + (int64_t)f.M * f.nblk64 * F5_SIDES;
    const int64_t tiles = ((f.M + G2_BM - 1) / G2_BM) * (int64_t)((f.N + G2_BN - 1) / G2_BN);
    const int cus = xc_num_cus();
    XC_ALLOW_LDS(filip5_kernel, G5_LDS_BYTES);
    hipLaunchKernelGGL(filip5_kernel, dim3((unsigned)(tiles < cus ? tiles : cus)), dim3(G2_THREADS), G5_LDS_BYTES, st, (const bf16_t*)X,
                       (const bf16_t*)Y, (int)d, f);
    hipLaunchKernelGGL(filip5_merge_rows_kernel, dim3((unsigned)bx, (unsigned)((yc + 255) / 256)), dim3(256), 0, st, f.rowpart, mask, log_temp, t2i,
                       (long)ldo, kmax, cnt, (int)nt, (int)ni, (int)yc, f.nblk64, (int)y0, (int)ytotal);
    const int ypb = (int)(256 / ni > 0 ? 256 / ni : 1);
    hipLaunchKernelGGL(filip5_merge_cols_kernel, dim3((unsigned)bx, (unsigned)((yc + ypb - 1) / ypb)), dim3(256), (size_t)ypb * ni * 4, st, f.colpart,
                       log_temp, i2t, (long)ldo, tmax, (int)nt, (int)ni, (int)yc, f.N, ypb, (int)y0, (int)ytotal);
    return check_launch(__func__);
}

int xclip_filip_route(void* P, int64_t ldp, const uint8_t* mask, const float* log_temp, const float* g1, const float* g2, int64_t ldg,
                      const int16_t* kmax, const int16_t* tmax, const float* cnt, int64_t bx, int64_t nt, int64_t yc, int64_t ni,
                      int64_t y0, int64_t ytotal, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(bx > 0 && nt > 0 && yc > 0 && ni > 0, "bad shape");
    XC_REQUIRE(ldp % vec_of(dtype) == 0 && ldp >= yc * ni && aligned16(P), "ldp must cover a chunk row, 16-byte chunk aligned");
    XC_REQUIRE(P && mask && log_temp && g1 && g2 && kmax && tmax && cnt, "null pointer");
    XC_REQUIRE(bx * nt < (1LL << 31) && ldp / vec_of(dtype) < (1LL << 31) && bx <= 65535, "chunk too large for the launch grid");
    // one work-group per (text x, slice of 256 chunks): blockIdx.y = x
    dim3 grid((unsigned)((ldp / vec_of(dtype) + 255) / 256), (unsigned)bx), block(256);
    if (dtype == XCLIP_BF16) {
        XC_ALLOW_LDS((filip_route_kernel<bf16_t>), ROUTE_LDS_BYTES);
        hipLaunchKernelGGL((filip_route_kernel<bf16_t>), grid, block, ROUTE_LDS_BYTES, (hipStream_t)stream, (bf16_t*)P, (long)ldp, mask, log_temp, g1, g2, (long)ldg, kmax, tmax, cnt, (int)bx, (int)nt, (int)yc, (int)ni, (int)y0, (int)ytotal);
    } else {
        XC_ALLOW_LDS((filip_route_kernel<float>), ROUTE_LDS_BYTES);
        hipLaunchKernelGGL((filip_route_kernel<float>), grid, block, ROUTE_LDS_BYTES, (hipStream_t)stream, (float*)P, (long)ldp, mask, log_temp, g1, g2, (long)ldg, kmax, tmax, cnt, (int)bx, (int)nt, (int)yc, (int)ni, (int)y0, (int)ytotal);
    }
    return check_launch(__func__);
}

int xclip_rowlse(const float* S, int64_t lds, int64_t rows, int64_t cols, int64_t diag_off, int dcl, float coef, float* lse,
                 float* loss_accum, void* stream) {
    XC_REQUIRE(S && lse && rows > 0 && cols > 0 && lds >= cols, "bad arguments");
    hipLaunchKernelGGL(rowlse_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, (long)lds, (int)rows, (int)cols, (int)diag_off, dcl, coef, lse, loss_accum);
    return check_launch(__func__);
}

int xclip_rowgrad(const float* S, int64_t lds, const float* lse, int64_t rows, int64_t cols, int64_t diag_off, int dcl, float coef,
                  const float* gmul, float* G, int64_t ldg, float* dtau_accum, void* stream) {
    XC_REQUIRE(S && lse && G && rows > 0 && cols > 0 && lds >= cols && ldg >= cols, "bad arguments");
    int64_t blocks = (rows * cols + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(rowgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, S, (long)lds, lse, (int)rows, (int)cols, (int)diag_off, dcl, coef, gmul, G, (long)ldg, dtau_accum);
    return check_launch(__func__);
}

int64_t xclip_simloss_workspace_bytes(int64_t nq, int64_t nk) { return 2 * ((nk + 63) / 64) * nq * 4; }

// the gemm3-based head kernels take bf16 problems at least a tile wide whose feature dim is a whole number of K steps
inline bool use_sim3(int64_t nq, int64_t nk, int64_t d, int dtype) {
    return dtype == XCLIP_BF16 && d % G2_BK == 0 && nq >= 128 && nk >= 128;
}
extern "C++" inline dim3 sim3_grid(int64_t nq, int64_t nk) {
    int64_t tiles = ((nq + G2_BM - 1) / G2_BM) * ((nk + G2_BN - 1) / G2_BN);
    const int cus = xc_num_cus();
    return dim3((unsigned)(tiles < cus ? tiles : cus));
}

int xclip_simloss_partial(const void* Q, const void* K, int64_t nq, int64_t nk, int64_t d, float scale, const float* log_scale,
                          int64_t diag_off, int dcl, void* workspace, int64_t tile_slot0, int64_t tile_slots, float* pos, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(nq > 0 && nk > 0 && d > 0 && d % vec_of(dtype) == 0, "bad shape (d must be a multiple of the 16-byte chunk)");
    XC_REQUIRE(aligned16(Q) && aligned16(K) && workspace != nullptr, "pointers must be 16-byte aligned / workspace required");
    SimParams p;
    memset(&p, 0, sizeof(p));
    p.Q = Q; p.K = K; p.nq = (int)nq; p.nk = (int)nk; p.d = (int)d; p.scale = scale; p.log_scale = log_scale;
    p.diag_off = (int)diag_off; p.dcl = dcl;
    p.tiles_m = (int)((nq + 127) / 128); p.tiles_n = (int)((nk + 127) / 128);
    XC_REQUIRE(tile_slot0 >= 0 && tile_slot0 + (nk + 63) / 64 <= tile_slots, "column slots out of range");
    p.part_m = (float*)workspace + tile_slot0 * nq; p.part_l = (float*)workspace + (tile_slots + tile_slot0) * nq; p.pos = pos;
    hipStream_t st = (hipStream_t)stream;
    if (use_sim3(nq, nk, d, dtype)) {
        static const int gen = measure_env("XCLIP_SIM", 5);      // measurement build: 3 = the round-1 two-stage loop (simloss3.h)
        if (gen == 3) {
            XC_ALLOW_LDS(sim3_lse_kernel, G2_LDS_BYTES);
            hipLaunchKernelGGL(sim3_lse_kernel, sim3_grid(nq, nk), dim3(G2_THREADS), G2_LDS_BYTES, st, p);
        } else {
            XC_ALLOW_LDS(sim5_lse_kernel, G5_LDS_BYTES);
            hipLaunchKernelGGL(sim5_lse_kernel, sim3_grid(nq, nk), dim3(G2_THREADS), G5_LDS_BYTES, st, p);
        }
        return check_launch(__func__);
    }
    dim3 grid(p.tiles_m * p.tiles_n), block(256);
    if (dtype == XCLIP_BF16) {
        XC_ALLOW_LDS((sim_lse_partial_kernel<bf16_t>), GemmCfg<bf16_t>::LDS_BYTES);
        hipLaunchKernelGGL((sim_lse_partial_kernel<bf16_t>), grid, block, GemmCfg<bf16_t>::LDS_BYTES, st, p);
    } else {
        XC_ALLOW_LDS((sim_lse_partial_kernel<float>), GemmCfg<float>::LDS_BYTES);
        hipLaunchKernelGGL((sim_lse_partial_kernel<float>), grid, block, GemmCfg<float>::LDS_BYTES, st, p);
    }
    return check_launch(__func__);
}

int xclip_simloss_combine(const void* workspace, int64_t nq, int64_t tile_slots, const float* pos, float* lse, float* loss_accum,
                          float coef, void* stream) {
    XC_REQUIRE(nq > 0 && tile_slots > 0 && workspace != nullptr && pos != nullptr && lse != nullptr, "bad arguments");
    const float* part_m = (const float*)workspace;
    hipLaunchKernelGGL(sim_lse_combine_kernel, dim3((unsigned)((nq + 63) / 64)), dim3(1024), 2 * 16 * 64 * 4, (hipStream_t)stream, part_m,
                       part_m + tile_slots * nq, pos, lse, loss_accum, (int)nq, (int)tile_slots, coef);
    return check_launch(__func__);
}

int xclip_simloss_fwd(const void* Q, const void* K, int64_t nq, int64_t nk, int64_t d, float scale, const float* log_scale,
                      int64_t diag_off, int dcl, float coef, void* workspace, float* pos, float* lse, float* loss_accum, int dtype,
                      void* stream) {
    const int64_t slots = (nk + 63) / 64;
    const int rc = xclip_simloss_partial(Q, K, nq, nk, d, scale, log_scale, diag_off, dcl, workspace, 0, slots, pos, dtype, stream);
    if (rc != 0) return rc;
    return xclip_simloss_combine(workspace, nq, slots, pos, lse, loss_accum, coef, stream);
}

int xclip_simloss_grad(const void* Q, const void* K, int64_t nq, int64_t nk, int64_t d, float scale, const float* log_scale,
                       int64_t diag_off, int dcl, float a, float c, float e, const float* gmul, int g_times_scale,
                       const float* lse_q, const float* lse_k, void* G, int64_t ldg, float* dtau_accum, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    const int vec = vec_of(dtype);
    XC_REQUIRE(nq > 0 && nk > 0 && d > 0 && d % vec == 0, "bad shape (d must be a multiple of the 16-byte chunk)");
    XC_REQUIRE(ldg % vec == 0 && ldg >= (nk + vec - 1) / vec * vec, "ldg must cover nk rounded up to the chunk");
    XC_REQUIRE(aligned16(Q) && aligned16(K) && aligned16(G), "pointers must be 16-byte aligned");
    SimParams p;
    memset(&p, 0, sizeof(p));
    p.Q = Q; p.K = K; p.nq = (int)nq; p.nk = (int)nk; p.d = (int)d; p.scale = scale; p.log_scale = log_scale;
    p.gmul = gmul; p.g_times_scale = g_times_scale; p.diag_off = (int)diag_off; p.dcl = dcl;
    p.tiles_m = (int)((nq + 127) / 128); p.tiles_n = (int)((nk + 127) / 128);
    p.lse_q = lse_q; p.lse_k = lse_k; p.a = a; p.c = c; p.e = e; p.G = G; p.ldg = ldg; p.dtau = dtau_accum;
    hipStream_t st = (hipStream_t)stream;
    if (use_sim3(nq, nk, d, dtype)) {
        // simloss5.h: every full 256 x 256 tile (on the diagonal or off it) on the ring loop with a spill-free epilogue of its own; tiles at
        // a ragged edge, if there are any, through simloss3.h's general epilogue over a tile list in a second launch.  XCLIP_SIM
        // (measurement build): 3 = simloss3.h alone (one launch, every tile through the general epilogue: 355 - 368 us at
        // 4096 x 32768 x 512), 5 = the first ring form (whole-line epilogue with the general tile in the same function: 423 us)
#ifdef XCLIP_MEASURE
        static const int gen = measure_env("XCLIP_SIM", 0);
        if (gen == 5) {
            XC_ALLOW_LDS(sim5_grad_kernel, G5_LDS_BYTES);
            hipLaunchKernelGGL(sim5_grad_kernel, sim3_grid(nq, nk), dim3(G2_THREADS), G5_LDS_BYTES, st, p);
            return check_launch(__func__);
        }
        if (gen == 3) {
            XC_ALLOW_LDS(sim3_grad_kernel, G2_LDS_BYTES);
            hipLaunchKernelGGL(sim3_grad_kernel, sim3_grid(nq, nk), dim3(G2_THREADS), G2_LDS_BYTES, st, p);
            return check_launch(__func__);
        }
        // 7 = the full-tile launch alone, 8 = the edge launch alone (the split of the two)
        const bool skip_fast = gen == 8, skip_edge = gen == 7;
#else
        const bool skip_fast = false, skip_edge = false;
#endif
#ifdef XCLIP_MEASURE
        static const int simg = measure_env("XCLIP_SIMG", 0);      // A/B of this round's additions to the G kernel (simloss5.h VAR), streamed form only
        if (!skip_fast && simg >= 1 && simg <= 4 && nq * ldg * 2 > (48LL << 20)) {
#define XC_SIMG(V) case V: XC_ALLOW_LDS((sim5_grad_fast_kernel<true, V>), G5_LDS_BYTES); hipLaunchKernelGGL((sim5_grad_fast_kernel<true, V>), sim3_grid(nq, nk), dim3(G2_THREADS), G5_LDS_BYTES, st, p); break;
            switch (simg) { XC_SIMG(1) XC_SIMG(2) XC_SIMG(3) XC_SIMG(4) }
#undef XC_SIMG
        } else
#endif
        if (skip_fast) {
        } else if (nq * ldg * 2 > (48LL << 20)) {                   // G larger than the L2s can hold anyway: streamed stores
            XC_ALLOW_LDS(sim5_grad_fast_kernel<true>, G5_LDS_BYTES);
            hipLaunchKernelGGL(sim5_grad_fast_kernel<true>, sim3_grid(nq, nk), dim3(G2_THREADS), G5_LDS_BYTES, st, p);
        } else {
            XC_ALLOW_LDS(sim5_grad_fast_kernel<false>, G5_LDS_BYTES);
            hipLaunchKernelGGL(sim5_grad_fast_kernel<false>, sim3_grid(nq, nk), dim3(G2_THREADS), G5_LDS_BYTES, st, p);
        }
        const int64_t tm = (nq + G2_BM - 1) / G2_BM, tn = (nk + G2_BN - 1) / G2_BN;
        const int64_t nedge = ((nk % G2_BN) ? tm : 0) + ((nq % G2_BM) ? tn : 0);    // Sim5EdgeTiles::count
        if (skip_edge || nedge == 0) return check_launch(__func__);
        const int cus = xc_num_cus();
        XC_ALLOW_LDS(sim5_grad_edge_kernel, G2_LDS_BYTES);
        hipLaunchKernelGGL(sim5_grad_edge_kernel, dim3((unsigned)(nedge < cus ? nedge : cus)), dim3(G2_THREADS), G2_LDS_BYTES, st, p);
        return check_launch(__func__);
    }
    dim3 grid(p.tiles_m * p.tiles_n), block(256);
    if (dtype == XCLIP_BF16) {
        XC_ALLOW_LDS((sim_grad_kernel<bf16_t>), GemmCfg<bf16_t>::LDS_BYTES);
        hipLaunchKernelGGL((sim_grad_kernel<bf16_t>), grid, block, GemmCfg<bf16_t>::LDS_BYTES, st, p);
    } else {
        XC_ALLOW_LDS((sim_grad_kernel<float>), GemmCfg<float>::LDS_BYTES);
        hipLaunchKernelGGL((sim_grad_kernel<float>), grid, block, GemmCfg<float>::LDS_BYTES, st, p);
    }
    return check_launch(__func__);
}

int xclip_simreg_diff(const void* A, int64_t lda, const void* C, int64_t ldc, void* D, int64_t ldd, int64_t rows, int64_t cols,
                      int64_t diag_off, float* sumsq_accum, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    const int vec = vec_of(dtype);
    XC_REQUIRE(rows >= 0 && cols > 0 && cols % vec == 0 && lda % vec == 0 && ldc % vec == 0 && ldd % vec == 0,
               "cols and the row strides must be multiples of the 16-byte chunk");
    XC_REQUIRE(lda >= cols && ldc >= cols && ldd >= cols, "row stride smaller than the row");
    XC_REQUIRE(A && C && D && aligned16(A) && aligned16(C) && aligned16(D), "null or misaligned pointer");
    if (rows == 0) return 0;
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    dim3 grid((unsigned)blocks), block(256);
    if (dtype == XCLIP_BF16)
        hipLaunchKernelGGL((simreg_diff_kernel<bf16_t>), grid, block, 16, (hipStream_t)stream, (const bf16_t*)A, (long)lda, (const bf16_t*)C,
                           (long)ldc, (bf16_t*)D, (long)ldd, (int)rows, (int)cols, (int)diag_off, sumsq_accum);
    else
        hipLaunchKernelGGL((simreg_diff_kernel<float>), grid, block, 16, (hipStream_t)stream, (const float*)A, (long)lda, (const float*)C,
                           (long)ldc, (float*)D, (long)ldd, (int)rows, (int)cols, (int)diag_off, sumsq_accum);
    return check_launch(__func__);
}

namespace {
constexpr int BN_MAX_SLICES = 256;
inline ColNormGeom colnorm_geom(int64_t rows, int64_t cols, int dtype) {
    const int nch = (int)(cols / vec_of(dtype));
    ColNormGeom g;
    g.cw = 64;
    if (nch < 64) { g.cw = 1; while (g.cw < nch) g.cw *= 2; }
    g.slabs = (nch + g.cw - 1) / g.cw;
    const int64_t rows_per_wg = 4 * (64 / g.cw);
    int64_t slices = 2048 / g.slabs;
    const int64_t useful = (rows + rows_per_wg - 1) / rows_per_wg;
    if (slices > useful) slices = useful;
    if (slices > BN_MAX_SLICES) slices = BN_MAX_SLICES;
    if (slices < 1) slices = 1;
    g.slices = (int)slices;
    return g;
}
}  // namespace

int64_t xclip_batchnorm_workspace_bytes(int64_t rows, int64_t cols) {
    (void)rows;
    return (int64_t)(2 * BN_MAX_SLICES + 2) * cols * (int64_t)sizeof(float);
}

int xclip_batchnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, float* running_mean,
                        float* running_var, float momentum, float eps, int training, int relu, int64_t rows, int64_t cols,
                        void* workspace, int64_t workspace_bytes, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(rows > 0 && cols > 0 && cols % vec_of(dtype) == 0 && rows < (1LL << 31) && cols < (1LL << 24), "cols must be a multiple of the 16-byte chunk");
    XC_REQUIRE(x && y && mean && rstd && aligned16(x) && aligned16(y), "null or misaligned pointer");
    XC_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "running_mean and running_var come together");
    XC_REQUIRE(training || running_mean, "evaluation mode needs the running statistics");
    XC_REQUIRE(!training || rows > 1, "Expected more than 1 value per channel when training");
    XC_REQUIRE(!training || (workspace && workspace_bytes >= xclip_batchnorm_workspace_bytes(rows, cols)), "workspace too small");
    const ColNormGeom g = colnorm_geom(rows, cols, dtype);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)g.slabs, (unsigned)g.slices), block(256);
    const int R = (int)rows, C = (int)cols;
    const int fin_blocks = (C + 63) / 64;              // finalize: 64 columns x 4 slice strips per work-group
    float* part = (float*)workspace;
    if (training) {
        const int lds = 256 * 2 * vec_of(dtype) * (int)sizeof(float);
        if (dtype == XCLIP_BF16) {
            hipLaunchKernelGGL((bn_stats_kernel<bf16_t>), grid, block, lds, st, (const bf16_t*)x, part, R, C, g.cw);
            hipLaunchKernelGGL((bn_finalize_kernel<bf16_t>), dim3(fin_blocks), block, 2048, st, (const bf16_t*)x, part, g.slices, R, C, eps, momentum,
                               mean, rstd, running_mean, running_var);
        } else {
            hipLaunchKernelGGL((bn_stats_kernel<float>), grid, block, lds, st, (const float*)x, part, R, C, g.cw);
            hipLaunchKernelGGL((bn_finalize_kernel<float>), dim3(fin_blocks), block, 2048, st, (const float*)x, part, g.slices, R, C, eps, momentum,
                               mean, rstd, running_mean, running_var);
        }
    } else {
        hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(fin_blocks), block, 0, st, running_mean, running_var, eps, mean, rstd, C);
    }
#define BN_APPLY(T, RELU) hipLaunchKernelGGL((bn_apply_kernel<T, RELU>), grid, block, 0, st, (const T*)x, mean, rstd, gamma, beta, (T*)y, R, C, g.cw)
    if (dtype == XCLIP_BF16) { if (relu) BN_APPLY(bf16_t, true); else BN_APPLY(bf16_t, false); }
    else                     { if (relu) BN_APPLY(float, true); else BN_APPLY(float, false); }
#undef BN_APPLY
    return check_launch(__func__);
}

int xclip_batchnorm_bwd(const void* x, const void* dy, const float* gamma, const float* beta, const float* mean, const float* rstd,
                        void* dx, float* dgamma, float* dbeta, int training, int relu, int64_t rows, int64_t cols, void* workspace,
                        int64_t workspace_bytes, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(rows > 0 && cols > 0 && cols % vec_of(dtype) == 0 && rows < (1LL << 31) && cols < (1LL << 24), "cols must be a multiple of the 16-byte chunk");
    XC_REQUIRE(x && dy && dx && mean && rstd && aligned16(x) && aligned16(dy) && aligned16(dx), "null or misaligned pointer");
    XC_REQUIRE(workspace && workspace_bytes >= xclip_batchnorm_workspace_bytes(rows, cols), "workspace too small");
    const ColNormGeom g = colnorm_geom(rows, cols, dtype);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)g.slabs, (unsigned)g.slices), block(256);
    const int R = (int)rows, C = (int)cols;
    float* part = (float*)workspace;
    float* coef = part + (int64_t)2 * BN_MAX_SLICES * cols;
    const BnCols cols_p{mean, rstd, gamma, beta};
    const int lds = 256 * 2 * vec_of(dtype) * (int)sizeof(float);
#define BN_BWD(T, RELU)                                                                                                              \
    do {                                                                                                                             \
        hipLaunchKernelGGL((bn_bwd_sums_kernel<T, RELU>), grid, block, lds, st, (const T*)x, (const T*)dy, cols_p, part, R, C, g.cw); \
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), block, 2048, st, part, g.slices, R, C, training, dgamma, dbeta, coef); \
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T, RELU>), grid, block, 0, st, (const T*)x, (const T*)dy, cols_p, coef, (T*)dx, R, C, g.cw); \
    } while (0)
    if (dtype == XCLIP_BF16) { if (relu) BN_BWD(bf16_t, true); else BN_BWD(bf16_t, false); }
    else                     { if (relu) BN_BWD(float, true); else BN_BWD(float, false); }
#undef BN_BWD
    return check_launch(__func__);
}

int xclip_neg_cosine_fwd(const void* p, const void* z, int64_t rows, int64_t dim, float coef, float* cosv, float* rp, float* rz,
                         float* loss_accum, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(rows >= 0 && dim > 0 && dim % vec_of(dtype) == 0, "dim must be a multiple of the 16-byte chunk");
    if (rows == 0) return 0;                                  // (an empty tensor has a null data pointer)
    XC_REQUIRE(p && z && cosv && rp && rz && loss_accum && aligned16(p) && aligned16(z), "null or misaligned pointer");
    int64_t blocks = (rows + 3) / 4;
    if (blocks > ROWLOSS_MAX_BLOCKS) blocks = ROWLOSS_MAX_BLOCKS;
    dim3 grid((unsigned)blocks), block(256);
    if (dtype == XCLIP_BF16)
        hipLaunchKernelGGL((neg_cosine_fwd_kernel<bf16_t>), grid, block, 16, (hipStream_t)stream, (const bf16_t*)p, (const bf16_t*)z, (int)rows, (int)dim, coef, cosv, rp, rz, loss_accum);
    else
        hipLaunchKernelGGL((neg_cosine_fwd_kernel<float>), grid, block, 16, (hipStream_t)stream, (const float*)p, (const float*)z, (int)rows, (int)dim, coef, cosv, rp, rz, loss_accum);
    return check_launch(__func__);
}

int xclip_neg_cosine_bwd(const void* p, const void* z, const float* cosv, const float* rp, const float* rz, const float* gmul, float coef,
                         void* dp, int64_t rows, int64_t dim, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(rows >= 0 && dim > 0 && dim % vec_of(dtype) == 0, "dim must be a multiple of the 16-byte chunk");
    if (rows == 0) return 0;                                  // (an empty tensor has a null data pointer)
    XC_REQUIRE(p && z && cosv && rp && rz && gmul && dp && aligned16(p) && aligned16(z) && aligned16(dp), "null or misaligned pointer");
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dtype == XCLIP_BF16)
        hipLaunchKernelGGL((neg_cosine_bwd_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)p, (const bf16_t*)z, cosv, rp, rz, gmul, coef, (bf16_t*)dp, (int)rows, (int)dim);
    else
        hipLaunchKernelGGL((neg_cosine_bwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)p, (const float*)z, cosv, rp, rz, gmul, coef, (float*)dp, (int)rows, (int)dim);
    return check_launch(__func__);
}

}  // extern "C"
