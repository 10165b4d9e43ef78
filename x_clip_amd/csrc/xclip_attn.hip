// xclip_attn.hip -- the attention entry points of include/xclip.h.  A translation unit of its own because it is compiled with
// -mllvm -amdgpu-mfma-vgpr-form: the attention kernels consume every MFMA result with VALU right away, and the default
// heuristic parked those accumulators in AGPRs (32 v_accvgpr_read per 12 MFMAs in the backward's inner loop); the same flag
// makes the contrastive-head kernels of the other unit spill.
#include "xc_device.h"

#include "api_common.h"
#include "kernels/attention.h"
#include "kernels/attention2.h"
#include "kernels/attention3.h"
#include "kernels/attention4.h"
#include "kernels/attention5.h"
#include "kernels/attention6.h"
#include "kernels/attention7.h"
#include "kernels/attention_pool.h"

using namespace xc;
using namespace xcapi;

namespace {

template <typename T, int NW, int NH = 1>
void launch_attn_fwd(AttnParams p, hipStream_t st) {
    p.chunks = (p.n + NW * 32 - 1) / (NW * 32);
    constexpr int lds = attn_fwd_lds_bytes<T, NW, NH>();
    XC_ALLOW_LDS((attn_fwd_kernel<T, NW, NH>), lds);
    hipLaunchKernelGGL((attn_fwd_kernel<T, NW, NH>), dim3(p.batch * p.heads * p.chunks), dim3(NW * 64), lds, st, p);
}
template <typename T, int NW, int NH = 1>
void launch_attn_bwd(AttnParams p, hipStream_t st) {
    p.chunks = (p.n + NW * 32 - 1) / (NW * 32);
    constexpr int lds_q = attn_dq_lds_bytes<T, NW, NH>(), lds_kv = attn_dkv_lds_bytes<T, NW, NH>();
    XC_ALLOW_LDS((attn_dq_kernel<T, NW, NH>), lds_q);
    XC_ALLOW_LDS((attn_dkv_kernel<T, NW, NH>), lds_kv);
    hipLaunchKernelGGL((attn_dq_kernel<T, NW, NH>), dim3(p.batch * p.heads * p.chunks), dim3(NW * 64), lds_q, st, p);
    hipLaunchKernelGGL((attn_dkv_kernel<T, NW, NH>), dim3(p.batch * p.heads * p.chunks), dim3(NW * 64), lds_kv, st, p);
}
template <int NW>
void launch_attn2_fwd(AttnParams p, hipStream_t st) {
    p.chunks = (p.n + NW * 32 - 1) / (NW * 32);
    constexpr int lds = attn2_lds_bytes<NW>();
    hipLaunchKernelGGL((attn2_fwd_kernel<NW>), dim3(p.batch * p.heads * p.chunks), dim3(NW * 64), lds, st, p);
}
template <int NW>
void launch_attn2_bwd(AttnParams p, hipStream_t st) {
    p.chunks = (p.n + NW * 32 - 1) / (NW * 32);
    constexpr int lds = attn2_lds_bytes<NW>();
    hipLaunchKernelGGL((attn2_dq_kernel<NW>), dim3(p.batch * p.heads * p.chunks), dim3(NW * 64), lds, st, p);
    hipLaunchKernelGGL((attn2_dkv_kernel<NW>), dim3(p.batch * p.heads * p.chunks), dim3(NW * 64), lds, st, p);
}
// waves per work-group: the NW in 1..4 that wastes the fewest padded rows (ties -> larger NW)
int attn_waves(int64_t n) {
    int best = 1;
    int64_t best_pad = -1;
    for (int nw = 1; nw <= 4; ++nw) {
        const int64_t rows = ((n + nw * 32 - 1) / (nw * 32)) * nw * 32;
        if (best_pad < 0 || rows <= best_pad) { best = nw; best_pad = rows; }
    }
    return best;
}

// ---- one query row per (sample, head): the last layer of a tower whose caller reads a single token row (kernels/attention_pool.h) ----
template <typename T>
int launch_attn_pool(const AttnPoolParams& p, int64_t head_dim, bool backward, hipStream_t st) {
    const dim3 grid((unsigned)(((int64_t)p.batch * p.heads + 3) / 4)), block(256);
    if (head_dim == 64) {
        if (backward) hipLaunchKernelGGL((attn_pool_bwd_kernel<T, 64>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((attn_pool_fwd_kernel<T, 64>), grid, block, 0, st, p);
    } else {
        if (backward) hipLaunchKernelGGL((attn_pool_bwd_kernel<T, 128>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((attn_pool_fwd_kernel<T, 128>), grid, block, 0, st, p);
    }
    return 0;
}
}  // namespace

extern "C" {

int xclip_attention_fwd(const void* qkv, const uint8_t* mask, void* out, float* lse, int64_t batch, int64_t n, int64_t heads,
                        int64_t head_dim, float scale, int causal, float dropout_p, uint64_t dropout_seed, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(batch >= 0 && n > 0 && heads > 0, "bad shape");
    XC_REQUIRE(head_dim == 64 || head_dim == 128, "head_dim must be 64 or 128 (narrower / in-between widths are zero-padded by the caller)");
    XC_REQUIRE(aligned16(qkv) && aligned16(out), "pointers must be 16-byte aligned");
    if (batch == 0) return 0;
    AttnParams p;
    memset(&p, 0, sizeof(p));
    p.qkv = qkv; p.mask = mask; p.out = out; p.lse = lse;
    p.batch = (int)batch; p.n = (int)n; p.heads = (int)heads; p.scale = scale; p.causal = causal != 0;
    p.first_round = 2 * xc_num_cus();
    XC_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p must lie in [0, 1)");
    p.drop_thresh = drop_thresh(dropout_p); p.drop_scale = 1.0f / (1.0f - dropout_p); p.drop_seed = dropout_seed;
    const bool tiled_only = p.drop_thresh != 0;               // dropout lives in the tiled kernels (attention.h)
    hipStream_t st = (hipStream_t)stream;
    if (head_dim == 128 && dtype == XCLIP_BF16 && n <= A3_MAX_N && !tiled_only) {   // wide heads, head-resident (attention4.h)
        XC_REQUIRE(scale > 0.f, "the head-resident kernels take the score maximum before scaling: scale must be positive");
        const int nwq = a4_fwd_waves((int)n);
        XC_REQUIRE(attn4_fwd_lds_bytes((int)n) <= 160 * 1024, "internal: wide-head forward LDS");
        if (causal) {
            XC_ALLOW_LDS(attn4_fwd_kernel<true>, 160 * 1024);
            hipLaunchKernelGGL(attn4_fwd_kernel<true>, dim3((unsigned)(batch * heads)), dim3(nwq * 64), attn4_fwd_lds_bytes((int)n), st, p);
        } else {
            XC_ALLOW_LDS(attn4_fwd_kernel<false>, 160 * 1024);
            hipLaunchKernelGGL(attn4_fwd_kernel<false>, dim3((unsigned)(batch * heads)), dim3(nwq * 64), attn4_fwd_lds_bytes((int)n), st, p);
        }
        return check_launch(__func__);
    }
    if (head_dim == 128) {                                     // wide heads, long sequences / fp32 / dropout: the tiled kernels with two 64-wide halves per head
        const int nw = attn_waves(n);
#define W(T) switch (nw) { case 1: launch_attn_fwd<T, 1, 2>(p, st); break; case 2: launch_attn_fwd<T, 2, 2>(p, st); break; \
                           case 3: launch_attn_fwd<T, 3, 2>(p, st); break; default: launch_attn_fwd<T, 4, 2>(p, st); break; }
        if (dtype == XCLIP_BF16) { W(bf16_t) } else { W(float) }
#undef W
        return check_launch(__func__);
    }
    if (dtype == XCLIP_BF16 && n <= A3_MAX_N && !tiled_only) {               // head-resident kernel: one work-group per (batch, head)
        XC_REQUIRE(scale > 0.f, "the head-resident kernels take the score maximum before scaling: scale must be positive");
        const int nwq = a3_waves((int)n);
        // half a head's time (13 us at n = 257, growing with n^2) between the two work-groups of a CU: 0.413 -> 0.376 ms at
        // b = 1024, n = 257 (profiles/r02_run20_attention_stagger_sweep.log); XCLIP_ATTN_STAGGER_FWD=<10 ns ticks> overrides
        static const int stag = measure_env("XCLIP_ATTN_STAGGER_FWD", -1);
        const int half_head = (int)(1300.0 * (double)n * (double)n / (257.0 * 257.0));
        p.stagger_10ns = (attn3_fwd_lds_bytes((int)n) <= 80 * 1024 && batch * heads >= 1024) ? (stag >= 0 ? stag : half_head) : 0;
        static const int abl_f = measure_env("XCLIP_ATTN_ABL", 0);  // measurement build only: 4 = the tail row as a 33rd block (round-3 form)
        p.chunks = abl_f;
        if (causal) {
            XC_ALLOW_LDS(attn3_fwd_kernel<true>, 160 * 1024);
            hipLaunchKernelGGL(attn3_fwd_kernel<true>, dim3((unsigned)(batch * heads)), dim3(nwq * 64), attn3_fwd_lds_bytes((int)n), st, p);
        } else {
            XC_ALLOW_LDS(attn3_fwd_kernel<false>, 160 * 1024);
            hipLaunchKernelGGL(attn3_fwd_kernel<false>, dim3((unsigned)(batch * heads)), dim3(nwq * 64), attn3_fwd_lds_bytes((int)n), st, p);
        }
        return check_launch(__func__);
    }
    const int nw = attn_waves(n);
#define F(T) switch (nw) { case 1: launch_attn_fwd<T, 1>(p, st); break; case 2: launch_attn_fwd<T, 2>(p, st); break; \
                           case 3: launch_attn_fwd<T, 3>(p, st); break; default: launch_attn_fwd<T, 4>(p, st); break; }
    if (dtype == XCLIP_BF16 && tiled_only) { F(bf16_t) }
    else if (dtype == XCLIP_BF16) {
        switch (nw) { case 1: launch_attn2_fwd<1>(p, st); break; case 2: launch_attn2_fwd<2>(p, st); break;
                      case 3: launch_attn2_fwd<3>(p, st); break; default: launch_attn2_fwd<4>(p, st); break; }
    } else { F(float) }
#undef F
    return check_launch(__func__);
}

int xclip_attention_bwd(const void* qkv, const uint8_t* mask, const void* out, const void* dout, const float* lse,
                        float* delta_ws, void* dqkv, int64_t batch, int64_t n, int64_t heads, int64_t head_dim, float scale, int causal,
                        float dropout_p, uint64_t dropout_seed, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(batch >= 0 && n > 0 && heads > 0, "bad shape");
    XC_REQUIRE(head_dim == 64 || head_dim == 128, "head_dim must be 64 or 128 (narrower / in-between widths are zero-padded by the caller)");
    XC_REQUIRE(aligned16(qkv) && aligned16(out) && aligned16(dout) && aligned16(dqkv), "pointers must be 16-byte aligned");
    if (batch == 0) return 0;
    AttnParams p;
    memset(&p, 0, sizeof(p));
    p.qkv = qkv; p.mask = mask; p.out = const_cast<void*>(out); p.lse = const_cast<float*>(lse); p.dout = dout;
    p.delta = delta_ws; p.dqkv = dqkv;
    p.batch = (int)batch; p.n = (int)n; p.heads = (int)heads; p.scale = scale; p.causal = causal != 0;
    p.first_round = 2 * xc_num_cus();
    XC_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p must lie in [0, 1)");
    p.drop_thresh = drop_thresh(dropout_p); p.drop_scale = 1.0f / (1.0f - dropout_p); p.drop_seed = dropout_seed;
    const bool tiled_only = p.drop_thresh != 0;               // dropout lives in the tiled kernels (attention.h)
    hipStream_t st = (hipStream_t)stream;
    if (head_dim == 128 && dtype == XCLIP_BF16 && n <= A3_MAX_N && !tiled_only) {   // wide heads, head-resident (attention4.h; computes delta itself)
        XC_REQUIRE(scale > 0.f, "the head-resident kernels take the score maximum before scaling: scale must be positive");
        const int nwq = a3_bwd_waves((int)n);
        if (causal) {
            XC_ALLOW_LDS(attn4_bwd_kernel<true>, 160 * 1024);
            hipLaunchKernelGGL(attn4_bwd_kernel<true>, dim3((unsigned)(batch * heads)), dim3(nwq * 64), attn4_bwd_lds_bytes((int)n), st, p);
        } else {
            XC_ALLOW_LDS(attn4_bwd_kernel<false>, 160 * 1024);
            hipLaunchKernelGGL(attn4_bwd_kernel<false>, dim3((unsigned)(batch * heads)), dim3(nwq * 64), attn4_bwd_lds_bytes((int)n), st, p);
        }
        return check_launch(__func__);
    }
    if (head_dim == 128) {                                     // wide heads, long sequences / fp32 / dropout: delta pass + the tiled dQ / dK, dV kernels on two halves
        XC_REQUIRE(delta_ws != nullptr, "wide heads need the [batch, heads, n] fp32 delta workspace");
        dim3 wgrid((unsigned)((batch * n + 3) / 4)), wblock(256);
        if (dtype == XCLIP_BF16)
            hipLaunchKernelGGL((attn_delta_kernel<bf16_t, 2>), wgrid, wblock, 0, st, (const bf16_t*)out, (const bf16_t*)dout, delta_ws, (int)batch, (int)n, (int)heads);
        else
            hipLaunchKernelGGL((attn_delta_kernel<float, 2>), wgrid, wblock, 0, st, (const float*)out, (const float*)dout, delta_ws, (int)batch, (int)n, (int)heads);
        const int nw = attn_waves(n);
#define W(T) switch (nw) { case 1: launch_attn_bwd<T, 1, 2>(p, st); break; case 2: launch_attn_bwd<T, 2, 2>(p, st); break; \
                           case 3: launch_attn_bwd<T, 3, 2>(p, st); break; default: launch_attn_bwd<T, 4, 2>(p, st); break; }
        if (dtype == XCLIP_BF16) { W(bf16_t) } else { W(float) }
#undef W
        return check_launch(__func__);
    }
    if (dtype == XCLIP_BF16 && n <= A3_MAX_N && !tiled_only) {               // merged head-resident backward (computes delta itself)
        XC_REQUIRE(scale > 0.f, "the head-resident kernels take the score maximum before scaling: scale must be positive");
        // the single-pass form (attention5.h) for the sequences it takes; XCLIP_ATTN_BWD=3 (measurement build) keeps the two-phase kernel for the A/B
        // (XCLIP_ATTN5_MIN=2: every sequence it can take, for the record of where it loses)
        // round 6: the streaming persistent form (attention6.h) for n = 256 / 257, XCLIP_ATTN_BWD=6 on the measurement build: correct on the
        // hardware and SLOWER than the single pass -- 975 against 813 us at n = 256, 1305 against 937 at n = 257 (b = 1024, 8 heads;
        // profiles/r06_f_attn_bwd_ab.log, ablations profiles/r06_g_attn6_abl.log) -- the product keeps attention5.h
        static const int bwd_gen = measure_env("XCLIP_ATTN_BWD", 5), min5 = measure_env("XCLIP_ATTN5_MIN", A5_MIN_BLOCKS);
        if (bwd_gen == 6 && a6_takes((int)n, causal)) {
            XC_ALLOW_LDS(attn6_bwd_kernel, 160 * 1024);
            const int64_t heads_total = batch * heads;
            const int cus = xc_num_cus();
            int64_t g6 = heads_total < cus ? heads_total : cus;
            static const int abl6 = measure_env("XCLIP_ATTN6_ABL", 0);    // measurement build only: attention6.h's ablation mask
            p.chunks = abl6;
            if (g6 >= 8) g6 &= ~(int64_t)7;                     // whole XCD rounds: a work-group's heads stay on one XCD's slice of the order
            hipLaunchKernelGGL(attn6_bwd_kernel, dim3((unsigned)g6), dim3(512), A6_LDS_BYTES, st, p);
            return check_launch(__func__);
        }
        if (bwd_gen == 7 && a5_takes((int)n, causal) && (n >> 5) >= min5 && attn5_bwd_lds_bytes((int)n) <= 160 * 1024) {
            // attention7.h (round 6, measurement build): attention5.h persistent, the next head's images requested (asm-issued DMA) under this
            // head's stores -- bit-identical, and SLOWER: 1018 against 942 us at n = 257, 846 against 808 at n = 256 (profiles/r06_n_attn7_ab.log);
            // requesting the images at the top of the head instead changes nothing (r06_o_attn7_abl.log, mask 2)
            XC_ALLOW_LDS(attn7_bwd_kernel, 160 * 1024);
            const int64_t heads_total = batch * heads;
            const int cus = xc_num_cus();
            int64_t g7 = heads_total < cus ? heads_total : cus;
            if (g7 >= 8) g7 &= ~(int64_t)7;
            static const int abl7 = measure_env("XCLIP_ATTN7_ABL", 0);    // measurement build only: attention7.h's ablation mask
            p.chunks = abl7;
            hipLaunchKernelGGL(attn7_bwd_kernel, dim3((unsigned)g7), dim3((unsigned)((n >> 5) * 64)), attn5_bwd_lds_bytes((int)n), st, p);
            return check_launch(__func__);
        }
        if (bwd_gen >= 5 && a5_takes((int)n, causal) && (n >> 5) >= min5 && attn5_bwd_lds_bytes((int)n) <= 160 * 1024) {
            XC_ALLOW_LDS(attn5_bwd_kernel, 160 * 1024);
            static const int var5 = measure_env("XCLIP_ATTN5_VAR", 0);     // measurement build only: attention5.h's variant mask
            p.chunks = var5;
            const int64_t g5 = batch * heads;
            hipLaunchKernelGGL(attn5_bwd_kernel, dim3((unsigned)g5), dim3((unsigned)((n >> 5) * 64)), attn5_bwd_lds_bytes((int)n), st, p);
            return check_launch(__func__);
        }
        const int nwq = a3_bwd_waves((int)n);
        static const int abl = measure_env("XCLIP_ATTN_ABL", 0);   // measurement build only
        p.chunks = abl;
        // (measured: no effect on the backward -- its two work-groups are latency-bound chains that already overlap)
        static const int stag = measure_env("XCLIP_ATTN_STAGGER_BWD", 0);
        p.stagger_10ns = (attn3_bwd_lds_bytes((int)n) <= 80 * 1024 && batch * heads >= 1024) ? stag : 0;
        if (causal) {
            XC_ALLOW_LDS(attn3_bwd_kernel<true>, 160 * 1024);
            hipLaunchKernelGGL(attn3_bwd_kernel<true>, dim3((unsigned)(batch * heads)), dim3(nwq * 64), attn3_bwd_lds_bytes((int)n), st, p);
        } else {
            XC_ALLOW_LDS(attn3_bwd_kernel<false>, 160 * 1024);
            hipLaunchKernelGGL(attn3_bwd_kernel<false>, dim3((unsigned)(batch * heads)), dim3(nwq * 64), attn3_bwd_lds_bytes((int)n), st, p);
        }
        return check_launch(__func__);
    }
    dim3 dgrid((unsigned)((batch * n + 3) / 4)), dblock(256);
    if (dtype == XCLIP_BF16)
        hipLaunchKernelGGL((attn_delta_kernel<bf16_t>), dgrid, dblock, 0, st, (const bf16_t*)out, (const bf16_t*)dout, delta_ws, (int)batch, (int)n, (int)heads);
    else
        hipLaunchKernelGGL((attn_delta_kernel<float>), dgrid, dblock, 0, st, (const float*)out, (const float*)dout, delta_ws, (int)batch, (int)n, (int)heads);
    const int nw = attn_waves(n);
#define F(T) switch (nw) { case 1: launch_attn_bwd<T, 1>(p, st); break; case 2: launch_attn_bwd<T, 2>(p, st); break; \
                           case 3: launch_attn_bwd<T, 3>(p, st); break; default: launch_attn_bwd<T, 4>(p, st); break; }
    if (dtype == XCLIP_BF16 && tiled_only) { F(bf16_t) }
    else if (dtype == XCLIP_BF16) {
        switch (nw) { case 1: launch_attn2_bwd<1>(p, st); break; case 2: launch_attn2_bwd<2>(p, st); break;
                      case 3: launch_attn2_bwd<3>(p, st); break; default: launch_attn2_bwd<4>(p, st); break; }
    } else { F(float) }
#undef F
    return check_launch(__func__);
}

int xclip_attention_pool_fwd(const void* q, const void* kv, const uint8_t* mask, void* out, float* lse, int64_t batch, int64_t n,
                             int64_t heads, int64_t head_dim, float scale, int64_t visible_keys, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(batch >= 0 && n > 0 && heads > 0 && visible_keys >= 1 && visible_keys <= n, "bad shape");
    XC_REQUIRE(head_dim == 64 || head_dim == 128, "head_dim must be 64 or 128 (narrower / in-between widths are zero-padded by the caller)");
    XC_REQUIRE(q && kv && out && lse && aligned16(q) && aligned16(kv) && aligned16(out), "pointers must be non-null and 16-byte aligned");
    if (batch == 0) return 0;
    AttnPoolParams p;
    memset(&p, 0, sizeof(p));
    p.q = q; p.kv = kv; p.mask = mask; p.out = out; p.lse = lse;
    p.batch = (int)batch; p.n = (int)n; p.heads = (int)heads; p.nvis = (int)visible_keys; p.scale = scale;
    if (dtype == XCLIP_BF16) launch_attn_pool<bf16_t>(p, head_dim, false, (hipStream_t)stream);
    else launch_attn_pool<float>(p, head_dim, false, (hipStream_t)stream);
    return check_launch(__func__);
}

int xclip_attention_pool_bwd(const void* q, const void* kv, const uint8_t* mask, const void* out, const void* dout, const float* lse,
                             void* dq, void* dkv, int64_t batch, int64_t n, int64_t heads, int64_t head_dim, float scale,
                             int64_t visible_keys, int dtype, void* stream) {
    XC_REQUIRE(dtype_ok(dtype), "bad dtype");
    XC_REQUIRE(batch >= 0 && n > 0 && heads > 0 && visible_keys >= 1 && visible_keys <= n, "bad shape");
    XC_REQUIRE(head_dim == 64 || head_dim == 128, "head_dim must be 64 or 128 (narrower / in-between widths are zero-padded by the caller)");
    XC_REQUIRE(q && kv && out && dout && lse && dq && dkv, "null pointer");
    XC_REQUIRE(aligned16(q) && aligned16(kv) && aligned16(out) && aligned16(dout) && aligned16(dq) && aligned16(dkv), "pointers must be 16-byte aligned");
    if (batch == 0) return 0;
    AttnPoolParams p;
    memset(&p, 0, sizeof(p));
    p.q = q; p.kv = kv; p.mask = mask; p.out = const_cast<void*>(out); p.lse = const_cast<float*>(lse); p.dout = dout; p.dq = dq; p.dkv = dkv;
    p.batch = (int)batch; p.n = (int)n; p.heads = (int)heads; p.nvis = (int)visible_keys; p.scale = scale;
    if (dtype == XCLIP_BF16) launch_attn_pool<bf16_t>(p, head_dim, true, (hipStream_t)stream);
    else launch_attn_pool<float>(p, head_dim, true, (hipStream_t)stream);
    return check_launch(__func__);
}

}  // extern "C"
