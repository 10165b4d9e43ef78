/* xclip.h -- C ABI of libxclip_hip.so, the MI355X (gfx950) kernel library behind x_clip_amd.
 *
 * The reference (lucidrains/x-clip) is pure Python/PyTorch and has no FFI of its own; every entry point below
 * replaces a sequence of ATen calls issued by the reference lines cited next to it.
 * Conventions:
 *   - plain pointers + sizes, no torch types.  All tensor pointers are DEVICE pointers, row-major, 16-byte
 *     aligned, contiguous dims a multiple of the 16-byte chunk (8 bf16 / 4 fp32) unless stated otherwise.
 *   - the caller owns every buffer (inputs, outputs, workspaces, accumulators); the library never allocates.
 *   - `dtype`: XCLIP_F32 or XCLIP_BF16 = storage type of activations / parameters; arithmetic is fp32.
 *   - `stream` is a hipStream_t (NULL = default stream).  Calls are asynchronous on it and never synchronise.
 *   - return 0 on success; otherwise a non-zero code and xclip_last_error() describes it (thread local).
 *   - "accum" outputs are fp32 accumulators the kernel ADDS into (caller zeroes them).
 */
#ifndef XCLIP_H
#define XCLIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { XCLIP_F32 = 0, XCLIP_BF16 = 1 };
#define XCLIP_ABI_VERSION 23

int xclip_abi_version(void);
const char* xclip_last_error(void);
/* which toolchain built this library, e.g. "hipcc HIP version: 7.2.26015-fc0010cf6a".  The hand-placed wait states around the asm
 * buffer stores (csrc/hw/xc_device.h buf_st16) were validated with ROCm 7.2's code generation: after a toolchain change re-run the GPU
 * gate tests (tests/test_kernels_gpu.py::test_gemm_full_size_every_element_and_repeatable, test_gemm_layouts, test_gemm_residual_epilogue). */
const char* xclip_build_info(void);
/* Diagnostics, not on the training path (no reference counterpart): one wave counts shader cycles against the constant 100 MHz counter for
 * `ticks_10ns` ticks and writes out2[0] = shader cycles, out2[1] = ticks elapsed (device memory, two uint64).  Launched between the
 * kernels of a step it reads the clock the part sustains under that load; bench.py puts it into its JSON line beside the step times. */
int xclip_clock_sample(uint64_t* out2, int64_t ticks_10ns, void* stream);

/* ---- LayerNorm family (reference LayerNorm x_clip.py:112-121; GEGLU x_clip.py:180-183) -----------------------
 * y[r,:] = (v - mean) * rstd * g (+ res[r,:]),  v = x[r,:dim]               (geglu = 0, ldx >= dim)
 *                                               v = x[r,:dim] * gelu(x[r,dim:2dim])   (geglu = 1, ldx >= 2 dim)
 * mean/rstd [rows] fp32 are saved for the backward.  eps: 1e-5 for fp32 models, 1e-3 otherwise (x_clip.py:118). */
int xclip_layernorm_fwd(const void* x, int64_t ldx, const void* g, const void* res, void* y, int64_t ldy, int64_t y_grp,
                        float* mean, float* rstd, int64_t rows, int64_t dim, float eps, int geglu, int dtype, void* stream);
/*   y row r lives at y + (r + (y_grp ? r / y_grp + 1 : 0)) * ldy: y_grp = n leaves the CLS slot of a [b, 1+n, dim]
 *   encoder output free (VisionTransformer.forward x_clip.py:389-390); res [rows, dim] is contiguous.
 * bwd: dx [rows, dim] (lddx) -- or [rows, 2 dim] = (d value | d gate) with geglu; dg_accum [dim] fp32 += dy * xhat;
 *   dres [rows, dim] (optional, not with geglu) is added to dx: the skip-path gradient of x + f(LN(x)) (x_clip.py:288-289);
 *   workspace: xclip_layernorm_bwd_workspace_bytes(rows, dim) bytes of scratch (per-work-group gain-gradient partials). */
int64_t xclip_layernorm_bwd_workspace_bytes(int64_t rows, int64_t dim);
int xclip_layernorm_bwd(const void* dy, const void* x, int64_t ldx, const void* g, const float* mean, const float* rstd,
                        const void* dres, void* dx, int64_t lddx, float* dg_accum, void* workspace, int64_t workspace_bytes,
                        int64_t rows, int64_t dim, int geglu, int dtype, void* stream);

/* The residual-block boundary of the Transformer as ONE pass over the rows (x_clip.py:245,288-289 followed by :126):
 *   x1 = LayerNorm(p) g1 + res   (to_out's LayerNorm + skip),   h2 = LayerNorm(x1) g2   (the feed-forward PreNorm),
 * contiguous [rows, dim] tensors, both pairs of statistics out.  The second LayerNorm sees x1 as stored (rounded to the dtype), so
 * the results equal two xclip_layernorm_fwd calls; the pass saves re-reading x1.  chain_bwd: dx1 = LN2'(dh2) + dres and
 * dp = LN1'(dx1) in one pass (dx1 is still written: the next residual junction adds it), gain gradients into dg2_accum / dg1_accum
 * (fp32 [dim] each) through per-work-group partial rows in `workspace`. */
int xclip_layernorm_chain_fwd(const void* p, const void* g1, const void* res, void* x1, float* mean1, float* rstd1, const void* g2,
                              void* h2, float* mean2, float* rstd2, int64_t rows, int64_t dim, float eps, int dtype, void* stream);
int64_t xclip_layernorm_chain_bwd_workspace_bytes(int64_t rows, int64_t dim);
int xclip_layernorm_chain_bwd(const void* dh2, const void* x1, const void* g2, const float* mean2, const float* rstd2, const void* dres,
                              void* dx1, const void* p, const void* g1, const float* mean1, const float* rstd1, void* dp, float* dg2_accum,
                              float* dg1_accum, void* workspace, int64_t workspace_bytes, int64_t rows, int64_t dim, int dtype, void* stream);
/* ---- l2 normalisation (reference l2norm = F.normalize, x_clip.py:54-55,715) ------------------------------------ */
int xclip_l2norm_fwd(const void* x, void* y, float* rnorm, int64_t rows, int64_t dim, int dtype, void* stream);
int xclip_l2norm_bwd(const void* dy, const void* y, const float* rnorm, void* dx, int64_t rows, int64_t dim, int dtype, void* stream);

/* ---- text embedding (reference TextTransformer.forward x_clip.py:320-335) ---------------------------------------
 * out[b,0] = cls ; out[b,1+j] = E[tok[b,j]] + P[j].  cls / P may be NULL (no CLS row / no absolute positions).
 * E has `vocab` rows.  A token id outside [0, vocab) -- where the reference's nn.Embedding raises IndexError (CPU) or trips a
 * device assert (x_clip.py:320) -- reads and writes nothing out of bounds: its output row is NaN and *bad_token_flag (device
 * int32, may be NULL) is set to 1 for the host to report; the backward entry points skip such ids. */
int xclip_text_embed_fwd(const int64_t* tokens, const void* E, const void* P, const void* cls, void* out,
                         int64_t batch, int64_t n, int64_t dim, int64_t vocab, int32_t* bad_token_flag, int dtype, void* stream);
/* fp32 accumulators: dE [vocab, dim], dP [n, dim] (may be NULL), dcls [dim] (NULL when has_cls = 0) */
int xclip_text_embed_bwd(const void* dout, const int64_t* tokens, float* dE_accum, float* dP_accum, float* dcls_accum,
                         int64_t batch, int64_t n, int64_t dim, int64_t vocab, int has_cls, int dtype, void* stream);

/* ---- patchify 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' (x_clip.py:357) with the PatchDropout keep-set
 * (x_clip.py:140-151) folded in: out[(b,i), :] = patch keep[b*nkeep + i] of image b (keep NULL: identity, nkeep =
 * number of patches).  Rows are zero padded from p*p*C up to ldo. */
int xclip_patchify(const void* image, const int32_t* keep, void* out, int64_t ldo, int64_t batch, int64_t channels,
                   int64_t height, int64_t width, int64_t patch, int64_t nkeep, int dtype, void* stream);

/* ---- vision CLS pooling: mean over tokens (x_clip.py:366-370) --------------------------------------------------
 * fwd: out[b] = mean_t x[b, t]   with x[b, t] at x + b * x_batch_stride + t * dim (elements).
 * bwd: dx[b, t] = dout[b] / n + (dsrc ? dsrc[b, t] : 0), dsrc rows at dsrc + b * src_batch_stride + t * dim; dx contiguous. */
int xclip_token_mean_fwd(const void* x, int64_t x_batch_stride, void* out, int64_t batch, int64_t n, int64_t dim, int dtype,
                         void* stream);
int xclip_token_mean_bwd(const void* dout, const void* dsrc, int64_t src_batch_stride, void* dx, int64_t batch, int64_t n,
                         int64_t dim, int dtype, void* stream);

/* dst[r, :dim] = src[r, :dim], rows at src + r * lds / dst + r * ldd (CLS select / scatter: enc[:, 0], x_clip.py:708-709) */
int xclip_copy_rows(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t dim, int dtype, void* stream);

/* out[i] = a[i] + b[i] (fp32 arithmetic, one rounding): sums the latent gradients of a view that takes part in several
 * multiview pairs (x_clip.py:750-755,851-868) */
int xclip_add(const void* a, const void* b, void* out, int64_t count, int dtype, void* stream);

/* table_accum[idx[r], :] += src[r, :] (fp32; NULL: skipped) and colsum_accum[:] += sum_r src[r, :] (fp32; NULL: skipped):
 * the gradients of the position table gathered by the kept-patch index and of the patch-embedding bias
 * (x_clip.py:358,382-385).  workspace (optional, xclip_rows_scatter_add_workspace_bytes): per-wave partial rows for the column sum. */
int64_t xclip_rows_scatter_add_workspace_bytes(int64_t rows, int64_t dim);
int xclip_rows_scatter_add(const void* src, int64_t lds, const int32_t* idx, float* table_accum, float* colsum_accum,
                           int64_t rows, int64_t dim, void* workspace, int64_t workspace_bytes, int dtype, void* stream);

/* table_accum[sorted_ids[e], :] += src[row(perm[e]), :], e in [0, count): `sorted_ids` ascending (int64), perm[e] = the flat
 * index entry e had before sorting, row(p) = (p / n_in) * n_out + p % n_in + row_off.  Equal ids are summed in registers and
 * flushed once per run: the token-embedding gradient (nn.Embedding backward of x_clip.py:320) with n_in = n, n_out = n + cls,
 * row_off = cls; the position-table gradient of the kept patches (x_clip.py:382-385) with n_in = n_out = 1, row_off = 0.
 * xclip_text_embed_bwd may be called with dE_accum = NULL when the embedding gradient is produced this way.  The table has
 * `table_rows` rows; entries whose id lies outside [0, table_rows) are dropped. */
int xclip_scatter_add_sorted(const void* src, int64_t lds, const int64_t* sorted_ids, const int64_t* perm, float* table_accum,
                             int64_t table_rows, int64_t count, int64_t dim, int64_t n_in, int64_t n_out, int64_t row_off, int dtype,
                             void* stream);

/* The sort in front of xclip_scatter_add_sorted: sorted_ids = ids ascending, perm[e] = the index entry e had in `ids`; STABLE (equal ids keep
 * their order: the segment sums of the scatter then add their rows in one fixed order).  Replaces the torch.sort(tokens.flatten()) the
 * nn.Embedding backward of x_clip.py:320 amounts to.  ids in [0, id_limit), id_limit <= 2^32, count < 2^32 (an id outside that range keeps a
 * deterministic place decided by its low bits; the scatter drops it).  Least-significant-digit radix passes of 8 bits over (id, position)
 * pairs: ceil(log2(id_limit) / 8) passes of three launches.  workspace: xclip_sort_ids_workspace_bytes(count) bytes, 16-byte aligned. */
int64_t xclip_sort_ids_workspace_bytes(int64_t count);
int xclip_sort_ids(const int64_t* ids, int64_t count, int64_t id_limit, int64_t* sorted_ids, int64_t* perm, void* workspace,
                   int64_t workspace_bytes, void* stream);

/* dst[i] = (dtype) (src[i] * scale) : fp32 gradient accumulators -> parameter dtype */
int xclip_cast_from_f32(const float* src, void* dst, int64_t count, float scale, int dtype, void* stream);

/* ---- GEMM: every nn.Linear on the path and its dgrad / wgrad (x_clip.py:191-195,209-210,358,368,556,570) --------
 * C[M,N] = alpha * op(A) op(B) + bias[n] + addrows[rowidx[m], n] + residual[m, n]      (each term optional / NULL)
 *   a_kmajor = 0: A[m*lda + k]   1: A[k*lda + m]        b_kmajor = 0: B[n*ldb + k] (Linear weight)   1: B[k*ldb + n]
 * forward (0,0), dgrad (0,1), wgrad (1,1).  `workspace` (fp32 split-K slabs, may be NULL) of
 * xclip_gemm_workspace_bytes(...) bytes lets long contractions (wgrad) fill the chip. */
int64_t xclip_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int dtype);
int xclip_gemm(int a_kmajor, int b_kmajor, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
               int64_t M, int64_t N, int64_t K, float alpha, const void* bias, const void* residual, int64_t ldr,
               const void* addrows, const int32_t* rowidx, int64_t ld_add, void* workspace, int64_t workspace_bytes,
               int dtype, void* stream);
/* bf16 products with M, N, K multiples of 64, no bias / gathered rows, an output of at most 64 tiles of 256 x 256 and at most `max_flop`
 * (2 M N K; default 6e9) go to a 64 x 64-tile kernel built for latency (the [batch, d] rows of a pooled last layer and of the latent
 * projections, x_clip.py:713-715): one launch, no split-K.  Sets the bound (0 = never; < 0 = only ask) and returns the previous one.
 * Process-wide; results do not depend on it beyond the summation order of the contraction. */
int64_t xclip_gemm_small_limit(int64_t max_flop);

/* ---- the inference returns (reference CLIP.forward with return_loss = False, x_clip.py:740-746) --------------------------------
 * xclip_gemm_batched: `batch` independent products C_z[M,N] = alpha * op(A_z) op(B_z) with xclip_gemm's operand layouts, problem z at
 *   A + z*stride_a (elements) etc.: einsum('b t d, b i d -> b t i') of the fine-grained (FILIP) similarities (0, 0) and its two gradients
 *   (0, 1) / (1, 1).  No optional terms.
 * xclip_rowdot: out[r] = <a[r,:], b[r,:]> in the operands' dtype: einsum('b d, b d -> b') of the matched pairs. */
int xclip_gemm_batched(int a_kmajor, int b_kmajor, const void* A, int64_t lda, int64_t stride_a, const void* B, int64_t ldb, int64_t stride_b,
                       void* C, int64_t ldc, int64_t stride_c, int64_t batch, int64_t M, int64_t N, int64_t K, float alpha, int dtype,
                       void* stream);
int xclip_rowdot(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t rows, int64_t dim, int dtype, void* stream);

/* ---- the feed-forward block's backward through net.4 and net.2 in one product (reference FeedForward, x_clip.py:180-199) ----------------
 * d(u | t) [M, 2F] = the GEGLU-LayerNorm backward of dh = dout [M, D] w2 [D, F], with dh never written: the two row statistics of the
 * LayerNorm backward follow from dout, the block's input x1 and output x2 (= x1 + h w2^T) and the vector w2 gamma (csrc/kernels/gemm9.h).
 * x [M, 2F] = FF1's output (value | gate), mean / rstd = the forward LayerNorm's statistics, dg_accum [F] fp32 += the gain's gradient.
 * bf16, M and F multiples of 256, D a multiple of 64 (xclip_ffn_dgrad_geglu_ok); the caller runs xclip_gemm + xclip_layernorm_bwd otherwise. */
int xclip_ffn_dgrad_geglu_ok(int64_t M, int64_t F, int64_t D, int dtype);
int64_t xclip_ffn_dgrad_geglu_workspace_bytes(int64_t M, int64_t F, int64_t D);
int xclip_ffn_dgrad_geglu(const void* dout, int64_t ldd, const void* w2, int64_t ldw, const void* x, int64_t ldx, const void* gamma,
                          const float* mean, const float* rstd, const void* x2, int64_t ld2, const void* x1, int64_t ld1, void* dx,
                          int64_t lddx, float* dg_accum, void* workspace, int64_t workspace_bytes, int64_t M, int64_t F, int64_t D,
                          int dtype, void* stream);
/* Round 6 (ABI 22): the row pass of that backward folded into the kernel that PRODUCES dout.  In a pre-norm stack (x_clip.py:285-289) the
 * gradient of a block's output is written by the LayerNorm backward of the block above it (or of norm_out), whose own input row IS the lower
 * block's output x2: xclip_layernorm_bwd_ffnstats = xclip_layernorm_bwd (no geglu) that also writes the lower block's four per-row constants
 * rowc [rows, 4] = {rstd4, -mean4 rstd4, (dx . wg) rstd4 / F, (dx . (x - x1_below)) rstd4 / F} from the row it holds (dx as stored), given
 * wg = xclip_ffn_wgamma(w2, gamma) [D] fp32, the lower block's input x1_below and its inner LayerNorm's statistics mean4 / rstd4, inv_f = 1 / F;
 * xclip_ffn_dgrad_geglu_rowc then runs the product with those constants (no x1 / x2 pass).  Same results as xclip_ffn_dgrad_geglu. */
int xclip_ffn_wgamma(const void* w2, int64_t ldw, const void* gamma, float* wg, int64_t D, int64_t F, int dtype, void* stream);
int xclip_layernorm_bwd_ffnstats(const void* dy, const void* x, int64_t ldx, const void* g, const float* mean, const float* rstd,
                                 const void* dres, void* dx, int64_t lddx, float* dg_accum, void* workspace, int64_t workspace_bytes,
                                 int64_t rows, int64_t dim, const void* x1_below, int64_t ld1, const float* wg, const float* mean4,
                                 const float* rstd4, float inv_f, float* rowc, int dtype, void* stream);
int xclip_ffn_dgrad_geglu_rowc(const void* dout, int64_t ldd, const void* w2, int64_t ldw, const void* x, int64_t ldx, const void* gamma,
                               const float* rowc, void* dx, int64_t lddx, float* dg_accum, void* workspace, int64_t workspace_bytes,
                               int64_t M, int64_t F, int64_t D, int dtype, void* stream);

/* ---- fused attention (reference Attention.forward x_clip.py:201-245; any dim_head up to 128) ----------------------
 * head_dim = the width of a head slot in memory: 64 (the reference default; the head-resident kernels) or 128 (wide heads: two
 * 64-wide halves per head through the tiled kernels).  A model head narrower than its slot is zero-padded by the caller.
 * qkv [batch, n, 3, heads, head_dim] = output of the to_qkv Linear; mask [batch, n] bytes (1 = attend) or NULL;
 * out [batch, n, heads*head_dim]; lse [batch, heads, n] fp32 saved for the backward.  scale = dim_head^-0.5 (the MODEL's dim_head).
 * causal != 0: key j is hidden from query i when j > i (the causal text encoder, x_clip.py:231-234), on top of the key mask.
 * dropout_p > 0: attention dropout on the softmax probabilities (Attention.dropout, x_clip.py:212,241): probability (b, h, i, j) is kept,
 * and scaled by 1 / (1 - p), iff the 32-bit mix of (dropout_seed, ((b heads + h) n + i) n + j) is >= p 2^32 (csrc/kernels/common.h
 * drop_hash) -- the backward (and a checkpointed re-run of the forward) must be given the same seed.  Runs the tiled kernels.
 * A query with no visible key gets output 0 (the reference's softmax over all -max scores gives the uniform average there). */
int xclip_attention_fwd(const void* qkv, const uint8_t* mask, void* out, float* lse, int64_t batch, int64_t n,
                        int64_t heads, int64_t head_dim, float scale, int causal, float dropout_p, uint64_t dropout_seed, int dtype,
                        void* stream);
/* delta_ws: [batch, heads, n] fp32 scratch; dqkv [batch, n, 3, heads, head_dim] fully overwritten */
int xclip_attention_bwd(const void* qkv, const uint8_t* mask, const void* out, const void* dout, const float* lse,
                        float* delta_ws, void* dqkv, int64_t batch, int64_t n, int64_t heads, int64_t head_dim, float scale, int causal,
                        float dropout_p, uint64_t dropout_seed, int dtype, void* stream);
/* The same attention for ONE query row per (sample, head) -- the last layer of a tower whose caller reads a single token row (the CLS head,
 * x_clip.py:708 `enc_text[:, 0]`; Attention.forward x_clip.py:213-245 restricted to that query):
 * q [batch, heads, head_dim] = to_qkv's first third applied to the pooled rows; kv [batch, n, 2, heads, head_dim] = its other two thirds
 * for every row; mask as above; keys [0, visible_keys) are visible (n, or the pooled row's index + 1 under a causal mask).
 * out [batch, heads, head_dim]; lse [batch, heads] fp32 (natural log of the scaled scores' sum), saved for the backward.
 * Backward: dq [batch, heads, head_dim] and dkv [batch, n, 2, heads, head_dim], fully overwritten (hidden keys: zeros). */
int xclip_attention_pool_fwd(const void* q, const void* kv, const uint8_t* mask, void* out, float* lse, int64_t batch, int64_t n,
                             int64_t heads, int64_t head_dim, float scale, int64_t visible_keys, int dtype, void* stream);
int xclip_attention_pool_bwd(const void* q, const void* kv, const uint8_t* mask, const void* out, const void* dout, const float* lse,
                             void* dq, void* dkv, int64_t batch, int64_t n, int64_t heads, int64_t head_dim, float scale,
                             int64_t visible_keys, int dtype, void* stream);
/* Feed-forward dropout (nn.Dropout between the inner LayerNorm and the second Linear, x_clip.py:193-194): y[i] = x[i] keep(i) / (1 - p)
 * over n contiguous elements (n a multiple of the 16-byte chunk), keep(i) iff drop_hash(seed, i) >= p 2^32.  The same call on the
 * gradient is the backward; y == x is allowed. */
int xclip_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int dtype, void* stream);

/* ---- contrastive head (similarity + InfoNCE / DCL, x_clip.py:813-847) ---------------------------------------------
 * S = scale * exp(*log_scale) * Q K^T, Q [nq, d], K [nk, d]; log_scale (device fp32 scalar, may be NULL) is the
 * temperature parameter (x_clip.py:574,736) -- it never visits the host.  Row i's positive is column i + diag_off.
 * fwd: lse[i] = log sum_j exp S_ij (positive excluded when dcl); pos[i] = S_{i,i+diag_off};
 *      *loss_accum += coef * sum_i (lse[i] - pos[i]).  workspace: xclip_simloss_workspace_bytes(nq, nk) bytes.
 * grad: G[i,j] = gmul {[a exp(S_ij - lse_q[i]) + c exp(S_ij - lse_k[j])] (1 - dcl*[j==i+diag_off]) - e [j==i+diag_off]}
 *      (gmul: device fp32 scalar = upstream d loss, may be NULL), written as G or, with g_times_scale, as
 *      scale*exp(*log_scale)*G, in `dtype` to G [nq, ldg] (columns nk .. roundup(nk, chunk) written as 0);
 *      *dtau_accum += sum_ij G_ij S_ij (NULL: skipped).  dQ = scale * G K and dK = scale * G^T Q are xclip_gemm calls. */
int64_t xclip_simloss_workspace_bytes(int64_t nq, int64_t nk);
/* The forward in two steps, so the K side can be consumed in chunks as they arrive (the local latents first, every
 * peer's all-gathered block later, reference x_clip.py:759-764 + distributed.py:14-39): `partial` reduces the columns of
 * one K chunk into per-64-column (max, sum) pairs stored in slots [tile_slot0, tile_slot0 + ceil(nk/64)) of a
 * workspace holding 2 * tile_slots * nq floats; diag_off is relative to the chunk (global offset - first column of the
 * chunk); pos must be zeroed by the caller and is written by the chunk that holds a row's positive.  `combine` folds all
 * slots into lse and adds coef * sum(lse - pos) to *loss_accum (may be NULL). */
int xclip_simloss_partial(const void* Q, const void* K, int64_t nq, int64_t nk, int64_t d, float scale, const float* log_scale,
                          int64_t diag_off, int dcl, void* workspace, int64_t tile_slot0, int64_t tile_slots, float* pos,
                          int dtype, void* stream);
int xclip_simloss_combine(const void* workspace, int64_t nq, int64_t tile_slots, const float* pos, float* lse, float* loss_accum,
                          float coef, void* stream);
int xclip_simloss_fwd(const void* Q, const void* K, int64_t nq, int64_t nk, int64_t d, float scale, const float* log_scale,
                      int64_t diag_off, int dcl, float coef, void* workspace, float* pos, float* lse, float* loss_accum, int dtype,
                      void* stream);
int xclip_simloss_grad(const void* Q, const void* K, int64_t nq, int64_t nk, int64_t d, float scale, const float* log_scale,
                       int64_t diag_off, int dcl, float a, float c, float e, const float* gmul, int g_times_scale,
                       const float* lse_q, const float* lse_k, void* G, int64_t ldg, float* dtau_accum, int dtype, void* stream);

/* ---- fine-grained (FILIP) head, use_all_token_embeds (x_clip.py:797-811) ----------------------------------------------------
 * The token similarity blocks come from xclip_gemm in chunks of `yc` images: S[(x, t), (y, k)] = <T[x,t], I[y0+y,k]> (no
 * temperature), row stride lds.  reduce: t2i[x, y0+y] = sum_t w[x,t] max_k temp*s / max(sum_t w, 1e-6), i2t[x, y0+y] = mean_k
 * max_{t: w[x,t]} temp*s (both [bx, ldo] fp32), plus the arg-max positions kmax [bx, nt, ytotal], tmax [bx, ytotal, ni] (int16).
 * route: the chunk of d loss / d s, P[(x,t),(y,k)] = temp (g1[x,y0+y] w[x,t] / cnt[x] [k == kmax] + g2[x,y0+y] / ni [t == tmax]),
 * in `dtype`, row stride ldp (padding columns zero); the backward then is dT += P I and dI = P^T T through xclip_gemm.
 * mask: [bx, nt] bytes (text != pad_id, x_clip.py:614); cnt [bx] fp32 = number of real tokens per text (written by reduce when
 * its chunk holds global column 0); log_temp: device fp32 scalar (the temperature parameter). */
int xclip_filip_reduce(const void* S, int64_t lds, const uint8_t* mask, const float* log_temp, float* t2i, float* i2t, int64_t ldo,
                       int16_t* kmax, int16_t* tmax, float* cnt, int64_t bx, int64_t nt, int64_t yc, int64_t ni, int64_t y0,
                       int64_t ytotal, int dtype, void* stream);
int xclip_filip_route(void* P, int64_t ldp, const uint8_t* mask, const float* log_temp, const float* g1, const float* g2, int64_t ldg,
                      const int16_t* kmax, const int16_t* tmax, const float* cnt, int64_t bx, int64_t nt, int64_t yc, int64_t ni,
                      int64_t y0, int64_t ytotal, int dtype, void* stream);
/* The same forward with the reductions INSIDE the token-similarity GEMM (x_clip.py:797-811 fused: the 'x t d, y i d -> x y t i' block is
 * never written): X [bx * nt, d] text-token latents, Y [yc * ni, d] image-token latents of images [y0, y0 + yc), both bf16 row-major
 * with row stride d; outputs as xclip_filip_reduce.  Needs dtype bf16, d a multiple of 64, nt >= 32, ni >= 32 (xclip_filip_fused_ok);
 * workspace: xclip_filip_fused_workspace_bytes(bx, nt, yc, ni) bytes of 4-byte partials {bf16 max | int16 arg-max}. */
int xclip_filip_fused_ok(int64_t nt, int64_t ni, int64_t d, int dtype);
int64_t xclip_filip_fused_workspace_bytes(int64_t bx, int64_t nt, int64_t yc, int64_t ni);
int xclip_filip_fused_fwd(const void* X, const uint8_t* mask, const void* Y, const float* log_temp, float* t2i, float* i2t, int64_t ldo,
                          int16_t* kmax, int16_t* tmax, float* cnt, void* workspace, int64_t workspace_bytes, int64_t bx, int64_t nt,
                          int64_t yc, int64_t ni, int64_t d, int64_t y0, int64_t ytotal, int dtype, void* stream);
/* InfoNCE / DCL over the rows of a MATERIALISED fp32 logit matrix S [rows, cols] (x_clip.py:821-847): lse[r] = log sum_c exp
 * S[r,c] (column r + diag_off left out when dcl), *loss_accum += coef * sum_r (lse[r] - S[r, r+diag_off]);
 * grad: G[r,c] = gmul * coef * (exp(S[r,c] - lse[r]) (1 - dcl [c == r+diag_off]) - [c == r+diag_off]); *dtau_accum += sum G o S. */
int xclip_rowlse(const float* S, int64_t lds, int64_t rows, int64_t cols, int64_t diag_off, int dcl, float coef, float* lse,
                 float* loss_accum, void* stream);
int xclip_rowgrad(const float* S, int64_t lds, const float* lse, int64_t rows, int64_t cols, int64_t diag_off, int dcl, float coef,
                  const float* gmul, float* G, int64_t ldg, float* dtau_accum, void* stream);
/* Masked-language-model head (MLM.forward, mlm.py:96-109; `to_logits` itself is xclip_gemm with a bias row).
 * gather_rows: out[r, :] = src[idx[r], :] -- only the masked positions of the encoder output are projected onto the vocabulary.
 * cross_entropy_fwd: lse[r] = log sum_{c < cols} exp(x[r, c]); *loss_accum += sum_r (lse[r] - x[r, labels[r]]) (the caller
 *   divides by the number of rows: F.cross_entropy(..., ignore_index) averages over the non-ignored positions).
 * cross_entropy_bwd: in place, x[r, c] <- (*gmul / rows) (softmax(x[r])[c] - [c == labels[r]]) for c < cols, 0 in the padding
 *   columns [cols, ld). */
int xclip_gather_rows(const void* src, int64_t lds, const int32_t* idx, void* out, int64_t rows, int64_t dim, int dtype, void* stream);
int xclip_cross_entropy_fwd(const void* logits, int64_t ld, const int64_t* labels, int64_t rows, int64_t cols, float* lse, float* loss_accum,
                            int dtype, void* stream);
int xclip_cross_entropy_bwd(void* logits, int64_t ld, const int64_t* labels, const float* lse, const float* gmul, int64_t rows, int64_t cols,
                            int dtype, void* stream);
/* `downsample_image_embeds` (x_clip.py:560-568), first stage: the depthwise Conv2d(C, C, 4, stride 2, padding 1, groups C, no bias)
 * over the image tokens laid out as an h x h grid.  x [batch, h*h, C] token-major, w [C, 16] (= the Conv2d weight [C, 1, 4, 4]),
 * y [batch, (h/2)^2, C].  (The 1 x 1 Conv2d with bias that follows is xclip_gemm with a bias row.)  bwd: dx [batch, h*h, C] and
 * dw_accum[C * 16] += the weight gradient (fp32), through per-wave partial rows in `workspace`. */
int64_t xclip_dwconv4s2_workspace_bytes(int64_t batch, int64_t h, int64_t C, int dtype);
int xclip_dwconv4s2_fwd(const void* x, const void* w, void* y, int64_t batch, int64_t h, int64_t C, int dtype, void* stream);
int xclip_dwconv4s2_bwd(const void* dy, const void* x, const void* w, void* dx, float* dw_accum, void* workspace, int64_t workspace_bytes,
                        int64_t batch, int64_t h, int64_t C, int dtype, void* stream);
/* Rotary position embedding (RotaryEmbedding / apply_rotary_pos_emb, x_clip.py:155-176; applied to q, k and v, :221-223), in
 * place on rows of `slots` head slots of `slot_width` (64 or 128) features (the packed qkv activation: slots = 3 * heads).  Token
 * position = row % n; in every slot the first `rot` = min(dim_head, 32) features (x_clip.py:311; even, 2 .. 32) are rotated pairwise
 * (j, j + rot / 2) by pos * inv_freq[j]; inv_freq: rot / 2 device fp32 values, the module's `inv_freq` buffer 10000^(-2 j / rot).
 * inverse != 0 applies the transposed rotation = the backward of the forward call. */
int xclip_rotary(void* x, int64_t ld, int64_t rows, int64_t n, int64_t slots, int64_t slot_width, int64_t rot, const float* inv_freq, int inverse,
                 int dtype, void* stream);
/* Similarity regularisation (x_clip.py:773-784): D[r,c] = A[r,c] - C[r,c] for two materialised similarity blocks (text-text and
 * image-image, [rows, cols] in the model dtype, row strides lda / ldc), 0 where c == r + diag_off (the global diagonal);
 * *sumsq_accum += sum D^2 (fp32, of the unrounded differences).  D (row stride ldd, model dtype) is the gradient factor:
 * d loss / d text_latent[r] = (2 w / N) sum_c D[r,c] text_latent[c], the image side with the opposite sign.  cols, lda, ldc, ldd
 * multiples of the 16-byte chunk. */
int xclip_simreg_diff(const void* A, int64_t lda, const void* C, int64_t ldc, void* D, int64_t ldd, int64_t rows, int64_t cols,
                      int64_t diag_off, float* sumsq_accum, int dtype, void* stream);

/* ---- visual self-supervision head (reference x_clip/visual_ssl.py; the Linear layers of its MLPs are xclip_gemm) -----------------
 * BatchNorm1d over the rows of x [rows, cols] (contiguous, model dtype), optionally with the ReLU that follows it in SimSiamMLP / MLP
 * (visual_ssl.py:112-136):  y = relu?((x - mean) rstd gamma + beta).  gamma / beta / running_* / mean / rstd are fp32 [cols];
 * gamma, beta NULL = BatchNorm1d(affine = False) (visual_ssl.py:135).
 *   training != 0: mean / rstd are the batch statistics (biased variance, written out for the backward) and, when running_mean is
 *     given, running = (1 - momentum) running + momentum stat with the unbiased variance -- nn.BatchNorm1d in train() mode;
 *     rows must be > 1 (the reference asserts the same, visual_ssl.py:238).
 *   training == 0: mean = running_mean, rstd = 1 / sqrt(running_var + eps).
 * bwd: dx, and dgamma / dbeta [cols] fp32 WRITTEN (either may be NULL); x, dy as in the forward call, the ReLU mask is re-evaluated
 *   from x.  workspace: xclip_batchnorm_workspace_bytes(rows, cols) bytes (per-slice column sums; no atomics, deterministic). */
int64_t xclip_batchnorm_workspace_bytes(int64_t rows, int64_t cols);
int xclip_batchnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, float* running_mean,
                        float* running_var, float momentum, float eps, int training, int relu, int64_t rows, int64_t cols,
                        void* workspace, int64_t workspace_bytes, int dtype, void* stream);
int xclip_batchnorm_bwd(const void* x, const void* dy, const float* gamma, const float* beta, const float* mean, const float* rstd,
                        void* dx, float* dgamma, float* dbeta, int training, int relu, int64_t rows, int64_t cols, void* workspace,
                        int64_t workspace_bytes, int dtype, void* stream);
/* SimSiam's loss_fn (visual_ssl.py:104-107): *loss_accum += coef sum_r (2 - 2 cos(p_r, z_r)), cos over F.normalize'd rows (eps 1e-12);
 * cosv / rp / rz [rows] fp32 (the cosine and the two reciprocal norms) are kept for the backward.  z is the stop-gradient target
 * (visual_ssl.py:243-249): bwd writes dp = *gmul coef d(2 - 2 cos) / dp only. */
int xclip_neg_cosine_fwd(const void* p, const void* z, int64_t rows, int64_t dim, float coef, float* cosv, float* rp, float* rz,
                         float* loss_accum, int dtype, void* stream);
int xclip_neg_cosine_bwd(const void* p, const void* z, const float* cosv, const float* rp, const float* rz, const float* gmul, float coef,
                         void* dp, int64_t rows, int64_t dim, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif
