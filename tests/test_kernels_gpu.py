"""MI355X run of every kernel through the C ABI of libxclip_hip.so (same cases as the emulator suite, larger
shapes incl. the BASELINE cfg2 widths)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
import kernel_cases as K  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
IDS = ["fp32", "bf16"]


@pytest.fixture(scope="module", autouse=True)
def hip_library():
    from x_clip_amd import _lib
    _lib._use_library_for_tests(None)
    assert not _lib.is_emulator()
    _lib.lib()          # raises if libxclip_hip.so is missing -- no fallback
    yield


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
@pytest.mark.parametrize("rows,dim,geglu,res", [(5, 64, False, False), (1031, 512, False, True), (517, 2048, True, False),
                                                (66, 4096, True, False), (7, 1544, True, False), (130, 768, False, False), (9000, 512, False, False),
                                                (13001, 2048, True, False)])     # (GEGLU backward: 3072 work-groups of the 3251 the workspace is sized for, 2 - 3 passes each)
def test_layernorm(dtype, rows, dim, geglu, res):
    K.case_layernorm(DEV, dtype, rows, dim, geglu, res)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
@pytest.mark.parametrize("batch,n,heads,masked,hd,row,causal", [(4, 257, 8, True, 64, 0, False), (3, 77, 12, True, 64, 0, False), (2, 290, 3, False, 128, 0, False), (2, 1000, 2, True, 64, 5, True), (5, 33, 2, True, 128, 0, False)])
def test_attention_pool(dtype, batch, n, heads, masked, hd, row, causal):
    """attention for one query row per (sample, head): key counts that are not multiples of the keys per wave-load (8 / 4 / 16), masks with a
    hole and a padded tail, both head-slot widths, a pooled row in the middle under a causal mask"""
    K.case_attention_pool(DEV, dtype, batch, n, heads, masked, hd, row, causal)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
def test_l2norm(dtype):
    K.case_l2norm(DEV, dtype, 1027, 512)
    K.case_l2norm(DEV, dtype, 33, 64)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
def test_text_embed(dtype):
    K.case_text_embed(DEV, dtype, 37, 256, 512, 1000)
    K.case_text_embed(DEV, dtype, 2, 5, 72, 11, has_pos=False, has_cls=False)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
def test_text_embed_out_of_range_ids(dtype):
    K.case_text_embed_bad_ids(DEV, dtype)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
def test_patchify(dtype):
    K.case_patchify(DEV, dtype, 5, 3, 256, 32, 0.5)
    K.case_patchify(DEV, dtype, 3, 3, 224, 16, 1.0)
    K.case_patchify(DEV, dtype, 2, 3, 28, 14, 1.0)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
def test_token_mean(dtype):
    K.case_token_mean(DEV, dtype, 33, 32, 512)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
@pytest.mark.parametrize("M,N,K_", [(136, 72, 96), (1000, 1536, 512), (520, 512, 2048)])
def test_gemm_layouts(dtype, layout, M, N, K_):
    K.case_gemm(DEV, dtype, M, N, K_, layout)


@pytest.mark.parametrize("layout,M,N,K_,res", [("nt", 263168, 1536, 512, False), ("nn", 263168, 512, 1536, False), ("nt", 131584, 512, 512, False),
                                               ("tn", 1536, 512, 263168, False), ("nt", 263168, 512, 2048, True), ("tn", 512, 2048, 263168, False),
                                               ("nt", 65536, 4096, 512, False),
                                               # round 4: the row tail as a split-K problem (xclip_api.hip gemm2_tail_cut) -- text rows 1028 row tiles,
                                               # vision rows 132: FF1 input gradient, FF2 forward + skip through the reduction's residual term
                                               ("nn", 263168, 512, 4096, False), ("nn", 33792, 512, 4096, False), ("nt", 33792, 512, 2048, True),
                                               # round 5: gemm8.h (the asm unit) takes every plain interior NT / NN product with >= 8 K steps: the
                                               # FF2 input gradient (k-major B at K = 512: all eight steps carry the previous tile's stores)
                                               ("nn", 263168, 2048, 512, False), ("nt", 33792, 1536, 512, False)])
def test_gemm_full_size_every_element_and_repeatable(layout, M, N, K_, res):
    """text-tower shapes at full size, every CU streaming (the regime the counted DMA waits and the stores left in flight across the
    tile boundary have to be right in -- the emulator lands every DMA piece at once and cannot see an early read, nor a missing wait
    state in front of an asm store): every output element against an fp32-accumulated reference product of the same bf16 operands,
    and ten launches bit-identical.  The plain NT / NN products run on gemm8.h's hand-scheduled body (held line stores under the next tile's
    MFMAs: no emulator twin, this test and tools/probe_gemm8.py are its gate).  Covers every interior-tile epilogue of gemm4.h: plain whole-line stores (nt / nn), the fp32
    split-K slab (tn), the residual form (res: FF2 + skip), and the streamed (> 48 MiB) + banded (16 N tiles) FF1 output."""
    from x_clip_amd import ops
    a_k, b_k = layout == "tn", layout in ("nn", "tn")
    g = torch.Generator(device="cpu").manual_seed(3)
    a = torch.randn((K_, M) if a_k else (M, K_), generator=g).to(torch.bfloat16).to(DEV)
    b = torch.randn((K_, N) if b_k else (N, K_), generator=g).to(torch.bfloat16).to(DEV)
    r = torch.randn(M, N, generator=g).to(torch.bfloat16).to(DEV) if res else None
    outs = [ops.gemm(a, b, M, N, K_, a_k, b_k, residual=r) for _ in range(10 if M * N <= (1 << 28) else 4)]
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "the same launch must give the same bits"
    A = a.t() if a_k else a
    B = b if b_k else b.t()
    rows = 16384                                                 # reference in row blocks (fp32 accumulate on the device, checked on the fly)
    worst = 0.0
    for r0 in range(0, M, rows):
        ref = A[r0: r0 + rows].float() @ B.float()
        if res:
            ref = ref + r[r0: r0 + rows].float()
        got = outs[0][r0: r0 + rows].float()
        scale = float(ref.abs().max())
        worst = max(worst, float((got - ref.to(torch.bfloat16).float()).abs().max()) / (scale * 2.0 ** -8))
    # element-wise 1 bf16 ulp holds per element (kernel_cases.close); here: no element off by more than 2 ulps OF THE SCALE -- a tile
    # computed from a stale LDS stage is off by the size of the output itself
    assert worst <= 2.0, worst


@pytest.mark.parametrize("layout,M,N,K_,res", [("nt", 1024, 512, 512, False), ("nn", 1024, 512, 512, False), ("tn", 512, 512, 1024, False),
                                               ("nt", 1024, 512, 2048, True), ("nt", 1024, 4096, 512, False), ("nn", 1024, 512, 2048, False),
                                               ("tn", 4096, 512, 1024, False), ("tn", 512, 2048, 1024, False), ("nn", 1024, 2048, 512, False),
                                               ("nt", 64, 64, 64, False), ("tn", 1024, 512, 1024, True)])
def test_gemm_small_products_every_element_and_repeatable(layout, M, N, K_, res):
    """gemm_small.h at the shapes of the pooled last layer / latent projections (b = 1024): counted waits on a ring of LDS-DMA stages that
    the emulator cannot see (it lands every piece at once) -- every element against an fp32 product, ten launches bit-identical, and the same
    product through the 256 x 256 kernels (xclip_gemm_small_limit(0)) within the bf16 rounding of the two summation orders"""
    from x_clip_amd import ops
    assert ops.gemm_small_limit() >= 2 * M * N * K_
    a_k, b_k = layout == "tn", layout in ("nn", "tn")
    g = torch.Generator(device="cpu").manual_seed(5)
    a = torch.randn((K_, M) if a_k else (M, K_), generator=g).to(torch.bfloat16).to(DEV)
    b = torch.randn((K_, N) if b_k else (N, K_), generator=g).to(torch.bfloat16).to(DEV)
    r = torch.randn(M, N, generator=g).to(torch.bfloat16).to(DEV) if res else None
    outs = [ops.gemm(a, b, M, N, K_, a_k, b_k, residual=r) for _ in range(10)]
    was = ops.gemm_small_limit(0)
    try:
        big = ops.gemm(a, b, M, N, K_, a_k, b_k, residual=r)
    finally:
        ops.gemm_small_limit(was)
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "the same launch must give the same bits"
    ref = (a.t() if a_k else a).float() @ (b if b_k else b.t()).float()
    if res:
        ref = ref + r.float()
    scale = float(ref.abs().max())
    assert float((outs[0].float() - ref.to(torch.bfloat16).float()).abs().max()) <= 2.0 * scale * 2.0 ** -8
    assert float((outs[0].float() - big.float()).abs().max()) <= 2.0 * scale * 2.0 ** -8


@pytest.mark.parametrize("M,F,D", [(2048, 512, 128), (33792, 2048, 512), (263168, 2048, 512)])
def test_ffn_dgrad_geglu_fused(M, F, D):
    """gemm9.h: net.4's input gradient + the GEGLU-LayerNorm backward in one kernel, at the vision / text towers' full sizes (every CU streaming)"""
    K.case_ffn_dgrad_geglu(DEV, M, F, D)


@pytest.mark.parametrize("M,F,D", [(2048, 512, 128), (33792, 2048, 512), (263168, 2048, 512)])
def test_ffn_rowstats_in_layernorm_bwd(M, F, D):
    """round 6: the fused feed-forward backward's row constants written by the LayerNorm backward above the block, at the towers' sizes"""
    K.case_ffn_rowstats_in_layernorm_bwd(DEV, M, F, D)


@pytest.mark.parametrize("resid_scale", [16.0, 100.0])
def test_ffn_dgrad_geglu_fused_large_residual_stream(resid_scale):
    """the residual stream 16 x / 100 x the feed-forward block's own output, at the vision tower's size"""
    K.case_ffn_dgrad_geglu(DEV, 33792, 2048, 512, resid_scale=resid_scale)


@pytest.mark.parametrize("layout,M,N,K_,alpha,in_place", [("nt", 4104, 512, 2048, 1.0, False), ("nn", 1024, 520, 256, 0.5, True), ("nt", 65792, 512, 2048, 1.0, False)])
def test_gemm_residual_epilogue(layout, M, N, K_, alpha, in_place):
    K.case_gemm(DEV, torch.bfloat16, M, N, K_, layout, alpha=alpha, residual_only=True, in_place=in_place)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
def test_gemm_epilogue_and_splitk(dtype):
    K.case_gemm(DEV, dtype, 1030, 520, 3072, "nt", epilogue=True, alpha=0.5)
    K.case_gemm(DEV, dtype, 1536, 512, 33000, "tn")          # wgrad shape: long token contraction, split-K
    K.case_gemm(DEV, dtype, 64, 64, 1536, "tn")


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
@pytest.mark.parametrize("n,heads,masked", [(33, 2, False), (70, 2, True), (257, 8, True), (32, 8, False), (197, 3, True), (258, 4, True), (288, 2, False), (66, 2, False)])
def test_attention(dtype, n, heads, masked):
    K.case_attention(DEV, dtype, 3, n, heads, masked)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
@pytest.mark.parametrize("n,heads,masked", [(65, 2, False), (256, 8, True), (257, 4, True), (97, 3, True), (32, 2, True), (288, 2, False), (320, 2, True)])
def test_attention_causal(dtype, n, heads, masked):
    K.case_attention(DEV, dtype, 3, n, heads, masked, causal=True)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
@pytest.mark.parametrize("n,heads,masked,causal", [(33, 2, False, False), (70, 3, True, False), (257, 8, True, False), (288, 2, False, False),
                                                   (97, 2, True, True), (320, 2, True, True)])
def test_attention_wide_heads(dtype, n, heads, masked, causal):
    """128-feature head slots (reference Attention accepts any dim_head, x_clip.py:201-212): two 64-wide halves per head"""
    K.case_attention(DEV, dtype, 3, n, heads, masked, causal=causal, hd=128)


def test_attention_streaming_backward_measurement_build():
    """attention6.h on the hardware (round 6; MEASUREMENT build, XCLIP_ATTN_BWD=6: slower than attention5.h, kept as a measured alternative):
    the streamed ring, the counted waits that leave a step's requests in flight, the K / V image hand-over across heads and the integer
    LDS atomics against the fp64 reference -- 8192 heads on 256 persistent work-groups (32 heads each) -- and bit-reproducible dQ"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from x_clip_amd import _lib, ops\n"
        "_lib.use_measurement_build()\n"
        "import kernel_cases as K\n"
        "dev = torch.device('cuda:0')\n"
        "K.case_attention(dev, torch.bfloat16, 3, 256, 2, True)\n"
        "K.case_attention_single_tail(dev, torch.bfloat16, n=257, heads=8)\n"
        "K.case_attention(dev, torch.bfloat16, 1024, 257, 8, True)\n"
        "torch.manual_seed(0)\n"
        "qkv = torch.randn(300, 257, 3 * 8 * 64, device=dev).bfloat16(); do = torch.randn(300, 257, 8 * 64, device=dev).bfloat16()\n"
        "out, lse = ops.attention_fwd(qkv, None, 8, 0.125)\n"
        "a = ops.attention_bwd(qkv, None, out, do, lse, 8, 0.125); b = ops.attention_bwd(qkv, None, out, do, lse, 8, 0.125)\n"
        "assert torch.equal(a, b) and torch.isfinite(a.float()).all()\n"
        "print('attn6 ok')\n") % (here, os.path.dirname(here))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, XCLIP_ATTN_BWD="6"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "attn6 ok" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])


def test_attention_persistent_single_pass_measurement_build():
    """attention7.h on the hardware (round 6; MEASUREMENT build, XCLIP_ATTN_BWD=7: attention5.h persistent with the next head's images requested
    by asm-issued DMA under this head's stores -- measured slower, kept as the recorded alternative): against the fp64 reference with 8 heads
    per work-group, and the SAME bits as the product kernel where the fp32 operation order is the same (n = 256).  (attention5.h's round-5
    forms, XCLIP_ATTN5_VAR=3, are compared on the emulator: tests/test_kernels_emu.py)"""
    import os
    import subprocess
    import sys
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    head = (
        "import sys, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from x_clip_amd import _lib, ops\n"
        "_lib.use_measurement_build()\n"
        "import kernel_cases as K\n"
        "dev = torch.device('cuda:0')\n") % (here, os.path.dirname(here))
    cases = (
        "K.case_attention(dev, torch.bfloat16, 3, 256, 2, True)\n"
        "K.case_attention_single_tail(dev, torch.bfloat16, n=257, heads=8)\n"
        "K.case_attention(dev, torch.bfloat16, 256, 257, 8, True)\n")
    tail = (
        "torch.manual_seed(0)\n"
        "qkv = torch.randn(300, 256, 3 * 8 * 64, device=dev).bfloat16(); do = torch.randn(300, 256, 8 * 64, device=dev).bfloat16()\n"
        "mask = torch.rand(300, 256, device=dev) > 0.2\n"
        "out, lse = ops.attention_fwd(qkv, mask, 8, 0.125)\n"
        "torch.save(ops.attention_bwd(qkv, mask, out, do, lse, 8, 0.125).cpu(), sys.argv[1])\n"
        "print('attn ok')\n")
    with tempfile.TemporaryDirectory() as tmp:
        got = {}
        for name, code, env in [("product", head + tail, dict(XCLIP_ATTN_BWD="5")), ("persistent", head + cases + tail, dict(XCLIP_ATTN_BWD="7"))]:
            path = os.path.join(tmp, name + ".pt")
            out = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
            assert out.returncode == 0 and "attn ok" in out.stdout, (name, out.stdout[-500:], out.stderr[-3000:])
            got[name] = torch.load(path)
        assert torch.equal(got["product"], got["persistent"])


def test_attention_single_tail_row():
    """257 = 8 x 32 + 1 tokens, not causal: the tail key / query as the accumulators' initial values (no 33rd block), with and without masks"""
    K.case_attention_single_tail(DEV, torch.bfloat16)
    K.case_attention_single_tail(DEV, torch.bfloat16, n=33, heads=2)          # the vision tower's 32 kept patches + CLS: ONE wave per head
    K.case_attention_single_tail(DEV, torch.bfloat16, n=129, heads=3)
    K.case_attention_single_tail(DEV, torch.bfloat16, n=257, heads=8)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
def test_attention_rescale_spike(dtype):
    K.case_attention_spike(DEV, dtype)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
@pytest.mark.parametrize("dcl", [False, True])
def test_simloss(dtype, dcl):
    K.case_simloss(DEV, dtype, 12, 12, 64, dcl)
    K.case_simloss(DEV, dtype, 300, 1100, 512, dcl, diag_off=600)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
@pytest.mark.parametrize("dcl", [False, True])
def test_simloss_closed_form(dtype, dcl):
    K.case_simloss_closed_form(DEV, dtype, 12, 64, dcl)
    K.case_simloss_closed_form(DEV, dtype, 520, 512, dcl)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
def test_layernorm_residual_paths(dtype):
    K.case_layernorm_residual_paths(DEV, dtype, 640, 512, 32)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
def test_row_moves(dtype):
    K.case_row_moves(DEV, dtype)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
@pytest.mark.parametrize("dcl", [False, True])
def test_simloss_chunked(dtype, dcl):
    K.case_simloss_chunked(DEV, dtype, dcl)


@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
@pytest.mark.parametrize("M,N,K_", [(264, 136, 128), (1032, 1536, 512), (2048, 512, 2048), (520, 776, 192)])
def test_gemm2_layouts(layout, M, N, K_):
    K.case_gemm(DEV, torch.bfloat16, M, N, K_, layout)


def test_gemm2_epilogue_and_splitk():
    K.case_gemm(DEV, torch.bfloat16, 1032, 520, 3072, "nt", epilogue=True, alpha=0.5)
    K.case_gemm(DEV, torch.bfloat16, 1536, 512, 64 * 700, "tn")      # wgrad shape: long token contraction, split-K
    K.case_gemm(DEV, torch.bfloat16, 512, 2048, 64 * 333, "tn")
    K.case_gemm(DEV, torch.bfloat16, 4096, 512, 64 * 129, "nn")


def test_sort_ids_stable_radix():
    """sort.h (round 6): the stable radix sort of (id, position) pairs in front of the segmented embedding-gradient sums, against
    torch.sort(stable=True) -- replaces the torch.sort of the reference-shaped backward (nn.Embedding backward, x_clip.py:320)"""
    K.case_sort_ids(DEV)
    K.case_sort_ids(DEV, sizes=((263168, 49408), (32768, 64), (1 << 20, 1 << 18)))                 # the step's two sorts at full size, and a million ids


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
def test_scatter_sorted_and_gelu(dtype):
    K.case_scatter_sorted(DEV, dtype)
    K.case_gelu_accuracy(DEV, dtype)


@pytest.mark.parametrize("dcl", [False, True])
def test_simloss_on_gemm_loop(dcl):
    K.case_simloss(DEV, torch.bfloat16, 264, 392, 64, dcl, diag_off=100)
    K.case_simloss(DEV, torch.bfloat16, 264, 776, 128, dcl, diag_off=100)
    for nq, nk, off in [(520, 1100, 300), (512, 1024, 384), (300, 700, -100), (256, 768, 5000), (1030, 520, 512), (4096, 8200, 4096)]:
        K.case_simloss(DEV, torch.bfloat16, nq, nk, 512, dcl, diag_off=off)        # interior + edge launches of G (simloss5.h)
    K.case_simloss(DEV, torch.bfloat16, 1024, 4096, 512, dcl, diag_off=2048)
    K.case_simloss(DEV, torch.bfloat16, 512, 3072, 512, dcl, diag_off=1000)          # > 8 column tiles: banded tile order (6 / 5 / 5)
    K.case_simloss(DEV, torch.bfloat16, 300, 2560, 512, dcl, diag_off=0)
    K.case_simloss(DEV, torch.bfloat16, 256, 2400, 512, dcl, diag_off=2100)
    for d in (64, 512):                                                             # exp(tau) = 200: no exp(scale - lse) anywhere
        K.case_simloss(DEV, torch.bfloat16, 512, 1024, d, dcl, diag_off=384, tau=5.3)
        K.case_simloss(DEV, torch.bfloat16, 264, 392, d, dcl, diag_off=100, tau=5.3)
    K.case_simloss_closed_form(DEV, torch.bfloat16, 1032, 512, dcl)


@pytest.mark.parametrize("dcl", [False, True], ids=["infonce", "dcl"])
@pytest.mark.parametrize("nq,nk,d,off", [(256, 256, 64, 0), (264, 392, 64, 100), (512, 1024, 512, 384), (1024, 4096, 512, 2048)])
def test_simloss_grad_with_spread_lse(nq, nk, d, off, dcl):
    """exp(tau) = 200 and four perfectly matched pairs: log-sum-exps 150 apart inside one wave block (NaN with the round-3 G)"""
    K.case_simloss_spread(DEV, torch.bfloat16, nq, nk, d, dcl, diag_off=off)
    # spread > 300: beyond what ANY single reference point bridges -- the wave blocks with matched rows take the two-exponential form
    K.case_simloss_spread(DEV, torch.bfloat16, nq, nk, d, dcl, diag_off=off, temp=400.0)
    if nq <= 264:
        K.case_simloss_spread(DEV, torch.float32, nq, nk, d, dcl, diag_off=off)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
@pytest.mark.parametrize("rows,cols,diag_off", [(5, 16, 0), (1031, 4104, 512), (4096, 32768, 8192)])
def test_simreg_diff(dtype, rows, cols, diag_off):
    K.case_simreg_diff(DEV, dtype, rows, cols, diag_off)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
@pytest.mark.parametrize("batch,n,heads", [(2, 5, 1), (7, 257, 8), (3, 33, 4)])
def test_rotary(dtype, batch, n, heads):
    K.case_rotary(DEV, dtype, batch, n, heads)
    K.case_rotary(DEV, dtype, batch, n, heads, hd=128)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
@pytest.mark.parametrize("batch,h,C", [(2, 4, 64), (3, 14, 512), (5, 8, 1024)])
def test_dwconv(dtype, batch, h, C):
    K.case_dwconv(DEV, dtype, batch, h, C)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
@pytest.mark.parametrize("rows,cols,ld", [(5, 24, 24), (700, 10000, 10000), (33, 1003, 1008), (9000, 40, 40)])
def test_cross_entropy(dtype, rows, cols, ld):
    K.case_cross_entropy(DEV, dtype, rows, cols, ld)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
@pytest.mark.parametrize("rows,dim", [(5, 64), (1031, 512), (4100, 1024)])
def test_layernorm_chain(dtype, rows, dim):
    K.case_layernorm_chain(DEV, dtype, rows, dim)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("rows,cols,relu,affine,training,offset", [(70000, 4096, True, True, True, 0.0), (130, 512, True, True, True, 0.0), (37, 64, False, False, True, 50.0),
                                                                    (8, 96, True, True, True, 0.0), (50, 256, True, True, False, 0.0),
                                                                    (2, 1024, False, True, True, 0.0)])
def test_batchnorm(dtype, rows, cols, relu, affine, training, offset):
    K.case_batchnorm(DEV, dtype, rows, cols, relu, affine, training, offset)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("rows,dim", [(130, 256), (3, 64), (17, 1024), (4300, 64)])
def test_neg_cosine(dtype, rows, dim):
    K.case_neg_cosine(DEV, dtype, rows, dim)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("rows,dim,temperature", [(20, 32, 0.5), (70, 128, 1.0), (5, 12, 0.3), (1000, 128, 2.0), (4100, 64, 1.0)])
def test_nt_xent(dtype, rows, dim, temperature):
    K.case_nt_xent(DEV, dtype, rows, dim, temperature)


def test_gemm_splitk_uneven_slices():
    K.case_gemm_splitk_uneven(DEV)


def test_gemm_random_shapes():
    K.case_gemm_fuzz(DEV, 150, seed=1)


def test_attention_random_lengths_bf16():
    """sequence lengths the fixed list does not hit (every residue of n mod 32 changes which sub-tiles are masked, which blocks are
    cooperative tails and how the software-pipelined sweeps end), with and without key padding / causal masking"""
    g = torch.Generator().manual_seed(5)
    for _ in range(14):
        n = int(torch.randint(33, 289, (1,), generator=g))
        heads = int(torch.randint(1, 4, (1,), generator=g))
        masked = bool(torch.randint(0, 2, (1,), generator=g))
        causal = bool(torch.randint(0, 3, (1,), generator=g) == 0)
        K.case_attention(DEV, torch.bfloat16, 2, n, heads, masked, causal=causal)


@pytest.mark.parametrize("layout,M,N,K_", [("nt", 520, 4096, 128), ("nn", 264, 3072, 64), ("nt", 256, 6144, 64)])
def test_gemm_banded_tile_order(layout, M, N, K_):
    """more than 8 N tiles: the ring kernel's banded tile order (bands of 8 / 6 / 8 tiles) visits every tile exactly once"""
    K.case_gemm(DEV, torch.bfloat16, M, N, K_, layout)


@pytest.mark.parametrize("bx,nt,by,ni,d,chunks", [(5, 77, 4, 98, 64, 1), (4, 64, 7, 64, 128, 1), (3, 130, 3, 200, 64, 2), (64, 77, 96, 98, 512, 1),
                                                  (48, 77, 40, 196, 512, 3), (16, 77, 24, 288, 768, 1), (33, 40, 50, 33, 512, 1), (24, 256, 96, 32, 512, 2)])
def test_filip_fused(bx, nt, by, ni, d, chunks):
    """the FILIP forward with its reductions inside the GEMM epilogue (filip5.h): emulator shapes, then configs[3]-like token counts
    (77 x 98, 77 x 196 in three image chunks) over hundreds of tiles and the ViT-L FILIP token count (288) at d = 768"""
    K.case_filip_fused(DEV, bx, nt, by, ni, d, chunks=chunks)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
@pytest.mark.parametrize("n,heads,masked,causal,hd", [(33, 2, False, False, 64), (257, 8, True, False, 64), (97, 3, True, True, 64), (288, 2, True, False, 128)])
def test_attention_dropout(dtype, n, heads, masked, causal, hd):
    K.case_attention(DEV, dtype, 3, n, heads, masked, causal=causal, hd=hd, drop=(0.25, 0xC0FFEE1234567))


@pytest.mark.parametrize("dtype", K.DTYPES, ids=IDS)
def test_dropout(dtype):
    K.case_dropout(DEV, dtype)
    K.case_dropout(DEV, dtype, n=(1 << 25) + 4096, p=0.1)              # many work-groups, grid-stride loop
