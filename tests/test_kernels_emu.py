"""CPU run of every kernel: the product's kernel sources + C ABI compiled for the host against the wave64 emulator
(tests/emu), driven through the same Python wrappers (x_clip_amd.ops) and checked against the oracle expressions.
Sizes are kept small (the emulator executes every GPU thread as a fibre)."""
import os
import sys

import pytest
import torch

from x_clip_amd import _lib

sys.path.insert(0, os.path.dirname(__file__))
import kernel_cases as K  # noqa: E402
from emu.build_emu import build  # noqa: E402

DEV = torch.device("cpu")


@pytest.fixture(scope="module", autouse=True)
def emulator_library():
    path = build()
    _lib._use_library_for_tests(path)
    yield
    _lib._use_library_for_tests(None)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("rows,dim,geglu,res", [(5, 64, False, False), (9, 512, False, True), (6, 256, True, False),
                                                (3, 2048, True, False), (2, 1096, False, False)])
def test_layernorm(dtype, rows, dim, geglu, res):
    K.case_layernorm(DEV, dtype, rows, dim, geglu, res)


@pytest.mark.parametrize("M,F,D", [(256, 256, 128), (512, 512, 192)])
def test_ffn_dgrad_geglu_fused(M, F, D):
    K.case_ffn_dgrad_geglu(DEV, M, F, D)


@pytest.mark.parametrize("M,F,D", [(256, 256, 128), (512, 256, 512)])
def test_ffn_rowstats_in_layernorm_bwd(M, F, D):
    K.case_ffn_rowstats_in_layernorm_bwd(DEV, M, F, D)


@pytest.mark.parametrize("resid_scale", [16.0, 100.0])
def test_ffn_dgrad_geglu_fused_large_residual_stream(resid_scale):
    """|x1| = 16 x / 100 x the block's output (the second row statistic comes from the bf16 difference x2 - x1); the fused kernel
    must stay within the bound the two-kernel path defines"""
    K.case_ffn_dgrad_geglu(DEV, 256, 256, 128, resid_scale=resid_scale)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
def test_l2norm(dtype):
    K.case_l2norm(DEV, dtype, 7, 64)
    K.case_l2norm(DEV, dtype, 3, 512)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
def test_text_embed(dtype):
    K.case_text_embed(DEV, dtype, 3, 9, 64, 50)
    K.case_text_embed(DEV, dtype, 2, 5, 72, 11, has_pos=False, has_cls=False)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
def test_text_embed_out_of_range_ids(dtype):
    K.case_text_embed_bad_ids(DEV, dtype)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
def test_patchify(dtype):
    K.case_patchify(DEV, dtype, 2, 3, 64, 32, 1.0)
    K.case_patchify(DEV, dtype, 3, 3, 32, 8, 0.5)
    K.case_patchify(DEV, dtype, 1, 3, 28, 14, 1.0)      # 588-wide rows: zero padded to the chunk


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
def test_token_mean(dtype):
    K.case_token_mean(DEV, dtype, 3, 5, 64)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
def test_gemm_layouts(dtype, layout):
    K.case_gemm(DEV, dtype, 136, 72, 96, layout)            # ragged M / N tiles, K tail


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
def test_gemm_epilogue_and_splitk(dtype, big_gemm_kernels):
    K.case_gemm(DEV, dtype, 40, 136, 64, "nt", epilogue=True, alpha=0.5)
    K.case_gemm(DEV, dtype, 64, 64, 1536, "tn")             # long contraction -> split-K slabs + reduce


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("n,masked", [(33, False), (70, True), (97, True), (66, False)])
def test_attention(dtype, n, masked):
    K.case_attention(DEV, dtype, 2, n, 2, masked)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("n,masked", [(65, False), (97, True), (32, True)])
def test_attention_causal(dtype, n, masked):
    K.case_attention(DEV, dtype, 2, n, 2, masked, causal=True)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("n,masked,causal", [(33, False, False), (70, True, False), (97, True, True), (130, False, True), (257, True, False), (258, False, True)])
def test_attention_wide_heads(dtype, n, masked, causal):
    """128-feature head slots (reference Attention accepts any dim_head, x_clip.py:201-212): two 64-wide halves per head"""
    K.case_attention(DEV, dtype, 2, n, 2, masked, causal=causal, hd=128)


@pytest.mark.parametrize("n", [257, 320])
def test_attention_causal_long_bf16(n):
    """n = 257: the cooperative-tail path of the head-resident kernels; n = 320: the tiled kernels for long sequences"""
    K.case_attention(DEV, torch.bfloat16, 1, n, 1, True, causal=True)


@pytest.mark.parametrize("n,heads,masked", [(224, 2, True), (225, 1, False), (256, 1, True), (257, 2, True), (257, 1, False)])
def test_attention_single_pass_backward(n, heads, masked):
    """attention5.h: the backward as one pass over the (query block, key block) pairs -- the sequences of 7 or 8 whole 32-blocks (+ one tail row)
    the host sends there (shorter ones stay on attention3.h: measured slower, xclip_attn.hip)"""
    K.case_attention(DEV, torch.bfloat16, 2, n, heads, masked)


def test_attention_single_tail_row():
    """257 = 8 x 32 + 1 tokens, not causal: the tail key / query as the accumulators' initial values (no 33rd block), with and without masks"""
    K.case_attention_single_tail(DEV, torch.bfloat16)
    K.case_attention_single_tail(DEV, torch.bfloat16, n=33, heads=2)          # the vision tower's 32 kept patches + CLS: ONE wave per head
    K.case_attention_single_tail(DEV, torch.bfloat16, n=129, heads=1)


def test_attention_rescale_spike():
    K.case_attention_spike(DEV, torch.float32)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("dcl", [False, True])
def test_simloss(dtype, dcl):
    K.case_simloss(DEV, dtype, 12, 12, 64, dcl)
    K.case_simloss(DEV, dtype, 20, 140, 32, dcl, diag_off=100)       # a rank's row block of a larger global batch


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("dcl", [False, True])
def test_simloss_closed_form(dtype, dcl):
    K.case_simloss_closed_form(DEV, dtype, 12, 64, dcl)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
def test_layernorm_residual_paths(dtype):
    K.case_layernorm_residual_paths(DEV, dtype, 12, 64, 4)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
def test_row_moves(dtype):
    K.case_row_moves(DEV, dtype)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("dcl", [False, True])
def test_simloss_chunked(dtype, dcl):
    K.case_simloss_chunked(DEV, dtype, dcl)


@pytest.fixture
def big_gemm_kernels():
    """shapes that gemm_small.h would take stay on the 256 x 256 kernels these tests are about"""
    was = K.ops.gemm_small_limit(0)
    yield
    K.ops.gemm_small_limit(was)


@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
@pytest.mark.parametrize("M,N,K_", [(128, 64, 64), (64, 192, 512), (192, 128, 640), (64, 64, 1536)])
def test_gemm_small_layouts(layout, M, N, K_):
    """gemm_small.h: one K step (two stages), the whole contraction requested up front (eight steps: the ring does not wrap), a ring that
    wraps twice, and a long one (24 steps through eight stages); every layout"""
    assert K.ops.gemm_small_limit() > 2 * M * N * K_
    K.case_gemm(DEV, torch.bfloat16, M, N, K_, layout)


@pytest.mark.parametrize("layout,alpha,in_place", [("nt", 1.0, False), ("nn", 0.5, True), ("tn", 0.25, False)])
def test_gemm_small_residual_and_alpha(layout, alpha, in_place):
    K.case_gemm(DEV, torch.bfloat16, 128, 128, 256, layout, alpha=alpha, residual_only=True, in_place=in_place)
    K.case_gemm(DEV, torch.bfloat16, 64, 128, 128, layout, alpha=alpha)


def test_gemm_small_many_tiles_four_stages():
    """more than 256 tiles: four stages (two work-groups per CU on the part), the ring wraps"""
    K.case_gemm(DEV, torch.bfloat16, 1088, 1024, 384, "nt")


def test_gemm_small_limit_is_honoured():
    was = K.ops.gemm_small_limit(0)
    try:
        assert K.ops.gemm_small_limit() == 0
        K.case_gemm(DEV, torch.bfloat16, 128, 64, 128, "nt")           # (the same product through the other kernels)
    finally:
        K.ops.gemm_small_limit(was)
    assert K.ops.gemm_small_limit() == was


@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
def test_gemm2_layouts(layout):
    """256x256 DMA-staged bf16 kernel (gemm2.h): ragged M and N tiles, two K steps"""
    K.case_gemm(DEV, torch.bfloat16, 264, 136, 128, layout)


@pytest.mark.parametrize("layout", ["nt", "tn"])
def test_gemm3_persistent_wraparound(layout):
    """6 tiles on the emulator's 3 'CUs': every work-group walks two tiles (cross-tile DMA prefetch, stage parity carry-over)"""
    K.case_gemm(DEV, torch.bfloat16, 520, 264, 192, layout)


@pytest.mark.parametrize("layout,M,N,K_,alpha", [("nt", 1024, 512, 128, 1.0), ("nn", 768, 512, 64, 0.5), ("nt", 776, 512, 64, 1.0)])
def test_gemm4_interior_tiles_in_a_row(layout, M, N, K_, alpha, big_gemm_kernels):
    """interior tiles in a row on one work-group (3 emulated CUs, 6-8 tiles): the straight-line descriptor epilogue, the C = 0 first
    k-block of every tile, a single K step per tile (the three-stage A ring wraps inside the prologue), a ragged last row of tiles"""
    K.case_gemm(DEV, torch.bfloat16, M, N, K_, layout, alpha=alpha)


@pytest.mark.parametrize("layout,M,N,K_,alpha,in_place", [("nt", 776, 512, 128, 1.0, False), ("nn", 512, 264, 64, 0.5, True), ("nt", 264, 136, 64, 1.0, False)])
def test_gemm4_residual_epilogue(layout, M, N, K_, alpha, in_place, big_gemm_kernels):
    """C = alpha A B + R through the straight-line residual epilogue (interior tiles) and the general one (ragged tiles); in place too"""
    K.case_gemm(DEV, torch.bfloat16, M, N, K_, layout, alpha=alpha, residual_only=True, in_place=in_place)


def test_gemm5_banded_tile_order():
    """16 N tiles: the ring kernel walks them in two bands of 8 (every tile exactly once, interior + a ragged last row of tiles)"""
    K.case_gemm(DEV, torch.bfloat16, 264, 4096, 64, "nt")


def test_gemm4_slab_epilogue_interior_tiles(big_gemm_kernels):
    """split-K weight-gradient shape whose tiles are all interior: the fp32 slab leaves through the wave-private LDS transposition
    (whole-line stores) -- only the LOGIC can be checked here; the store hazard this path once hit exists on the hardware only
    (tests/test_kernels_gpu.py::test_gemm_layouts[520-512-2048-tn], tools/debug/slab_epilogue_check.py)"""
    K.case_gemm(DEV, torch.bfloat16, 512, 256, 1024, "tn")


def test_gemm2_epilogue_and_splitk():
    K.case_gemm(DEV, torch.bfloat16, 136, 264, 64, "nt", epilogue=True, alpha=0.5)
    K.case_gemm(DEV, torch.bfloat16, 776, 264, 64, "nt", epilogue=True, alpha=0.5)      # persistent + epilogue terms
    K.case_gemm(DEV, torch.bfloat16, 128, 136, 1024, "tn")           # split-K slabs + reduce


def test_sort_ids_stable_radix():
    """sort.h (round 6): the stable radix sort of (id, position) pairs in front of the segmented embedding-gradient sums, against
    torch.sort(stable=True) -- replaces the torch.sort of the reference-shaped backward (nn.Embedding backward, x_clip.py:320)"""
    K.case_sort_ids(DEV)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
def test_scatter_sorted_and_gelu(dtype):
    K.case_scatter_sorted(DEV, dtype)
    K.case_gelu_accuracy(DEV, dtype)


@pytest.mark.parametrize("dcl", [False, True])
def test_simloss_on_gemm_loop(dcl):
    """bf16 problems at least a tile wide run on the production GEMM loop (simloss5.h on g5_run): ragged rows and columns, 2 x 2
    tiles; then 2 x 4 tiles of which (0, 2) is interior and off the diagonal -- the whole-line G epilogue -- over two K steps"""
    K.case_simloss(DEV, torch.bfloat16, 264, 392, 64, dcl, diag_off=100)
    K.case_simloss(DEV, torch.bfloat16, 264, 776, 128, dcl, diag_off=100)


@pytest.mark.parametrize("nq,nk,diag_off", [(520, 1100, 300), (512, 1024, 0), (512, 1024, 384), (300, 700, -100), (256, 768, 5000), (768, 256, 0),
                                            (1030, 520, 512)])
def test_simloss_grad_interior_and_edge_launches(nq, nk, diag_off):
    """G (simloss5.h): every full tile -- with a piece of the positive diagonal or without -- through the spill-free ring-loop kernel,
    the tiles at a ragged edge through the tile list of the second launch -- every tile exactly once (d tau counts each logit once),
    for diagonals that start inside, before or past the columns, ragged rows and / or columns, more rows than columns"""
    K.case_simloss(DEV, torch.bfloat16, nq, nk, 64, True, diag_off=diag_off)
    K.case_simloss(DEV, torch.bfloat16, nq, nk, 64, False, diag_off=diag_off)


def test_simloss_banded_tile_order():
    """more than 8 column tiles: the ring-loop kernels walk the tiles in bands of 4..8 column tiles (12 tiles -> bands of 6; 10 -> 5;
    9 tiles + a ragged one = 10), every tile exactly once (lse / G / d tau all see each logit once)"""
    K.case_simloss(DEV, torch.bfloat16, 512, 3072, 64, True, diag_off=1000)
    K.case_simloss(DEV, torch.bfloat16, 300, 2560, 64, False, diag_off=0)
    K.case_simloss(DEV, torch.bfloat16, 256, 2400, 64, False, diag_off=2100)


@pytest.mark.parametrize("d", [64, 512])
def test_simloss_grad_at_high_temperature(d):
    """exp(tau) = 200 (the reference never clamps its temperature, x_clip.py:574,736): the one-exponential-per-logit form of G must not
    depend on exp(scale - lse) being representable (it overflowed when the reference point was `scale`)"""
    K.case_simloss(DEV, torch.bfloat16, 512, 1024, d, False, diag_off=384, tau=5.3)
    K.case_simloss(DEV, torch.bfloat16, 264, 392, d, True, diag_off=100, tau=5.3)


@pytest.mark.parametrize("dcl", [False, True], ids=["infonce", "dcl"])
@pytest.mark.parametrize("nq,nk,d,off", [(256, 256, 64, 0), (264, 392, 64, 100)])
def test_simloss_grad_with_spread_lse(nq, nk, d, off, dcl):
    """exp(tau) = 200 and four perfectly matched pairs: log-sum-exps 150 apart inside one wave block (NaN with the round-3 G)"""
    K.case_simloss_spread(DEV, torch.bfloat16, nq, nk, d, dcl, diag_off=off)
    # spread > 300: beyond what ANY single reference point bridges -- the wave blocks with matched rows take the two-exponential form
    K.case_simloss_spread(DEV, torch.bfloat16, nq, nk, d, dcl, diag_off=off, temp=400.0)
    if nq <= 264:
        K.case_simloss_spread(DEV, torch.float32, nq, nk, d, dcl, diag_off=off)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("rows,cols,diag_off", [(5, 16, 0), (7, 24, 8)])
def test_simreg_diff(dtype, rows, cols, diag_off):
    K.case_simreg_diff(DEV, dtype, rows, cols, diag_off)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("batch,n,heads", [(2, 5, 1), (1, 33, 2)])
def test_rotary(dtype, batch, n, heads):
    K.case_rotary(DEV, dtype, batch, n, heads)
    K.case_rotary(DEV, dtype, batch, n, heads, hd=128)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("batch,h,C", [(2, 4, 64), (1, 2, 16)])
def test_dwconv(dtype, batch, h, C):
    K.case_dwconv(DEV, dtype, batch, h, C)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("rows,cols,ld", [(5, 24, 24), (9, 1000, 1000), (3, 13, 16), (4200, 24, 24)])
def test_cross_entropy(dtype, rows, cols, ld):
    K.case_cross_entropy(DEV, dtype, rows, cols, ld)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("rows,dim", [(5, 64), (9, 512)])
def test_layernorm_chain(dtype, rows, dim):
    K.case_layernorm_chain(DEV, dtype, rows, dim)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("rows,cols,relu,affine,training,offset", [(130, 512, True, True, True, 0.0), (37, 64, False, False, True, 50.0),
                                                                    (8, 96, True, True, True, 0.0), (50, 256, True, True, False, 0.0),
                                                                    (2, 1024, False, True, True, 0.0)])
def test_batchnorm(dtype, rows, cols, relu, affine, training, offset):
    K.case_batchnorm(DEV, dtype, rows, cols, relu, affine, training, offset)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("rows,dim", [(130, 256), (3, 64), (17, 1024), (4300, 64)])
def test_neg_cosine(dtype, rows, dim):
    K.case_neg_cosine(DEV, dtype, rows, dim)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("rows,dim,temperature", [(20, 32, 0.5), (70, 128, 1.0), (5, 12, 0.3)])
def test_nt_xent(dtype, rows, dim, temperature):
    K.case_nt_xent(DEV, dtype, rows, dim, temperature)


def test_gemm_splitk_uneven_slices(big_gemm_kernels):
    K.case_gemm_splitk_uneven(DEV, M=256, N=256, K=2880)


def test_edge_cases_of_the_row_kernels():
    """zero rows are a no-op (an empty tensor has a null data pointer); NT-Xent with a single pair; a zero prediction row in the SimSiam
    loss follows F.normalize's clamp; causal attention whose leading keys are padding (the first queries see no key: output 0, finite
    gradients -- the reference's softmax over all-masked scores returns the uniform average there, include/xclip.h)"""
    from x_clip_amd import ops
    from x_clip_amd.visual_ssl import _NegCosineFn, nt_xent_loss
    F = torch.nn.functional
    assert tuple(ops.gather_rows(torch.randn(4, 8), torch.zeros(0, dtype=torch.int32)).shape) == (0, 8)
    acc = torch.zeros(1)
    assert ops.cross_entropy_fwd(torch.randn(0, 8), 8, torch.zeros(0, dtype=torch.int64), acc).numel() == 0 and float(acc) == 0.0
    q, k = torch.randn(1, 16, requires_grad=True), torch.randn(1, 16, requires_grad=True)
    loss = nt_xent_loss(q, k, 0.5)                       # one pair: each row's only candidate is its partner
    loss.backward()
    assert float(loss.detach()) == 0.0 and float(q.grad.abs().max()) == 0.0
    p = torch.randn(6, 32)
    p[2] = 0
    p.requires_grad_(True)
    z = torch.randn(6, 32)
    _NegCosineFn.apply(p, z, 1.0 / 6).backward()
    p64 = p.detach().double().requires_grad_(True)
    (2 - 2 * (F.normalize(p64, dim=-1) * F.normalize(z.double(), dim=-1)).sum(-1)).mean().backward()
    torch.testing.assert_close(p.grad.double(), p64.grad, rtol=1e-5, atol=1e-6)
    qkv = K.rnd((2, 40, 3 * 64), torch.bfloat16, 5)
    mask = torch.ones(2, 40, dtype=torch.bool)
    mask[0, :3] = False
    out, lse = ops.attention_fwd(qkv, mask, 1, 0.125, True)
    dq = ops.attention_bwd(qkv, mask, out, torch.ones_like(out), lse, 1, 0.125, True)
    assert torch.isfinite(out.float()).all() and float(out[0, :3].float().abs().max()) == 0.0 and torch.isfinite(dq.float()).all()


def test_gemm6_gemm7_experimental_kernels():
    """measure/gemm6.h / gemm7.h (MEASUREMENT build only, XCLIP_GEMM=6 / 7; measured slower than the ring kernel: DESIGN_APPENDIX.md section 6b) still
    compute the product -- the switch is read once per process, so the check runs in its own interpreter; and the PRODUCT build ignores the
    switch altogether"""
    import subprocess
    import sys
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from x_clip_amd import _lib, ops\n"
        "from emu.build_emu import build\n"
        "_lib._use_library_for_tests(build(measure=True))\n"
        "for (M, N, K, alpha) in [(512, 256, 128, 1.0), (768, 384, 256, 0.5), (256, 4096, 128, 1.0)]:\n"
        "    torch.manual_seed(0)\n"
        "    a = torch.randn(M, K).bfloat16(); b = torch.randn(N, K).bfloat16()\n"
        "    got = ops.gemm(a, b, M, N, K, alpha=alpha).float()\n"
        "    want = alpha * (a.float() @ b.float().t())\n"
        "    assert float((got - want).abs().max()) <= float(want.abs().max()) * 2.0 ** -8, (M, N, K)\n"
        "print('gemm6 ok')\n") % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for gen in ("6", "7"):                                   # 7: gemm7.h, four waves of 128 x 128 (takes the 256-multiples among these)
        env = dict(os.environ, XCLIP_GEMM=gen)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "gemm6 ok" in out.stdout, (gen, out.stderr[-2000:])


def test_attention_streaming_backward_measurement_build():
    """attention6.h (round 6: the attention backward as a stream through a three-slot LDS ring, one persistent work-group per CU, dQ summed in
    fixed point with integer LDS atomics; MEASUREMENT build only, XCLIP_ATTN_BWD=6 -- measured slower than attention5.h) against the fp64
    reference: n = 256 / 257, masks incl. a padded tail key, more heads than emulated CUs (a work-group walks several heads: the ring, the K / V
    image hand-over and the lse / mask registers cross head boundaries); its dQ is bit-reproducible (integer adds commute)"""
    import subprocess
    import sys
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from x_clip_amd import _lib, ops\n"
        "from emu.build_emu import build\n"
        "_lib._use_library_for_tests(build(measure=True))\n"
        "import kernel_cases as K\n"
        "dev = torch.device('cpu')\n"
        "K.case_attention(dev, torch.bfloat16, 2, 256, 1, True)\n"
        "K.case_attention_single_tail(dev, torch.bfloat16)\n"
        "torch.manual_seed(0)\n"
        "qkv = torch.randn(2, 257, 3 * 64).bfloat16(); do = torch.randn(2, 257, 64).bfloat16()\n"
        "out, lse = ops.attention_fwd(qkv, None, 1, 0.125)\n"
        "a = ops.attention_bwd(qkv, None, out, do, lse, 1, 0.125); b = ops.attention_bwd(qkv, None, out, do, lse, 1, 0.125)\n"
        "assert torch.equal(a, b)\n"
        "print('attn6 ok')\n") % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, XCLIP_ATTN_BWD="6"), capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0 and "attn6 ok" in out.stdout, out.stderr[-3000:]


def test_attention_persistent_single_pass_measurement_build():
    """attention7.h (round 6, MEASUREMENT build, XCLIP_ATTN_BWD=7: attention5.h as a persistent kernel, the next head's images requested under
    this head's stores -- measured slower) and attention5.h's round-5 forms (XCLIP_ATTN5_VAR=3: quarter-line stores,
    images before the delta rows): against the fp64 reference, and the same bits as the product form (delta rows first, whole-line stores
    through the wave's spent exchange tile) -- n = 256 / 257, masks, more heads than emulated CUs"""
    import subprocess
    import sys
    import tempfile
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from x_clip_amd import _lib, ops\n"
        "from emu.build_emu import build\n"
        "_lib._use_library_for_tests(build(measure=True))\n"
        "import kernel_cases as K\n"
        "dev = torch.device('cpu')\n"
        "K.case_attention(dev, torch.bfloat16, 2, 256, 1, True)\n"
        "K.case_attention_single_tail(dev, torch.bfloat16)\n"
        "torch.manual_seed(0)\n"
        "qkv = torch.randn(3, 257, 3 * 64).bfloat16(); do = torch.randn(3, 257, 64).bfloat16(); mask = torch.rand(3, 257) > 0.2\n"
        "out, lse = ops.attention_fwd(qkv, mask, 1, 0.125)\n"
        "torch.save(ops.attention_bwd(qkv, mask, out, do, lse, 1, 0.125), sys.argv[1])\n"
        "print('attn ok')\n") % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    with tempfile.TemporaryDirectory() as tmp:
        got = {}
        for name, env in [("product", dict(XCLIP_ATTN_BWD="5")), ("round5", dict(XCLIP_ATTN_BWD="5", XCLIP_ATTN5_VAR="3")), ("persistent", dict(XCLIP_ATTN_BWD="7"))]:
            path = os.path.join(tmp, name + ".pt")
            out = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, **env), capture_output=True, text=True, timeout=1800)
            assert out.returncode == 0 and "attn ok" in out.stdout, (name, out.stderr[-3000:])
            got[name] = torch.load(path)
        assert torch.equal(got["product"], got["round5"]) and torch.equal(got["product"], got["persistent"])


def test_gemm_split_k_whole_slices_per_xcd():
    """weight-gradient (TN) split-K launches run on a 1-D grid whose work-groups place themselves: the (slice, tile) pairs slice-major, one
    contiguous eighth per XCD (gemm2.h g2_where: the work-groups that read the same K range share an L2).  16 slices x 2 tiles, 8 x 8, 16 x 4
    (whole slices per XCD) and 10 slices x 12 tiles (slices shared by two XCDs, 120 pairs = 15 per XCD) against an fp32 product"""
    from x_clip_amd import ops
    torch.manual_seed(0)
    for (M, N, K, slices) in [(512, 256, 4096, 16), (1024, 512, 2048, 8), (512, 512, 4096, 16), (1536, 512, 2624, 10)]:
        a, b = torch.randn(K, M).bfloat16(), torch.randn(K, N).bfloat16()
        assert _lib.lib().xclip_gemm_workspace_bytes(M, N, K, 1) == slices * M * N * 4
        got = ops.gemm(a, b, M, N, K, a_kmajor=True, b_kmajor=True).float()
        want = a.float().t() @ b.float()
        assert float((got - want.bfloat16().float()).abs().max()) <= float(want.abs().max()) * 2.0 ** -7, (M, N, K)


def test_gemm_row_tail_as_split_k():
    """xclip_api.hip gemm2_tail_cut: a persistent launch whose last round would fill only a few CUs cuts its rows at the last whole round and
    runs the row tail as a split-K problem (fp32 slabs + the reduction, which also applies alpha and the skip term).  The policy plans for
    the device's CU count, so the emulator is told it has 4 (its own interpreter: the count is read once): 1280 rows = 5 row tiles x 2
    column tiles = 2 rounds + 2 tiles -> 4 row tiles in the main launch, 1 as slabs; NT / NN, with and without a skip term, in place"""
    import subprocess
    import sys
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from x_clip_amd import _lib, ops\n"
        "from emu.build_emu import build\n"
        "_lib._use_library_for_tests(build())\n"
        "torch.manual_seed(0)\n"
        "L = _lib.lib()\n"
        "assert L.xclip_gemm_workspace_bytes(1280, 512, 1536, 1) == 2 * 256 * 512 * 4        # the tail: 2 tiles x 2 K slices fill the 4 CUs\n"
        "assert L.xclip_gemm_workspace_bytes(1280, 512, 512, 1) == 0                         # short K: a tile is not worth cutting\n"
        "for (M, N, K, bk, res, inplace) in [(1280, 512, 1536, False, False, False), (1280, 512, 2048, False, True, False), (1280, 512, 2048, True, True, True), (1416, 256, 1024, True, False, False)]:\n"
        "    a = torch.randn(M, K).bfloat16(); b = (torch.randn(K, N) if bk else torch.randn(N, K)).bfloat16()\n"
        "    r = torch.randn(M, N).bfloat16() if res else None\n"
        "    want = 0.5 * (a.float() @ (b.float() if bk else b.float().t())) + (r.float() if res else 0)\n"
        "    out = r.clone() if inplace else None\n"
        "    got = ops.gemm(a, b, M, N, K, b_kmajor=bk, alpha=0.5, residual=(out if inplace else r), out=out).float()\n"
        "    assert float((got - want.bfloat16().float()).abs().max()) <= float(want.abs().max()) * 2.0 ** -7, (M, N, K)   # one bf16 ulp at the scale\n"
        "print('tail ok')\n") % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, XCLIP_EMU_CUS="4"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "tail ok" in out.stdout, out.stderr[-2000:]


def test_geglu_layernorm_bwd_grid_smaller_than_its_workspace():
    """xclip_api.hip ln_geglu_bwd_blocks: the GEGLU-LayerNorm backward runs two rounds of six work-groups per CU, fewer than the rows of dg
    partials its workspace is sized for; a one-CU emulator (12 work-groups) walks 203 rows in nine passes with a dead slot in the last one,
    and the fold must read exactly the partial rows that were written"""
    import subprocess
    import sys
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from x_clip_amd import _lib\n"
        "from emu.build_emu import build\n"
        "_lib._use_library_for_tests(build())\n"
        "import kernel_cases as K\n"
        "for dtype in K.DTYPES:\n"
        "    K.case_layernorm(torch.device('cpu'), dtype, 203, 2048, True, False)\n"
        "    K.case_layernorm(torch.device('cpu'), dtype, 50, 1024, True, False)\n"
        "print('grid ok')\n") % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, XCLIP_EMU_CUS="1"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "grid ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("bx,nt,by,ni,d,chunks", [(5, 77, 4, 98, 64, 1), (4, 64, 7, 64, 128, 1), (3, 130, 3, 200, 64, 2), (4, 70, 3, 65, 64, 1),
                                                  (9, 32, 9, 32, 64, 1), (7, 40, 8, 33, 64, 1), (3, 256, 6, 32, 64, 1)])
def test_filip_fused(bx, nt, by, ni, d, chunks):
    """the FILIP forward with its reductions inside the GEMM epilogue: 385 x 392 (2 x 2 tiles, ragged both ways, text / image segments
    cut by wave blocks and tiles), segment = wave block (64), long segments spanning tiles + two image chunks, 280 rows (the second
    row tile's lower wave block lies wholly past M: no text exists there -- an out-of-bounds partial once) x 195 columns (odd); the
    smallest segments (32 tokens a side, aligned: 288 x 288), misaligned short segments (40 x 33: three images per wave block, five
    texts per 128 rows), and the README's FILIP model under patch dropout (256 text tokens x 32 kept patches)"""
    K.case_filip_fused(DEV, bx, nt, by, ni, d, chunks=chunks)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("n,masked,causal,hd", [(33, False, False, 64), (97, True, True, 64), (70, True, False, 128)])
def test_attention_dropout(dtype, n, masked, causal, hd):
    """attention dropout (Attention.dropout, x_clip.py:212,241) in the tiled kernels, forward and backward, against the reference
    arithmetic on the same keep-mask"""
    K.case_attention(DEV, dtype, 2, n, 2, masked, causal=causal, hd=hd, drop=(0.25, 0xC0FFEE1234567))


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
def test_dropout(dtype):
    K.case_dropout(DEV, dtype)


@pytest.mark.parametrize("dtype", K.DTYPES, ids=["fp32", "bf16"])
@pytest.mark.parametrize("batch,n,heads,masked,hd,row,causal", [(2, 37, 2, True, 64, 0, False), (2, 20, 1, False, 128, 0, False), (1, 70, 2, True, 64, 5, True), (3, 9, 3, True, 128, 0, False)])
def test_attention_pool(dtype, batch, n, heads, masked, hd, row, causal):
    """attention for one query row per (sample, head): key counts that are not multiples of the keys per wave-load (8 / 4 / 16), masks with a
    hole and a padded tail, both head-slot widths, a pooled row in the middle under a causal mask"""
    K.case_attention_pool(DEV, dtype, batch, n, heads, masked, hd, row, causal)
