"""CPU end-to-end run of the product `x_clip_amd.CLIP` (host mirror + the C-ABI kernels compiled against the wave64
emulator) on BASELINE config 0 (cfg1: dim 64, depth 2/2, image 64 patch 32, seq 32, batch 4), checked against the
reference-generated golden fixtures and against the fp64 oracle."""
import os
import sys

import pytest
import torch

from x_clip_amd import _lib

sys.path.insert(0, os.path.dirname(__file__))
import clip_cases as C
from x_clip_amd import ops as ops_mod  # noqa: E402
from emu.build_emu import build  # noqa: E402
from oracle import clip_oracle as O  # noqa: E402

DEV = torch.device("cpu")


@pytest.fixture(scope="module", autouse=True)
def emulator_library():
    _lib._use_library_for_tests(build())
    yield
    _lib._use_library_for_tests(None)


# the emulator runs every fixture in 15-20 s; the plain variants whose code paths are a subset of a combined fixture below
# (cfg1_dcl, cfg1_extra_dcl, cfg1_multiview, cfg1_multiview_m3n1, cfg1_filip, cfg1_simreg_extra, cfg1_rotary, cfg1_filip_downsample, cfg1_mlm,
# cfg1_simsiam, cfg1_simclr (4096-wide projector: minutes on the emulator), cfg1_causal) are exercised on the GPU only
# (cfg1_filip_dcl and cfg1_simsiam_mlm_dcl run in test_no_kernel_reads_unwritten_memory below, with poisoned allocations)
@pytest.mark.parametrize("name", ["cfg1_infonce", "cfg1_patchdrop",
                                  "cfg1_simreg_extra_dcl", "cfg1_rotary_dcl_multiview", "cfg1_filip_downsample_extra_dcl", "cfg1_mlm_dcl_multiview", "cfg1_causal_dcl_multiview", "cfg1_wide_heads", "cfg1_wide_heads_rotary_dcl", "cfg1_rotary_narrow24", "cfg1_rotary_narrow16_dcl"])
def test_clip_matches_reference_fixture(name):
    C.case_golden(DEV, name)


def test_clip_bf16_vs_oracle():
    C.case_vs_oracle(DEV, torch.bfloat16, O.CFG1, 4, bf16_loss=1.4e-3)      # (the dim-64 toy model: measured 6.6e-4)


def test_fused_ffn_backward_vs_oracle_and_two_kernel_path():
    import dataclasses
    """round 5: the feed-forward block's net.4 input gradient + net.2 backward as one kernel (csrc/kernels/gemm9.h) takes stacks whose rows and
    hidden width are whole 256-tiles (bf16).  A 128-wide model with 8 x 32 text rows / 8 x 32 image tokens: every non-pooled layer of both towers
    goes through it; against the fp64 oracle, and against the same step with the fusion off (gradients equal up to the rounding of d a)"""
    from x_clip_amd import ops
    cfg = dataclasses.replace(O.CFG1, dim_text=128, dim_image=128, text_seq_len=31, text_enc_depth=3, visual_image_size=128, visual_patch_size=16,
                              visual_enc_depth=2)                      # image: 64 patches per sample, mean-pooled into the CLS slot: 8 x 64 rows
    calls = []
    orig = ops.ffn_dgrad_geglu
    ops.ffn_dgrad_geglu = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        C.case_vs_oracle(DEV, torch.bfloat16, cfg, 8, bf16_loss=1.4e-3)
    finally:
        ops.ffn_dgrad_geglu = orig
    assert len(calls) >= 3, calls                                      # (the text tower's two dense layers + the vision tower's)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_live_rows_vs_oracle(dtype):
    """the size-independent form of the GPU suite's full-size step test, at emulator size: 7 samples, 3 with a live upstream gradient"""
    # (the 64-wide toy latents have elements of ~1/8, a bf16 ulp of 5e-4 to 1e-3: measured 2.6e-3; the 512-wide model holds 1e-3)
    C.case_live_rows(DEV, O.CFG1, 7, [0, 3, 6], dtype, label="cfg1 b=7 (emulator)", bf16_latent_bar=5e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_high_temperature_vs_oracle(dtype):
    """exp(tau) = 200 -- the reference multiplies by exp(tau) without a clamp (x_clip.py:574,736) and exponentiates without subtracting a
    maximum (:826), so in fp32 it is at its own limit there; the product's log-sum-exp forms and one-exponential gradient must not be"""
    import dataclasses
    loose = dict(bf16_cos=0.97, bf16_rel=0.3, bf16_loss=5e-2)   # (bf16 logits of magnitude 100: the loss bar is absolute)
    C.case_vs_oracle(DEV, dtype, O.CFG1, 5, temperature=5.3, **loose)
    C.case_vs_oracle(DEV, dtype, dataclasses.replace(O.CFG1, decoupled_contrastive_learning=True, use_all_token_embeds=True), 5, temperature=5.3, **loose)


def test_narrow_heads_vs_oracle():
    """dim_head below the kernels' 64 (x_clip.py:201-211 accepts any): heads run zero-padded, the scale stays dim_head^-0.5, and the
    gradients of the real to_qkv / to_out weights come back through the padding; rotary with 32-wide heads rotates the whole head,
    with narrower ones all of their features"""
    import dataclasses
    C.case_vs_oracle(DEV, torch.float32, dataclasses.replace(O.CFG1, text_dim_head=32, visual_dim_head=24, text_rotary_pos_emb=True), 4)
    with pytest.raises(NotImplementedError):
        C.build_clip(dataclasses.replace(O.CFG1, text_dim_head=160), {}, DEV, torch.float32)
    # rotary heads narrower than 32 (round 4): min(dim_head, 32) rotated features (x_clip.py:311) -- 16 and 24 (12 pairs: not a 16-byte chunk)
    C.case_vs_oracle(DEV, torch.float32, dataclasses.replace(O.CFG1, text_dim_head=16, text_rotary_pos_emb=True), 4)
    C.case_vs_oracle(DEV, torch.bfloat16, dataclasses.replace(O.CFG1, text_dim_head=24, text_rotary_pos_emb=True), 4, bf16_loss=1.4e-3)
    with pytest.raises(NotImplementedError):
        C.build_clip(dataclasses.replace(O.CFG1, text_dim_head=15, text_rotary_pos_emb=True), {}, DEV, torch.float32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_wide_heads_vs_oracle(dtype):
    """dim_head above 64 (ordinary ViT shapes: 80, 96, 128; x_clip.py:201-212 accepts any): 128-feature head slots = two 64-wide
    halves per head, narrower-than-slot heads zero-padded; rotary on the first 32 features of the 96-wide text heads"""
    import dataclasses
    C.case_vs_oracle(DEV, dtype, dataclasses.replace(O.CFG1, text_dim_head=96, visual_dim_head=128, text_rotary_pos_emb=True, text_heads=2,
                                                     visual_heads=2), 4, bf16_loss=1.4e-3)


def test_bare_transformer_with_rotary_table_and_mask():
    """Transformer.forward(x, rotary_pos_emb = RotaryEmbedding(32)(n), mask) as a user of the reference's building block calls it
    (x_clip.py:274-291,155-166), forward and input gradient against the oracle's stack; a table that is not position x frequency is
    refused"""
    from x_clip_amd.clip import RotaryEmbedding, Transformer
    torch.manual_seed(3)
    dim, depth, heads, b, n = 64, 2, 2, 3, 9
    net = Transformer(dim, depth=depth, heads=heads, dim_head=32)
    net.train()
    x = torch.randn(b, n, dim, requires_grad=True)
    mask = torch.ones(b, n, dtype=torch.bool)
    mask[1, 6:] = False
    table = RotaryEmbedding(32)(n, DEV)
    y = net(x, rotary_pos_emb=table, mask=mask)
    g = torch.randn_like(y)
    y.backward(g)
    sd = {"t." + k: v.detach().double() for k, v in net.state_dict().items()}
    x64 = x.detach().double().requires_grad_(True)
    want = O.transformer(x64, sd, "t.", depth, heads, 32, mask, O.rotary_freqs(n, 32, torch.float64), False)
    want.backward(g.double())
    assert float((y.detach().double() - want.detach()).abs().max()) < 2e-5
    assert float((x.grad.double() - x64.grad).abs().max() / x64.grad.abs().max()) < 2e-5
    with pytest.raises(NotImplementedError):
        net(x, rotary_pos_emb=table * table, mask=mask)


@pytest.mark.parametrize("over", [dict(), dict(text_causal_mask=True, text_eos_id=7), dict(text_rotary_pos_emb=True), dict(use_mlm=True),
                                  dict(extra_latent_projection=True), dict(use_all_token_embeds=True, downsample_image_embeds=True, visual_patch_size=16),
                                  dict(use_visual_ssl=True, ssl_projection_size=32, ssl_projection_hidden_size=64),
                                  dict(use_visual_ssl=True, visual_ssl_type="simclr", ssl_projection_size=32)],
                         ids=["default", "causal", "rotary", "mlm", "extra", "downsample", "simsiam", "simclr"])
def test_state_dict_keys_and_shapes(over):
    """the product's state_dict (keys, shapes, the aliases under which shared towers are listed again) against the map the reference's
    strict load_state_dict pinned in oracle/make_golden.py"""
    import dataclasses
    cfg = dataclasses.replace(O.CFG1, **over)
    model = C.build_clip(cfg, O.make_state_dict(cfg, 1, torch.float32), DEV, torch.float32)     # strict load inside
    shapes = O.state_dict_shapes(cfg)
    sd = model.state_dict()
    assert set(sd) == set(shapes), set(sd) ^ set(shapes)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k


def test_batchnorm_module_semantics():
    """x_clip_amd.visual_ssl.BatchNorm1d against torch.nn.BatchNorm1d: train / eval, running statistics and num_batches_tracked, cumulative
    average (momentum = None), affine = False, buffers cast to bf16"""
    from x_clip_amd.visual_ssl import BatchNorm1d
    g = torch.Generator().manual_seed(3)
    for kw in (dict(), dict(momentum=None), dict(affine=False), dict(momentum=0.3, eps=1e-3)):
        ours, ref = BatchNorm1d(64, **kw), torch.nn.BatchNorm1d(64, **kw).double()
        if kw.get("affine", True):
            with torch.no_grad():
                ours.weight.copy_(1 + 0.1 * torch.randn(64, generator=g))
                ours.bias.copy_(0.1 * torch.randn(64, generator=g))
                ref.weight.copy_(ours.weight.double())
                ref.bias.copy_(ours.bias.double())
        for step in range(3):
            x = torch.randn(10 + step, 64, generator=g) * (1 + step) + step
            y, yr = ours(x), ref(x.double())
            torch.testing.assert_close(y.double(), yr, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(ours.running_mean.double(), ref.running_mean, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(ours.running_var.double(), ref.running_var, rtol=1e-5, atol=1e-6)
        assert int(ours.num_batches_tracked) == int(ref.num_batches_tracked) == 3
        ours.eval(), ref.eval()
        x = torch.randn(5, 64, generator=g)
        torch.testing.assert_close(ours(x).double(), ref(x.double()), rtol=1e-5, atol=1e-5)
        assert int(ours.num_batches_tracked) == 3                      # eval mode leaves the statistics alone
    b16 = BatchNorm1d(64).to(torch.bfloat16)
    b16(torch.randn(12, 64, generator=g).to(torch.bfloat16))
    assert b16.running_mean.dtype == torch.bfloat16 and torch.isfinite(b16.running_var.float()).all() and float(b16.running_mean.float().abs().max()) > 0


def test_eos_pooling_indices():
    """XF.eos_to_front against the oracle's restatement (first eos per row to the front, the rest in order) and its gradient (a row permutation)"""
    from x_clip_amd import functional as XF
    g = torch.Generator().manual_seed(5)
    enc = torch.randn(5, 9, 64, generator=g, requires_grad=True)
    tokens = torch.randint(1, 50, (5, 9), generator=g)
    eos = 77
    for r, pos in enumerate((8, 0, 4, 3, 6)):
        tokens[r, pos] = eos
    tokens[2, 7] = eos                                                   # a second eos later in the row is ignored
    out = XF.eos_to_front(enc, tokens, eos)
    assert torch.equal(out.detach(), O.eos_to_front(enc.detach(), tokens, eos))
    w = torch.randn(5, 9, 64, generator=g)
    (out * w).sum().backward()
    enc2 = enc.detach().clone().requires_grad_(True)
    (O.eos_to_front(enc2, tokens, eos) * w).sum().backward()
    assert torch.equal(enc.grad, enc2.grad)
    with pytest.raises(AssertionError, match="does not have the eos id"):
        XF.eos_to_front(enc.detach(), torch.ones(5, 9, dtype=torch.int64), eos)


def test_no_cpu_fallback_without_library():
    """the product refuses CPU tensors when the real HIP library (not the emulator) is bound"""
    from x_clip_amd import ops
    _lib._use_library_for_tests(None)
    try:
        with pytest.raises(RuntimeError):
            ops.l2norm_fwd(torch.randn(4, 64))
    finally:
        _lib._use_library_for_tests(build())


def test_short_and_fully_padded_text():
    C.case_short_and_padded_text(DEV)


def test_freeze_and_early_returns():
    C.case_freeze_and_early_returns(DEV, O.CFG1, batch=4)


def test_pluggable_encoders_head_only():
    C.case_pluggable_encoders_head_only(DEV)


def test_filip_odd_batch_and_token_counts():
    """fine-grained head where (images x image tokens) is not a whole 16-byte chunk of similarity columns: 5 images x 9 patches = 45
    (fp32 chunk 4, bf16 chunk 8) -- the reference runs these shapes, the token-similarity GEMM needs its N / K padded"""
    import dataclasses
    cfg = dataclasses.replace(O.CFG1, use_all_token_embeds=True, visual_image_size=96)
    C.case_vs_oracle(DEV, torch.float32, cfg, 5)
    C.case_vs_oracle(DEV, torch.bfloat16, cfg, 5, bf16_cos=0.98, bf16_rel=0.25, bf16_loss=1.4e-3)   # measured: rel 0.141, cosine 0.990 (arg-max ties under bf16 scores); dim-64 toy model: loss as test_clip_bf16_vs_oracle


def test_backward_twice_and_inplace_edits_fail_loudly():
    """the activation tape is released while the backward walks it: a second backward says so; parameters are registered with
    autograd, so an in-place edit between forward and backward trips its version check instead of giving silently wrong gradients"""
    from x_clip_amd import CLIP
    m = CLIP(**O.CFG1.ctor_kwargs(), visual_patch_dropout=0.0).train()
    text, image, _, _ = O.make_inputs(O.CFG1, 4, 3)
    loss = m(text, image.float(), return_loss=True)
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="already back-propagated"):
        loss.backward()
    m.zero_grad()
    loss = m(text, image.float(), return_loss=True)
    with torch.no_grad():
        m.text_transformer.transformer.layers[0][0].fn.to_qkv.weight.mul_(1.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        loss.backward()


def test_no_grad_pass_keeps_no_tape():
    """inference / frozen-tower passes must not hold the training tape (needs_input_grad is True for parameters even under no_grad)"""
    from x_clip_amd import functional as XF
    kept = []
    orig = XF.stack_forward

    def spy(*a, **k):
        kept.append(k.get("keep_tape", True))
        return orig(*a, **k)
    XF.stack_forward = spy
    try:
        from x_clip_amd import CLIP
        m = CLIP(**O.CFG1.ctor_kwargs(), visual_patch_dropout=0.0).train()
        text, image, _, _ = O.make_inputs(O.CFG1, 4, 3)
        with torch.no_grad():
            m(text, image.float(), return_latents=True)
        assert kept == [False, False], kept
        kept.clear()
        m(text, image.float(), return_loss=True, freeze_image_encoder=True)
        assert sorted(kept) == [False, True], kept
    finally:
        XF.stack_forward = orig


def test_text_micro_batches_same_result():
    """CLIP.text_micro_batches = 2: the text batch goes through the tower in two slices (on the GPU: on two streams); same loss
    (encoders are row independent), gradients equal up to the order of the weight-gradient sums"""
    from x_clip_amd import CLIP
    torch.manual_seed(4)
    a = CLIP(**O.CFG1.ctor_kwargs(), visual_patch_dropout=0.0).train()
    b = CLIP(**O.CFG1.ctor_kwargs(), visual_patch_dropout=0.0).train()
    b.load_state_dict(a.state_dict())
    b.text_micro_batches, b._micro_batch_min_rows = 2, 1
    text, image, _, _ = O.make_inputs(O.CFG1, 4, 9)
    la, lb = a(text, image.float(), return_loss=True), b(text, image.float(), return_loss=True)
    la.backward(), lb.backward()
    assert abs(float(la.detach()) - float(lb.detach())) < 1e-6
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if pa.grad is None:
            assert pb.grad is None, k
        else:
            torch.testing.assert_close(pa.grad, pb.grad, rtol=2e-4, atol=1e-6, msg=k)


def test_image_micro_batches_same_result():
    """CLIP.image_micro_batches = 2 (the memory knob of the ViT-L line): the image batch, with an injected PatchDropout draw and an
    augmented view, goes through the vision tower in two sequential slices; same loss, same gradients up to summation order"""
    from x_clip_amd import CLIP
    torch.manual_seed(4)
    kw = dict(O.CFG1.ctor_kwargs(), visual_image_size=128)                     # 16 patches, 8 kept
    a = CLIP(**kw, visual_patch_dropout=0.5, checkpoint_during_training=True).train()
    b = CLIP(**kw, visual_patch_dropout=0.5, checkpoint_during_training=True).train()
    b.load_state_dict(a.state_dict())
    b.image_micro_batches = 2
    cfg = O.ClipConfig(**{k: v for k, v in kw.items()})
    text, image, aug_t, aug_i = O.make_inputs(cfg, 4, 9, 0, 1)
    keep = torch.randn(8, 16, generator=torch.Generator().manual_seed(3)).topk(8, dim=-1).indices.to(torch.int32)
    losses = []
    for m in (a, b):
        m.visual_transformer.keep_indices_override = keep
        l = m(text, image.float(), return_loss=True, aug_image=[aug_i[0].float()])
        l.backward()
        losses.append(float(l.detach()))
    assert abs(losses[0] - losses[1]) < 1e-6
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if pa.grad is None:
            assert pb.grad is None, k
        else:
            torch.testing.assert_close(pa.grad, pb.grad, rtol=2e-4, atol=1e-6, msg=k)


def test_no_kernel_reads_unwritten_memory():
    """reference fixtures again (FILIP head; SimSiam + MLM side losses) with every torch.empty the product makes poisoned with NaN"""
    with C.poisoned_empty():
        C.case_golden(DEV, "cfg1_filip_dcl")
        C.case_golden(DEV, "cfg1_simsiam_mlm_dcl")


def test_filip_fused_path_vs_oracle_and_chunked():
    """a FILIP configuration whose shape takes the fused forward (64 image tokens, 70 text tokens, bf16): against the fp64 oracle, and
    against the same model with the fused path switched off (chunked similarities + reduction passes) -- the two forwards choose their
    arg-max tokens from fp32 / bf16-rounded scores respectively, so the losses agree to bf16 precision, not bit for bit"""
    import dataclasses
    from x_clip_amd import losses
    cfg = dataclasses.replace(O.CFG1, use_all_token_embeds=True, visual_image_size=256, text_seq_len=70, text_enc_depth=1, visual_enc_depth=1,
                              decoupled_contrastive_learning=True)
    calls = []
    orig = ops_mod.filip_fused_fwd
    ops_mod.filip_fused_fwd = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        l_fused = C.case_vs_oracle(DEV, torch.bfloat16, cfg, 3, n_aug_text=1, bf16_cos=0.98, bf16_rel=0.25, bf16_loss=1.4e-3)
        assert len(calls) >= 2, "the fused FILIP forward was not taken"
        n = len(calls)
        losses.FILIP_FUSED = False
        l_chunk = C.case_vs_oracle(DEV, torch.bfloat16, cfg, 3, n_aug_text=1, bf16_cos=0.98, bf16_rel=0.25, bf16_loss=1.4e-3)
        assert len(calls) == n
    finally:
        losses.FILIP_FUSED = True
        ops_mod.filip_fused_fwd = orig
    assert abs(l_fused - l_chunk) < 2e-3, (l_fused, l_chunk)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_transformer_dropout_vs_oracle(dtype):
    """Transformer(attn_dropout, ff_dropout) (x_clip.py:185-212,241,247-291) in training mode: forward and every gradient against the
    oracle evaluating the reference's dropout arithmetic on the product's keep-masks (rebuilt on the host from the pass's seed);
    activation checkpointing re-runs the layer with the same masks (bit-identical gradients); eval mode applies no dropout"""
    from x_clip_amd import functional as XF
    from x_clip_amd.clip import Transformer
    torch.manual_seed(5)
    dim, depth, heads, dh, b, n = 64, 2, 2, 32, 3, 20
    net = Transformer(dim, depth=depth, heads=heads, dim_head=dh, attn_dropout=0.2, ff_dropout=0.1).to(dtype).train()
    ck = Transformer(dim, depth=depth, heads=heads, dim_head=dh, attn_dropout=0.2, ff_dropout=0.1, checkpoint_during_training=True).to(dtype).train()
    ck.load_state_dict(net.state_dict())
    x = torch.randn(b, n, dim).to(dtype)
    mask = torch.ones(b, n, dtype=torch.bool)
    mask[1, 13:] = False
    g = torch.randn(b, n, dim).to(dtype)
    seed = 0x2545F4914F6CDD1
    orig = XF._draw_seed
    XF._draw_seed = lambda: seed
    try:
        outs = []
        for m in (net, ck):
            xi = x.clone().requires_grad_(True)
            y = m(xi, mask=mask)
            y.backward(g)
            outs.append((y.detach(), xi.grad, {k: p.grad.clone() for k, p in m.named_parameters()}))
    finally:
        XF._draw_seed = orig
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for k in outs[0][2]:
        if k.endswith(".g"):
            torch.testing.assert_close(outs[0][2][k], outs[1][2][k], rtol=1e-4, atol=1e-6)      # fp32 atomics: order varies
        else:
            assert torch.equal(outs[0][2][k], outs[1][2][k]), k
    sd = {"t." + k: v.detach().double().requires_grad_(True) for k, v in net.state_dict().items()}
    x64 = x.double().requires_grad_(True)
    fp32 = dtype == torch.float32
    with O.layer_norm_eps(1e-5 if fp32 else 1e-3):
        ref = O.transformer(x64, sd, "t.", depth, heads, dh, mask, dropout=(0.2, 0.1, seed))
        ref.backward(g.double())
    y, dx, grads = outs[0]
    tol = 2e-5 if fp32 else 3e-2
    assert float((y.double() - ref.detach()).abs().max()) < tol * float(ref.detach().abs().max()), float((y.double() - ref.detach()).abs().max())
    assert float((dx.double() - x64.grad).norm() / x64.grad.norm()) < (2e-4 if fp32 else 5e-2)
    for k, gr in grads.items():
        rg = sd["t." + k].grad
        rel = float((gr.double() - rg).norm() / rg.norm())
        assert rel < (2e-4 if fp32 else 8e-2), (k, rel)
    # without the masks the same oracle is far away: the dropout is really applied
    with O.layer_norm_eps(1e-5 if fp32 else 1e-3):
        plain = O.transformer(x.double(), {k: v.detach() for k, v in sd.items()}, "t.", depth, heads, dh, mask)
    assert float((y.double() - plain).abs().max()) > 0.05
    net.eval()
    with torch.no_grad():
        ye = net(x, mask=mask)
    assert float((ye.double() - plain).abs().max()) < (1e-4 if fp32 else 6e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_pruned_text_rows_equal_dense_last_layer(dtype):
    """the CLS head reads one row of the text encoding: the tower's last layer runs its row-wise part on that row alone (CLIP.prune_unused_rows);
    plain, DCL + CLOOB projections with an augmented text view under activation checkpointing, two text slices, rotary, frozen text tower"""
    import dataclasses
    C.case_pruned_rows_equal_dense(DEV, dtype, O.CFG1, 5)
    C.case_pruned_rows_equal_dense(DEV, dtype, dataclasses.replace(O.CFG1, decoupled_contrastive_learning=True, extra_latent_projection=True), 4, n_aug_text=1,
                                   checkpoint=True)
    C.case_pruned_rows_equal_dense(DEV, dtype, dataclasses.replace(O.CFG1, text_rotary_pos_emb=True), 6, micro=2)
    C.case_pruned_rows_equal_dense(DEV, dtype, O.CFG1, 4, freeze_text=True)


def test_pruned_text_rows_inference_paths():
    """the early returns (x_clip.py:697-746) with the text tower asked for its CLS row: latents and similarities under torch.no_grad() in eval mode
    equal the dense tower's; return_encodings still hands back every row"""
    cfg = O.CFG1
    sd = O.make_state_dict(cfg, 41, torch.float32)
    text, image, _, _ = O.make_inputs(cfg, 3, 42)
    outs = []
    for prune in (True, False):
        model = C.build_clip(cfg, sd, DEV, torch.float32)
        model.prune_unused_rows = prune
        model.eval()
        with torch.no_grad():
            tl, il = model(text, image.float(), return_latents=True)[:2]
            sim = model(text, image.float())
            enc_t, enc_i = model(text, image.float(), return_encodings=True)
        assert enc_t.shape == (3, cfg.text_seq_len + 1, cfg.dim_text) and enc_i.ndim == 3
        outs.append((tl, il, sim, enc_t))
    for a, b in zip(*outs):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
