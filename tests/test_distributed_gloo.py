"""World-size-2 run of the product on CPU (gloo backend, kernels from the wave64 emulator build): the rank-sharded loss
must reproduce the reference's distributed semantics recorded in tests/golden/dist2_*.json -- every rank's loss equals the
single-process global-batch loss, the rank-summed parameter gradients equal the single-process gradients (temperature:
every rank holds the full gradient) -- and the public `all_gather` keeps the reference contract.  The same workers run on
the MI355X in tests/test_distributed_gpu.py."""
import dataclasses
import json
import os
import sys

import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import dist_cases as D  # noqa: E402


@pytest.mark.parametrize("name,sizes", [("dist2_infonce", [5, 3]), ("dist2_dcl", [5, 3]), ("dist2_simreg_extra", [5, 3])])
def test_two_ranks_match_reference_semantics(name, sizes, tmp_path):
    with open(os.path.join(HERE, "golden", name + ".json")) as f:
        rec = json.load(f)
    if sum(sizes) != sum(rec["sizes"]):
        pytest.skip("fixture batch differs")
    port = D.free_port()
    mp.spawn(D.worker_fixture, args=(2, port, name, sizes, str(tmp_path)), nprocs=2, join=True)
    D.check_fixture(str(tmp_path), name)


def test_two_ranks_even_batches_gradsync_vs_oracle(tmp_path):
    """aligned per-rank batches (chunked G path), CLOOB extra projections + DCL + one augmented text view, and the bucketed
    gradient all-reduce: after GradSync every rank holds (1/W) x the single-process global-batch gradient -- what
    DDP-mean gives the reference."""
    from oracle import clip_oracle as O
    cfg = dataclasses.replace(O.CFG1, decoupled_contrastive_learning=True, extra_latent_projection=True)
    batch, world = 8, 2
    port = D.free_port()
    mp.spawn(D.worker_even, args=(world, port, dataclasses.asdict(cfg), batch, str(tmp_path)), nprocs=world, join=True)
    D.check_even(str(tmp_path), cfg, batch, world)


@pytest.mark.parametrize("dcl", [False, True])
def test_two_ranks_filip_vs_oracle(tmp_path, dcl):
    """fine-grained (FILIP) head across 2 ranks -- a configuration the reference cannot run (torch.stack of text and image
    latents): every rank's loss = single-process global-batch oracle loss, rank-summed gradients = oracle gradients"""
    from oracle import clip_oracle as O
    cfg = dataclasses.replace(O.CFG1, use_all_token_embeds=True, decoupled_contrastive_learning=dcl)
    batch, world = 4, 2
    port = D.free_port()
    mp.spawn(D.worker_filip, args=(world, port, dataclasses.asdict(cfg), batch, str(tmp_path)), nprocs=world, join=True)
    D.check_filip(str(tmp_path), cfg, batch, world)


# ---- more than two ranks, ragged per-rank batches: the peer-chunk loops of the loss kernels iterate W - 1 times,
# the padded-on-the-wire gathers carry different sizes per rank, one rank holds a single sample ----------------------------------------
RAGGED = {
    # name: (world sizes, config overrides, augmented text views, augmented image views, GradSync)
    "w2_dcl_gradsync": ([5, 3], dict(decoupled_contrastive_learning=True), 1, 0, True),
    "w4_dcl": ([3, 1, 4, 2], dict(decoupled_contrastive_learning=True), 0, 0, False),
    "w4_simreg_extra_dcl": ([3, 1, 4, 2], dict(decoupled_contrastive_learning=True, extra_latent_projection=True, sim_reg_loss_weight=0.5), 0, 0, False),
    "w4_multiview_m3n2": ([2, 3, 1, 2], dict(), 2, 1, False),
    "w4_filip_dcl": ([2, 1, 3, 2], dict(use_all_token_embeds=True, decoupled_contrastive_learning=True), 0, 0, False),
    "w4_filip_multiview_m2n3": ([2, 1, 2, 1], dict(use_all_token_embeds=True), 1, 2, False),
    "w8_infonce_gradsync": ([2, 1, 3, 2, 1, 2, 4, 1], dict(), 0, 0, True),
    "w8_dcl_extra_multiview_m2n2_gradsync": ([1, 2, 1, 3, 2, 1, 1, 2], dict(decoupled_contrastive_learning=True, extra_latent_projection=True), 1, 1, True),
    "w8_filip": ([1, 2, 1, 1, 2, 1, 1, 1], dict(use_all_token_embeds=True), 0, 0, False),
}


@pytest.mark.parametrize("name", list(RAGGED))
def test_many_ranks_ragged_vs_oracle(tmp_path, name):
    from oracle import clip_oracle as O
    sizes, over, n_t, n_i, gs = RAGGED[name]
    cfg = dataclasses.replace(O.CFG1, **over)
    world = len(sizes)
    port = D.free_port()
    mp.spawn(D.worker_ragged, args=(world, port, dataclasses.asdict(cfg), sizes, str(tmp_path), "cpu", n_t, n_i, gs), nprocs=world, join=True)
    D.check_ragged(str(tmp_path), cfg, sizes, n_t, n_i, gs)


def test_gradsync_with_a_frozen_tower(tmp_path):
    """freeze_text_encoder = True under GradSync (3 ranks, ragged): the text tower's bucket has no gradient on any rank -- its slices go
    onto the wire as zeros and `.grad` stays None -- while the other buckets reduce as usual (a bucket whose hook count never completes
    is flushed by finish(); every rank still issues the same collectives of the same size)"""
    from oracle import clip_oracle as O
    cfg = dataclasses.replace(O.CFG1, decoupled_contrastive_learning=True)
    sizes = [2, 3, 1]
    port = D.free_port()
    mp.spawn(D.worker_ragged, args=(3, port, dataclasses.asdict(cfg), sizes, str(tmp_path), "cpu", 0, 0, True, "float32", True), nprocs=3, join=True)
    D.check_ragged(str(tmp_path), cfg, sizes, gradsync=True, freeze_text=True)


def test_two_ranks_filip_fused_forward_bf16(tmp_path):
    """the rank-sharded FILIP head on a shape that takes the fused forward (64 image tokens, 70 text tokens, bf16; filip5.h) in both of
    its uses: the row block (local texts x all images) and, in the backward, the column block (all texts x local images); ragged 3 + 2"""
    from oracle import clip_oracle as O
    import torch
    cfg = dataclasses.replace(O.CFG1, use_all_token_embeds=True, visual_image_size=256, text_seq_len=70, text_enc_depth=1, visual_enc_depth=1)
    sizes = [3, 2]
    port = D.free_port()
    mp.spawn(D.worker_ragged, args=(2, port, dataclasses.asdict(cfg), sizes, str(tmp_path), "cpu", 0, 0, False, "bfloat16"), nprocs=2, join=True)
    # (bf16 FILIP bars of the single-process toy-model test: arg-max ties under bf16 scores)
    worst = D.check_ragged(str(tmp_path), cfg, sizes, dtype=torch.bfloat16, rel_bar=0.25, loss_bar=1.4e-3)
    print("worst gradient relative error (2 ranks, fused FILIP, bf16):", worst)


def test_gradsync_tower_walked_twice_per_step(tmp_path):
    """image_micro_batches = 2 under GradSync (2 ragged ranks, 3 steps): every vision-tower parameter's hook fires twice per step, so
    "all parameters fired once" must not launch the bucket -- the first step reduces in finish() and learns the count, the later steps
    launch from the hook that completes it; gradients = (1 / W) x the oracle's"""
    from oracle import clip_oracle as O
    cfg = dataclasses.replace(O.CFG1, decoupled_contrastive_learning=True)
    sizes = [4, 2]
    port = D.free_port()
    mp.spawn(D.worker_ragged, args=(2, port, dataclasses.asdict(cfg), sizes, str(tmp_path), "cpu", 0, 0, True, "float32", False, 2), nprocs=2, join=True)
    D.check_ragged(str(tmp_path), cfg, sizes, gradsync=True)


def test_gradsync_fp32_wire_under_bf16_parameters_and_two_sinks(tmp_path):
    """GradSync(reduce_dtype=float32) on a bf16 model: bf16 gradients are cast into fp32 slices, summed in fp32, and come back into `.grad`
    in the parameters' dtype; three steps (the later ones launch from the hooks).  Two more GradSync objects of other modules are alive in
    the same process: the grad sinks are a list, none displaces another"""
    from oracle import clip_oracle as O
    import torch
    cfg = dataclasses.replace(O.CFG1, decoupled_contrastive_learning=True)
    port = D.free_port()
    mp.spawn(D.worker_even, args=(2, port, dataclasses.asdict(cfg), 4, str(tmp_path), "cpu", "bfloat16", None, "gloo", 3, False, "float32", True),
             nprocs=2, join=True)
    D.check_even(str(tmp_path), cfg, 4, 2, dtype=torch.bfloat16, rel_bar=0.2, loss_bar=2e-3, cos_bar=0.98)


def test_gradsync_in_place_with_two_sinks(tmp_path):
    """same wire dtype as the parameters (fp32): every weight-gradient GEMM still writes into its bucket slice with two other sinks alive"""
    from oracle import clip_oracle as O
    cfg = dataclasses.replace(O.CFG1, decoupled_contrastive_learning=True, extra_latent_projection=True)
    port = D.free_port()
    mp.spawn(D.worker_even, args=(2, port, dataclasses.asdict(cfg), 8, str(tmp_path), "cpu", "float32", None, "gloo", 3, False, None, True), nprocs=2, join=True)
    D.check_even(str(tmp_path), cfg, 8, 2)


def test_gradsync_ranks_that_disagree_fall_back_together(tmp_path):
    """if the ranks walk their towers differently (here: rank 1 freezes its text tower), launching buckets from the hooks pairs flat buffers of
    different buckets (NCCL: a hang or silent corruption).  The first step's counts and completion order are compared across the ranks;
    on disagreement every rank turns the overlap off and reduces in finish(), in index order -- the gradients stay right"""
    from oracle import clip_oracle as O
    cfg = dataclasses.replace(O.CFG1, decoupled_contrastive_learning=True)
    port = D.free_port()
    mp.spawn(D.worker_disagreeing_ranks, args=(2, port, dataclasses.asdict(cfg), 4, str(tmp_path)), nprocs=2, join=True)
    # (rank 1's frozen text tower contributed zeros: the vision side of the model is what both ranks differentiated)
    D.check_even(str(tmp_path), cfg, 4, 2, only_prefix=("visual_transformer.", "to_visual_latent", "temperature"))


# ---- round 6: layer-sized gradient buckets and the announced unfreeze ---------------------------------
@pytest.mark.parametrize("steps", [3])
def test_gradsync_layer_sized_buckets_two_ranks(tmp_path, steps):
    """the toy towers cut into many small buckets (bucket_bytes = 48 KB: embeddings, each layer, the projections): three steps, the later
    ones launch every bucket from the hook that completes it in the agreed order (last layer first); gradients = (1 / W) x the oracle's,
    the same bits on both ranks; both ranks agreed on one launch order and issued the same number of collectives"""
    from oracle import clip_oracle as O
    import torch
    cfg = dataclasses.replace(O.CFG1, decoupled_contrastive_learning=True, extra_latent_projection=True)
    port = D.free_port()
    mp.spawn(D.worker_even, args=(2, port, dataclasses.asdict(cfg), 8, str(tmp_path), "cpu", "float32", None, "gloo", steps, False, None, False, 48 << 10),
             nprocs=2, join=True)
    D.check_even(str(tmp_path), cfg, 8, 2)
    outs = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"), weights_only=False) for r in range(2)]
    assert outs[0]["buckets"] == outs[1]["buckets"] >= 5
    assert outs[0]["order"] == outs[1]["order"] and outs[0]["launched"] == outs[1]["launched"] == steps * outs[0]["buckets"]
    assert outs[0]["overlap"] and outs[1]["overlap"]


def test_gradsync_layer_sized_buckets_eight_ragged_ranks(tmp_path):
    """eight ragged ranks, DCL + extra projections + multiview, small buckets, GradSync over three steps"""
    from oracle import clip_oracle as O
    sizes, over, n_t, n_i, gs = RAGGED["w8_dcl_extra_multiview_m2n2_gradsync"]
    cfg = dataclasses.replace(O.CFG1, **over)
    port = D.free_port()
    mp.spawn(D.worker_ragged, args=(8, port, dataclasses.asdict(cfg), sizes, str(tmp_path), "cpu", n_t, n_i, True, "float32", False, 1, 48 << 10),
             nprocs=8, join=True)
    D.check_ragged(str(tmp_path), cfg, sizes, n_t, n_i, True)


def test_gradsync_partition_of_the_default_model():
    """the default architecture in bf16 at the default bucket size: one bucket per transformer layer (8.4 MB), the embeddings apart, the
    small top-level parameters last; every trainable parameter exactly once; nothing above 2 x bucket_bytes"""
    import torch
    from x_clip_amd import CLIP
    from x_clip_amd.distributed import GradSync
    model = CLIP().to(torch.bfloat16)
    buckets = GradSync._partition(model, 12 << 20)
    ids = [id(p) for b in buckets for p in b]
    want = [id(p) for p in model.parameters() if p.requires_grad]
    assert len(ids) == len(set(ids)) and set(ids) == set(want)
    sizes = [sum(p.numel() * 2 for p in b) for b in buckets]
    assert max(sizes) <= 24 << 20, sizes
    assert len(buckets) >= 14, sizes                      # 6 + 6 layers, the embeddings / patch embedding, the projections
    layer_sized = [s for s in sizes if 8 << 20 <= s <= 9 << 20]
    assert len(layer_sized) >= 10, sizes


def test_gradsync_unfreeze_needs_rearm(tmp_path):
    from oracle import clip_oracle as O
    cfg = dataclasses.replace(O.CFG1, decoupled_contrastive_learning=True)
    port = D.free_port()
    mp.spawn(D.worker_unfreeze, args=(2, port, dataclasses.asdict(cfg), 4, str(tmp_path), "cpu", 48 << 10), nprocs=2, join=True)
    D.check_even(str(tmp_path), cfg, 4, 2)
