"""The multi-rank path on a real MI355X: two processes share cuda:0, every kernel is libxclip_hip.so, the gathers / side streams /
gradient hooks run on real HIP streams (what the CPU emulator run cannot see: ordering of the collective against the vision side
stream, the weight-gradient side stream and workspace reuse).  gloo carries the collectives (it stages device tensors through the
host), because RCCL refuses two ranks on one device -- `test_rccl_two_ranks_one_device_probe` records what it says.  Same workers and
same checks as tests/test_distributed_gloo.py; every worker additionally runs with NaN-poisoned torch.empty."""
import dataclasses
import json
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import dist_cases as D  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def poisoned_workers(monkeypatch):
    monkeypatch.setenv("XCLIP_TEST_POISON", "1")


@pytest.mark.parametrize("name,sizes", [("dist2_infonce", [5, 3]), ("dist2_dcl", [5, 3]), ("dist2_simreg_extra", [5, 3])])
def test_two_ranks_on_gpu_match_reference_semantics(name, sizes, tmp_path):
    port = D.free_port()
    mp.spawn(D.worker_fixture, args=(2, port, name, sizes, str(tmp_path), "cuda"), nprocs=2, join=True)
    D.check_fixture(str(tmp_path), name)


def test_two_ranks_on_gpu_even_batches_gradsync_vs_oracle(tmp_path):
    from oracle import clip_oracle as O
    cfg = dataclasses.replace(O.CFG1, decoupled_contrastive_learning=True, extra_latent_projection=True)
    port = D.free_port()
    mp.spawn(D.worker_even, args=(2, port, dataclasses.asdict(cfg), 8, str(tmp_path), "cuda"), nprocs=2, join=True)
    D.check_even(str(tmp_path), cfg, 8, 2)


def test_two_ranks_on_gpu_mid_bf16_overlap_streams(tmp_path):
    """a model large enough for the fast kernels and the side streams to matter (dim 512, 2 layers, 71 text positions, patch dropout),
    bf16, DCL + an augmented text view, GradSync hooks firing under the wgrad side stream: vs the fp64 oracle of the global batch"""
    from oracle import clip_oracle as O
    cfg = O.ClipConfig(dim_text=512, dim_image=512, dim_latent=512, num_text_tokens=2000, text_enc_depth=2, text_seq_len=70,
                       text_heads=8, visual_enc_depth=2, visual_image_size=128, visual_patch_size=32, visual_heads=8,
                       decoupled_contrastive_learning=True)
    port = D.free_port()
    mp.spawn(D.worker_even, args=(2, port, dataclasses.asdict(cfg), 16, str(tmp_path), "cuda", "bfloat16", 8), nprocs=2, join=True)
    worst = D.check_even(str(tmp_path), cfg, 16, 2, dtype=torch.bfloat16, patch_keep=8, rel_bar=0.08, loss_bar=3e-4, cos_bar=0.999)      # the single-process bf16 bars (clip_cases.case_vs_oracle)
    print("worst gradient relative error (2 ranks, bf16):", worst)


@pytest.mark.parametrize("dcl", [False, True])
def test_two_ranks_on_gpu_filip_vs_oracle(tmp_path, dcl):
    from oracle import clip_oracle as O
    cfg = dataclasses.replace(O.CFG1, use_all_token_embeds=True, decoupled_contrastive_learning=dcl)
    port = D.free_port()
    mp.spawn(D.worker_filip, args=(2, port, dataclasses.asdict(cfg), 4, str(tmp_path), "cuda"), nprocs=2, join=True)
    D.check_filip(str(tmp_path), cfg, 4, 2)


@pytest.mark.parametrize("name", ["w4_dcl", "w4_simreg_extra_dcl", "w4_multiview_m3n2", "w4_filip_dcl", "w8_dcl_extra_multiview_m2n2_gradsync"])
def test_many_ranks_on_gpu_ragged_vs_oracle(tmp_path, name):
    """four / eight processes on cuda:0 (libxclip_hip.so, real HIP streams, gloo carrying the bytes), ragged per-rank batches: the
    peer-chunk loops run W - 1 times, GradSync reduces persistent flat buckets whose slices the weight-gradient GEMMs wrote in place"""
    from oracle import clip_oracle as O
    from test_distributed_gloo import RAGGED
    sizes, over, n_t, n_i, gs = RAGGED[name]
    cfg = dataclasses.replace(O.CFG1, **over)
    world = len(sizes)
    port = D.free_port()
    mp.spawn(D.worker_ragged, args=(world, port, dataclasses.asdict(cfg), sizes, str(tmp_path), "cuda", n_t, n_i, gs), nprocs=world, join=True)
    D.check_ragged(str(tmp_path), cfg, sizes, n_t, n_i, gs)


def test_four_ranks_on_gpu_mid_bf16_gradsync(tmp_path):
    """the dim-512 bf16 model of the 2-rank test on FOUR ranks (aligned batches: the chunked G path with three peer chunks)"""
    from oracle import clip_oracle as O
    cfg = O.ClipConfig(dim_text=512, dim_image=512, dim_latent=512, num_text_tokens=2000, text_enc_depth=2, text_seq_len=70,
                       text_heads=8, visual_enc_depth=2, visual_image_size=128, visual_patch_size=32, visual_heads=8,
                       decoupled_contrastive_learning=True)
    port = D.free_port()
    mp.spawn(D.worker_even, args=(4, port, dataclasses.asdict(cfg), 8, str(tmp_path), "cuda", "bfloat16", 8), nprocs=4, join=True)
    # (cosine 0.995: the patch-embedding bias gradient is a column sum of cancelling terms; here it is additionally summed over four
    #  ranks in bf16 by the all-reduce -- the 2-rank test above holds 0.999)
    worst = D.check_even(str(tmp_path), cfg, 8, 4, dtype=torch.bfloat16, patch_keep=8, rel_bar=0.08, loss_bar=3e-4, cos_bar=0.995)
    print("worst gradient relative error (4 ranks, bf16):", worst)


MID = dict(dim_text=512, dim_image=512, dim_latent=512, num_text_tokens=2000, text_enc_depth=2, text_seq_len=70, text_heads=8,
           visual_enc_depth=2, visual_image_size=128, visual_patch_size=32, visual_heads=8, decoupled_contrastive_learning=True)


def test_rccl_world_size_one_head_and_gradsync(tmp_path):
    """RCCL ITSELF on the one GPU of the box: a world-size-1 `nccl` process group, the rank-sharded head forced on
    (all-gather of the latents, log-sum-exp gather, scalar all-reduces) and GradSync(overlap=True) over persistent buckets -- bf16, dim 512,
    vision tower and weight gradients on their side streams, three steps (the later ones launch buckets from the hooks).  With one rank the
    collectives are copies, but they run on RCCL's own stream, `Work.wait()` has real stream semantics, and every event edge GradSync /
    GatheredViews record is exercised against the backend they were written for; NaN-poisoned allocations make a missing edge visible.
    Result = the fp64 oracle's single-process loss and gradients (the single-process bf16 bars)."""
    from oracle import clip_oracle as O
    cfg = O.ClipConfig(**MID)
    port = D.free_port()
    mp.spawn(D.worker_even, args=(1, port, dataclasses.asdict(cfg), 16, str(tmp_path), "cuda", "bfloat16", 8, "nccl", 3, True), nprocs=1, join=True)
    worst = D.check_even(str(tmp_path), cfg, 16, 1, dtype=torch.bfloat16, patch_keep=8, rel_bar=0.08, loss_bar=3e-4, cos_bar=0.999)
    print("worst gradient relative error (world-size-1 nccl, bf16):", worst)


def test_eight_ranks_on_gpu_mid_bf16_gradsync_wire_dtype(tmp_path):
    """the dim-512 bf16 model on EIGHT ranks: what the bucket all-reduce costs in accuracy when the wire is
    bf16 (the running sum is rounded at every hop) against GradSync(reduce_dtype=float32).  The measured worst gradient error and cosine
    of both go into gpurun_out/gradsync_wire_dtype.json; both are held to the bars of the 4-rank test.  (Round 6: ONE spawn of the eight
    processes runs both wires, and the fp64 oracle of the global batch is shared with the 2- and 4-rank tests: 217 s of the suite -> one run.)"""
    from oracle import clip_oracle as O
    cfg = O.ClipConfig(**MID)
    port = D.free_port()
    mp.spawn(D.worker_even, args=(8, port, dataclasses.asdict(cfg), 4, str(tmp_path), "cuda", "bfloat16", 8, "gloo", 2, False, "model+float32"), nprocs=8, join=True)
    # measured (profiles/r04_a_gradsync_wire_dtype.json): worst gradient (the patch-embedding bias, a column sum of cancelling terms) 4.59 % /
    # cosine 0.99895 on a bf16 wire, 4.56 % / 0.99896 on an fp32 wire -- the error is the ranks' own bf16 arithmetic, not the reduction
    bars = dict(rel_bar=0.08, loss_bar=3e-4, cos_bar=0.995)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "gradsync_wire_dtype.json")
    rec = {}
    for wire, suffix in (("bfloat16", "_model"), ("float32", "_float32")):
        measured = {}
        try:
            worst = D.check_even(str(tmp_path), cfg, 4, 8, dtype=torch.bfloat16, patch_keep=8, measured=measured, suffix=suffix, **bars)
        finally:
            rec[wire] = measured
            with open(path, "w") as f:
                json.dump(rec, f, indent=1)
        print("worst gradient relative error (8 ranks, bf16 model, wire", wire, "):", worst)


# ---- RCCL on DISTINCT devices: rank r on cuda:r, the `nccl` backend.  On the one-GPU boxes of the pool these SKIP
# (and say so in the report); on a multi-GPU node they are the parity check of exactly what `bench.py --gpus N` runs: the latents' RCCL
# all-gather consumed per rank chunk, the log-sum-exp gather, the scalar all-reduces and GradSync's bucketed all-reduces launched from the
# hooks, all against the fp64 oracle of the concatenated global batch. ---------------------------------------------------------------------
def _need_devices(world, core=True):
    """core tests run wherever the devices exist; the others only with XCLIP_RCCL_FULL=1 -- on a multi-GPU box every one of these is a spawn of
    W interpreters with an RCCL start-up each, and the suite is timed (the core seven: even batches at W = 2 / 4 / 8, one ragged case per W,
    one reference fixture)"""
    have = torch.cuda.device_count()
    if have < world:
        pytest.skip(f"RCCL on distinct devices needs {world} GPUs, this box has {have}")
    if not core and os.environ.get("XCLIP_RCCL_FULL") != "1":
        pytest.skip("extended RCCL case: set XCLIP_RCCL_FULL=1")


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_ranks_vs_oracle_even_mid_bf16_gradsync(tmp_path, world):
    """the dim-512 bf16 model, DCL + one augmented text view, aligned per-rank batches, three steps (the later ones launch the gradient
    buckets from the hooks in the agreed order): every rank's loss = the oracle's on the global batch, every rank's averaged gradients =
    (1 / W) x the oracle's and the same bits on every rank"""
    _need_devices(world)
    from oracle import clip_oracle as O
    cfg = O.ClipConfig(**MID)
    batch = 32 // world if world > 2 else 8
    port = D.free_port()
    mp.spawn(D.worker_even, args=(world, port, dataclasses.asdict(cfg), batch, str(tmp_path), "rccl", "bfloat16", 8, "nccl", 3), nprocs=world, join=True)
    worst = D.check_even(str(tmp_path), cfg, batch, world, dtype=torch.bfloat16, patch_keep=8, rel_bar=0.08, loss_bar=3e-4, cos_bar=0.995)
    print(f"worst gradient relative error ({world} ranks over RCCL, bf16):", worst)


@pytest.mark.parametrize("world,name", [(2, "w2_dcl_gradsync"), (4, "w4_dcl"), (4, "w4_simreg_extra_dcl"), (4, "w4_multiview_m3n2"), (4, "w4_filip_dcl"),
                                        (8, "w8_infonce_gradsync"), (8, "w8_dcl_extra_multiview_m2n2_gradsync"), (8, "w8_filip")])
def test_rccl_ranks_vs_oracle_ragged(tmp_path, world, name):
    """ragged per-rank batches (padded on the wire, one rank with a single sample), every head: fp32, the oracle's bars of the gloo suite"""
    _need_devices(world, core=name in ("w2_dcl_gradsync", "w4_dcl", "w8_infonce_gradsync"))
    from oracle import clip_oracle as O
    from test_distributed_gloo import RAGGED
    sizes, over, n_t, n_i, gs = RAGGED[name]
    assert len(sizes) == world
    cfg = dataclasses.replace(O.CFG1, **over)
    port = D.free_port()
    mp.spawn(D.worker_ragged, args=(world, port, dataclasses.asdict(cfg), sizes, str(tmp_path), "rccl", n_t, n_i, gs), nprocs=world, join=True)
    D.check_ragged(str(tmp_path), cfg, sizes, n_t, n_i, gs)


@pytest.mark.parametrize("name,sizes", [("dist2_infonce", [5, 3]), ("dist2_dcl", [5, 3]), ("dist2_simreg_extra", [5, 3])])
def test_rccl_two_ranks_match_reference_semantics(name, sizes, tmp_path):
    """the reference's own distributed semantics (fixtures generated from the reference under gloo) over RCCL on two devices"""
    _need_devices(2, core=name == "dist2_dcl")
    port = D.free_port()
    mp.spawn(D.worker_fixture, args=(2, port, name, sizes, str(tmp_path), "rccl"), nprocs=2, join=True)
    D.check_fixture(str(tmp_path), name)


def test_rccl_two_ranks_one_device_probe(tmp_path):
    """RCCL (`nccl` backend) with both ranks on cuda:0: recorded, not required -- the single-GPU box cannot give each rank its own
    device; the driver's multi-GPU scaling run is where RCCL itself executes"""
    port = D.free_port()
    try:
        mp.spawn(D.worker_nccl_probe, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    except Exception as e:                                   # noqa: BLE001
        outcome = {"spawn_error": str(e)[:400]}
    else:
        outcome = {f"rank{r}": torch.load(os.path.join(tmp_path, f"nccl_rank{r}.pt"), weights_only=False) for r in range(2)
                   if os.path.exists(os.path.join(tmp_path, f"nccl_rank{r}.pt"))}
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "rccl_one_device_probe.json"), "w") as f:
        json.dump(outcome, f, indent=1, default=str)
    print("RCCL two ranks on one device:", outcome)
