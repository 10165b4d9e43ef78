"""World-size-2 run of the product on CPU (gloo backend, kernels from the wave64 emulator build): the rank-sharded loss
must reproduce the reference's distributed semantics recorded in tests/golden/dist2_*.json -- every rank's loss equals the
single-process global-batch loss, the rank-summed parameter gradients equal the single-process gradients (temperature:
every rank holds the full gradient) -- and the public `all_gather` keeps the reference contract."""
import json
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def _maybe_poison():
    """XCLIP_TEST_POISON=1 (diagnostics): NaN-poison every torch.empty in the worker, see clip_cases.poisoned_empty"""
    if os.environ.get("XCLIP_TEST_POISON") == "1":
        import clip_cases
        clip_cases.poisoned_empty().__enter__()


def _worker(rank, world, port, name, sizes, tmp):
    _maybe_poison()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emu.build_emu import build
    from x_clip_amd import CLIP, _lib
    from x_clip_amd.distributed import all_gather
    from oracle import clip_oracle as O
    _lib._use_library_for_tests(build())
    with open(os.path.join(HERE, "golden", name + ".json")) as f:
        rec = json.load(f)
    cfg = O.ClipConfig(**rec["config"])
    sd = O.make_state_dict(cfg, rec["param_seed"], torch.float32)
    total = sum(sizes)
    text, image, _, _ = O.make_inputs(cfg, total, rec["input_seed"])
    lo = sum(sizes[:rank])
    text, image = text[lo: lo + sizes[rank]], image[lo: lo + sizes[rank]].float()
    model = CLIP(**cfg.ctor_kwargs(), visual_patch_dropout=0.0)        # AFTER init_process_group (x_clip.py:591)
    assert model.requires_all_gather
    model.load_state_dict(sd)
    model.train()
    loss = model(text, image, return_loss=True)
    loss.backward()
    grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()}
    # reference-contract all_gather: uneven sizes along dim 1, backward keeps the local slice
    x = (torch.arange(2 * sizes[rank] * 3, dtype=torch.float32).view(2, sizes[rank], 3) + 100 * rank).requires_grad_(True)
    gathered, szs = all_gather(x, 1, None)
    assert szs.tolist() == sizes and gathered.shape == (2, total, 3)
    assert torch.equal(gathered[:, lo: lo + sizes[rank]], x.detach())
    (gathered * torch.arange(total, dtype=torch.float32).view(1, -1, 1)).sum().backward()
    assert torch.equal(x.grad, torch.arange(lo, lo + sizes[rank], dtype=torch.float32).view(1, -1, 1).expand(2, -1, 3))
    torch.save({"loss": float(loss.detach()), "grads": grads}, os.path.join(tmp, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("name,sizes", [("dist2_infonce", [5, 3]), ("dist2_dcl", [5, 3]), ("dist2_simreg_extra", [5, 3])])
def test_two_ranks_match_reference_semantics(name, sizes, tmp_path):
    with open(os.path.join(HERE, "golden", name + ".json")) as f:
        rec = json.load(f)
    if sum(sizes) != sum(rec["sizes"]):
        pytest.skip("fixture batch differs")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, name, sizes, str(tmp_path)), nprocs=2, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=False) for r in range(2)]
    single = rec["single_process"]
    for o in outs:
        assert abs(o["loss"] - single["loss"]) < 1e-5, (o["loss"], single["loss"])
    for k, ref_norm in single["grad_norm"].items():
        if ref_norm is None:
            continue
        g0, g1 = outs[0]["grads"][k], outs[1]["grads"][k]
        if k == "temperature":
            for g in (g0, g1):          # downstream of the gather: every rank holds the full gradient
                assert abs(float(g.norm()) - ref_norm) <= 1e-4 * ref_norm + 1e-7
            continue
        tot = (g0 + g1).double().norm()
        assert abs(float(tot) - ref_norm) <= 5e-4 * ref_norm + 1e-7, (k, float(tot), ref_norm)


def _worker_even(rank, world, port, cfg_kwargs, batch, tmp):
    _maybe_poison()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emu.build_emu import build
    from x_clip_amd import CLIP, _lib
    from x_clip_amd.distributed import GradSync
    from oracle import clip_oracle as O
    _lib._use_library_for_tests(build())
    cfg = O.ClipConfig(**cfg_kwargs)
    sd = O.make_state_dict(cfg, 5, torch.float32)
    text, image, aug_t, aug_i = O.make_inputs(cfg, batch * world, 6, 1, 0)
    sl = slice(rank * batch, (rank + 1) * batch)
    model = CLIP(**cfg.ctor_kwargs(), visual_patch_dropout=0.0)
    model.load_state_dict(sd)
    model.train()
    model.assume_equal_batch = True
    sync = GradSync(model)
    loss = model(text[sl], image[sl].float(), return_loss=True, aug_text=[aug_t[0][sl]])
    loss.backward()
    sync.finish()
    grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()}
    torch.save({"loss": float(loss.detach()), "grads": grads}, os.path.join(tmp, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_two_ranks_even_batches_gradsync_vs_oracle(tmp_path):
    """aligned per-rank batches (chunked G path), CLOOB extra projections + DCL + one augmented text view, and the bucketed
    gradient all-reduce: after GradSync every rank holds (1/W) x the single-process global-batch gradient -- what
    DDP-mean gives the reference."""
    from oracle import clip_oracle as O
    import dataclasses
    cfg = dataclasses.replace(O.CFG1, decoupled_contrastive_learning=True, extra_latent_projection=True)
    batch, world = 8, 2
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_even, args=(world, port, dataclasses.asdict(cfg), batch, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=False) for r in range(world)]
    sd = {k: v.double().requires_grad_(True) for k, v in O.make_state_dict(cfg, 5, torch.float32).items()}
    text, image, aug_t, _ = O.make_inputs(cfg, batch * world, 6, 1, 0)
    ref = O.clip_forward(sd, cfg, text, image.float().double(), aug_t, [])
    ref.backward()
    for o in outs:
        assert abs(o["loss"] - float(ref.detach())) < 1e-5, (o["loss"], float(ref.detach()))
    for k, v in sd.items():
        if v.grad is None:
            continue
        # temperature: every rank computes the full d tau (like the reference), so its mean is the full gradient
        want = v.grad if k == "temperature" else v.grad / world
        for o in outs:
            g = o["grads"][k].double()
            rel = float((g - want).norm() / want.norm().clamp_min(1e-30))
            assert rel < 3e-4, (k, rel)
        assert torch.equal(outs[0]["grads"][k], outs[1]["grads"][k]), k


def _worker_filip(rank, world, port, cfg_kwargs, batch, tmp):
    _maybe_poison()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emu.build_emu import build
    from x_clip_amd import CLIP, _lib
    from oracle import clip_oracle as O
    _lib._use_library_for_tests(build())
    cfg = O.ClipConfig(**cfg_kwargs)
    sd = O.make_state_dict(cfg, 15, torch.float32)
    text, image, _, _ = O.make_inputs(cfg, batch * world, 16)
    sl = slice(rank * batch, (rank + 1) * batch)
    model = CLIP(**cfg.ctor_kwargs(), visual_patch_dropout=0.0)
    model.load_state_dict(sd)
    model.train()
    loss = model(text[sl], image[sl].float(), return_loss=True)
    loss.backward()
    grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()}
    torch.save({"loss": float(loss.detach()), "grads": grads}, os.path.join(tmp, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("dcl", [False, True])
def test_two_ranks_filip_vs_oracle(tmp_path, dcl):
    """fine-grained (FILIP) head across 2 ranks -- a configuration the reference cannot run (torch.stack of text and image
    latents): every rank's loss = single-process global-batch oracle loss, rank-summed gradients = oracle gradients"""
    from oracle import clip_oracle as O
    import dataclasses
    cfg = dataclasses.replace(O.CFG1, use_all_token_embeds=True, decoupled_contrastive_learning=dcl)
    batch, world = 4, 2
    port = 33500 + (os.getpid() % 2000) + (1 if dcl else 0)
    mp.spawn(_worker_filip, args=(world, port, dataclasses.asdict(cfg), batch, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=False) for r in range(world)]
    sd = {k: v.double().requires_grad_(True) for k, v in O.make_state_dict(cfg, 15, torch.float32).items()}
    text, image, _, _ = O.make_inputs(cfg, batch * world, 16)
    ref = O.clip_forward(sd, cfg, text, image.float().double())
    ref.backward()
    for o in outs:
        assert abs(o["loss"] - float(ref.detach())) < 1e-5, (o["loss"], float(ref.detach()))
    for k, v in sd.items():
        if v.grad is None or float(v.grad.abs().max()) == 0.0:
            continue
        g0, g1 = outs[0]["grads"][k], outs[1]["grads"][k]
        tot = (g0 + g1).double() / (world if k == "temperature" else 1)
        rel = float((tot - v.grad).norm() / v.grad.norm().clamp_min(1e-30))
        assert rel < 3e-4, (k, rel)
