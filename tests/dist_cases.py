"""Workers of the multi-rank runs of the product, shared by the CPU suite (tests/test_distributed_gloo.py: gloo backend, kernels from the
wave64 emulator build) and the GPU suite (tests/test_distributed_gpu.py: libxclip_hip.so, real HIP streams).  `kind` =
  "cpu"  -- emulator kernels, gloo;
  "cuda" -- every rank on cuda:0 of a one-GPU box, gloo carrying the collectives (RCCL does not accept two ranks on one device);
  "rccl" -- rank r on cuda:r, the `nccl` backend (= RCCL over xGMI): what `bench.py --gpus N` runs on a multi-GPU node.  The tests that
            use it skip where torch.cuda.device_count() < world."""
import json
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def free_port() -> int:
    """a TCP port nobody listens on right now (asked from the kernel): the parent test picks it and hands it to its workers -- ports
    derived from the process id collided between parallel pytest workers once the suite grew"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _maybe_poison():
    """XCLIP_TEST_POISON=1: NaN-poison every torch.empty in the worker, see clip_cases.poisoned_empty"""
    if os.environ.get("XCLIP_TEST_POISON") == "1":
        import clip_cases
        clip_cases.poisoned_empty().__enter__()


def setup(rank, world, port, kind, backend="gloo"):
    """process group + kernel library for this worker -> torch.device"""
    _maybe_poison()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from x_clip_amd import _lib
    if kind in ("cuda", "rccl"):
        index = rank if kind == "rccl" else 0
        if kind == "rccl":
            assert torch.cuda.device_count() >= world, (torch.cuda.device_count(), world)
            backend = "nccl"
        torch.cuda.set_device(index)
        dev = torch.device("cuda", index)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        _lib._use_library_for_tests(None)
        _lib.lib()                                   # libxclip_hip.so or an exception: no fallback
        assert not _lib.is_emulator()
    else:
        dev = torch.device("cpu")
        dist.init_process_group(backend, rank=rank, world_size=world)
        from emu.build_emu import build
        _lib._use_library_for_tests(build())
    return dev


def worker_fixture(rank, world, port, name, sizes, tmp, kind="cpu"):
    dev = setup(rank, world, port, kind)
    from x_clip_amd import CLIP
    from x_clip_amd.distributed import all_gather
    from oracle import clip_oracle as O
    with open(os.path.join(HERE, "golden", name + ".json")) as f:
        rec = json.load(f)
    cfg = O.ClipConfig(**rec["config"])
    sd = O.make_state_dict(cfg, rec["param_seed"], torch.float32)
    total = sum(sizes)
    text, image, _, _ = O.make_inputs(cfg, total, rec["input_seed"])
    lo = sum(sizes[:rank])
    text, image = text[lo: lo + sizes[rank]].to(dev), image[lo: lo + sizes[rank]].float().to(dev)
    model = CLIP(**cfg.ctor_kwargs(), visual_patch_dropout=0.0)        # AFTER init_process_group (x_clip.py:591)
    assert model.requires_all_gather
    model.load_state_dict(sd)
    model = model.to(dev).train()
    loss = model(text, image, return_loss=True)
    loss.backward()
    grads = {k: (p.grad.detach().cpu().clone() if p.grad is not None else None) for k, p in model.named_parameters()}
    # reference-contract all_gather: uneven sizes along dim 1, backward keeps the local slice
    x = (torch.arange(2 * sizes[rank] * 3, dtype=torch.float32).view(2, sizes[rank], 3) + 100 * rank).to(dev).requires_grad_(True)
    gathered, szs = all_gather(x, 1, None)
    assert szs.tolist() == sizes and gathered.shape == (2, total, 3)
    assert torch.equal(gathered[:, lo: lo + sizes[rank]], x.detach())
    (gathered * torch.arange(total, dtype=torch.float32, device=dev).view(1, -1, 1)).sum().backward()
    assert torch.equal(x.grad.cpu(), torch.arange(lo, lo + sizes[rank], dtype=torch.float32).view(1, -1, 1).expand(2, -1, 3))
    torch.save({"loss": float(loss.detach()), "grads": grads}, os.path.join(tmp, f"rank{rank}.pt"))
    dist.destroy_process_group()


def worker_even(rank, world, port, cfg_kwargs, batch, tmp, kind="cpu", dtype_name="float32", patch_keep=None, backend="gloo", steps=1,
                force_gather=False, reduce_dtype_name=None, second_sink=False, bucket_bytes=None):
    """backend="nccl" with world = 1: RCCL itself on the one GPU of the box -- its collectives run on the process group's own stream and
    Work.wait() has real stream semantics (force_gather: CLIP latches requires_all_gather only for world > 1, x_clip.py:591)"""
    dev = setup(rank, world, port, kind, backend)
    from x_clip_amd import CLIP
    from x_clip_amd.distributed import GradSync
    from oracle import clip_oracle as O
    dtype = getattr(torch, dtype_name)
    cfg = O.ClipConfig(**cfg_kwargs)
    sd = O.make_state_dict(cfg, 5, torch.float32)
    sd = {k: (v.to(dtype).float() if v.is_floating_point() else v) for k, v in sd.items()}
    text, image, aug_t, aug_i = O.make_inputs(cfg, batch * world, 6, 1, 0)
    sl = slice(rank * batch, (rank + 1) * batch)
    model = CLIP(**cfg.ctor_kwargs(), visual_patch_dropout=0.5 if patch_keep else 0.0)
    model.load_state_dict(sd)
    model = model.to(dtype).to(dev).train()
    model.assume_equal_batch = True
    if patch_keep:
        g = torch.Generator().manual_seed(8)
        keep = torch.randn(batch * world * 2, cfg.num_patches, generator=g).topk(patch_keep, dim=-1).indices
        # rows of this rank's images: the main view, then the augmented views in the order CLIP.forward concatenates them
        model.visual_transformer.keep_indices_override = keep[sl].to(torch.int32).to(dev)
    if force_gather:
        model.requires_all_gather = True
    other = None
    if second_sink:                                           # a second GradSync (another model) must not displace the first
        other = GradSync(torch.nn.Linear(8, 8).to(dev))
    # reduce_dtype_name may name SEVERAL wire dtypes ("bfloat16+float32"): the same processes then run the steps once per wire (a GradSync
    # each) and save rank{r}_{wire}.pt -- one spawn of eight interpreters instead of two (the GPU suite's two slowest tests)
    wires = [None] if reduce_dtype_name is None else reduce_dtype_name.split("+")
    for wire in wires:
        wire_dtype = None if wire in (None, "model") else getattr(torch, wire)
        for p_ in model.parameters():
            p_.grad = None
        sync = GradSync(model, reduce_dtype=wire_dtype, **({"bucket_bytes": bucket_bytes} if bucket_bytes else {}))
        if bucket_bytes:                                          # the towers cut into several buckets (toy models: ask for small ones)
            assert len(sync.buckets) >= 5, [sum(p.numel() for p in b) for b in sync.buckets]
        if second_sink:
            other2 = GradSync(torch.nn.Linear(8, 8).to(dev))      # ... whichever was registered last
        for step in range(steps):                                 # steps > 1: the later steps launch buckets from the hooks, in the frozen order
            model.zero_grad(set_to_none=True)
            loss = model(text[sl].to(dev), image[sl].to(dtype).to(dev), return_loss=True, aug_text=[aug_t[0][sl].to(dev)])
            loss.backward()
            sync.finish()
            if wire_dtype is None:
                assert sync.stats["in_place"] >= 4 * (cfg.text_enc_depth + cfg.visual_enc_depth), sync.stats     # the weight-gradient GEMMs wrote into the bucket slices
            if step > 0:
                assert sync._agreed and all(e is not None for e in sync._expected)
        if kind != "cpu":
            torch.cuda.synchronize()
        grads = {k: (p.grad.detach().float().cpu().clone() if p.grad is not None else None) for k, p in model.named_parameters()}
        for p in model.parameters():
            assert p.grad is None or p.grad.dtype == p.dtype
        suffix = "" if len(wires) == 1 else "_" + str(wire)
        torch.save({"loss": float(loss.detach()), "grads": grads, "overlap": sync.overlap, "buckets": len(sync.buckets), "launched": sync.launched,
                    "order": list(sync._order)}, os.path.join(tmp, f"rank{rank}{suffix}.pt"))
        sync.remove()
        if second_sink:
            other2.remove()
    dist.destroy_process_group()


def worker_disagreeing_ranks(rank, world, port, cfg_kwargs, batch, tmp, kind="cpu"):
    """rank 1 freezes its text tower (fewer hook firings than rank 0 -- what a data-dependent side loss skipped on one rank looks like):
    the first step's agreement check must turn the overlap off on BOTH ranks; every bucket is then reduced in finish(), in index order"""
    import warnings
    dev = setup(rank, world, port, kind)
    from x_clip_amd import CLIP
    from x_clip_amd.distributed import GradSync
    from oracle import clip_oracle as O
    cfg = O.ClipConfig(**cfg_kwargs)
    sd = O.make_state_dict(cfg, 5, torch.float32)
    text, image, aug_t, _ = O.make_inputs(cfg, batch * world, 6, 1, 0)
    sl = slice(rank * batch, (rank + 1) * batch)
    model = CLIP(**cfg.ctor_kwargs(), visual_patch_dropout=0.0)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    model.assume_equal_batch = True
    sync = GradSync(model)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        for step in range(3):
            model.zero_grad(set_to_none=True)
            loss = model(text[sl].to(dev), image[sl].float().to(dev), return_loss=True, aug_text=[aug_t[0][sl].to(dev)],
                         freeze_text_encoder=(rank == 1))
            loss.backward()
            sync.finish()
    assert sync._agreed and sync.overlap is False, (sync._agreed, sync.overlap, sync._expected, rank)
    assert any("walked their towers differently" in str(w.message) for w in caught)
    # which buckets go on the wire may not depend on what this rank fired -- every rank issued every bucket in all three steps
    assert sync.launched == 3 * len(sync.buckets), (rank, sync.launched, len(sync.buckets))
    grads = {k: (p.grad.detach().float().cpu().clone() if p.grad is not None else None) for k, p in model.named_parameters()}
    torch.save({"loss": float(loss.detach()), "grads": grads}, os.path.join(tmp, f"rank{rank}.pt"))
    dist.destroy_process_group()


def worker_unfreeze(rank, world, port, cfg_kwargs, batch, tmp, kind="cpu", bucket_bytes=None):
    """LiT-style unlock: the text tower is frozen for two steps and trained from the third.  Without rearm() the third step's
    finish() raises on every rank (gradients for buckets no rank reduced in the first step) and leaves a usable object; after rearm() the
    step is a first step again and the averaged gradients of BOTH towers are the oracle's (data and checks of worker_even)."""
    dev = setup(rank, world, port, kind)
    from x_clip_amd import CLIP
    from x_clip_amd.distributed import GradSync
    from oracle import clip_oracle as O
    cfg = O.ClipConfig(**cfg_kwargs)
    sd = O.make_state_dict(cfg, 5, torch.float32)
    text, image, aug_t, _ = O.make_inputs(cfg, batch * world, 6, 1, 0)
    sl = slice(rank * batch, (rank + 1) * batch)
    model = CLIP(**cfg.ctor_kwargs(), visual_patch_dropout=0.0)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    model.assume_equal_batch = True
    sync = GradSync(model, **({"bucket_bytes": bucket_bytes} if bucket_bytes else {}))

    def step(freeze):
        model.zero_grad(set_to_none=True)
        loss = model(text[sl].to(dev), image[sl].float().to(dev), return_loss=True, aug_text=[aug_t[0][sl].to(dev)], freeze_text_encoder=freeze)
        loss.backward()
        return loss

    for _ in range(2):
        step(True)
        sync.finish()
    step(False)
    raised = False
    try:
        sync.finish()
    except RuntimeError as e:
        raised = "rearm" in str(e)
    assert raised
    assert sync._works == [] and all(c == 0 for c in sync._count) and not sync._claimed      # the step's state was reset before the raise
    sync.rearm()
    launched0 = sync.launched
    for _ in range(2):                                        # a first step again, then one launched from the hooks
        loss = step(False)
        sync.finish()
    assert sync._agreed and all((e or 0) > 0 for e in sync._expected), sync._expected
    assert sync.launched - launched0 == 2 * len(sync.buckets)
    if kind != "cpu":
        torch.cuda.synchronize()
    grads = {k: (p.grad.detach().float().cpu().clone() if p.grad is not None else None) for k, p in model.named_parameters()}
    torch.save({"loss": float(loss.detach()), "grads": grads}, os.path.join(tmp, f"rank{rank}.pt"))
    dist.destroy_process_group()


def worker_filip(rank, world, port, cfg_kwargs, batch, tmp, kind="cpu"):
    dev = setup(rank, world, port, kind)
    from x_clip_amd import CLIP
    from oracle import clip_oracle as O
    cfg = O.ClipConfig(**cfg_kwargs)
    sd = O.make_state_dict(cfg, 15, torch.float32)
    text, image, _, _ = O.make_inputs(cfg, batch * world, 16)
    sl = slice(rank * batch, (rank + 1) * batch)
    model = CLIP(**cfg.ctor_kwargs(), visual_patch_dropout=0.0)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    loss = model(text[sl].to(dev), image[sl].float().to(dev), return_loss=True)
    loss.backward()
    grads = {k: (p.grad.detach().cpu().clone() if p.grad is not None else None) for k, p in model.named_parameters()}
    torch.save({"loss": float(loss.detach()), "grads": grads}, os.path.join(tmp, f"rank{rank}.pt"))
    dist.destroy_process_group()


def worker_ragged(rank, world, port, cfg_kwargs, sizes, tmp, kind="cpu", n_aug_text=0, n_aug_image=0, gradsync=False, dtype_name="float32",
                  freeze_text=False, image_slices=1, bucket_bytes=None):
    """any world size, any per-rank batch sizes, any head: rank r holds rows sum(sizes[:r]) ... of the global batch of every view"""
    dev = setup(rank, world, port, kind)
    from x_clip_amd import CLIP
    from x_clip_amd.distributed import GradSync
    from oracle import clip_oracle as O
    dtype = getattr(torch, dtype_name)
    cfg = O.ClipConfig(**cfg_kwargs)
    sd = O.make_state_dict(cfg, 25, torch.float32)
    sd = {k: (v.to(dtype).float() if v.is_floating_point() else v) for k, v in sd.items()}
    text, image, aug_t, aug_i = O.make_inputs(cfg, sum(sizes), 26, n_aug_text, n_aug_image)
    lo = sum(sizes[:rank])
    sl = slice(lo, lo + sizes[rank])
    model = CLIP(**cfg.ctor_kwargs(), visual_patch_dropout=0.0)
    model.load_state_dict(sd)
    model = model.to(dtype).to(dev).train()
    sync = GradSync(model, **({"bucket_bytes": bucket_bytes} if bucket_bytes else {})) if gradsync else None
    model.image_micro_batches = image_slices              # > 1: the vision tower's parameters see that many backward passes per step
    kw = {}
    if n_aug_text:
        kw["aug_text"] = [a[sl].to(dev) for a in aug_t]
    if n_aug_image:
        kw["aug_image"] = [a[sl].to(dtype).to(dev) for a in aug_i]
    for step in range(3 if gradsync else 1):                  # GradSync: later steps reuse the persistent flat buffers and launch from hooks
        model.zero_grad(set_to_none=True)
        loss = model(text[sl].to(dev), image[sl].to(dtype).to(dev), return_loss=True, freeze_text_encoder=freeze_text, **kw)
        loss.backward()
        if sync is not None:
            sync.finish()
            if freeze_text:                                   # a whole bucket without gradients: zeros on the wire, .grad stays None
                assert all(p.grad is None for p in model.text_transformer.parameters())
                assert sync.stats["unused"] >= sum(1 for _ in model.text_transformer.parameters())
            # every weight-gradient GEMM of both towers and the latent projections wrote straight into its bucket slice
            n_gemm = 4 * ((0 if freeze_text else cfg.text_enc_depth) + (cfg.visual_enc_depth if image_slices == 1 else 0)) + 2 * (2 if cfg.extra_latent_projection else 1)
            if step > 0 and not freeze_text:
                assert all(e is not None for e in sync._expected)      # buckets complete on a learned firing count from step 2 on
            assert sync.stats["in_place"] >= n_gemm, (sync.stats, n_gemm)
            for p in model.parameters():
                if p.grad is not None:
                    assert p.grad.data_ptr() == sync._view(p).data_ptr()
    if kind != "cpu":
        torch.cuda.synchronize()
    grads = {k: (p.grad.detach().float().cpu().clone() if p.grad is not None else None) for k, p in model.named_parameters()}
    torch.save({"loss": float(loss.detach()), "grads": grads}, os.path.join(tmp, f"rank{rank}.pt"))
    dist.destroy_process_group()


def worker_nccl_probe(rank, world, port, tmp):
    """does RCCL accept two ranks on ONE device?  Records the outcome; never raises."""
    out = {"rank": rank}
    try:
        dev = setup(rank, world, port, "cuda", backend="nccl")
        t = torch.full((4,), float(rank + 1), device=dev)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        out["ok"], out["sum"] = True, t.tolist()
        dist.destroy_process_group()
    except Exception as e:                                    # noqa: BLE001 -- the point is to record what RCCL says
        out["ok"], out["error"] = False, f"{type(e).__name__}: {str(e)[:400]}"
    torch.save(out, os.path.join(tmp, f"nccl_rank{rank}.pt"))


# ---- the checks (run in the parent on the workers' files) ------------------------------------------------------------------------
def check_fixture(tmp, name, world=2):
    with open(os.path.join(HERE, "golden", name + ".json")) as f:
        rec = json.load(f)
    outs = [torch.load(os.path.join(tmp, f"rank{r}.pt"), weights_only=False) for r in range(world)]
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


_EVEN_ORACLE = {}                                             # (config, global batch, dtype, kept patches) -> (fp64 loss, fp64 gradients): the session's cache


def _even_oracle(cfg, total, dtype, patch_keep):
    """the fp64 oracle of worker_even's global batch -- evaluated once per (configuration, global batch) and session: the 2 / 4 / 8-rank runs of
    the dim-512 model share a global batch of 32, and the wire-dtype variants share everything"""
    import dataclasses
    from oracle import clip_oracle as O
    key = (tuple(sorted(dataclasses.asdict(cfg).items())), total, str(dtype), patch_keep)
    if key not in _EVEN_ORACLE:
        sd = O.make_state_dict(cfg, 5, torch.float32)
        sd = {k: (v.to(dtype).double().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
        text, image, aug_t, _ = O.make_inputs(cfg, total, 6, 1, 0)
        keep = None
        if patch_keep:
            g = torch.Generator().manual_seed(8)
            keep = torch.randn(total * 2, cfg.num_patches, generator=g).topk(patch_keep, dim=-1).indices[:total]
        with O.layer_norm_eps(1e-5 if dtype == torch.float32 else 1e-3):
            ref = O.clip_forward(sd, cfg, text, image.to(dtype).double(), aug_t, [], keep)
            ref.backward()
        _EVEN_ORACLE[key] = (ref.detach(), sd)
    return _EVEN_ORACLE[key]


def check_even(tmp, cfg, batch, world=2, dtype=torch.float32, patch_keep=None, rel_bar=3e-4, loss_bar=1e-5, cos_bar=None, measured=None, only_prefix=None,
               suffix=""):
    outs = [torch.load(os.path.join(tmp, f"rank{r}{suffix}.pt"), weights_only=False) for r in range(world)]
    ref, sd = _even_oracle(cfg, batch * world, dtype, patch_keep)
    for o in outs:
        assert abs(o["loss"] - float(ref.detach())) < loss_bar * max(1.0, abs(float(ref.detach()))), (o["loss"], float(ref.detach()))
    worst = (0.0, "")
    worst_cos = (2.0, "")
    failures = []
    for k, v in sd.items():
        if not torch.is_tensor(v) or not v.is_floating_point() or v.grad is None or float(v.grad.abs().max()) == 0.0:
            continue
        if only_prefix is not None and not k.startswith(tuple(only_prefix)):
            continue
        # temperature: every rank computes the full d tau (like the reference), so its mean is the full gradient
        want = v.grad if k == "temperature" else v.grad / world
        for o in outs:
            g = o["grads"][k].double()
            rel = float((g - want).norm() / want.norm().clamp_min(1e-30))
            worst = max(worst, (rel, k))
            cos = float((g * want).sum() / (g.norm() * want.norm()))
            worst_cos = min(worst_cos, (cos, k))
            if not rel < rel_bar:
                failures.append((k, "rel", rel))
            if cos_bar is not None and not cos > cos_bar:
                failures.append((k, "cos", cos))
    if measured is not None:
        measured.update(worst_rel=worst[0], worst_rel_param=worst[1], worst_cos=worst_cos[0], worst_cos_param=worst_cos[1], world=world)
    assert not failures, failures[:6]
    for k, v in sd.items():
        if not torch.is_tensor(v) or not v.is_floating_point() or v.grad is None or float(v.grad.abs().max()) == 0.0:
            continue
        if only_prefix is not None and not k.startswith(tuple(only_prefix)):
            continue
        for o in outs[1:]:
            assert torch.equal(outs[0]["grads"][k], o["grads"][k]), k       # the all-reduced gradients are the same bits on every rank
    return worst


def check_filip(tmp, cfg, batch, world=2):
    from oracle import clip_oracle as O
    outs = [torch.load(os.path.join(tmp, f"rank{r}.pt"), weights_only=False) for r in range(world)]
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


def check_ragged(tmp, cfg, sizes, n_aug_text=0, n_aug_image=0, gradsync=False, dtype=torch.float32, rel_bar=3e-4, loss_bar=1e-5,
                 freeze_text=False):
    """every rank's loss = the oracle's loss on the concatenated global batch; without GradSync the rank-SUMMED parameter gradients
    equal the oracle's (temperature sits downstream of the gather: every rank holds the full gradient, x_clip/distributed.py:51-54);
    with GradSync every rank holds (1/W) x that sum, bit-identical across ranks"""
    from oracle import clip_oracle as O
    world = len(sizes)
    outs = [torch.load(os.path.join(tmp, f"rank{r}.pt"), weights_only=False) for r in range(world)]
    sd = O.make_state_dict(cfg, 25, torch.float32)
    sd = {k: (v.to(dtype).double().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    text, image, aug_t, aug_i = O.make_inputs(cfg, sum(sizes), 26, n_aug_text, n_aug_image)
    with O.layer_norm_eps(1e-5 if dtype == torch.float32 else 1e-3):
        ref = O.clip_forward(sd, cfg, text, image.to(dtype).double(), aug_t, [a.to(dtype).double() for a in aug_i])
        ref.backward()
    for o in outs:
        assert abs(o["loss"] - float(ref.detach())) < loss_bar * max(1.0, abs(float(ref.detach()))), (o["loss"], float(ref.detach()))
    worst = (0.0, "")
    for k, v in sd.items():
        if not torch.is_tensor(v) or not v.is_floating_point() or v.grad is None or float(v.grad.abs().max()) == 0.0:
            continue
        if freeze_text and k.startswith("text_transformer."):
            assert all(o["grads"][k] is None for o in outs), k   # LiT-style frozen tower (x_clip.py:394-408): no gradient on any rank
            continue
        gs = [o["grads"][k].double() for o in outs]
        if gradsync:
            for g in gs[1:]:
                assert torch.equal(g, gs[0]), k
            got = gs[0] * (1 if k == "temperature" else world)
        else:
            got = sum(gs) / (world if k == "temperature" else 1)
        rel = float((got - v.grad).norm() / v.grad.norm().clamp_min(1e-30))
        worst = max(worst, (rel, k))
        assert rel < rel_bar, (k, rel)
    return worst
