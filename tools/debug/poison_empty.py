"""Debug aid: NaN-poisons every torch.empty / empty_like allocation, then runs one end-to-end case; a kernel that reads memory it
was supposed to write first shows up as a non-finite gradient.  python tools/debug/poison_empty.py [cpu|cuda]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
dev = torch.device("cuda:0" if (len(sys.argv) < 2 or sys.argv[1] == "cuda") else "cpu")
if dev.type == "cpu":
    from x_clip_amd import _lib
    from emu.build_emu import build
    _lib._use_library_for_tests(build())
import clip_cases as C
from oracle import clip_oracle as O
_empty, _empty_like = torch.empty, torch.empty_like


def poison(t):
    if t.is_floating_point():
        t.fill_(float("nan"))
    elif t.dtype == torch.uint8:
        t.fill_(0xFF)
    return t


torch.empty = lambda *a, **k: poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: poison(_empty_like(*a, **k))
cfg = O.ClipConfig(dim_text=768, dim_image=1024, dim_latent=768, num_text_tokens=3000, text_enc_depth=1, text_seq_len=77,
                   text_heads=12, visual_enc_depth=2, visual_image_size=56, visual_patch_size=14, visual_heads=16)
sd = O.make_state_dict(cfg, 7, torch.float32)
text, image, aug_t, aug_i = O.make_inputs(cfg, 8, 8, 1, 1)
g = torch.Generator().manual_seed(9)
keep = torch.randn(16, cfg.num_patches, generator=g).topk(8, dim=-1).indices
for rep in range(2):
    model = C.build_clip(cfg, sd, dev, torch.bfloat16, patch_dropout=0.5)
    loss = C.run_product(model, text, image.to(torch.bfloat16), aug_t, [a.to(torch.bfloat16) for a in aug_i], dev, torch.bfloat16, keep)
    bad = [k for k, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    big = [(k, float(p.grad.abs().max())) for k, p in model.named_parameters() if p.grad is not None and float(p.grad.float().abs().max()) > 1e3]
    print("rep", rep, "loss", float(loss), "non-finite grads:", bad[:8], "huge:", big[:8])

# ---- second pass: wrap every x_clip_amd.ops function, report the first call whose outputs contain NaN although its tensor inputs are finite
import inspect
from x_clip_amd import ops
state = {"found": False, "n": 0}


def tensors(o):
    if isinstance(o, torch.Tensor):
        return [o]
    if isinstance(o, (tuple, list)):
        return [t for x in o for t in tensors(x)]
    return []


def wrap(name, fn):
    def inner(*a, **k):
        out = fn(*a, **k)
        if state["found"]:
            return out
        if dev.type == "cuda":
            torch.cuda.synchronize()
        ins = [t for t in tensors(list(a) + list(k.values())) if t.is_floating_point()]
        outs = [t for t in tensors(out) if t.is_floating_point()]
        if any(not torch.isfinite(t).all() for t in outs):
            fin_in = all(torch.isfinite(t).all() for t in ins if t.data_ptr() not in {o.data_ptr() for o in outs})
            print(f"NaN out of ops.{name}: inputs finite = {fin_in}; in shapes {[tuple(t.shape) for t in ins]}; out shapes {[tuple(t.shape) for t in outs]}; "
                  f"non-finite per output {[int((~torch.isfinite(t)).sum()) for t in outs]}", flush=True)
            if fin_in and k.get("out") is None:
                for t in outs:
                    bad = (~torch.isfinite(t)).nonzero()
                    if bad.numel():
                        print("   first / last bad index:", bad[0].tolist(), bad[-1].tolist(), flush=True)
                state["n"] += 1
                state["found"] = state["n"] >= 4
        return out
    return inner


for name, fn in list(vars(ops).items()):
    if inspect.isfunction(fn) and not name.startswith("_") and name not in ("workspace", "dtype_code", "vec", "ln_eps"):
        setattr(ops, name, wrap(name, fn))
from x_clip_amd import functional
functional.OVERLAP_WGRAD = os.environ.get("WGRAD", "1") == "1"
model = C.build_clip(cfg, sd, dev, torch.bfloat16, patch_dropout=0.5)
model.overlap_towers = os.environ.get("TOWERS", "1") == "1"
loss = C.run_product(model, text, image.to(torch.bfloat16), aug_t, [a.to(torch.bfloat16) for a in aug_i], dev, torch.bfloat16, keep)
bad = [k for k, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
print("wrapped run: non-finite grads", len(bad), bad[:4])
