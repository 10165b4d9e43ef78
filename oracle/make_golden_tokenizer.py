"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Golden vectors for the tokenizer.

Loads the *unmodified reference* tokenizer module (/root/reference/x_clip/tokenizer.py) in the build container and records what its
SimpleTokenizer returns
  * with tests/golden/bpe_toy_merges.txt (this repository's own small merges file)  -> tests/golden/tokenizer_toy.json
  * with the CLIP vocabulary the reference ships (x_clip/data/bpe_simple_vocab_16e6.txt) -> tests/golden/tokenizer_clip_vocab.json
    (ids only; the vocabulary itself is not copied -- the tests that need it run where XCLIP_BPE_VOCAB points at a copy and are
    skipped elsewhere).
Two imports the image lacks are stubbed, neither on the arithmetic path: `ftfy.fix_text` as the identity (the texts below are
well-formed Unicode, for which it is the identity anyway) and `beartype` as a no-op decorator.

    python oracle/make_golden_tokenizer.py
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_FILE = "/root/reference/x_clip/tokenizer.py"
REF_VOCAB = "/root/reference/x_clip/data/bpe_simple_vocab_16e6.txt"

TEXTS = [
    "a photo of a cat sitting on the mat",
    "A Photo of TWO dogs,   running in the park!!",
    "it's a bird, it's a plane; they've seen it & we'll see it -- I'm sure you'd like it.",
    "naïve café über el niño, 1997 and 2015",
    "photographs of photographers photographing photographs",
    "日本語 のテキスト and emoji 🙂🙂 ...",
    "aaaa aaa aa a aaaaaaaa banana bandana",
    "&lt;b&gt; html &amp;amp; entities &quot;quoted&quot;",
    "<|startoftext|> the quick brown fox <|endoftext|>",
    "x",
    "tabs\tand\nnewlines   collapse",
    "supercalifragilisticexpialidocious antidisestablishmentarianism 3.14159",
]


def load_reference():
    ftfy = types.ModuleType("ftfy")
    ftfy.fix_text = lambda t: t
    bt = types.ModuleType("beartype")
    bt.beartype = lambda f: f
    btt = types.ModuleType("beartype.typing")
    import typing
    for n in ("Optional", "Union", "List"):
        setattr(btt, n, getattr(typing, n))
    sys.modules.setdefault("ftfy", ftfy)
    sys.modules.setdefault("beartype", bt)
    sys.modules.setdefault("beartype.typing", btt)
    spec = importlib.util.spec_from_file_location("reference_tokenizer", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                         # (builds the module-level tokenizer from the CLIP vocabulary)
    return mod


def record(tok, name, source):
    rows = [tok.encode(t) for t in TEXTS]
    padded = tok.tokenize(TEXTS[:5], context_length=32, truncate_text=True, pad_to_context_length=True)
    natural = tok.tokenize(TEXTS[:3])
    out = {
        "source": source,
        "texts": TEXTS,
        "encode": rows,
        "tokenize_ctx32_truncate_pad": padded.to(torch.int64).tolist(),
        "tokenize_natural": natural.to(torch.int64).tolist(),
        "decode": [tok.decode(r) for r in rows],
        "bpe": {w: tok.bpe(w) for w in ("photographs", "aaaa", "x", "sunset")},
        "n_symbols": len(tok.encoder),
    }
    try:
        tok.tokenize(TEXTS, context_length=8)
        out["too_long_raises"] = False
    except RuntimeError as e:
        out["too_long_raises"] = str(e)
    path = os.path.join(GOLDEN, name)
    with open(path, "w", encoding="utf8") as f:
        json.dump(out, f, ensure_ascii=True, indent=0)
    print(name, sum(len(r) for r in rows), "ids")


def main():
    ref = load_reference()
    record(ref.SimpleTokenizer(os.path.join(GOLDEN, "bpe_toy_merges.txt")), "tokenizer_toy.json",
           "reference SimpleTokenizer(bpe_path=tests/golden/bpe_toy_merges.txt)")
    record(ref.SimpleTokenizer(REF_VOCAB), "tokenizer_clip_vocab.json", "reference SimpleTokenizer() with the CLIP vocabulary it ships")


if __name__ == "__main__":
    main()
