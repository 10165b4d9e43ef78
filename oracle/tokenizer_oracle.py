"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package (x_clip_amd).

A plain restatement of the reference's byte-level BPE tokenizer (x_clip/tokenizer.py:27-167, itself OpenAI CLIP's
simple_tokenizer) as free functions over a `Vocab` record: the quadratic "find the best pair, merge every occurrence, repeat" loop,
written for clarity, against which the product's heap-based x_clip_amd/tokenizer.py is checked.

Parity pin: tests/test_tokenizer.py compares these functions with tests/golden/tokenizer_*.json, which
oracle/make_golden_tokenizer.py produced by importing and running the reference's own SimpleTokenizer in the build container
(ftfy and beartype, absent from the image, stubbed as identity / no-op decorators).
"""
from __future__ import annotations

import html
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Tuple

import regex


def byte_to_char() -> Dict[int, str]:
    """tokenizer.py:27-38: printable latin-1 bytes map to themselves, the rest to U+0100.. in byte order"""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    out, extra = {}, 0
    for b in keep:
        out[b] = chr(b)
    for b in range(256):
        if b not in out:
            out[b] = chr(256 + extra)
            extra += 1
    return out


@dataclass
class Vocab:
    encoder: Dict[str, int]
    ranks: Dict[Tuple[str, str], int]
    b2c: Dict[int, str]


def load_vocab(path: str) -> Vocab:
    """tokenizer.py:59-75"""
    b2c = byte_to_char()
    lines = Path(path).read_text(encoding="utf8").split("\n")[1:49152 - 256 - 2 + 1]
    merges = [tuple(ln.split()) for ln in lines]
    alphabet = list(b2c.values())                        # insertion order = printable bytes first (as the reference's dict)
    symbols = alphabet + [c + "</w>" for c in alphabet] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
    return Vocab(dict(zip(symbols, range(len(symbols)))), dict(zip(merges, range(len(merges)))), b2c)


def merge_word(v: Vocab, token: str) -> List[str]:
    """tokenizer.py:81-120"""
    if token in ("<|startoftext|>", "<|endoftext|>"):
        return [token]
    word = list(token[:-1]) + [token[-1] + "</w>"]
    while len(word) > 1:
        pairs = {(word[i], word[i + 1]) for i in range(len(word) - 1)}
        best = min(pairs, key=lambda p: v.ranks.get(p, float("inf")))
        if best not in v.ranks:
            break
        merged, i = [], 0
        while i < len(word):
            if i + 1 < len(word) and (word[i], word[i + 1]) == best:
                merged.append(word[i] + word[i + 1])
                i += 2
            else:
                merged.append(word[i])
                i += 1
        word = merged
    return word


_PAT = regex.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""", regex.IGNORECASE)


def encode(v: Vocab, text: str) -> List[int]:
    """tokenizer.py:122-128 (basic_clean :46-49 without ftfy, whitespace_clean :51-54)"""
    text = html.unescape(html.unescape(text)).strip()
    text = regex.sub(r"\s+", " ", text).strip().lower()
    ids: List[int] = []
    for tok in regex.findall(_PAT, text):
        tok = "".join(v.b2c[b] for b in tok.encode("utf-8"))
        ids.extend(v.encoder[s] for s in merge_word(v, tok))
    return ids


def tokenize(v: Vocab, texts: List[str], context_length: int = 256, truncate_text: bool = False,
             pad_to_context_length: bool = False) -> List[List[int]]:
    """tokenizer.py:140-167 as nested lists (pad id 0)"""
    rows = [encode(v, t) for t in texts]
    longest = max(len(r) for r in rows)
    if longest > context_length:
        if not truncate_text:
            raise RuntimeError(f"One of the inputs is too long for context length {context_length}")
        rows = [r[:context_length] for r in rows]
        longest = context_length
    width = context_length if pad_to_context_length else longest
    return [r + [0] * (width - len(r)) for r in rows]
