"""Host-side text path: byte-level BPE tokenizer -> int64 [b, n] -> pinned host ring -> device.

Replaces reference x_clip/tokenizer.py:58-169 (`SimpleTokenizer`, the module-level `tokenizer`) with the same public surface --
`encode`, `decode`, `tokenize(texts, context_length=256, truncate_text=False, pad_to_context_length=False)` returning the int64
id matrix `CLIP.forward(text, ...)` consumes (pad id 0 = CLIP's `text_pad_id`) -- and adds what the reference leaves to the user:
`TokenPipeline`, which tokenizes batches on a worker thread into a ring of page-locked host buffers and uploads them on its own
HIP stream, so the ids of step s + 1 are resident in HBM while step s runs (bench.py's contract: inputs resident when the timed
region starts).

The vocabulary is data, not code, and is NOT shipped here: pass `bpe_path`, set XCLIP_BPE_VOCAB, or drop the reference's
`bpe_simple_vocab_16e6.txt` into x_clip_amd/data/.  Any merges file in that format works (header line, then `left right` per
line, most frequent first); tests/golden/bpe_toy_merges.txt is a small one made by oracle/make_toy_merges.py.

Differences from the reference, all deliberate:
  * the merge loop is a rank-ordered heap over a doubly linked symbol list (O(n log n) per word) instead of rescanning the word
    for its best pair after every merge; the result is the same sequence of merges (tests/test_tokenizer.py against the reference's
    own output);
  * `ftfy.fix_text` (reference tokenizer.py:47) is applied when ftfy is importable and skipped otherwise (it is not in this image):
    for text that is already well-formed Unicode it is the identity;
  * `tokenize` always returns int64 (the reference returns float32 for a batch whose every text is empty);
  * `vocab_size` is the real size of the vocabulary built from the file (49408 for the CLIP file, which the reference hard-codes).
`decode(remove_start_end=True)` drops ids 49406, 40407 and 0 exactly as reference tokenizer.py:133 does (40407 sic).
"""
from __future__ import annotations

import heapq
import html
import os
import queue
import threading
from pathlib import Path
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple, Union

import regex
import torch

try:                                                   # optional: mojibake repair (reference tokenizer.py:47)
    import ftfy as _ftfy
except ImportError:                                    # pragma: no cover - not in this image
    _ftfy = None

_WORD_END = "</w>"
_SPECIALS = ("<|startoftext|>", "<|endoftext|>")
_MAX_MERGES = 49152 - 256 - 2                          # the CLIP vocabulary: 256 bytes + 256 word-final bytes + merges + 2 = 49408

# pre-tokenisation: the two specials, English contractions, letter runs, single digits, runs of anything else that is not space
_SPLIT = regex.compile(
    "|".join([regex.escape(s) for s in _SPECIALS] + [r"'s", r"'t", r"'re", r"'ve", r"'m", r"'ll", r"'d",
                                                     r"[\p{L}]+", r"[\p{N}]", r"[^\s\p{L}\p{N}]+"]),
    regex.IGNORECASE)
_SPACES = regex.compile(r"\s+")


def byte_alphabet() -> Tuple[List[str], List[str]]:
    """one printable character per byte value (reference bytes_to_unicode, tokenizer.py:27-38): bytes that already print keep their
    code point, the other 68 take 256, 257, ... in byte order.  Returned in VOCABULARY order (printable first)."""
    printable = [b for b in range(256) if 33 <= b <= 126 or 161 <= b <= 172 or 174 <= b <= 255]
    rest = [b for b in range(256) if b not in set(printable)]
    order = printable + rest
    chars = [chr(b) for b in printable] + [chr(256 + i) for i in range(len(rest))]
    table = [""] * 256
    for b, c in zip(order, chars):
        table[b] = c
    return [table[b] for b in order], table             # (in vocabulary order, indexed by byte value)


def default_bpe() -> str:
    cands = [os.environ.get("XCLIP_BPE_VOCAB"), str(Path(__file__).with_name("data") / "bpe_simple_vocab_16e6.txt")]
    for c in cands:
        if c and os.path.isfile(c):
            return c
    raise FileNotFoundError(
        "no BPE merges file: pass SimpleTokenizer(bpe_path=...), set XCLIP_BPE_VOCAB, or copy bpe_simple_vocab_16e6.txt "
        "(shipped with the reference under x_clip/data/) to " + cands[1])


class SimpleTokenizer:
    def __init__(self, bpe_path: Optional[str] = None):
        path = bpe_path if bpe_path is not None else default_bpe()
        lines = Path(path).read_text(encoding="utf8").split("\n")
        merges = [tuple(ln.split()) for ln in lines[1:1 + _MAX_MERGES]]
        vocab_order, self._byte_char = byte_alphabet()
        vocab = list(vocab_order) + [c + _WORD_END for c in vocab_order] + ["".join(m) for m in merges] + list(_SPECIALS)
        # (a file shorter than the CLIP one ends in an empty line, which becomes an empty symbol with an id of its own, as it does in
        #  the reference: the ids of the two specials depend on it)
        self.encoder: Dict[str, int] = {s: i for i, s in enumerate(vocab)}
        if len(self.encoder) != len(vocab):                       # a merges file that creates a symbol twice: last one wins, as in
            self.encoder = dict(zip(vocab, range(len(vocab))))    # the reference's dict(zip(...)) (tokenizer.py:72)
        self.decoder: Dict[int, str] = {i: s for s, i in self.encoder.items()}
        self.vocab_size = len(vocab)
        self.bpe_ranks: Dict[Tuple[str, str], int] = {}
        for r, m in enumerate(merges):
            if len(m) == 2:
                self.bpe_ranks[m] = r                              # (a repeated pair keeps its LAST rank: dict(zip(...)) again)
        self._char_byte = {c: b for b, c in enumerate(self._byte_char)}
        self._cache: Dict[str, List[int]] = {s: [self.encoder[s]] for s in _SPECIALS}
        self.sot_id, self.eot_id = self.encoder[_SPECIALS[0]], self.encoder[_SPECIALS[1]]

    # ---- one pre-token -> ids ---------------------------------------------------------------------------------------------
    def _merge_word(self, symbols: List[str]) -> List[str]:
        """apply the learned merges to one word, lowest rank first, leftmost occurrence first"""
        n = len(symbols)
        if n < 2:
            return symbols
        ranks = self.bpe_ranks
        prev = list(range(-1, n - 1))
        nxt = list(range(1, n + 1))
        nxt[-1] = -1
        alive = [True] * n
        heap = []
        for i in range(n - 1):
            r = ranks.get((symbols[i], symbols[i + 1]))
            if r is not None:
                heap.append((r, i, symbols[i], symbols[i + 1]))
        heapq.heapify(heap)
        while heap:
            # every queued occurrence of the best pair, left to right, BEFORE anything their merges create is looked at: with a
            # merges file in which a later line outranks an earlier one this is what keeps the result the reference's
            best = heap[0][0]
            batch = []
            while heap and heap[0][0] == best:
                batch.append(heapq.heappop(heap))
            for _, i, left, right in batch:
                j = nxt[i] if alive[i] else -1
                if j < 0 or symbols[i] != left or symbols[j] != right:   # one of the two has been merged away since this was queued
                    continue
                symbols[i] = left + right
                alive[j] = False
                k = nxt[j]
                nxt[i] = k
                if k >= 0:
                    prev[k] = i
                    r2 = ranks.get((symbols[i], symbols[k]))
                    if r2 is not None:
                        heapq.heappush(heap, (r2, i, symbols[i], symbols[k]))
                h = prev[i]
                if h >= 0:
                    r2 = ranks.get((symbols[h], symbols[i]))
                    if r2 is not None:
                        heapq.heappush(heap, (r2, h, symbols[h], symbols[i]))
        return [s for s, a in zip(symbols, alive) if a]

    def _word_ids(self, token: str) -> List[int]:
        ids = self._cache.get(token)
        if ids is None:
            symbols = list(token[:-1]) + [token[-1] + _WORD_END]
            ids = [self.encoder[s] for s in self._merge_word(symbols)]
            self._cache[token] = ids
        return ids

    def bpe(self, token: str) -> str:
        """the merged symbols of one pre-token, space separated (reference SimpleTokenizer.bpe, tokenizer.py:81-120)"""
        return " ".join(self.decoder[i] for i in self._word_ids(token))

    # ---- text <-> ids -----------------------------------------------------------------------------------------------------
    @staticmethod
    def clean(text: str) -> str:
        if _ftfy is not None:
            text = _ftfy.fix_text(text)
        text = html.unescape(html.unescape(text)).strip()
        return _SPACES.sub(" ", text).strip().lower()

    def encode(self, text: str) -> List[int]:
        out: List[int] = []
        table = self._byte_char
        for tok in _SPLIT.findall(self.clean(text)):
            out.extend(self._word_ids("".join(table[b] for b in tok.encode("utf-8"))))
        return out

    def decode(self, tokens, remove_start_end: bool = True, pad_tokens=frozenset()) -> str:
        if torch.is_tensor(tokens):
            tokens = tokens.tolist()
        if remove_start_end:
            tokens = [t for t in tokens if t not in (49406, 40407, 0)]
        text = "".join(self.decoder[t] for t in tokens if t not in pad_tokens)
        return bytearray(self._char_byte[c] for c in text).decode("utf-8", errors="replace").replace(_WORD_END, " ")

    def tokenize(self, texts: Union[str, Sequence[str]], context_length: int = 256, truncate_text: bool = False,
                 pad_to_context_length: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """-> int64 [b, n], n = the longest text (or context_length with pad_to_context_length), zero padded on the right.
        `out`: an int64 [b, context_length] host tensor to fill instead of allocating (implies pad_to_context_length)."""
        if isinstance(texts, str):
            texts = [texts]
        if not isinstance(texts, (list, tuple)) or not all(isinstance(t, str) for t in texts):
            raise TypeError("texts must be a string or a list of strings")
        rows = [self.encode(t) for t in texts]
        longest = max((len(r) for r in rows), default=0)
        if longest > context_length:
            if not truncate_text:
                raise RuntimeError(f"One of the inputs is too long for context length {context_length}")
            rows = [r[:context_length] for r in rows]
            longest = context_length
        width = context_length if (pad_to_context_length or out is not None) else longest
        if out is None:
            out = torch.zeros(len(rows), width, dtype=torch.int64)
        else:
            assert out.dtype == torch.int64 and tuple(out.shape) == (len(rows), context_length) and out.device.type == "cpu"
            out.zero_()
        for i, r in enumerate(rows):
            if r:
                out[i, : len(r)] = torch.tensor(r, dtype=torch.int64)
        return out


class TokenPipeline:
    """texts -> device-resident int64 [batch, context_length] batches, tokenized and uploaded ahead of the consumer.

    A worker thread pulls `batch_size` strings at a time from `texts`, tokenizes them into the next of `depth` page-locked host
    buffers and enqueues an asynchronous host-to-device copy on the pipeline's own HIP stream; iterating yields the device tensor
    after making the CURRENT stream wait for that copy (no host synchronisation).  A host buffer is reused only after the copy
    that read it has completed (event).  On a CPU device the buffers are ordinary memory and the 'copy' is the buffer itself."""

    def __init__(self, texts: Iterable[str], batch_size: int, context_length: int = 256, device="cuda", tokenizer=None,
                 truncate_text: bool = True, depth: int = 3, drop_last: bool = False):
        self.tok = tokenizer if tokenizer is not None else get_tokenizer()
        self.device = torch.device(device)
        self.batch_size, self.context_length, self.truncate, self.drop_last = batch_size, context_length, truncate_text, drop_last
        self._gpu = self.device.type == "cuda"
        self._host = [torch.zeros(batch_size, context_length, dtype=torch.int64, pin_memory=self._gpu) for _ in range(depth)]
        self._free_evt = [None] * depth                 # copy-done event of the last upload that read host buffer i
        self._stream = torch.cuda.Stream(self.device) if self._gpu else None
        self._q: "queue.Queue" = queue.Queue(maxsize=max(1, depth - 1))
        self._src = iter(texts)
        self._err = None
        self._thread = threading.Thread(target=self._work, daemon=True)
        self._thread.start()

    def _work(self):
        try:
            slot = 0
            while True:
                batch: List[str] = []
                for t in self._src:
                    batch.append(t)
                    if len(batch) == self.batch_size:
                        break
                if not batch or (self.drop_last and len(batch) < self.batch_size):
                    break
                host = self._host[slot]
                if self._free_evt[slot] is not None:
                    self._free_evt[slot].synchronize()   # (worker thread only: the consumer never blocks on this)
                view = host[: len(batch)]
                self.tok.tokenize(batch, self.context_length, truncate_text=self.truncate, out=view)
                if self._gpu:
                    with torch.cuda.stream(self._stream):
                        dev = view.to(self.device, non_blocking=True)
                        evt = torch.cuda.Event()
                        evt.record(self._stream)
                    self._free_evt[slot] = evt
                    self._q.put((dev, evt))
                else:
                    self._q.put((view.clone(), None))
                slot = (slot + 1) % len(self._host)
        except BaseException as e:                        # surfaces in the consumer
            self._err = e
        finally:
            self._q.put(None)

    def __iter__(self) -> Iterator[torch.Tensor]:
        return self

    def __next__(self) -> torch.Tensor:
        item = self._q.get()
        if item is None:
            self._q.put(None)
            if self._err is not None:
                raise self._err
            raise StopIteration
        dev, evt = item
        if evt is not None:
            torch.cuda.current_stream(self.device).wait_event(evt)
            dev.record_stream(torch.cuda.current_stream(self.device))
        return dev


_default: Optional[SimpleTokenizer] = None


def get_tokenizer() -> SimpleTokenizer:
    global _default
    if _default is None:
        _default = SimpleTokenizer()
    return _default


def __getattr__(name):                                   # `from x_clip.tokenizer import tokenizer` (reference tokenizer.py:169), built lazily
    if name == "tokenizer":
        return get_tokenizer()
    raise AttributeError(name)
