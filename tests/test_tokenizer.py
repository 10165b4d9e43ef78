"""Host text path: x_clip_amd.tokenizer against (1) golden vectors the REFERENCE's own
SimpleTokenizer produced (oracle/make_golden_tokenizer.py, reference x_clip/tokenizer.py:58-167), (2) the oracle restatement on
random text, and the pinned-buffer upload pipeline.  The CLIP-vocabulary cases need the vocabulary file itself, which is not
shipped: they run where XCLIP_BPE_VOCAB (or x_clip_amd/data/) provides it and are skipped elsewhere."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import tokenizer_oracle as TO
from x_clip_amd import tokenizer as T

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOY = os.path.join(GOLDEN, "bpe_toy_merges.txt")


def _clip_vocab():
    try:
        return T.default_bpe()
    except FileNotFoundError:
        return None


def _check_fixture(name, path):
    g = json.load(open(os.path.join(GOLDEN, name), encoding="utf8"))
    tok, voc = T.SimpleTokenizer(path), TO.load_vocab(path)
    assert tok.vocab_size == g["n_symbols"] == len(voc.encoder)
    for text, want in zip(g["texts"], g["encode"]):
        assert tok.encode(text) == want, text                       # product == reference
        assert TO.encode(voc, text) == want, text                   # oracle == reference
    assert [tok.decode(r) for r in g["encode"]] == g["decode"]
    assert tok.decode(torch.tensor(g["encode"][0])) == g["decode"][0]
    got = tok.tokenize(g["texts"][:5], context_length=32, truncate_text=True, pad_to_context_length=True)
    assert got.dtype == torch.int64 and got.tolist() == g["tokenize_ctx32_truncate_pad"]
    assert TO.tokenize(voc, g["texts"][:5], 32, True, True) == g["tokenize_ctx32_truncate_pad"]
    assert tok.tokenize(g["texts"][:3]).tolist() == g["tokenize_natural"]
    assert {w: tok.bpe(w) for w in g["bpe"]} == g["bpe"]
    with pytest.raises(RuntimeError) as e:
        tok.tokenize(g["texts"], context_length=8)
    assert str(e.value) == g["too_long_raises"]


def test_reference_fixture_toy_vocabulary():
    _check_fixture("tokenizer_toy.json", TOY)


@pytest.mark.skipif(_clip_vocab() is None, reason="CLIP BPE vocabulary not available (set XCLIP_BPE_VOCAB)")
def test_reference_fixture_clip_vocabulary():
    _check_fixture("tokenizer_clip_vocab.json", _clip_vocab())
    tok = T.get_tokenizer()
    assert tok.vocab_size == 49408 and (tok.sot_id, tok.eot_id) == (49406, 49407)
    assert T.tokenizer is tok                                       # the reference's module-level instance, built lazily
    from x_clip.tokenizer import tokenizer as alias
    assert alias is tok


def test_heap_merge_equals_rescan_merge_on_random_words():
    """the product's heap over a linked list and the oracle's rescan loop perform the same merges -- also with a merges file whose
    lines are out of frequency order (bpe_toy_merges.txt line 6 needs a symbol that is only created later)"""
    tok, voc = T.SimpleTokenizer(TOY), TO.load_vocab(TOY)
    rng = np.random.RandomState(0)
    alphabet = list("aabnotheprszz") + ["é", "日", "🙂"]
    for _ in range(3000):
        n = rng.randint(1, 14)
        word = "".join(alphabet[i] for i in rng.randint(0, len(alphabet), n))
        assert tok.encode(word) == TO.encode(voc, word), word
    assert tok.encode("aazz aaaazzzz") == TO.encode(voc, "aazz aaaazzzz")
    sent = " ".join("".join(alphabet[i] for i in rng.randint(0, len(alphabet), rng.randint(1, 9))) for _ in range(200))
    assert tok.encode(sent) == TO.encode(voc, sent)


def test_tokenize_shapes_padding_and_errors():
    tok = T.SimpleTokenizer(TOY)
    a = tok.tokenize("a photo of a cat")
    assert a.shape[0] == 1 and a.dtype == torch.int64 and (a != 0).all()
    b = tok.tokenize(["a photo", "a photo of a cat sitting on the mat"], context_length=16, pad_to_context_length=True)
    assert tuple(b.shape) == (2, 16) and (b[0, 2:] == 0).all() and b[1, 0] == b[0, 0]
    c = tok.tokenize(["", ""], context_length=4, pad_to_context_length=True)          # (the reference returns float32 here)
    assert c.dtype == torch.int64 and tuple(c.shape) == (2, 4) and (c == 0).all()
    out = torch.full((2, 16), 7, dtype=torch.int64)
    assert tok.tokenize(["a photo", "a cat"], context_length=16, out=out) is out and (out[:, 4:] == 0).all()
    with pytest.raises(TypeError):
        tok.tokenize([1, 2])
    with pytest.raises(FileNotFoundError):
        T.SimpleTokenizer("/nonexistent/merges.txt")
    ids = tok.encode("the quick brown fox")
    assert tok.decode(ids).strip() == "the quick brown fox"


def test_edge_cases_against_the_oracle():
    """specials inside text, a very long word (the heap keeps the merge loop O(n log n)), mixed scripts, entities, truncation"""
    tok, voc = T.SimpleTokenizer(TOY), TO.load_vocab(TOY)
    texts = ["<|startoftext|>a photo<|endoftext|> of <|startoftext|> a cat", "x" * 1500 + "photograph" * 40, "ünïcödé 日本語テキスト 🙂🙂🙂 naïve",
             "&amp;lt;tag&amp;gt; &#233; &quot;q&quot;", "it's they've we'll i'm you'd that's 'tis", "1234567890 3.14 1e-5", "   \t\n  ", ""]
    for t in texts:
        assert tok.encode(t) == TO.encode(voc, t), t[:40]
    assert tok.encode("<|startoftext|>")[0] == tok.sot_id and tok.encode("<|endoftext|>")[0] == tok.eot_id
    long = "a photo of a cat " * 40
    ids = tok.encode(long)
    cut = tok.tokenize([long, "a cat"], context_length=16, truncate_text=True)
    assert tuple(cut.shape) == (2, 16) and cut[0].tolist() == ids[:16] and cut[1, 2:].eq(0).all()
    assert tok.decode(ids[:8], pad_tokens={ids[1]}) == tok.decode([i for i in ids[:8] if i != ids[1]])
    assert tok.decode(torch.tensor([0, 0] + ids[:4] + [0])) == tok.decode(ids[:4])       # pad id 0 dropped (reference tokenizer.py:133)


def _corpus(n):
    words = "a photo of the cat dog park sunset mountains running sitting quick brown fox two three".split()
    rng = np.random.RandomState(1)
    return [" ".join(words[i] for i in rng.randint(0, len(words), rng.randint(1, 20))) for _ in range(n)]


def test_pipeline_host_device_batches_in_order():
    tok = T.SimpleTokenizer(TOY)
    texts = _corpus(37)
    pipe = T.TokenPipeline(texts, batch_size=8, context_length=24, device="cpu", tokenizer=tok, depth=2)
    got = list(pipe)
    assert [g.shape[0] for g in got] == [8, 8, 8, 8, 5]
    want = tok.tokenize(texts, context_length=24, truncate_text=True, pad_to_context_length=True)
    assert torch.equal(torch.cat(got), want)
    assert len(list(T.TokenPipeline(texts, 8, 24, "cpu", tok, drop_last=True))) == 4
    with pytest.raises(RuntimeError):                                                  # worker errors surface in the consumer
        list(T.TokenPipeline(texts, 8, 4, "cpu", tok, truncate_text=False))


@pytest.mark.gpu
def test_pipeline_uploads_to_hbm_and_feeds_the_text_encoder():
    from x_clip_amd import CLIP
    dev = torch.device("cuda:0")
    tok = T.SimpleTokenizer(TOY)
    texts = _corpus(64)
    pipe = T.TokenPipeline(texts, batch_size=16, context_length=32, device=dev, tokenizer=tok, depth=3)
    want = tok.tokenize(texts, context_length=32, truncate_text=True, pad_to_context_length=True)
    torch.manual_seed(0)
    clip = CLIP(dim_text=64, dim_image=64, dim_latent=64, num_text_tokens=tok.vocab_size, text_enc_depth=1, text_seq_len=32,
                text_heads=1, visual_enc_depth=1, visual_image_size=64, visual_patch_size=32, visual_heads=1,
                visual_patch_dropout=0.).to(dev)
    clip.train()
    n = 0
    for i, ids in enumerate(pipe):
        assert ids.device == dev and ids.dtype == torch.int64 and tuple(ids.shape) == (16, 32)
        assert torch.equal(ids.cpu(), want[16 * i: 16 * i + 16])
        loss = clip(ids, torch.randn(16, 3, 64, 64, device=dev), return_loss=True)
        loss.backward()
        assert torch.isfinite(loss)
        n += 1
    assert n == 4
