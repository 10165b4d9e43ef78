"""TEST INFRASTRUCTURE ONLY: learns a small BPE merges file (tests/golden/bpe_toy_merges.txt, the format of the CLIP vocabulary the
reference ships: one header line, then `left right` per line, most frequent first) from a few sentences, so that the tokenizer
tests have a vocabulary that is this repository's own.      python oracle/make_toy_merges.py"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.tokenizer_oracle import byte_to_char

CORPUS = """
a photo of a cat sitting on the mat . a photo of a dog running in the park . two dogs and three cats are playing together
the quick brown fox jumps over the lazy dog . she sells sea shells by the sea shore . an image of the mountains at sunset
it's a bird , it's a plane ! they've seen it , we'll see it , i'm sure you'd like it . that's what she said
naive cafe , the uber driver , el nino brings rain in 1997 and 2015 . banana bandana cabana , aaaa aaa aa a
photographs of photographers photographing photographs . running runner runs ran . testing tested tests tester
""" * 3 + " naïve café über el niño 日本語 のテキスト emoji 🙂🙂 "


def main(n_merges=400):
    b2c = byte_to_char()
    words = collections.Counter()
    for w in CORPUS.lower().split():
        sym = [b2c[b] for b in w.encode("utf-8")]
        sym[-1] += "</w>"
        words[tuple(sym)] += 1
    merges = []
    for _ in range(n_merges):
        pairs = collections.Counter()
        for w, c in words.items():
            for i in range(len(w) - 1):
                pairs[(w[i], w[i + 1])] += c
        if not pairs:
            break
        best = max(sorted(pairs), key=lambda p: pairs[p])
        merges.append(best)
        new = collections.Counter()
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i + 1 < len(w) and (w[i], w[i + 1]) == best:
                    out.append(w[i] + w[i + 1]); i += 2
                else:
                    out.append(w[i]); i += 1
            new[tuple(out)] += c
        words = new
    # two lines in the "wrong" order (a pair that needs a symbol created further down) exercise the tie / order rules
    merges.insert(5, ("aa</w>", "zz"))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bpe_toy_merges.txt")
    with open(path, "w", encoding="utf8") as f:
        f.write("#version: toy (oracle/make_toy_merges.py)\n")
        for a, b in merges:
            f.write(f"{a} {b}\n")
    print(len(merges), "merges ->", path)


if __name__ == "__main__":
    main()
