"""tests/golden/g9_wordpiece.npz: a synthetic BERT-style vocabulary + VQA-like questions encoded by HuggingFace's own
WordPiece implementation (`tokenizers.BertWordPieceTokenizer`, the engine behind BertTokenizerFast that the reference's
ViltProcessor / BertTokenizer resolve to), lower-casing on, truncation to max_length 40 (src/modeling/vilt.py:98) and 25
(src/modeling/albef.py:56).  Run in the build container:  python oracle/make_wordpiece_golden.py"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tokenizers import BertWordPieceTokenizer  # noqa: E402

SEP = "\x1e"      # record separator between the texts in the fixture

WORDS = """what is the color of a an are there how many people in this picture photo image on man woman wearing doing
kind type animal food sport room does do can you see where which who why left right top bottom white black red blue
green yellow brown gray orange pink purple one two three four five six seven eight nine ten yes no table chair dog cat
horse bird tree sky water grass street sign car bus train plane boat bike person child holding eating sitting standing
playing looking made used shown visible time day night weather sunny cloudy rain snow number name brand shape pattern
material wood metal glass plastic painting sculpture artist style century abstract figure background foreground""".split()
SUFFIXES = ["##s", "##ing", "##ed", "##er", "##est", "##ly", "##es", "##n", "##t", "##al", "##ion", "##y", "##e",
            "##a", "##b", "##c", "##d", "##g", "##h", "##i", "##k", "##l", "##m", "##o", "##p", "##r", "##u", "##w", "##x",
            "##z", "##0", "##1", "##2", "##5", "##th", "##ness", "##able", "##ful", "##ment"]
SINGLES = list("abcdefghijklmnopqrstuvwxyz0123456789") + list("?!.,;:'\"()-/&%$#@*+=<>[]{}_`~^|\\")
SPECIAL = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"]


def build_vocab():
    v = list(SPECIAL)
    for t in SINGLES + WORDS + SUFFIXES + ["un", "re", "pre", "play", "paint", "walk", "graph", "##graph", "cafe",
                                           "naive", "resume", "中", "文", "##ß"]:
        if t not in v:
            v.append(t)
    return v


def questions(n, seed=7):
    rnd = random.Random(seed)
    qs = ["What is the color of the cat?", "How many people are in this picture?", "Is there a dog on the left?",
          "what's the man's name", "  Which   artist painted\tthis?  ", "UNAFFABLE zzzqqq playing!!", "",
          "Café naïve résumé 中文 ok", "a" * 101 + " the", "is it 10:30 a.m. or 5pm (approx.)?",
          "the-cat/dog_bird&co", "x y�z   tab\there\x01now", "straße wood"]
    while len(qs) < n:
        k = rnd.randint(3, 60)
        ws = []
        for _ in range(k):
            w = rnd.choice(WORDS)
            r = rnd.random()
            if r < 0.15:
                w = w + rnd.choice(["s", "ing", "ed", "er", "ly", "ness", "able"])
            elif r < 0.2:
                w = w.upper()
            elif r < 0.25:
                w = w + rnd.choice("?!.,;:")
            elif r < 0.28:
                w = "".join(rnd.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(rnd.randint(1, 12)))
            ws.append(w)
        qs.append(" ".join(ws) + rnd.choice(["?", "", " ?", "??"]))
    return qs


def main():
    out = os.path.join(ROOT, "tests", "golden")
    vocab = build_vocab()
    path = "/tmp/g9_vocab.txt"
    open(path, "w", encoding="utf-8").write("\n".join(vocab) + "\n")
    tok = BertWordPieceTokenizer(path, lowercase=True)
    qs = questions(96)
    rec = {"vocab": np.array("\n".join(vocab)), "texts": np.array(SEP.join(qs))}
    for max_len in (40, 25):
        tok.enable_truncation(max_length=max_len)
        enc = [tok.encode(q).ids for q in qs]
        L = max(len(e) for e in enc)
        ids = np.zeros((len(enc), L), np.int64)
        mask = np.zeros((len(enc), L), np.int64)
        for i, e in enumerate(enc):
            ids[i, :len(e)] = e
            mask[i, :len(e)] = 1
        rec[f"ids{max_len}"], rec[f"mask{max_len}"] = ids, mask
    np.savez_compressed(os.path.join(out, "g9_wordpiece.npz"), **rec)
    print("wrote g9_wordpiece.npz:", len(vocab), "vocab entries,", len(qs), "texts")


if __name__ == "__main__":
    main()
