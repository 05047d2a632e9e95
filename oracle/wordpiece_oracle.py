"""CPU restatement of the text half of the reference's input pipeline -- TEST INFRASTRUCTURE ONLY (imported by tests/ and
never by feddat_amd/).

Reference call sites: ViltProcessor(images, texts, padding=True, truncation=True, max_length=40) in
src/modeling/vilt.py:98 and BertTokenizer(..., padding='longest', truncation=True, max_length=25) in
src/modeling/albef.py:56-57.  The tokenizer itself is third-party (HuggingFace transformers / tokenizers, not vendored):
BERT's BasicTokenizer (clean text, lower-case, NFD accent stripping, CJK spacing, punctuation splitting) followed by greedy
longest-match-first WordPiece ('##' continuation prefix, [UNK] for words longer than 100 characters or with an
unmatched remainder), then [CLS] tokens[:max_length - 2] [SEP], padding with [PAD] to the longest row of the batch.
Pinned against `tokenizers.BertWordPieceTokenizer` on a synthetic vocabulary (tests/golden/g9_wordpiece.npz, written
by oracle/make_wordpiece_golden.py): the real bert-base-uncased vocab.txt is not available offline, the algorithm is."""
import unicodedata
from typing import Dict, List, Sequence

import numpy as np


def _is_punct(ch: str) -> bool:
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_ws(ch: str) -> bool:
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch: str) -> bool:
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


def _is_cjk(cp: int) -> bool:
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


def normalize(text: str) -> str:
    """BertNormalizer(clean_text, handle_chinese_chars, strip_accents, lowercase): the result is what the pre-tokenizer
    splits on whitespace / punctuation."""
    out = []
    for ch in text:
        cp = ord(ch)
        if cp == 0 or cp == 0xFFFD or _is_control(ch):
            continue
        out.append(" " if _is_ws(ch) else ch)
    text = "".join(out)
    text = "".join(f" {c} " if _is_cjk(ord(c)) else c for c in text)
    text = unicodedata.normalize("NFD", text)
    text = "".join(c for c in text if unicodedata.category(c) != "Mn")
    return text.lower()


def basic_tokens(text: str) -> List[str]:
    words, cur = [], []
    for ch in normalize(text):
        if _is_ws(ch):
            if cur:
                words.append("".join(cur))
                cur = []
        elif _is_punct(ch):
            if cur:
                words.append("".join(cur))
                cur = []
            words.append(ch)
        else:
            cur.append(ch)
    if cur:
        words.append("".join(cur))
    return words


def wordpiece(word: str, vocab: Dict[str, int], unk: str = "[UNK]", max_chars: int = 100) -> List[str]:
    if len(word) > max_chars:
        return [unk]
    pieces, start = [], 0
    while start < len(word):
        end, cur = len(word), None
        while start < end:
            sub = ("##" if start > 0 else "") + word[start:end]
            if sub in vocab:
                cur = sub
                break
            end -= 1
        if cur is None:
            return [unk]
        pieces.append(cur)
        start = end
    return pieces


def encode_batch(texts: Sequence[str], vocab: Dict[str, int], max_length: int = 40, pad_to: int = None):
    """-> input_ids, attention_mask, token_type_ids (int64 [B, L]); L = longest row of the batch (padding=True) or
    pad_to."""
    rows = []
    for t in texts:
        toks = [p for w in basic_tokens(t) for p in wordpiece(w, vocab)]
        ids = [vocab["[CLS]"]] + [vocab[p] for p in toks[:max_length - 2]] + [vocab["[SEP]"]]
        rows.append(ids)
    L = pad_to or max(len(r) for r in rows)
    ids = np.full((len(rows), L), vocab["[PAD]"], np.int64)
    mask = np.zeros((len(rows), L), np.int64)
    for i, r in enumerate(rows):
        ids[i, :len(r)] = r
        mask[i, :len(r)] = 1
    return ids, mask, np.zeros_like(ids)
