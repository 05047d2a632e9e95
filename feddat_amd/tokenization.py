"""Device-side mirror of the text half of the reference's input pipeline: BERT WordPiece as ViltProcessor(text=...,
padding=True, truncation=True, max_length=40) (src/modeling/vilt.py:98) / BertTokenizer(..., padding='longest',
truncation=True, max_length=25) (src/modeling/albef.py:56-57) produce it, run once per batch by
`feddat_wordpiece_encode` instead of three times per batch on the host.

    tok = WordPieceTokenizer("./models/bert-base-uncased/vocab.txt", device)
    enc = tok(questions, padding=True, truncation=True, max_length=40)     # {'input_ids', 'attention_mask', 'token_type_ids'}

ASCII questions go to the device as raw bytes.  A question with non-ASCII or control characters is first normalised on the
host with the BertNormalizer rules that need Unicode tables (clean-up, CJK and non-ASCII punctuation spacing, NFD accent
stripping, lower-casing); the kernel does the rest.  The encodings match HuggingFace's tokenizer on every fixture text
(tests/golden/g9_wordpiece.npz); the vocabulary file itself is the caller's (the reference loads it from disk, there is no
network here)."""
from __future__ import annotations

import ctypes as C
import unicodedata
from typing import Dict, Iterable, List, Sequence, Union

import numpy as np
import torch

from . import lib as L


def _needs_host_normalisation(s: str) -> bool:
    return (not s.isascii()) or any((ord(c) < 32 and c not in "\t\n\r") or ord(c) == 127 for c in s)


def _is_cjk(cp: int) -> bool:
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


def host_normalise(text: str) -> str:
    """BertNormalizer for the texts the kernel does not take raw: drop control characters, fold Unicode whitespace, put
    spaces around CJK characters and non-ASCII punctuation, strip accents (NFD, category Mn), lower-case."""
    out: List[str] = []
    for ch in text:
        cp = ord(ch)
        cat = unicodedata.category(ch)
        if cp == 0 or cp == 0xFFFD or (cat.startswith("C") and ch not in "\t\n\r"):
            continue
        if ch in " \t\n\r" or cat == "Zs":
            out.append(" ")
        elif _is_cjk(cp):
            out.append(f" {ch} ")
        else:
            out.append(ch)
    s = unicodedata.normalize("NFD", "".join(out))
    s = "".join(c for c in s if unicodedata.category(c) != "Mn").lower()
    return "".join(f" {c} " if (ord(c) > 127 and unicodedata.category(c).startswith("P")) else c for c in s)


class WordPieceTokenizer:
    model_input_names = ["input_ids", "token_type_ids", "attention_mask"]

    def __init__(self, vocab: Union[str, Sequence[str]], device, unk_token="[UNK]", cls_token="[CLS]", sep_token="[SEP]",
                 pad_token="[PAD]"):
        L.load()
        self.device = torch.device(device)
        if isinstance(vocab, str):
            with open(vocab, encoding="utf-8") as f:
                vocab = [ln.rstrip("\n") for ln in f]
        self.vocab = list(vocab)
        index = {t: i for i, t in reversed(list(enumerate(self.vocab)))}
        try:
            self.unk_id, self.cls_id, self.sep_id, self.pad_id = (index[t] for t in (unk_token, cls_token, sep_token,
                                                                                     pad_token))
        except KeyError as e:
            raise L.FeddatHipError(f"vocabulary has no {e.args[0]} token") from None
        enc = [t.encode("utf-8") for t in self.vocab]
        offs = np.zeros(len(enc) + 1, np.int64)
        np.cumsum([len(b) for b in enc], out=offs[1:])
        blob = b"".join(enc)
        self.entries = int(L.load().feddat_wordpiece_table_entries(len(enc)))
        table = np.zeros(self.entries * 16, np.uint8)
        rc = L.load().feddat_wordpiece_table_build(C.c_char_p(blob), offs.ctypes.data_as(C.c_void_p), len(enc),
                                                   table.ctypes.data_as(C.c_void_p), self.entries)
        L._chk(rc, "feddat_wordpiece_table_build")
        self.table = torch.from_numpy(table).to(self.device)

    def __call__(self, text: Union[str, Iterable[str]], padding: Union[bool, str] = True, truncation: bool = True,
                 max_length: int = 40, return_tensors: str = "pt") -> Dict[str, torch.Tensor]:
        texts = [text] if isinstance(text, str) else list(text)
        if not truncation:
            raise L.FeddatHipError("the device tokenizer always truncates to max_length (the reference does: vilt.py:98)")
        raw = [(host_normalise(t) if _needs_host_normalisation(t) else t).encode("utf-8") for t in texts]
        offs = np.zeros(len(raw) + 1, np.int64)
        np.cumsum([len(b) for b in raw], out=offs[1:])
        blob = np.frombuffer(b"".join(raw) + b"\0", np.uint8)
        d_blob = torch.from_numpy(blob.copy()).to(self.device)
        d_offs = torch.from_numpy(offs).to(self.device)
        n = len(raw)
        ids = torch.empty(n, max_length, dtype=torch.int64, device=self.device)
        mask = torch.empty(n, max_length, dtype=torch.int64, device=self.device)
        lens = torch.empty(n, dtype=torch.int32, device=self.device)
        rc = L.load().feddat_wordpiece_encode(L._p(d_blob), L._p(d_offs), n, L._p(self.table), self.entries, self.unk_id,
                                              self.cls_id, self.sep_id, self.pad_id, max_length, L._p(ids), L._p(mask),
                                              L._p(lens), L._stream())
        L._chk(rc, "feddat_wordpiece_encode")
        if padding in (True, "longest"):       # the reference pads to the longest question of the batch (one host sync)
            lens_h = lens.cpu()
            if int(lens_h.min()) < 0:
                raise L.FeddatHipError("feddat_wordpiece_encode refused a text (longer than 2048 bytes)")
            longest = int(lens_h.max())
            ids, mask = ids[:, :longest].contiguous(), mask[:, :longest].contiguous()
        elif padding != "max_length":
            raise L.FeddatHipError("padding must be True / 'longest' / 'max_length'")
        return {"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": mask, "lengths": lens}
