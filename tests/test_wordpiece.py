"""Text half of the input pipeline: the CPU oracle against HuggingFace's own WordPiece implementation (fixture G9, CPU),
and the device tokenizer against both (GPU)."""
import numpy as np
import pytest
import torch

from oracle import wordpiece_oracle as W
from tests.golden_util import load

SEP = "\x1e"


def _fixture(golden_dir):
    g = load(golden_dir, "g9_wordpiece.npz")
    vocab = str(g["vocab"]).split("\n")
    texts = str(g["texts"]).split(SEP)
    return g, vocab, texts


@pytest.mark.parametrize("max_len", [40, 25])
def test_oracle_matches_huggingface_wordpiece(golden_dir, max_len):
    g, vocab, texts = _fixture(golden_dir)
    ids, mask, tt = W.encode_batch(texts, {t: i for i, t in enumerate(vocab)}, max_len)
    assert ids.shape == g[f"ids{max_len}"].shape
    assert (ids == g[f"ids{max_len}"]).all() and (mask == g[f"mask{max_len}"]).all() and not tt.any()


def test_oracle_edge_cases():
    vocab = {t: i for i, t in enumerate(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "a", "##b", "ab", "?", "##c"])}
    assert W.wordpiece("abc", vocab) == ["ab", "##c"]          # longest match first, not "a ##b ##c"
    assert W.wordpiece("abd", vocab) == ["[UNK]"]              # unmatched remainder -> the whole word is unknown
    assert W.wordpiece("a" * 101, vocab) == ["[UNK]"]
    ids, mask, _ = W.encode_batch(["", "ab?"], vocab, 8)
    assert ids.tolist() == [[2, 3, 0, 0], [2, 6, 7, 3]] and mask.tolist() == [[1, 1, 0, 0], [1, 1, 1, 1]]


@pytest.mark.gpu
@pytest.mark.parametrize("max_len", [40, 25])
def test_device_tokenizer_matches_huggingface_and_oracle(golden_dir, max_len):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from feddat_amd.tokenization import WordPieceTokenizer
    g, vocab, texts = _fixture(golden_dir)
    tok = WordPieceTokenizer(vocab, "cuda")
    enc = tok(texts, padding=True, truncation=True, max_length=max_len)
    assert enc["input_ids"].dtype == torch.int64 and enc["input_ids"].is_cuda
    assert np.array_equal(enc["input_ids"].cpu().numpy(), g[f"ids{max_len}"])
    assert np.array_equal(enc["attention_mask"].cpu().numpy(), g[f"mask{max_len}"])
    assert not enc["token_type_ids"].any()
    # fixed-frame padding (what the static-shape engine consumes): same tokens, [PAD] up to max_length
    enc2 = tok(texts, padding="max_length", max_length=max_len)
    ids_o, mask_o, _ = W.encode_batch(texts, {t: i for i, t in enumerate(vocab)}, max_len, pad_to=max_len)
    assert np.array_equal(enc2["input_ids"].cpu().numpy(), ids_o)
    assert np.array_equal(enc2["attention_mask"].cpu().numpy(), mask_o)


@pytest.mark.gpu
def test_device_tokenizer_long_words_and_batches():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from feddat_amd.tokenization import WordPieceTokenizer
    rng = np.random.default_rng(0)
    letters = "abcdefghij"
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "?", ","] + list(letters) + ["##" + c for c in letters]
    # random multi-letter pieces, incl. long ones (candidate ends beyond one 64-lane round)
    for _ in range(300):
        n = int(rng.integers(2, 90))
        w = "".join(rng.choice(list(letters), n))
        vocab.append(w if rng.random() < 0.5 else "##" + w)
    vocab = list(dict.fromkeys(vocab))
    index = {t: i for i, t in enumerate(vocab)}
    texts = []
    for _ in range(64):
        ws = []
        for _ in range(int(rng.integers(1, 12))):
            n = int(rng.integers(1, 120))
            ws.append("".join(rng.choice(list(letters), n)))
        texts.append(" ".join(ws) + "?")
    tok = WordPieceTokenizer(vocab, "cuda")
    got = tok(texts, padding="max_length", max_length=40)
    ids_o, mask_o, _ = W.encode_batch(texts, index, 40, pad_to=40)
    assert np.array_equal(got["input_ids"].cpu().numpy(), ids_o)
    assert np.array_equal(got["attention_mask"].cpu().numpy(), mask_o)
