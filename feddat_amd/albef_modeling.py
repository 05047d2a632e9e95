"""Drop-in mirror of the reference's ALBEF wrapper API (src/modeling/albef.py:88-193) on the MI355X engine.

    model = create_albef_continual_learner_model(params, device, batch_size, n_answers, ...)
    model.activate_gating(); model.set_active_adapter('adapter_0')      # albef.py:139-171 fan-out over the 30 Adapter modules
    loss, logits = model(task_key, batch)        # batch['train'] = True : ALBEF.forward(train=True)  (albef.py:52-60)
    ids, probs   = model(task_key, batch)        # batch['train'] = False: rank_answer over batch['answer_list'] (albef.py:61-73)

Batches are the reference's (convert_batch_to_albef_input_dict, albef.py:275-286): {"images": [B,3,R,R] tensor, "questions":
[str], "answers": [str], "weights": [n], "n": answers per question, "alpha", "train"} -- tokenised ONCE per batch by the device
WordPiece tokenizer (the reference's BertTokenizer runs inside every forward pass: albef.py:56-57,62-64) and embedded in the
engine's static frame -- or already-tokenised dicts (question_ids / question_mask, answer_ids / answer_mask, weights, k), of
any length up to the frame (AlbefDatEngine.set_batch)."""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch

from . import lib as L
from .albef_engine import AlbefDatEngine


class ALBEFContinualLearner:
    BERT_LOCAL_PATH = "./models/bert-base-uncased"       # albef.py:38

    def __init__(self, params: Dict[str, torch.Tensor], device, batch_size: int, n_answers: int, q_len: int = 25,
                 a_len: int = 4, lr: float = 1e-4, tokenizer_vocab=None, **dims):
        self.device = torch.device(device)
        self._vocab, self._tokenizer = tokenizer_vocab, None
        self.engine = AlbefDatEngine(params, self.device, batch=batch_size, n_answers=n_answers, q_len=q_len, a_len=a_len,
                                     lr=lr, **dims)
        self.gating, self.active = False, "adapter_1"
        # prepare_model's initial state (main.py:157-159 + adapter.py:55-58): adapter_0/1 trainable, adapter_2 frozen
        self.adapter_requires_grad = {0: True, 1: True, 2: False}
        self.comm_state_dict_names = [n for n in self.state_dict() if "adapter_1" in n]   # main.py:160-163

    def set_active_adapter(self, name):          # albef.py:139-147 -> adapter.py:60-85 (requires_grad toggling included)
        self.active = name
        if name == "adapter_0":
            self.adapter_requires_grad[0], self.adapter_requires_grad[1] = True, False
        elif name == "adapter_1":
            self.adapter_requires_grad[1], self.adapter_requires_grad[0] = True, False

    def activate_gating(self):                   # albef.py:161-169
        self.gating = True

    def deactivate_gating(self):                 # albef.py:150-158
        self.gating = False

    def optimizer_adapters(self) -> Sequence[int]:
        return tuple(a for a in (0, 1) if self.adapter_requires_grad[a])

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return self.engine.state_dict()

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = False):
        own = self.engine.state_dict()
        unknown = [k for k in sd if k not in own]
        if strict and unknown:
            raise KeyError(f"unexpected keys: {unknown[:3]}")
        self.engine.load_tensors({k: v for k, v in sd.items() if k in own})

    def after_load(self):
        for a in range(3):
            self.engine.repack_adapter(a)

    # ---- the reference's batch schema: questions / answers as strings (albef.py:52-73) ----
    @property
    def tokenizer(self):
        if self._tokenizer is None:
            import os
            from .tokenization import WordPieceTokenizer
            vocab = self._vocab
            if vocab is None:
                vocab = os.path.join(self.BERT_LOCAL_PATH, "vocab.txt")
                if not os.path.exists(vocab):
                    raise L.FeddatHipError(
                        f"questions / answers were given as strings but there is no BERT vocabulary: pass tokenizer_vocab=<vocab.txt "
                        f"path or token list>, or place bert-base-uncased at {self.BERT_LOCAL_PATH} as the reference does")
            self._tokenizer = WordPieceTokenizer(vocab, self.device)
        return self._tokenizer

    def _tok(self, texts: List[str], limit: int, truncate: bool, what: str):
        """padding='longest' inside the engine's frame: -> (ids [n, longest], mask); without truncation a text longer than
        `limit` tokens is an error (the reference's tokenizer would hand ALBEF a longer tensor than the engine holds)."""
        enc = self.tokenizer(texts, padding="max_length", truncation=True, max_length=limit if truncate else limit + 1)
        longest = int(enc["lengths"].max())
        if not truncate and longest > limit:
            raise L.FeddatHipError(f"{what} longer than the engine's frame ({limit} tokens)")
        return enc["input_ids"][:, :longest].contiguous(), enc["attention_mask"][:, :longest].contiguous(), enc["lengths"]

    def process_inputs(self, batch: Dict) -> Dict:
        """ALBEFWrapper.forward's tokenisation (albef.py:52-73) once per batch on the device: TRAIN questions padded to the
        longest and truncated at 25 tokens (albef.py:56), answers padded to the longest; EVAL questions are tokenised WITHOUT
        truncation (albef.py:62 passes no max_length) -- one longer than the engine's frame is an error, not a silent cut that
        would change rank_answer's result; the eval path appends [SEP] to every entry of the answer list (albef.py:63) and
        tokenises it the same way."""
        eng = self.engine
        out = {"image": batch["images"].to(self.device, torch.float32, non_blocking=True), "train": batch.get("train", True)}
        if out["train"]:
            qi, qm, _ = self._tok(list(batch["questions"]), min(25, eng.Lq), True, "question")
        else:
            qi, qm, _ = self._tok(list(batch["questions"]), eng.Lq, False, "eval question (the reference does not truncate it)")
        out["question_ids"], out["question_mask"] = qi, qm
        if out["train"]:
            ai, am, _ = self._tok(list(batch["answers"]), eng.La, False, "answer")
            out.update(answer_ids=ai, answer_mask=am, k=list(batch["n"]),
                       weights=batch["weights"].to(self.device, torch.float32, non_blocking=True))
            if "alpha" in batch:
                out["alpha"] = batch["alpha"]
        else:
            # "<answer>[SEP]" -> [CLS] pieces [SEP] [SEP]: the literal [SEP] is the special token (id 102), then the tokenizer's own
            ai, am, lens = self._tok(list(batch["answer_list"]), eng.La - 1, False, "answer-list entry")
            ids = torch.full((ai.shape[0], eng.La), self.tokenizer.pad_id, dtype=torch.int64, device=self.device)
            msk = torch.zeros_like(ids)
            ids[:, :ai.shape[1]], msk[:, :am.shape[1]] = ai, am
            pos = lens.long()[:, None]
            ids.scatter_(1, pos, self.tokenizer.sep_id)
            msk.scatter_(1, pos, 1)
            longest = int(lens.max()) + 1
            out.update(answer_list_ids=ids[:, :longest].contiguous(), answer_list_mask=msk[:, :longest].contiguous(),
                       k=int(batch["k"]))
        return out

    def forward(self, task_key: str, batch: Dict):
        mode = "gating" if self.gating else self.active
        if "questions" in batch:
            batch = self.process_inputs(batch)
        if batch.get("train", True):
            loss, logits = self.engine.forward_train_logits(batch, mode)
            return [loss, logits]
        if "answer_list_ids" not in batch:
            raise L.FeddatHipError("eval batches need answer_list_ids / answer_list_mask (tokenised answer list) and k")
        ids, probs = self.engine.rank_answer(batch, batch["answer_list_ids"], batch["answer_list_mask"], int(batch["k"]), mode)
        return [ids, probs]

    __call__ = forward


def create_albef_continual_learner_model(params: Dict[str, torch.Tensor], device, batch_size: int, n_answers: int,
                                         q_len: int = 25, a_len: int = 4, lr: float = 1e-4, tokenizer_vocab=None,
                                         **dims) -> ALBEFContinualLearner:
    """albef.py:255-273 (the ALBEF.pth checkpoint is passed in as a tensor dict: feddat_amd.weights.load_albef_pretrained
    reads it from a local file; there is no hub access here)."""
    return ALBEFContinualLearner(params, device, batch_size, n_answers, q_len, a_len, lr, tokenizer_vocab=tokenizer_vocab, **dims)


def convert_batch_to_albef_input_dict(batch):
    """albef.py:275-286: the collated list -> the dict ALBEFContinualLearner.forward / AlbefTaskTrainer.train_step take."""
    return {"images": batch[0], "questions": batch[1], "answers": batch[2], "weights": batch[3], "n": batch[4],
            "alpha": batch[5] if len(batch) > 5 else 0.0}
