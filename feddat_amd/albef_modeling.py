"""Drop-in mirror of the reference's ALBEF wrapper API (src/modeling/albef.py:88-193) on the MI355X engine.

    model = create_albef_continual_learner_model(params, device, batch_size, n_answers, ...)
    model.activate_gating(); model.set_active_adapter('adapter_0')      # albef.py:139-171 fan-out over the 30 Adapter modules
    loss, logits = model(task_key, batch)        # batch['train'] = True : ALBEF.forward(train=True)  (albef.py:52-60)
    ids, probs   = model(task_key, batch)        # batch['train'] = False: rank_answer over batch['answer_list'] (albef.py:61-73)

Batches carry the encodings the reference obtains from its BertTokenizer (question_ids / question_mask, answer_ids /
answer_mask; feddat_amd.tokenization.WordPieceTokenizer produces them on the device) instead of strings."""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch

from . import lib as L
from .albef_engine import AlbefDatEngine


class ALBEFContinualLearner:
    def __init__(self, params: Dict[str, torch.Tensor], device, batch_size: int, n_answers: int, q_len: int = 25,
                 a_len: int = 4, lr: float = 1e-4, **dims):
        self.device = torch.device(device)
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

    def forward(self, task_key: str, batch: Dict):
        mode = "gating" if self.gating else self.active
        if batch.get("train", True):
            loss, logits = self.engine.forward_train_logits(batch, mode)
            return [loss, logits]
        if "answer_list_ids" not in batch:
            raise L.FeddatHipError("eval batches need answer_list_ids / answer_list_mask (tokenised answer list) and k")
        ids, probs = self.engine.rank_answer(batch, batch["answer_list_ids"], batch["answer_list_mask"], int(batch["k"]), mode)
        return [ids, probs]

    __call__ = forward


def create_albef_continual_learner_model(params: Dict[str, torch.Tensor], device, batch_size: int, n_answers: int,
                                         q_len: int = 25, a_len: int = 4, lr: float = 1e-4, **dims) -> ALBEFContinualLearner:
    """albef.py:255-273 (the ALBEF.pth checkpoint is passed in as a tensor dict: there is no hub / disk access here)."""
    return ALBEFContinualLearner(params, device, batch_size, n_answers, q_len, a_len, lr, **dims)
