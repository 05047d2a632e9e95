"""Drop-in mirrors of the reference's module API for the hot path, backed by libfeddat_hip.so.

  reference                                              here
  src/modeling/models/adapter.py        Adapter                   -> Adapter
  src/modeling/adaptered_output.py      Adaptered_ViltOutput      -> Adaptered_ViltOutput
  src/modeling/vilt.py                  ViltContinualLearner      -> ViltContinualLearner (create_vilt_continual_learner_model)

Same names, same call signatures, same mode-switch semantics (including the requires_grad side effects of
set_active_adapter that decide optimizer membership, adapter.py:66-95).  Tensors are device tensors; every forward
is a HIP kernel launch through the C ABI -- there is no PyTorch arithmetic and no CPU path (a missing library or a
CPU tensor raises FeddatHipError).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from . import lib as L
from .engine import ViltDatEngine, ENC


class _Param:
    """Minimal stand-in for nn.Parameter: .data is a device tensor (a view into an engine buffer or an own
    allocation), .requires_grad is the flag the reference toggles."""

    def __init__(self, data: torch.Tensor, requires_grad: bool = True):
        self.data = data
        self.requires_grad = requires_grad

    @property
    def shape(self):
        return self.data.shape


class _Linear:
    def __init__(self, weight: torch.Tensor, bias: torch.Tensor):
        self.weight = _Param(weight)
        self.bias = _Param(bias)

    def parameters(self):
        return [self.weight, self.bias]


class Adapter:
    """adapter.py:16-163.  Bottleneck adapters `adapter_{i}_{down,up}` (model_dim -> model_dim/16 -> model_dim,
    ReLU); forward(hidden_states, input_tensor) in single-adapter or fixed 0.5/0.5 "gating" mode."""

    def __init__(self, names, device, model_dim: int = 768, adapter_reduction_factor: int = 16, _views=None):
        L.load()
        if model_dim != 768 or adapter_reduction_factor != 16:
            raise L.FeddatHipError("the HIP adapter kernel is specialised for model_dim=768, reduction=16 "
                                   "(the only configuration the reference ever instantiates, adapter.py:22)")
        self.actv = "relu"
        self.scaling = 1.0
        self.gating = False
        if isinstance(names, str):
            names = [names]
        self.names = [n for n in names if "adapter" in n]
        self.device = torch.device(device)
        r = model_dim // adapter_reduction_factor
        # the module belongs to the operand format (library) the constructing thread is bound to: lib.operands(...)
        self._fmt = L.current_operands()
        op = L.OPERAND_DTYPE[self._fmt]
        self._packs: Dict[str, dict] = {}
        for n in self.names:
            if _views is not None:
                wd, bd, wu, bu = _views[n]
            else:  # init_bert_weights: N(0, 0.02) weights, zero biases (adapter.py:5-14)
                wd = torch.randn(r, model_dim, device=self.device) * 0.02
                bd = torch.zeros(r, device=self.device)
                wu = torch.randn(model_dim, r, device=self.device) * 0.02
                bu = torch.zeros(model_dim, device=self.device)
            setattr(self, f"{n}_down", _Linear(wd, bd))
            setattr(self, f"{n}_up", _Linear(wu, bu))
            self._packs[n] = dict(
                wd=torch.empty(r, model_dim, dtype=op, device=self.device),
                wdT=torch.empty(model_dim, r, dtype=op, device=self.device),
                wu=torch.empty(model_dim, r, dtype=op, device=self.device),
                wuT=torch.empty(r, model_dim, dtype=op, device=self.device), bd=bd, bu=bu)
        if hasattr(self, "adapter_2_down"):                      # adapter.py:55-58
            for m in (self.adapter_2_down, self.adapter_2_up):
                for p in m.parameters():
                    p.requires_grad = False
        self.refresh()

    def refresh(self):
        """Re-derive the 16-bit operand copies after the fp32 weights changed."""
        with L.operands(self._fmt):
            for n, p in self._packs.items():
                L.adapter_pack(getattr(self, f"{n}_down").weight.data, getattr(self, f"{n}_up").weight.data, p["wd"],
                               p["wdT"], p["wu"], p["wuT"])

    def deactivate_gating(self):
        self.gating = False

    def activate_gating(self):
        self.gating = True

    def set_active_adapter(self, name):                           # adapter.py:66-95
        if isinstance(name, str):
            self.active_adapter_down = getattr(self, f"{name}_down")
            self.active_adapter_up = getattr(self, f"{name}_up")
            self._active = name

        def flag(n, v):
            for m in (getattr(self, f"{n}_down"), getattr(self, f"{n}_up")):
                for p in m.parameters():
                    p.requires_grad = v
        if name == "adapter_0":
            flag("adapter_0", True)
            flag("adapter_1", False)
        elif name == "adapter_1":
            flag("adapter_1", True)
            flag("adapter_0", False)
        elif isinstance(name, list):
            for n in name:
                flag(n, True)

    def named_parameters(self):
        for n in self.names:
            for part in ("down", "up"):
                lin = getattr(self, f"{n}_{part}")
                yield f"{n}_{part}.weight", lin.weight
                yield f"{n}_{part}.bias", lin.bias

    def forward(self, hidden_states: torch.Tensor, input_tensor: torch.Tensor) -> torch.Tensor:
        """adapter.py:124-163.  The reference always calls adapter(h, h) (adaptered_output.py:77); the fused kernel
        implements exactly that form, so a distinct input_tensor is rejected loudly."""
        if hidden_states.data_ptr() != input_tensor.data_ptr():
            raise L.FeddatHipError("fused adapter kernel requires input_tensor is hidden_states "
                                   "(adaptered_output.py:77 is the only call site)")
        x = hidden_states
        if x.dtype != torch.float32 or not x.is_contiguous():
            raise L.FeddatHipError("adapter forward expects a contiguous fp32 device tensor [..., 768]")
        T = x.numel() // 768
        if not self.gating:
            ads = [dict(self._packs[self._active], scale=1.0)]
        elif hasattr(self, "adapter_2_down"):
            ads = [dict(self._packs["adapter_0"], scale=0.5 * self.scaling),
                   dict(self._packs["adapter_2"], scale=0.5 * self.scaling)]
        else:
            ads = [dict(self._packs["adapter_0"], scale=0.5 * self.scaling),
                   dict(self._packs["adapter_1"], scale=0.5 * self.scaling)]
        out = torch.empty_like(x)
        with L.operands(self._fmt):
            L.adapter_fwd(x.view(T, 768), out.view(T, 768), L.make_segs([dict(row_begin=0, row_end=T, adapters=ads)]), T)
        return out

    __call__ = forward


class Adaptered_ViltOutput:
    """adaptered_output.py:67-78: h = dense(x) (+ dropout p=0) + input_tensor; return adapter(h, h).
    `layer` supplies the frozen ViltOutput dense as .dense.weight [768,3072] / .dense.bias [768] device tensors."""

    def __init__(self, layer, adapter_config) -> None:
        self.layer = layer
        self.adapter = Adapter(**adapter_config, model_dim=768)
        w = layer.dense.weight.data if hasattr(layer.dense.weight, "data") else layer.dense.weight
        self._fmt = self.adapter._fmt
        self._w16 = torch.empty(w.shape, dtype=L.OPERAND_DTYPE[self._fmt], device=w.device)
        with L.operands(self._fmt):
            L.cvt_f32_bf16(w.contiguous().float(), self._w16)
        b = layer.dense.bias
        self._b = (b.data if hasattr(b, "data") else b).contiguous().float()

    def forward(self, hidden_states: torch.Tensor, input_tensor: torch.Tensor) -> torch.Tensor:
        x = hidden_states.reshape(-1, hidden_states.shape[-1])
        res = input_tensor.reshape(-1, 768).contiguous().float()
        h = torch.empty_like(res)
        with L.operands(self._fmt):
            if x.dtype != self._w16.dtype:
                x16 = torch.empty(x.shape, dtype=self._w16.dtype, device=x.device)
                L.cvt_f32_bf16(x.contiguous().float(), x16)
                x = x16
            L.gemm_bf16_nt(x, self._w16, L.EPI_RESID_F32, bias=self._b, resid=res, out_f32=h)
        return self.adapter(h, h).view(input_tensor.shape)

    __call__ = forward


class ViltContinualLearner:
    """vilt.py:154-382 for the classification / single-image path, on the HIP engine.

    forward(task_key, images, texts) -> (pooled, logits) takes the reference's batch schema: `images` = list of decoded
    RGB images ([H, W, 3] uint8 arrays / tensors, or PIL images), `texts` = list of questions (vilt.py:244-264); they go
    through process_inputs -- the device image processor + device WordPiece tokenizer, ONCE per batch (the reference runs
    its host processor inside every one of the three forward passes of a train_step).  A ready-made HF ViLT encodings
    dict (pixel_values, pixel_mask, input_ids, attention_mask, token_type_ids) in `images` with texts=None is taken as is
    (the golden harness and the synthetic benchmarks feed tensors).

    vocab: path of the BERT vocab.txt (the reference loads ./models/bert-base-uncased, vilt.py:47-48) or the token list."""

    BERT_LOCAL_PATH = "./models/bert-base-uncased"       # vilt.py:47

    def __init__(self, ordered_cl_tasks: List[str], params: Dict[str, torch.Tensor], device, batch_size: int,
                 image_size: int = 384, num_layers: int = 12, lr: float = 1e-4, vocab=None, operands: str = None):
        """operands: 16-bit MFMA operand format of the engine, "f16" (default; the reference's mixed_precision: fp16,
        accelerate_config.yaml:8) or "bf16" (engine.ViltDatEngine)."""
        self.ordered_cl_tasks = list(ordered_cl_tasks)
        self.device = torch.device(device)
        self.engine = ViltDatEngine(params, self.ordered_cl_tasks, self.device, batch=batch_size, res=image_size,
                                    layers=num_layers, lr=lr, operands=operands)
        self.max_text_length = self.engine.Lt               # vilt.py:50: config.max_position_embeddings = 40
        self._vocab = vocab
        self._tokenizer = None
        self._image_processor = None
        self.gating = False
        self.active = "adapter_1"
        # requires_grad flags per adapter, toggled exactly like adapter.py:66-95; prepare_model's initial state
        # (main.py:157-159 + adapter.py:55-58): adapter_0/1 trainable, adapter_2 frozen
        self.adapter_requires_grad = {0: True, 1: True, 2: False}
        self.comm_state_dict_names = [n for n in self.state_dict() if "adapter_1" in n]   # main.py:160-163

    # ---- adapter switches (vilt.py:363-373) ----
    def set_active_adapter(self, name):
        self.active = name
        if name == "adapter_0":
            self.adapter_requires_grad[0], self.adapter_requires_grad[1] = True, False
        elif name == "adapter_1":
            self.adapter_requires_grad[1], self.adapter_requires_grad[0] = True, False

    def activate_gating(self):
        self.gating = True

    def deactivate_gating(self):
        self.gating = False

    def optimizer_adapters(self) -> Sequence[int]:
        """Which adapters a freshly created optimizer would hold (create_optimizer filters on requires_grad)."""
        return tuple(a for a in (0, 1) if self.adapter_requires_grad[a])

    # ---- state dict with the reference's keys ----
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

    # ---- the reference's batch schema (vilt.py:87-100): images + questions -> HF ViLT encodings, on the device ----
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
                        f"texts were given but there is no BERT vocabulary: pass vocab=<vocab.txt path or token list> to "
                        f"ViltContinualLearner, or place bert-base-uncased at {self.BERT_LOCAL_PATH} as the reference does")
            self._tokenizer = WordPieceTokenizer(vocab, self.device)
        return self._tokenizer

    @property
    def image_processor(self):
        if self._image_processor is None:
            from .image_processing import ViltImageProcessor
            # every batch is padded to the engine's static frame; patches outside an image's valid rectangle are masked keys
            self._image_processor = ViltImageProcessor(self.device, pad_to=self.engine.res)
        return self._image_processor

    def process_inputs(self, images: List, texts: List[str]) -> Dict[str, torch.Tensor]:
        """ViltEncoderWrapper.process_inputs (vilt.py:87-100): ViltProcessor(images, text, max_length=40, padding=True,
        truncation=True) -> {pixel_values, pixel_mask, input_ids, attention_mask, token_type_ids}, as device tensors in the
        engine's static frame: images resized by the ViLT rule (shorter edge 384, longer <= 640, multiples of 32) and
        zero-padded to the frame with their pixel_mask, questions padded to max_text_length with attention_mask 0 -- the
        pooled feature does not depend on how far a batch is padded (masked keys; golden G6)."""
        import numpy as np
        arrs = [np.asarray(im.convert("RGB")) if hasattr(im, "convert") else im for im in images]
        if len(arrs) != len(texts):
            raise L.FeddatHipError(f"{len(arrs)} images but {len(texts)} texts")
        enc = dict(self.image_processor(arrs))
        tok = self.tokenizer(list(texts), padding="max_length", truncation=True, max_length=self.max_text_length)
        enc.update(input_ids=tok["input_ids"], attention_mask=tok["attention_mask"], token_type_ids=tok["token_type_ids"])
        return enc

    def forward(self, task_key: str, images, texts=None):
        mode = "gating" if self.gating else self.active
        if not isinstance(images, dict):
            images = self.process_inputs(images, texts)
        return self.engine.forward(images, mode, task_key)

    __call__ = forward


def create_vilt_continual_learner_model(params: Dict[str, torch.Tensor], ordered_cl_tasks: List[str], device,
                                        batch_size: int, image_size: int = 384, num_layers: int = 12,
                                        lr: float = 1e-4, vocab=None, operands: str = None) -> ViltContinualLearner:
    """vilt.py:421-452 (the pretrained checkpoint is passed in as a tensor dict: feddat_amd.weights.load_vilt_pretrained
    reads it from a local HF directory; there is no hub access here)."""
    return ViltContinualLearner(ordered_cl_tasks, params, device, batch_size, image_size, num_layers, lr, vocab=vocab,
                                operands=operands)


def convert_batch_to_vilt_input_dict(batch: Dict):
    """vilt.py:455-459: what batch_collate yields -> the arguments of ViltContinualLearner.forward / process_inputs."""
    return {"images": batch["images"], "texts": batch["raw_texts"]}
