"""Pretrained-weight loading from LOCAL files into the state-dict keys of SURVEY.md section 8b.

What the reference does (nothing here touches the network; `--pretrained_model_name` must be a path):

  ViLT   load_vilt_encoder / ViltEncoderWrapper.__init__                   src/modeling/vilt.py:40-53,102-113,387-420
         ViltModel.from_pretrained(dir) -- a HF directory with model.safetensors / pytorch_model.bin whose keys are those
         of ViltModel, or of a head model around it (dandelin/vilt-b32-mlm is a ViltForMaskedLM: `vilt.` prefix, extra
         `mlm_score.*` tensors that from_pretrained drops) -- then expand_modality_type_embeddings(): the 2-row
         token_type_embeddings becomes 3 rows, row 2 a copy of row 1.  Adaptered_ViltOutput wraps every layer's
         ViltOutput (vilt.py:356-361), which renames `encoder.layer.i.output.dense.*` to `...output.layer.dense.*`.
         A ViltEncoderWrapper state dict saved with torch.save (keys `vilt.*`; the else-branch of load_vilt_encoder) loads too.
  ALBEF  load_albef                                                          src/modeling/albef.py:205-241
         torch.load(ALBEF.pth)['model']; interpolate_pos_embed (bicubic, align_corners=False: models/vit.py:193-217) to the
         configured image size; `bert.` stripped from the text-encoder keys; text-encoder layers 6..11, its embeddings and
         its MLM head also initialise the 6-layer text DECODER (layers 0..5, `text_decoder.bert.*`, `text_decoder.cls.*`);
         everything else in the file (momentum copies, projections, ITM head, queues) is dropped by strict=False.

Adapters (adapter.py:5-14,22-58: N(0, 0.02) weights, zero biases) and the per-task heads (vilt.py:202-209: nn.Linear /
nn.LayerNorm defaults) are not part of any pretrained file: init_trainable() creates them.

This module is host-side file conversion (the reference does it on the host, once); no training arithmetic happens here."""
from __future__ import annotations

import json
import math
import os
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import albef_spec, vilt_spec
from .lib import FeddatHipError

ENC = vilt_spec.ENC
PRE = albef_spec.PRE


# --------------------------------------------------------------------------------------------------------------- files
def _read_tensor_file(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return dict(load_file(path))
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:      # noqa: BLE001 -- the restricted unpickler's error is opaque; say what to do
        # the reference calls plain torch.load (albef.py:208): checkpoints that carry non-tensor objects (config, optimizer or
        # scheduler state) are refused by the tensors-only unpickler used here
        raise FeddatHipError(
            f"{path}: torch.load(weights_only=True) refused the file ({type(e).__name__}: {str(e)[:200]}).  Only tensor state "
            "dicts are read (no arbitrary unpickling of a downloaded file); re-save the checkpoint as tensors only, e.g. "
            "torch.save({'model': torch.load(p, weights_only=False)['model']}, out) in an environment you trust, or convert "
            "it to .safetensors") from e
    if isinstance(obj, dict) and "model" in obj and isinstance(obj["model"], dict):      # ALBEF.pth
        obj = obj["model"]
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        obj = obj["state_dict"]
    if not isinstance(obj, dict):
        raise FeddatHipError(f"{path}: expected a state dict, got {type(obj).__name__}")
    return {k: v for k, v in obj.items() if isinstance(v, torch.Tensor)}


def read_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    """A file (.safetensors / .bin / .pt / .pth) or a HuggingFace model directory (model.safetensors, pytorch_model.bin, or
    their sharded forms with an *.index.json)."""
    if not path or not os.path.exists(path):
        raise FeddatHipError(
            f"--pretrained_model_name {path!r} is not a local file or directory.  There is no network access and no silent "
            "random initialisation: pass a local HuggingFace directory / checkpoint file, or omit the flag to run on "
            "random weights of the real architecture (synthetic benchmarks).")
    if os.path.isfile(path):
        return _read_tensor_file(path)
    for name in ("model.safetensors", "pytorch_model.bin"):
        f = os.path.join(path, name)
        if os.path.exists(f):
            return _read_tensor_file(f)
        idx = f + ".index.json"
        if os.path.exists(idx):
            with open(idx) as fh:
                shards = sorted(set(json.load(fh)["weight_map"].values()))
            out: Dict[str, torch.Tensor] = {}
            for s in shards:
                out.update(_read_tensor_file(os.path.join(path, s)))
            return out
    raise FeddatHipError(f"{path}: no model.safetensors / pytorch_model.bin in this directory")


# ------------------------------------------------------------------------------------------------------- trainable init
def _linear_default(gen, out_f, in_f):
    """nn.Linear.reset_parameters: kaiming_uniform(a = sqrt(5)) = U(-1/sqrt(in), 1/sqrt(in)) for weight and bias."""
    b = 1.0 / math.sqrt(in_f)
    w = (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * b
    bias = (torch.rand(out_f, generator=gen) * 2 - 1) * b
    return w, bias


def init_trainable(shapes: Dict[str, tuple], seed: int = 0) -> Dict[str, torch.Tensor]:
    """Fresh adapters and task heads for every such key in `shapes` (the reference builds them at model construction)."""
    gen = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for k, shp in shapes.items():
        if "adapter_" in k:
            out[k] = torch.zeros(shp) if k.endswith("bias") else 0.02 * torch.randn(shp, generator=gen)
        elif k.startswith("task_layer.") and k.endswith("clf_norm0.weight"):
            out[k] = torch.ones(shp)
        elif k.startswith("task_layer.") and k.endswith("clf_norm0.bias"):
            out[k] = torch.zeros(shp)
    for k, shp in shapes.items():
        if k.startswith("task_layer.") and k.endswith(("clf_fc0.weight", "clf_fc1.weight")):
            w, b = _linear_default(gen, shp[0], shp[1])
            out[k], out[k[:-len("weight")] + "bias"] = w, b
    return out


# ----------------------------------------------------------------------------------------------------------------- ViLT
def convert_vilt_state_dict(sd: Dict[str, torch.Tensor], layers: int = 12) -> Dict[str, torch.Tensor]:
    """HF ViltModel / Vilt*Model / ViltEncoderWrapper keys -> `vilt_encoder.vilt.*` keys of the reference's
    ViltContinualLearner (frozen backbone only)."""
    body = {}
    for k, v in sd.items():
        if k.startswith("vilt_encoder.vilt."):
            k = k[len("vilt_encoder.vilt."):]
        elif k.startswith("vilt."):
            k = k[len("vilt."):]
        elif k.split(".")[0] not in ("embeddings", "encoder", "layernorm", "pooler"):
            continue                                    # mlm_score.*, classifier.* ...: dropped by ViltModel.from_pretrained
        if k.endswith("position_ids"):
            continue
        # Adaptered_ViltOutput holds the original ViltOutput as `.layer` (adaptered_output.py:60-70)
        parts = k.split(".")
        if len(parts) > 4 and parts[0] == "encoder" and parts[3] == "output" and parts[4] == "dense":
            k = ".".join(parts[:4] + ["layer"] + parts[4:])
        body[k] = v.to(torch.float32)
    out = {}
    want = [k for k in vilt_spec.param_shapes(layers, ()).keys() if "adapter_" not in k]
    tt = body.get("embeddings.token_type_embeddings.weight")
    if tt is not None and tt.shape[0] == 2:             # expand_modality_type_embeddings (vilt.py:102-113)
        body["embeddings.token_type_embeddings.weight"] = torch.cat([tt, tt[1:2]], 0)
    missing = []
    for full in want:
        k = full[len(ENC):]
        if k not in body:
            missing.append(k)
            continue
        out[full] = body[k].contiguous()
    if missing:
        raise FeddatHipError(f"ViLT checkpoint lacks {len(missing)} backbone tensors, e.g. {missing[:4]}")
    return out


def load_vilt_pretrained(path: str, tasks: Sequence[str], layers: int = 12, seed: int = 0,
                         num_labels: int = 100) -> Dict[str, torch.Tensor]:
    """Everything create_vilt_continual_learner_model needs: frozen backbone from `path`, fresh adapters and heads."""
    sd = convert_vilt_state_dict(read_checkpoint(path), layers)
    shapes = vilt_spec.param_shapes(layers, tasks, num_labels=num_labels)
    for k, shp in shapes.items():
        if k in sd and tuple(sd[k].shape) != tuple(shp):
            # any square grid is accepted for the position table (the engine resizes it per sample)
            if k.endswith("embeddings.position_embeddings") and sd[k].dim() == 3 and sd[k].shape[2] == shp[2] \
                    and int(round(math.sqrt(sd[k].shape[1] - 1))) ** 2 == sd[k].shape[1] - 1:
                continue
            raise FeddatHipError(f"{k}: checkpoint shape {tuple(sd[k].shape)} != expected {tuple(shp)}")
    sd.update(init_trainable({k: v for k, v in shapes.items() if k not in sd}, seed))
    lack = [k for k in shapes if k not in sd]
    if lack:
        raise FeddatHipError(f"no source for {lack[:4]}")
    return sd


# ---------------------------------------------------------------------------------------------------------------- ALBEF
def _cubic_weights(t: np.ndarray, A: float = -0.75):
    """torch's bicubic (upsample_bicubic2d) convolution coefficients for fractional offset t."""
    def c1(x):      # |x| <= 1
        return ((A + 2) * x - (A + 3)) * x * x + 1
    def c2(x):      # 1 < |x| < 2
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    return np.stack([c2(t + 1), c1(t), c1(1 - t), c2(2 - t)], -1)


def _bicubic_axis(x: np.ndarray, new: int, axis: int) -> np.ndarray:
    """Bicubic resize along one axis, align_corners=False, indices clamped to the border (torch semantics), fp32."""
    old = x.shape[axis]
    scale = np.float32(old) / np.float32(new)
    src = (np.arange(new, dtype=np.float32) + np.float32(0.5)) * scale - np.float32(0.5)
    i0 = np.floor(src).astype(np.int64)
    w = _cubic_weights((src - i0).astype(np.float32)).astype(np.float32)          # [new, 4]
    x = np.moveaxis(x, axis, 0)
    out = np.zeros((new,) + x.shape[1:], np.float32)
    for j in range(4):
        idx = np.clip(i0 - 1 + j, 0, old - 1)
        out += x[idx] * w[:, j].reshape((new,) + (1,) * (x.ndim - 1))
    return np.moveaxis(out, 0, axis)


def interpolate_pos_embed(pos: torch.Tensor, new_tokens: int, extra_tokens: int = 1) -> torch.Tensor:
    """models/vit.py:193-217: [1, extra + g*g, H] -> [1, extra + n*n, H], bicubic over the 2-D grid, class token kept."""
    n_old = pos.shape[-2] - extra_tokens
    g, n = int(n_old ** 0.5), int((new_tokens - extra_tokens) ** 0.5)
    if g == n:
        return pos
    grid = pos[0, extra_tokens:].to(torch.float32).numpy().reshape(g, g, -1)
    grid = _bicubic_axis(_bicubic_axis(grid, n, 1), n, 0)          # torch separates width first, then height
    return torch.cat([pos[:, :extra_tokens].to(torch.float32), torch.from_numpy(grid.reshape(1, n * n, -1))], 1)


def convert_albef_state_dict(sd: Dict[str, torch.Tensor], image: int = 384, patch: int = 16, **dims) -> Dict[str, torch.Tensor]:
    """ALBEF.pth['model'] keys -> `albef_model.albef.*` keys of the reference's ALBEFContinualLearner, with the key surgery
    of load_albef (albef.py:216-236)."""
    sd = {k: v for k, v in sd.items()}
    n_tok = (image // patch) ** 2 + 1
    if "visual_encoder.pos_embed" in sd:
        sd["visual_encoder.pos_embed"] = interpolate_pos_embed(sd["visual_encoder.pos_embed"], n_tok)
    for key in list(sd.keys()):
        if "bert" in key:
            sd[key.replace("bert.", "")] = sd[key]
        if "text_encoder" in key:
            if "layer" in key:
                parts = key.split(".")
                n = int(parts[4])
                if n < 6:
                    del sd[key]
                    continue
                parts[4] = str(n - 6)
                enc_key = ".".join(parts)
            else:
                enc_key = key
            sd[enc_key.replace("text_encoder", "text_decoder")] = sd[key]
            del sd[key]
    shapes = albef_spec.param_shapes(image=image, patch=patch, **dims)
    out, missing = {}, []
    for full, shp in shapes.items():
        if "adapter_" in full:
            continue
        k = full[len(PRE):]
        if k not in sd:
            missing.append(k)
            continue
        v = sd[k].to(torch.float32).contiguous()
        if tuple(v.shape) != tuple(shp):
            raise FeddatHipError(f"{k}: checkpoint shape {tuple(v.shape)} != expected {tuple(shp)}")
        out[full] = v
    if missing:
        raise FeddatHipError(f"ALBEF checkpoint lacks {len(missing)} tensors, e.g. {missing[:4]}")
    # the LM head's decoder matrix is tied to the decoder's word embeddings (xbert.py BertLMHeadModel); a file in which the
    # two differ cannot be represented
    dec_w = sd.get("text_decoder.cls.predictions.decoder.weight")
    emb_w = out.get(PRE + "text_decoder.bert.embeddings.word_embeddings.weight")
    if dec_w is not None and emb_w is not None and not torch.equal(dec_w.to(torch.float32), emb_w):
        raise FeddatHipError("text_decoder.cls.predictions.decoder.weight differs from the tied word embeddings")
    return out


def load_albef_pretrained(path: str, seed: int = 0, image: int = 384, **dims) -> Dict[str, torch.Tensor]:
    sd = convert_albef_state_dict(read_checkpoint(path), image=image, **dims)
    shapes = albef_spec.param_shapes(image=image, **dims)
    sd.update(init_trainable({k: v for k, v in shapes.items() if k not in sd}, seed))
    lack = [k for k in shapes if k not in sd]             # every expected key has a source (as in load_vilt_pretrained)
    if lack:
        raise FeddatHipError(f"no source for {lack[:4]}")
    return sd


def resolve(pretrained: Optional[str]):
    """None -> None (caller uses random weights of the real architecture); a string must be an existing local path."""
    if pretrained is None or pretrained == "":
        return None
    if not os.path.exists(pretrained):
        raise FeddatHipError(
            f"--pretrained_model_name {pretrained!r}: no such local file or directory (hub names cannot be fetched: there is "
            "no network).  Training does NOT fall back to random weights when a checkpoint was asked for.")
    return pretrained
