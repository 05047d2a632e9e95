"""Synthetic PRETRAINED checkpoints in the file formats the reference loads -- test infrastructure shared by
oracle/make_loader_golden.py (which proves them valid by loading them with HuggingFace `ViltModel.from_pretrained` and with the
reference's own `load_albef`) and tests/test_weights*.py (which load them with feddat_amd.weights).  Every tensor is filled
from its checkpoint key (oracle.feddat_oracle.seeded_value on "ckpt::" + key), so the 150-250 MB files never need committing:
both sides regenerate identical bytes.

  write_hf_vilt_checkpoint   a HuggingFace directory as `dandelin/vilt-b32-mlm` is laid out: config.json + model.safetensors
                             (or pytorch_model.bin) of a ViltForMaskedLM -- ViltModel keys under `vilt.`, the MLM head's
                             `mlm_score.*` tensors beside them, 2-row token_type_embeddings (vilt.py:102-113 expands to 3).
  write_albef_checkpoint     ALBEF.pth as published: {'model': state_dict} with the text encoder as a BertForMaskedLM
                             (`text_encoder.bert.*`, `text_encoder.cls.*`), 12-layer numbering with the fusion layers from 6,
                             position embeddings of a SMALLER pre-training resolution, and the tensors load_albef drops
                             (momentum copies, projections, ITM head, temperature, queues)."""
from __future__ import annotations

import json
import os
from typing import Dict

import torch

from oracle import feddat_oracle as O


def _fill(key: str, shape) -> torch.Tensor:
    return O.seeded_value("ckpt::" + key, tuple(shape), 0.02, 0.02)


def hf_vilt_shapes(layers: int, hidden: int = 768, inter: int = 3072, patch: int = 32, grid: int = 12, max_text: int = 40,
                   vocab: int = 30522) -> Dict[str, tuple]:
    """Parameter names / shapes of transformers' ViltForMaskedLM (ViltConfig defaults = dandelin/vilt-b32-mlm)."""
    H, I = hidden, inter
    s: Dict[str, tuple] = {}
    e = "vilt.embeddings."
    s[e + "cls_token"] = (1, 1, H)
    s[e + "position_embeddings"] = (1, grid * grid + 1, H)
    s[e + "text_embeddings.word_embeddings.weight"] = (vocab, H)
    s[e + "text_embeddings.position_embeddings.weight"] = (max_text, H)
    s[e + "text_embeddings.token_type_embeddings.weight"] = (2, H)
    s[e + "text_embeddings.LayerNorm.weight"] = (H,)
    s[e + "text_embeddings.LayerNorm.bias"] = (H,)
    s[e + "patch_embeddings.projection.weight"] = (H, 3, patch, patch)
    s[e + "patch_embeddings.projection.bias"] = (H,)
    s[e + "token_type_embeddings.weight"] = (2, H)
    for i in range(layers):
        L = f"vilt.encoder.layer.{i}."
        for n in ("query", "key", "value"):
            s[L + f"attention.attention.{n}.weight"] = (H, H)
            s[L + f"attention.attention.{n}.bias"] = (H,)
        for n, shp in (("attention.output.dense", (H, H)), ("intermediate.dense", (I, H)), ("output.dense", (H, I))):
            s[L + n + ".weight"] = shp
            s[L + n + ".bias"] = (shp[0],)
        for ln in ("layernorm_before", "layernorm_after"):
            s[L + ln + ".weight"] = (H,)
            s[L + ln + ".bias"] = (H,)
    s["vilt.layernorm.weight"] = (H,)
    s["vilt.layernorm.bias"] = (H,)
    s["vilt.pooler.dense.weight"] = (H, H)
    s["vilt.pooler.dense.bias"] = (H,)
    # MLM head (dropped by ViltModel.from_pretrained)
    s["mlm_score.bias"] = (vocab,)
    s["mlm_score.transform.dense.weight"] = (H, H)
    s["mlm_score.transform.dense.bias"] = (H,)
    s["mlm_score.transform.LayerNorm.weight"] = (H,)
    s["mlm_score.transform.LayerNorm.bias"] = (H,)
    return s


def write_hf_vilt_checkpoint(path: str, layers: int = 2, fmt: str = "safetensors", vocab: int = 30522) -> str:
    os.makedirs(path, exist_ok=True)
    sd = {k: _fill(k, shp).contiguous() for k, shp in hf_vilt_shapes(layers, vocab=vocab).items()}
    cfg = {"architectures": ["ViltForMaskedLM"], "model_type": "vilt", "num_hidden_layers": layers, "hidden_size": 768,
           "intermediate_size": 3072, "num_attention_heads": 12, "image_size": 384, "patch_size": 32, "num_channels": 3,
           "max_position_embeddings": 40, "vocab_size": vocab, "type_vocab_size": 2, "modality_type_vocab_size": 2,
           "layer_norm_eps": 1e-12, "hidden_act": "gelu", "hidden_dropout_prob": 0.0, "attention_probs_dropout_prob": 0.0,
           "qkv_bias": True, "max_image_length": -1, "num_images": -1, "tie_word_embeddings": False,
           "initializer_range": 0.02}
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    if fmt == "safetensors":
        from safetensors.torch import save_file
        save_file(sd, os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
    else:
        torch.save(sd, os.path.join(path, "pytorch_model.bin"))
    return path


def albef_pth_shapes(vit_depth: int, enc_layers: int, pre_image: int, patch: int = 16, hidden: int = 768, inter: int = 3072,
                     vocab: int = 30522, max_pos: int = 512, fusion_layer: int = 6) -> Dict[str, tuple]:
    H, I = hidden, inter
    s: Dict[str, tuple] = {}
    for v in ("visual_encoder.", "visual_encoder_m."):
        s[v + "cls_token"] = (1, 1, H)
        s[v + "pos_embed"] = (1, (pre_image // patch) ** 2 + 1, H)
        s[v + "patch_embed.proj.weight"] = (H, 3, patch, patch)
        s[v + "patch_embed.proj.bias"] = (H,)
        for i in range(vit_depth):
            b = f"{v}blocks.{i}."
            for n, shp in (("norm1", (H,)), ("attn.qkv", (3 * H, H)), ("attn.proj", (H, H)), ("norm2", (H,)),
                           ("mlp.fc1", (I, H)), ("mlp.fc2", (H, I))):
                s[b + n + ".weight"] = shp
                s[b + n + ".bias"] = (shp[0],)
        s[v + "norm.weight"] = (H,)
        s[v + "norm.bias"] = (H,)
        if v.endswith("_m."):
            for k in [k for k in s if k.startswith(v) and "blocks." in k]:       # keep the momentum copy tiny
                del s[k]
    t = "text_encoder.bert."
    e = t + "embeddings."
    s[e + "word_embeddings.weight"] = (vocab, H)
    s[e + "position_embeddings.weight"] = (max_pos, H)
    s[e + "token_type_embeddings.weight"] = (2, H)
    s[e + "LayerNorm.weight"] = (H,)
    s[e + "LayerNorm.bias"] = (H,)
    for i in range(enc_layers):
        L = f"{t}encoder.layer.{i}."
        for blk in (("attention",) + (("crossattention",) if i >= fusion_layer else ())):
            for n in ("query", "key", "value"):
                s[f"{L}{blk}.self.{n}.weight"] = (H, H)
                s[f"{L}{blk}.self.{n}.bias"] = (H,)
            s[f"{L}{blk}.output.dense.weight"] = (H, H)
            s[f"{L}{blk}.output.dense.bias"] = (H,)
            s[f"{L}{blk}.output.LayerNorm.weight"] = (H,)
            s[f"{L}{blk}.output.LayerNorm.bias"] = (H,)
        for n, shp in (("intermediate.dense", (I, H)), ("output.dense", (H, I))):
            s[L + n + ".weight"] = shp
            s[L + n + ".bias"] = (shp[0],)
        s[L + "output.LayerNorm.weight"] = (H,)
        s[L + "output.LayerNorm.bias"] = (H,)
    c = "text_encoder.cls.predictions."
    s[c + "bias"] = (vocab,)
    s[c + "transform.dense.weight"] = (H, H)
    s[c + "transform.dense.bias"] = (H,)
    s[c + "transform.LayerNorm.weight"] = (H,)
    s[c + "transform.LayerNorm.bias"] = (H,)
    # what load_albef's strict=False drops
    s["vision_proj.weight"], s["vision_proj.bias"] = (256, H), (256,)
    s["text_proj.weight"], s["text_proj.bias"] = (256, H), (256,)
    s["itm_head.weight"], s["itm_head.bias"] = (2, H), (2,)
    s["temp"] = ()
    s["image_queue"], s["text_queue"] = (256, 64), (256, 64)
    s["text_encoder_m.bert.embeddings.LayerNorm.weight"] = (H,)
    s["text_encoder_m.bert.encoder.layer.7.output.LayerNorm.bias"] = (H,)
    return s


def write_albef_checkpoint(path: str, vit_depth: int, enc_layers: int, pre_image: int, vocab: int, max_pos: int) -> str:
    shapes = albef_pth_shapes(vit_depth, enc_layers, pre_image, vocab=vocab, max_pos=max_pos)
    sd = {k: _fill(k, shp).contiguous() for k, shp in shapes.items()}
    sd["queue_ptr"] = torch.zeros(1, dtype=torch.long)
    sd["text_encoder.bert.embeddings.position_ids"] = torch.arange(max_pos).expand((1, -1)).clone()
    # the MLM decoder is tied to the word embeddings in the published file
    sd["text_encoder.cls.predictions.decoder.weight"] = sd["text_encoder.bert.embeddings.word_embeddings.weight"]
    sd["text_encoder.cls.predictions.decoder.bias"] = sd["text_encoder.cls.predictions.bias"]
    torch.save({"model": sd, "config": {"note": "synthetic"}, "epoch": 29}, path)
    return path
