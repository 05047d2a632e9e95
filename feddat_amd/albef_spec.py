"""State-dict keys / shapes of the reference's ALBEF dual-adapter model (ALBEFContinualLearner.albef_model.albef:
VisionTransformer with an Adapter per block, src/modeling/models/vit.py:79-110; BertModel / BertLMHeadModel whose BertOutput
holds an Adapter, src/modeling/models/xbert.py:429-445; built by src/modeling/models/albef_model.py:13-43) and a random
initialiser of that architecture (no network for ALBEF.pth / bert-base-uncased here: benchmarks use random-init weights of the
real shapes).  The LM-head decoder weight is the decoder's word-embedding matrix (tied), so it has no entry of its own."""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import torch

PRE = "albef_model.albef."


def param_shapes(vit_depth: int = 12, enc_layers: int = 12, fusion_layer: int = 6, dec_layers: int = 6, image: int = 384,
                 patch: int = 16, hidden: int = 768, inter: int = 3072, vocab: int = 30522, max_pos: int = 512,
                 bottleneck: int = 48) -> Dict[str, Tuple[int, ...]]:
    H, I, r = hidden, inter, bottleneck
    s: Dict[str, Tuple[int, ...]] = {}

    def adapters(base):
        for a in range(3):
            s[f"{base}adapter_{a}_down.weight"] = (r, H)
            s[f"{base}adapter_{a}_down.bias"] = (r,)
            s[f"{base}adapter_{a}_up.weight"] = (H, r)
            s[f"{base}adapter_{a}_up.bias"] = (H,)

    v = PRE + "visual_encoder."
    s[v + "cls_token"] = (1, 1, H)
    s[v + "pos_embed"] = (1, (image // patch) ** 2 + 1, H)
    s[v + "patch_embed.proj.weight"] = (H, 3, patch, patch)
    s[v + "patch_embed.proj.bias"] = (H,)
    for i in range(vit_depth):
        b = f"{v}blocks.{i}."
        for n, shp in (("norm1.weight", (H,)), ("norm1.bias", (H,)), ("attn.qkv.weight", (3 * H, H)), ("attn.qkv.bias", (3 * H,)),
                       ("attn.proj.weight", (H, H)), ("attn.proj.bias", (H,)), ("norm2.weight", (H,)), ("norm2.bias", (H,)),
                       ("mlp.fc1.weight", (I, H)), ("mlp.fc1.bias", (I,)), ("mlp.fc2.weight", (H, I)), ("mlp.fc2.bias", (H,))):
            s[b + n] = shp
        adapters(b + "adapter.")
    s[v + "norm.weight"] = (H,)
    s[v + "norm.bias"] = (H,)
    for tower, layers, fusion in ((PRE + "text_encoder.", enc_layers, fusion_layer), (PRE + "text_decoder.bert.", dec_layers, 0)):
        e = tower + "embeddings."
        s[e + "word_embeddings.weight"] = (vocab, H)
        s[e + "position_embeddings.weight"] = (max_pos, H)
        s[e + "token_type_embeddings.weight"] = (2, H)
        s[e + "LayerNorm.weight"] = (H,)
        s[e + "LayerNorm.bias"] = (H,)
        for i in range(layers):
            Lp = f"{tower}encoder.layer.{i}."
            for blk in (("attention",) + (("crossattention",) if i >= fusion else ())):
                for n in ("query", "key", "value"):
                    s[f"{Lp}{blk}.self.{n}.weight"] = (H, H)
                    s[f"{Lp}{blk}.self.{n}.bias"] = (H,)
                s[f"{Lp}{blk}.output.dense.weight"] = (H, H)
                s[f"{Lp}{blk}.output.dense.bias"] = (H,)
                s[f"{Lp}{blk}.output.LayerNorm.weight"] = (H,)
                s[f"{Lp}{blk}.output.LayerNorm.bias"] = (H,)
            s[Lp + "intermediate.dense.weight"] = (I, H)
            s[Lp + "intermediate.dense.bias"] = (I,)
            s[Lp + "output.dense.weight"] = (H, I)
            s[Lp + "output.dense.bias"] = (H,)
            s[Lp + "output.LayerNorm.weight"] = (H,)
            s[Lp + "output.LayerNorm.bias"] = (H,)
            adapters(Lp + "output.adapter.")
    c = PRE + "text_decoder.cls.predictions."
    s[c + "bias"] = (vocab,)
    s[c + "transform.dense.weight"] = (H, H)
    s[c + "transform.dense.bias"] = (H,)
    s[c + "transform.LayerNorm.weight"] = (H,)
    s[c + "transform.LayerNorm.bias"] = (H,)
    return s


def random_init(seed: int = 0, device="cpu", std: float = 0.02, bias_std: float = 0.02, **dims) -> Dict[str, torch.Tensor]:
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = {}
    for k, shp in param_shapes(**dims).items():
        is_ln = ("LayerNorm" in k) or (".norm" in k)
        if is_ln and k.endswith("weight"):
            out[k] = 1.0 + bias_std * torch.randn(shp, generator=g, device=device)
        elif k.endswith("bias"):
            out[k] = bias_std * torch.randn(shp, generator=g, device=device)
        else:
            out[k] = std * torch.randn(shp, generator=g, device=device)
    return out


def synthetic_batch(B: int, seed: int, image: int = 384, q_len: int = 25, a_len: int = 4, k: Sequence[int] = None,
                    vocab: int = 30522, device="cpu"):
    """N(0,1) images, questions of q_len tokens ([CLS] ... [SEP]), one (or k[b]) answers per
    question of a_len tokens ([CLS] a b [SEP]), weights 1 -- the reference's batch after tokenisation (albef.py:52-60)."""
    g = torch.Generator().manual_seed(seed)
    k = list(k) if k is not None else [1] * B
    n = sum(k)
    q = torch.randint(1000, min(30000, vocab), (B, q_len), generator=g)
    q[:, 0], q[:, -1] = 101, 102
    a = torch.randint(1000, min(30000, vocab), (n, a_len), generator=g)
    a[:, 0], a[:, -1] = 101, 102
    batch = {"image": torch.randn(B, 3, image, image, generator=g), "question_ids": q,
             "question_mask": torch.ones(B, q_len, dtype=torch.long), "answer_ids": a,
             "answer_mask": torch.ones(n, a_len, dtype=torch.long), "weights": torch.ones(n)}
    out = {kk: v.to(device) for kk, v in batch.items()}
    out["k"] = k
    return out
