"""State-dict keys / shapes of the reference model on the hot path (ViltContinualLearner with
Adaptered_ViltOutput in every layer: src/modeling/vilt.py:154-219,356-361; src/modeling/models/adapter.py:22-58;
HF ViltModel with ViltConfig defaults = dandelin/vilt-b32-mlm) and a random initialiser of that architecture
(there is no network for checkpoints: benchmarks use random-init weights of the real shapes)."""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import torch

ENC = "vilt_encoder.vilt."


def param_shapes(layers: int = 12, tasks: Sequence[str] = ("art",), hidden: int = 768, inter: int = 3072,
                 patch: int = 32, grid: int = 12, max_text: int = 40, vocab: int = 30522, num_labels: int = 100,
                 bottleneck: int = 48) -> Dict[str, Tuple[int, ...]]:
    H, I, r = hidden, inter, bottleneck
    s: Dict[str, Tuple[int, ...]] = {}
    e = ENC + "embeddings."
    s[e + "cls_token"] = (1, 1, H)
    s[e + "position_embeddings"] = (1, grid * grid + 1, H)
    s[e + "text_embeddings.word_embeddings.weight"] = (vocab, H)
    s[e + "text_embeddings.position_embeddings.weight"] = (max_text, H)
    s[e + "text_embeddings.token_type_embeddings.weight"] = (2, H)
    s[e + "text_embeddings.LayerNorm.weight"] = (H,)
    s[e + "text_embeddings.LayerNorm.bias"] = (H,)
    s[e + "patch_embeddings.projection.weight"] = (H, 3, patch, patch)
    s[e + "patch_embeddings.projection.bias"] = (H,)
    s[e + "token_type_embeddings.weight"] = (3, H)
    for i in range(layers):
        L = ENC + f"encoder.layer.{i}."
        for n in ("query", "key", "value"):
            s[L + f"attention.attention.{n}.weight"] = (H, H)
            s[L + f"attention.attention.{n}.bias"] = (H,)
        s[L + "attention.output.dense.weight"] = (H, H)
        s[L + "attention.output.dense.bias"] = (H,)
        s[L + "intermediate.dense.weight"] = (I, H)
        s[L + "intermediate.dense.bias"] = (I,)
        s[L + "output.layer.dense.weight"] = (H, I)
        s[L + "output.layer.dense.bias"] = (H,)
        for a in range(3):
            A = L + f"output.adapter.adapter_{a}_"
            s[A + "down.weight"] = (r, H)
            s[A + "down.bias"] = (r,)
            s[A + "up.weight"] = (H, r)
            s[A + "up.bias"] = (H,)
        for ln in ("layernorm_before", "layernorm_after"):
            s[L + ln + ".weight"] = (H,)
            s[L + ln + ".bias"] = (H,)
    s[ENC + "layernorm.weight"] = (H,)
    s[ENC + "layernorm.bias"] = (H,)
    s[ENC + "pooler.dense.weight"] = (H, H)
    s[ENC + "pooler.dense.bias"] = (H,)
    for t in tasks:
        s[f"task_layer.{t}.clf_fc0.weight"] = (2 * H, H)
        s[f"task_layer.{t}.clf_fc0.bias"] = (2 * H,)
        s[f"task_layer.{t}.clf_norm0.weight"] = (2 * H,)
        s[f"task_layer.{t}.clf_norm0.bias"] = (2 * H,)
        s[f"task_layer.{t}.clf_fc1.weight"] = (num_labels, 2 * H)
        s[f"task_layer.{t}.clf_fc1.bias"] = (num_labels,)
    return s


def random_init(layers: int = 12, tasks: Sequence[str] = ("art",), seed: int = 0, device="cpu",
                std: float = 0.02, bias_std: float = 0.02) -> Dict[str, torch.Tensor]:
    """BERT-style init (adapter.py:5-14: N(0, 0.02) weights, LayerNorm 1/0) -- biases get a small N(0, bias_std)
    instead of exact zeros so that every bias path of the kernels is exercised."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = {}
    for k, shp in param_shapes(layers, tasks).items():
        is_ln = ("LayerNorm" in k) or ("layernorm" in k) or ("clf_norm0" in k)
        if is_ln and k.endswith("weight"):
            out[k] = 1.0 + bias_std * torch.randn(shp, generator=g, device=device)
        elif k.endswith("bias"):
            out[k] = bias_std * torch.randn(shp, generator=g, device=device)
        else:
            out[k] = std * torch.randn(shp, generator=g, device=device)
    return out


def client_label_prior(client: int, num_labels: int = 100, alpha: float = 0.5, seed: int = 1234) -> torch.Tensor:
    """Heterogeneous clients: client k draws its answers from its own Dirichlet(alpha = 0.5) prior over
    the labels -- a few answers dominate each client, differently per client, so the clients' gradients disagree the way
    non-IID VQA domains do (the reference's clients are different datasets: vqa_utils.py:21-31,62-67 build the targets)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed + 7919 * int(client))
    p = torch._standard_gamma(torch.full((num_labels,), float(alpha)), generator=g)
    p = p.clamp_min(1e-12)
    return p / p.sum()


def synthetic_batch(B: int, res: int, seed: int, device="cpu", text_len: int = 40, num_labels: int = 100,
                    label_prior: torch.Tensor = None):
    """Synthetic VQA batch in the reference's schema (HF ViLT encodings + target_scores):
    N(0,1) pixels, [CLS] 38 random ids [SEP], 1-3 labels per row with scores in {0.3, 0.6, 0.9, 1.0}; labels uniform, or
    drawn without replacement from `label_prior` (client_label_prior: the heterogeneous clients of config 3)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    px = torch.randn(B, 3, res, res, generator=g)
    ids = torch.randint(1000, 30000, (B, text_len), generator=g)
    ids[:, 0], ids[:, -1] = 101, 102
    target = torch.zeros(B, num_labels)
    scores = torch.tensor([0.3, 0.6, 0.9, 1.0])
    for b in range(B):
        n = int(torch.randint(1, 4, (1,), generator=g))
        if label_prior is None:
            labs = torch.randperm(num_labels, generator=g)[:n]
        else:
            labs = torch.multinomial(label_prior, n, replacement=False, generator=g)
        target[b, labs] = scores[torch.randint(0, 4, (n,), generator=g)]
    batch = {"pixel_values": px, "pixel_mask": torch.ones(B, res, res, dtype=torch.long), "input_ids": ids,
             "attention_mask": torch.ones(B, text_len, dtype=torch.long),
             "token_type_ids": torch.zeros(B, text_len, dtype=torch.long), "target_scores": target}
    return {k: v.to(device) for k, v in batch.items()}
