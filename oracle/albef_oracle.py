"""CPU fp32 restatement of the reference's ALBEF dual-adapter path (configs[3]) -- TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by feddat_amd/.

Pinned against fixtures captured from the reference's own modules (oracle/make_albef_golden.py ->
tests/golden/g10_albef*.npz): VisionTransformer / Block (src/modeling/models/vit.py:99-110,176-190), BertModel /
BertLayer / BertOutput with the double-LayerNorm adapter variant (src/modeling/models/xbert.py:429-445,448-525;
src/modeling/models/adapter.py:97-116), BertLMHeadModel's shifted token loss (xbert.py:1283-1297), ALBEF.forward /
rank_answer (src/modeling/models/albef_model.py:69-156,171-228) and the dat branch of TaskTrainer.train_step
(src/train/visionlanguage_tasks/task_trainer.py:280-330 with the ALBEF loss wiring :296-297,316-317 and the vocabulary-axis
KL :506-516).

Parameters live in a flat dict keyed by the reference's state-dict names (prefix `albef_model.albef.`); the LM-head
decoder weight is the decoder's word-embedding matrix (HF tie_word_embeddings, as transformers 4.16.2 -- the reference's
pin -- ties it in init_weights) and `cls.predictions.decoder.bias` is `cls.predictions.bias`."""
from __future__ import annotations

import math
import re
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

from oracle import feddat_oracle as O

PRE = "albef_model.albef."
Params = Dict[str, torch.Tensor]


class AlbefDims:
    def __init__(self, vit_depth=12, enc_layers=12, fusion_layer=6, dec_layers=6, image=384, patch=16, hidden=768,
                 inter=3072, heads=12, vocab=30522, max_pos=512, bottleneck=48, pad_id=0):
        self.vit_depth, self.enc_layers, self.fusion_layer, self.dec_layers = vit_depth, enc_layers, fusion_layer, dec_layers
        self.image, self.patch, self.hidden, self.inter, self.heads = image, patch, hidden, inter, heads
        self.vocab, self.max_pos, self.r, self.pad_id = vocab, max_pos, bottleneck, pad_id
        self.n_img = (image // patch) ** 2 + 1


def _adapter_shapes(s, base, H, r):
    for a in range(3):
        s[f"{base}adapter_{a}_down.weight"] = (r, H)
        s[f"{base}adapter_{a}_down.bias"] = (r,)
        s[f"{base}adapter_{a}_up.weight"] = (H, r)
        s[f"{base}adapter_{a}_up.bias"] = (H,)


def _bert_layer_shapes(s, L, H, I, r, cross):
    for blk in (("attention",) + (("crossattention",) if cross else ())):
        for n in ("query", "key", "value"):
            s[f"{L}{blk}.self.{n}.weight"] = (H, H)
            s[f"{L}{blk}.self.{n}.bias"] = (H,)
        s[f"{L}{blk}.output.dense.weight"] = (H, H)
        s[f"{L}{blk}.output.dense.bias"] = (H,)
        s[f"{L}{blk}.output.LayerNorm.weight"] = (H,)
        s[f"{L}{blk}.output.LayerNorm.bias"] = (H,)
    s[L + "intermediate.dense.weight"] = (I, H)
    s[L + "intermediate.dense.bias"] = (I,)
    s[L + "output.dense.weight"] = (H, I)
    s[L + "output.dense.bias"] = (H,)
    s[L + "output.LayerNorm.weight"] = (H,)
    s[L + "output.LayerNorm.bias"] = (H,)
    _adapter_shapes(s, L + "output.adapter.", H, r)


def param_shapes(d: AlbefDims) -> Dict[str, tuple]:
    H, I, r = d.hidden, d.inter, d.r
    s: Dict[str, tuple] = {}
    v = PRE + "visual_encoder."
    s[v + "cls_token"] = (1, 1, H)
    s[v + "pos_embed"] = (1, d.n_img, H)
    s[v + "patch_embed.proj.weight"] = (H, 3, d.patch, d.patch)
    s[v + "patch_embed.proj.bias"] = (H,)
    for i in range(d.vit_depth):
        b = f"{v}blocks.{i}."
        for n, shp in (("norm1.weight", (H,)), ("norm1.bias", (H,)), ("attn.qkv.weight", (3 * H, H)), ("attn.qkv.bias", (3 * H,)),
                       ("attn.proj.weight", (H, H)), ("attn.proj.bias", (H,)), ("norm2.weight", (H,)), ("norm2.bias", (H,)),
                       ("mlp.fc1.weight", (I, H)), ("mlp.fc1.bias", (I,)), ("mlp.fc2.weight", (H, I)), ("mlp.fc2.bias", (H,))):
            s[b + n] = shp
        _adapter_shapes(s, b + "adapter.", H, r)
    s[v + "norm.weight"] = (H,)
    s[v + "norm.bias"] = (H,)
    for tower, layers, fusion in ((PRE + "text_encoder.", d.enc_layers, d.fusion_layer),
                                  (PRE + "text_decoder.bert.", d.dec_layers, 0)):
        e = tower + "embeddings."
        s[e + "word_embeddings.weight"] = (d.vocab, H)
        s[e + "position_embeddings.weight"] = (d.max_pos, H)
        s[e + "token_type_embeddings.weight"] = (2, H)
        s[e + "LayerNorm.weight"] = (H,)
        s[e + "LayerNorm.bias"] = (H,)
        for i in range(layers):
            _bert_layer_shapes(s, f"{tower}encoder.layer.{i}.", H, I, r, i >= fusion)
    c = PRE + "text_decoder.cls.predictions."
    s[c + "bias"] = (d.vocab,)
    s[c + "transform.dense.weight"] = (H, H)
    s[c + "transform.dense.bias"] = (H,)
    s[c + "transform.LayerNorm.weight"] = (H,)
    s[c + "transform.LayerNorm.bias"] = (H,)
    return s


def make_params(d: AlbefDims, std: float = 0.02, bias_std: float = 0.02) -> Params:
    """Name-seeded deterministic fill (same generator as the ViLT oracle): identical tensors in the golden generator, the
    tests and the engine without shipping 1.3 GB."""
    return {k: O.seeded_value(k, shp, std, bias_std) for k, shp in param_shapes(d).items()}


def trainable_names(P: Params, adapter: int) -> List[str]:
    return [n for n in P if f"adapter_{adapter}_" in n]


def comm_names(P: Params) -> List[str]:
    return [n for n in P if "adapter_1" in n]          # main.py:160-163


# ------------------------------------------------------------------------------------------------ adapters
def _ad(P, base, a):
    return tuple(P[f"{base}adapter_{a}_{t}"] for t in ("down.weight", "down.bias", "up.weight", "up.bias"))


def adapter(P: Params, base: str, h, inp, mode: str):
    """Adapter.forward (adapter.py:124-163): single adapter or the fixed 0.5 / 0.5 mix of adapter_0 and adapter_2."""
    if mode == "gating":
        return O.adapter_gated(h, inp, _ad(P, base, 0), _ad(P, base, 2))
    return O.adapter_single(h, inp, *_ad(P, base, int(mode.split("_")[1])))


def adapter_layer_forward_bert(P: Params, base: str, dense_out, inp, ln_w, ln_b, eps, mode: str):
    """adapter.py:97-116: residual = dense_out; x = LN(dense_out + inp); y = residual + A(x); out = LN(y + inp) -- the same
    LayerNorm applied twice."""
    x = F.layer_norm(dense_out + inp, (inp.shape[-1],), ln_w, ln_b, eps)
    y = adapter(P, base, x, dense_out, mode)
    return F.layer_norm(y + inp, (inp.shape[-1],), ln_w, ln_b, eps)


# ------------------------------------------------------------------------------------------------ train-mode dropout
# The reference trains under model.train() with hidden_dropout_prob = attention_probs_dropout_prob = 0.1
# (src/configs/model_configs.py:44-46) in the two BERT towers: after the embedding LayerNorm (xbert.py:216), on the attention
# probabilities (:333), on BertSelfOutput's dense output (:360, self- and cross-attention blocks) and on BertOutput's dense
# output ahead of the adapter (:440).  The ViT has no dropout (vit.py:120: all rates 0).  torch's generator cannot be
# reproduced on the device, so masks are COUNTER-BASED here and in libfeddat_hip.so (include/feddat_hip.h,
# feddat_attn2_fwd_dropout): element idx of the tensor a site drops is kept iff
#     fmix32(fmix32(idx * 0x9E3779B1 + key0) + key1 + step * 0x632BE5AB) >= p * 2^32,
# (key0, key1) = splitmix64(seed, pass, site).  pass = 0 / 1 / 2 for the P0 / P1 / P2 forward of one train_step
# (task_trainer.py:283-287,290-295,311-315: three independent draws), step = train_steps since the start of the local
# update.  SITE PLACEMENT is pinned to the reference: oracle/make_albef_golden.py runs the reference's own modules with
# nn.Dropout.forward replaced by this mask function keyed on the MODULE's name (fixture g12).
class _Drop:
    p, seed, step, pass_id = 0.0, 0, 0, 0


DROP = _Drop()
_KINDS = {"emb": 0, "self_probs": 1, "self_out": 2, "cross_probs": 3, "cross_out": 4, "out": 5}
_M32 = 0xFFFFFFFF


def dropout_site(name: str, kind: str) -> int:
    """Site number of a dropout module from any parameter / module name inside it: tower (0 encoder, 1 decoder), layer, kind."""
    tower = 1 if "text_decoder" in name else 0
    m = re.search(r"layer\.(\d+)\.", name)
    return (tower * 64 + (int(m.group(1)) if m else 0)) * 8 + _KINDS[kind]


def dropout_keys(seed: int, pass_id: int, site: int):
    M = (1 << 64) - 1
    z = (seed * 0x9E3779B97F4A7C15 + pass_id * 0xBF58476D1CE4E5B9 + site * 0x94D049BB133111EB + 0x2545F4914F6CDD1D) & M
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & M
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & M
    z ^= z >> 31
    return z & _M32, z >> 32


def _fmix32(x):
    x = x ^ (x >> 16)
    x = (x * 0x85EBCA6B) & _M32
    x = x ^ (x >> 13)
    x = (x * 0xC2B2AE35) & _M32
    return x ^ (x >> 16)


def dropout_keep(numel: int, p: float, key0: int, key1: int, step: int) -> torch.Tensor:
    """bool [numel]: the keep mask (int64 arithmetic masked to 32 bits = the device's uint32 wrap-around)."""
    idx = torch.arange(numel, dtype=torch.int64)
    x = _fmix32((idx * 0x9E3779B1 + key0) & _M32)
    x = _fmix32((x + ((key1 + step * 0x632BE5AB) & _M32)) & _M32)
    return x >= int(float(torch.tensor(p, dtype=torch.float32)) * 4294967296.0)


def drop(x, name: str, kind: str):
    """nn.Dropout(p) in train mode at the site named by (name, kind): x * keep / (1 - p)."""
    if DROP.p <= 0:
        return x
    k0, k1 = dropout_keys(DROP.seed, DROP.pass_id, dropout_site(name, kind))
    keep = dropout_keep(x.numel(), DROP.p, k0, k1, DROP.step).view(x.shape)
    scale = torch.tensor(1.0, dtype=torch.float32) / (torch.tensor(1.0, dtype=torch.float32) - torch.tensor(DROP.p, dtype=torch.float32))
    return x * (keep.to(x.dtype) * scale)


# ------------------------------------------------------------------------------------------------ ViT-B/16
def _mha(q, k, v, heads, add_mask=None, pdrop=None):
    B, Sq, H = q.shape
    Skv = k.shape[1]
    hd = H // heads
    q = q.view(B, Sq, heads, hd).transpose(1, 2)
    k = k.view(B, Skv, heads, hd).transpose(1, 2)
    v = v.view(B, Skv, heads, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2) / math.sqrt(hd)
    if add_mask is not None:
        s = s + add_mask
    p = s.softmax(-1)
    if pdrop is not None:
        p = pdrop(p)                       # attention_probs_dropped = self.dropout(attention_probs)   xbert.py:333
    return (p @ v).transpose(1, 2).reshape(B, Sq, H)


def vit_forward(P: Params, d: AlbefDims, image, mode: str):
    """VisionTransformer.forward (vit.py:176-190) with Block.forward (vit.py:99-110): adapter after the MLP residual."""
    v = PRE + "visual_encoder."
    H = d.hidden
    x = F.conv2d(image, P[v + "patch_embed.proj.weight"], P[v + "patch_embed.proj.bias"], stride=d.patch)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([P[v + "cls_token"].expand(x.shape[0], -1, -1), x], 1) + P[v + "pos_embed"][:, :x.shape[1] + 1]
    for i in range(d.vit_depth):
        b = f"{v}blocks.{i}."
        y = F.layer_norm(x, (H,), P[b + "norm1.weight"], P[b + "norm1.bias"], 1e-6)
        qkv = F.linear(y, P[b + "attn.qkv.weight"], P[b + "attn.qkv.bias"])
        q, k, vv = qkv.split(H, -1)
        x = x + F.linear(_mha(q, k, vv, d.heads), P[b + "attn.proj.weight"], P[b + "attn.proj.bias"])
        y = F.layer_norm(x, (H,), P[b + "norm2.weight"], P[b + "norm2.bias"], 1e-6)
        y = F.linear(F.gelu(F.linear(y, P[b + "mlp.fc1.weight"], P[b + "mlp.fc1.bias"])), P[b + "mlp.fc2.weight"],
                     P[b + "mlp.fc2.bias"])
        x = x + y
        x = adapter(P, b + "adapter.", x, x, mode)
    return F.layer_norm(x, (H,), P[v + "norm.weight"], P[v + "norm.bias"], 1e-6)


# ------------------------------------------------------------------------------------------------ BERT towers
def bert_embeddings(P: Params, tower: str, ids):
    e = tower + "embeddings."
    L = ids.shape[1]
    x = P[e + "word_embeddings.weight"][ids] + P[e + "token_type_embeddings.weight"][0] + \
        P[e + "position_embeddings.weight"][:L]
    x = F.layer_norm(x, (x.shape[-1],), P[e + "LayerNorm.weight"], P[e + "LayerNorm.bias"], 1e-12)
    return drop(x, e, "emb")                                                                  # xbert.py:216


def _bert_attention(P, base, h, kv_src, add_mask, heads):
    q = F.linear(h, P[base + "self.query.weight"], P[base + "self.query.bias"])
    k = F.linear(kv_src, P[base + "self.key.weight"], P[base + "self.key.bias"])
    v = F.linear(kv_src, P[base + "self.value.weight"], P[base + "self.value.bias"])
    cross = "crossattention" in base
    ctx = _mha(q, k, v, heads, add_mask, (lambda p: drop(p, base, "cross_probs" if cross else "self_probs")) if DROP.p > 0 else None)
    out = F.linear(ctx, P[base + "output.dense.weight"], P[base + "output.dense.bias"])
    out = drop(out, base, "cross_out" if cross else "self_out")                              # BertSelfOutput, xbert.py:360
    return F.layer_norm(out + h, (h.shape[-1],), P[base + "output.LayerNorm.weight"], P[base + "output.LayerNorm.bias"], 1e-12)


def bert_layer(P: Params, L: str, h, self_mask, enc, enc_mask, heads, mode: str, cross: bool):
    """BertLayer.forward (xbert.py:463-525): self-attention, optional cross-attention, FFN with the adapter-wrapped
    BertOutput (xbert.py:438-445)."""
    a = _bert_attention(P, L + "attention.", h, h, self_mask, heads)
    if cross:
        a = _bert_attention(P, L + "crossattention.", a, enc, enc_mask, heads)
    inter = F.gelu(F.linear(a, P[L + "intermediate.dense.weight"], P[L + "intermediate.dense.bias"]))
    dense = F.linear(inter, P[L + "output.dense.weight"], P[L + "output.dense.bias"])
    dense = drop(dense, L, "out")                                                             # BertOutput, xbert.py:440
    return adapter_layer_forward_bert(P, L + "output.adapter.", dense, a, P[L + "output.LayerNorm.weight"],
                                      P[L + "output.LayerNorm.bias"], 1e-12, mode)


def _pad_mask(m):           # get_extended_attention_mask / invert_attention_mask: 0 attend, -10000 masked
    return (1.0 - m[:, None, None, :].float()) * -10000.0


def text_encoder(P: Params, d: AlbefDims, ids, mask, image_embeds, mode: str):
    t = PRE + "text_encoder."
    h = bert_embeddings(P, t, ids)
    sm = _pad_mask(mask)
    for i in range(d.enc_layers):
        h = bert_layer(P, f"{t}encoder.layer.{i}.", h, sm, image_embeds, None, d.heads, mode, i >= d.fusion_layer)
    return h


def text_decoder(P: Params, d: AlbefDims, ids, mask, enc, enc_mask, mode: str):
    """BertLMHeadModel (is_decoder=True): causal x padding self-attention mask, cross-attention to the question states in
    every layer (fusion_layer = 0), BertOnlyMLMHead; returns logits [N, L, V]."""
    t = PRE + "text_decoder.bert."
    N, L = ids.shape
    h = bert_embeddings(P, t, ids)
    causal = torch.tril(torch.ones(L, L))
    sm = (1.0 - causal[None, None] * mask[:, None, None, :].float()) * -10000.0
    em = _pad_mask(enc_mask)
    for i in range(d.dec_layers):
        h = bert_layer(P, f"{t}encoder.layer.{i}.", h, sm, enc, em, d.heads, mode, True)
    c = PRE + "text_decoder.cls.predictions."
    x = F.gelu(F.linear(h, P[c + "transform.dense.weight"], P[c + "transform.dense.bias"]))
    x = F.layer_norm(x, (x.shape[-1],), P[c + "transform.LayerNorm.weight"], P[c + "transform.LayerNorm.bias"], 1e-12)
    return F.linear(x, P[t + "embeddings.word_embeddings.weight"], P[c + "bias"])


def lm_loss_per_answer(logits, ids, pad_id):
    """xbert.py:1283-1297 with reduction='none': shifted next-token CE summed over each answer's tokens (pads ignored)."""
    labels = ids.masked_fill(ids == pad_id, -100)[:, 1:]
    lg = logits[:, :-1]
    ce = F.cross_entropy(lg.reshape(-1, lg.shape[-1]), labels.reshape(-1), reduction="none", ignore_index=-100)
    return ce.view(ids.shape[0], -1).sum(1)


def albef_train_forward(P: Params, d: AlbefDims, batch, mode: str):
    """ALBEF.forward(train=True), distill off (albef_model.py:69-145): -> (loss, logits[:, :-1])."""
    img = vit_forward(P, d, batch["image"], mode)
    qs = text_encoder(P, d, batch["question_ids"], batch["question_mask"], img, mode)
    rep = torch.repeat_interleave(torch.arange(len(batch["k"])), torch.tensor(batch["k"]))
    logits = text_decoder(P, d, batch["answer_ids"], batch["answer_mask"], qs[rep], batch["question_mask"][rep], mode)
    loss = (batch["weights"] * lm_loss_per_answer(logits, batch["answer_ids"], d.pad_id)).sum() / batch["image"].shape[0]
    return loss, logits[:, :-1].contiguous()


def kl_loss(output, target, temp: float = 3.0):
    """task_trainer.py:506-516: softmax over the LAST axis when it is vocabulary-sized (> 3000: the ALBEF decoder logits
    [N, L-1, V]), over axis 1 otherwise (the reference's own branch); batchmean over the first axis."""
    dim = -1 if output.shape[-1] > 3000 else 1
    p = F.log_softmax(output / temp, dim=dim)
    q = F.softmax(target / temp, dim=dim)
    return F.kl_div(p, q, reduction="batchmean") * temp ** 2


def rank_answer(P: Params, d: AlbefDims, qs, q_mask, answer_ids, answer_mask, k: int, mode: str):
    """ALBEF.rank_answer (albef_model.py:171-228): first-token shortlist of k answers, re-ranked by sequence likelihood."""
    nq = qs.shape[0]
    start = answer_ids[0, 0].repeat(nq, 1)
    logits = text_decoder(P, d, start, torch.ones_like(start), qs, q_mask, mode)[:, 0]
    prob_first = F.softmax(logits, 1).index_select(1, answer_ids[:, 1])
    topk_probs, topk_ids = prob_first.topk(k, 1)
    ids = torch.cat([answer_ids.index_select(0, t) for t in topk_ids], 0)
    atts = torch.cat([answer_mask.index_select(0, t) for t in topk_ids], 0)
    rep = torch.arange(nq).repeat_interleave(k)
    lg = text_decoder(P, d, ids, atts, qs[rep], q_mask[rep], mode)
    loss = lm_loss_per_answer(lg, ids, d.pad_id).view(nq * k, 1)
    log_probs = torch.cat([topk_probs.view(-1, 1).log(), -loss], 1).sum(1).view(nq, k)
    probs = F.softmax(log_probs, -1)
    probs, rerank = probs.topk(k, 1)
    return torch.gather(topk_ids, 1, rerank), probs


def albef_eval_forward(P: Params, d: AlbefDims, batch, k: int, mode: str):
    img = vit_forward(P, d, batch["image"], mode)
    qs = text_encoder(P, d, batch["question_ids"], batch["question_mask"], img, mode)
    return rank_answer(P, d, qs, batch["question_mask"], batch["answer_list_ids"], batch["answer_list_mask"], k, mode)


# ------------------------------------------------------------------------------------------------ client update
class AlbefDatClient:
    """TaskTrainer.train prologue + train_step, dat branch, ALBEF wiring (task_trainer.py:36-59,280-330): only adapter
    parameters are trainable (main.py:138-159), so both optimizers hold adapter_0 and adapter_1 tensors only."""

    def __init__(self, P: Params, d: AlbefDims, lr: float, steps_per_epoch: int, num_epochs: int = 15,
                 warmup_ratio: float = 0.1, opt_adapters: Sequence[int] = (0, 1), dropout: float = 0.0, seed: int = 0):
        self.P, self.d, self.lr = P, d, lr
        self.dropout, self.seed, self.step_idx = dropout, seed, 0
        for n in list(P):
            if "adapter_1" in n:
                P[n.replace("adapter_1", "adapter_2")] = P[n].clone()
        self.total = steps_per_epoch * num_epochs
        self.warmup = int(self.total * warmup_ratio)
        self.names = [n for a in opt_adapters for n in trainable_names(P, a)]
        self.opt = O.AdamWState(self.names, lr)
        self.t = 0

    def _lr(self):
        return self.lr * O.poly_lr_lambda(self.t, self.warmup, self.total)

    def _set_pass(self, pass_id):
        DROP.p, DROP.seed, DROP.step, DROP.pass_id = self.dropout, self.seed, self.step_idx, pass_id

    def _sub_step(self, batch, mode, adapter_idx, teacher):
        self._set_pass(1 if mode == "adapter_1" else 2)
        names = [n for n in trainable_names(self.P, adapter_idx) if n in self.opt.t]
        for n in names:
            self.P[n].requires_grad_(True)
        loss, logits = albef_train_forward(self.P, self.d, batch, mode)
        L = (loss + kl_loss(logits, teacher.detach())) / 2
        grads = torch.autograd.grad(L, [self.P[n] for n in names]) if names else []
        for n in names:
            self.P[n].requires_grad_(False)
        with torch.no_grad():
            self.opt.step(self.P, dict(zip(names, grads)), self._lr())
        self.t += 1
        return loss.detach(), logits.detach(), float(L)

    def train_step(self, batch):
        self._set_pass(0)
        try:
            with torch.no_grad():
                _, logits_all = albef_train_forward(self.P, self.d, batch, "gating")
            loss_1, logits_1, self.last_L1 = self._sub_step(batch, "adapter_1", 1, logits_all)
            loss_0, _, self.last_L0 = self._sub_step(batch, "gating", 0, logits_1)
        finally:
            DROP.p = 0.0                   # dropout is a property of train_step only (eval / plain forwards: model.eval())
        self.step_idx += 1
        self.last_loss_1 = float(loss_1)
        return loss_0


def synthetic_batch(B: int, d: AlbefDims, seed: int, q_len: int = 25, a_len: int = 4, k: Sequence[int] = None,
                    ragged: bool = False):
    """N(0,1) images, questions of q_len tokens ([CLS] ... [SEP]), k answers per question of a_len
    tokens ([CLS] a b [SEP]), weights 1.  ragged=True shortens some questions / answers (padding + masks)."""
    g = torch.Generator().manual_seed(seed)
    k = list(k) if k is not None else [1] * B
    n = sum(k)
    img = torch.randn(B, 3, d.image, d.image, generator=g)
    q = torch.randint(1000, min(30000, d.vocab), (B, q_len), generator=g)
    q[:, 0], q[:, -1] = 101, 102
    qm = torch.ones(B, q_len, dtype=torch.long)
    a = torch.randint(1000, min(30000, d.vocab), (n, a_len), generator=g)
    a[:, 0], a[:, -1] = 101, 102
    am = torch.ones(n, a_len, dtype=torch.long)
    if ragged:
        for b in range(B):
            cut = q_len - (b % 4) * 3
            if cut < q_len:
                q[b, cut - 1] = 102
                q[b, cut:] = d.pad_id
                qm[b, cut:] = 0
        for i in range(n):
            if i % 3 == 1 and a_len > 3:
                a[i, a_len - 2] = 102
                a[i, a_len - 1] = d.pad_id
                am[i, a_len - 1] = 0
    w = torch.ones(n) if not ragged else (0.5 + torch.rand(n, generator=g))
    return {"image": img, "question_ids": q, "question_mask": qm, "answer_ids": a, "answer_mask": am, "weights": w, "k": k}
