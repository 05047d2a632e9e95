"""CPU restatement (PyTorch-CPU, fp32) of the FedDAT hot path -- TEST INFRASTRUCTURE ONLY.

This file is the parity oracle.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package ``feddat_amd``
never does (it fails loudly when its HIP library is missing instead of falling back).

Pinned: every function below is checked against fixtures captured from the reference's
own modules (``oracle/make_golden.py`` imports /root/reference and writes
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` replays them).  The frozen
backbone arithmetic lives in HuggingFace ``transformers`` (reference pins 4.16.2,
requirements.txt:5; not vendored) -- restated here from its published ViLT algorithm and
anchored on the reference's call sites vilt.py:98,127 and the captured goldens.

Each function cites the reference file:line (relative to /root/reference) it follows.
All parameters live in a flat ``dict[str, Tensor]`` keyed by the reference's state-dict
names.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

ENC = "vilt_encoder.vilt."
LAYER = ENC + "encoder.layer.{i}."
ADAPTER = LAYER + "output.adapter."


# ----------------------------------------------------------------------------------------
# configuration (HF ViltConfig defaults == dandelin/vilt-b32-mlm; vilt.py:401-405)
# ----------------------------------------------------------------------------------------
class ViltDims:
    def __init__(self, hidden=768, layers=12, heads=12, inter=3072, patch=32, image_size=384,
                 max_text=40, vocab=30522, num_labels=100, reduction=16, ln_eps=1e-12):
        self.hidden = hidden
        self.layers = layers
        self.heads = heads
        self.head_dim = hidden // heads
        self.inter = inter
        self.patch = patch
        self.image_size = image_size
        self.grid = image_size // patch
        self.max_text = max_text
        self.vocab = vocab
        self.num_labels = num_labels
        self.bottleneck = hidden // reduction  # adapter.py:22,34
        self.ln_eps = ln_eps


# ----------------------------------------------------------------------------------------
# deterministic parameters (name-seeded; SURVEY.md section 8c "Deterministic weights")
# ----------------------------------------------------------------------------------------
def _seeded_normal(name: str, shape, std: float) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return torch.randn(shape, generator=g, dtype=torch.float32) * std


def param_shapes(d: ViltDims, tasks: Sequence[str]) -> Dict[str, Tuple[int, ...]]:
    """State-dict keys and shapes of the reference model (vilt.py:154-219,356-361;
    adapter.py:22-58; HF ViltModel)."""
    H, I, r = d.hidden, d.inter, d.bottleneck
    s: Dict[str, Tuple[int, ...]] = {}
    e = ENC + "embeddings."
    s[e + "cls_token"] = (1, 1, H)
    s[e + "position_embeddings"] = (1, d.grid * d.grid + 1, H)
    s[e + "text_embeddings.word_embeddings.weight"] = (d.vocab, H)
    s[e + "text_embeddings.position_embeddings.weight"] = (d.max_text, H)
    s[e + "text_embeddings.token_type_embeddings.weight"] = (2, H)
    s[e + "text_embeddings.LayerNorm.weight"] = (H,)
    s[e + "text_embeddings.LayerNorm.bias"] = (H,)
    s[e + "patch_embeddings.projection.weight"] = (H, 3, d.patch, d.patch)
    s[e + "patch_embeddings.projection.bias"] = (H,)
    s[e + "token_type_embeddings.weight"] = (3, H)  # expanded, vilt.py:102-113
    for i in range(d.layers):
        L = LAYER.format(i=i)
        for n in ("query", "key", "value"):
            s[L + f"attention.attention.{n}.weight"] = (H, H)
            s[L + f"attention.attention.{n}.bias"] = (H,)
        s[L + "attention.output.dense.weight"] = (H, H)
        s[L + "attention.output.dense.bias"] = (H,)
        s[L + "intermediate.dense.weight"] = (I, H)
        s[L + "intermediate.dense.bias"] = (I,)
        s[L + "output.layer.dense.weight"] = (H, I)  # adaptered_output.py:70 wraps .layer
        s[L + "output.layer.dense.bias"] = (H,)
        for a in range(3):
            A = ADAPTER.format(i=i)
            s[A + f"adapter_{a}_down.weight"] = (r, H)
            s[A + f"adapter_{a}_down.bias"] = (r,)
            s[A + f"adapter_{a}_up.weight"] = (H, r)
            s[A + f"adapter_{a}_up.bias"] = (H,)
        s[L + "layernorm_before.weight"] = (H,)
        s[L + "layernorm_before.bias"] = (H,)
        s[L + "layernorm_after.weight"] = (H,)
        s[L + "layernorm_after.bias"] = (H,)
    s[ENC + "layernorm.weight"] = (H,)
    s[ENC + "layernorm.bias"] = (H,)
    s[ENC + "pooler.dense.weight"] = (H, H)
    s[ENC + "pooler.dense.bias"] = (H,)
    for t in tasks:  # vilt.py:202-209
        s[f"task_layer.{t}.clf_fc0.weight"] = (2 * H, H)
        s[f"task_layer.{t}.clf_fc0.bias"] = (2 * H,)
        s[f"task_layer.{t}.clf_norm0.weight"] = (2 * H,)
        s[f"task_layer.{t}.clf_norm0.bias"] = (2 * H,)
        s[f"task_layer.{t}.clf_fc1.weight"] = (d.num_labels, 2 * H)
        s[f"task_layer.{t}.clf_fc1.bias"] = (d.num_labels,)
    return s


def seeded_value(name: str, shape, std: float = 0.02, bias_std: float = 0.0) -> torch.Tensor:
    """Deterministic fill rule shared by the golden generator, the oracle and the GPU tests.
    LayerNorm weight=1/bias=0 (adapter.py:5-14); other biases N(0, bias_std) (0 by default,
    matching init_bert_weights; tests use a non-zero bias_std so bias paths are exercised)."""
    is_ln = ("LayerNorm" in name) or ("layernorm" in name) or ("clf_norm0" in name)
    if is_ln and name.endswith("weight"):
        return torch.ones(shape) + (_seeded_normal(name, shape, bias_std) if bias_std else 0)
    if name.endswith("bias"):
        return _seeded_normal(name, shape, bias_std) if bias_std else torch.zeros(shape)
    return _seeded_normal(name, shape, std)


def make_params(d: ViltDims, tasks: Sequence[str], std: float = 0.02, bias_std: float = 0.0) -> Params:
    return {k: seeded_value(k, shp, std, bias_std) for k, shp in param_shapes(d, tasks).items()}


# ----------------------------------------------------------------------------------------
# synthetic batches (SURVEY.md section 8d "Synthetic inputs")
# ----------------------------------------------------------------------------------------
def synthetic_batch(B: int, res: int, seed: int, text_len: int = 40, num_labels: int = 100,
                    vocab_lo: int = 1000, vocab_hi: int = 30000, label_prior: Optional[torch.Tensor] = None):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    pixel_values = torch.randn(B, 3, res, res, generator=g)
    ids = torch.randint(vocab_lo, vocab_hi, (B, text_len), generator=g)
    ids[:, 0] = 101
    ids[:, -1] = 102
    target = torch.zeros(B, num_labels)
    scores = torch.tensor([0.3, 0.6, 0.9, 1.0])  # vqa_utils.py:21-31
    for b in range(B):
        n = int(torch.randint(1, 4, (1,), generator=g))
        if label_prior is None:
            labs = torch.randperm(num_labels, generator=g)[:n]
        else:
            labs = torch.multinomial(label_prior, n, replacement=False, generator=g)
        sc = scores[torch.randint(0, 4, (n,), generator=g)]
        target[b, labs] = sc  # vqa_utils.py:62-67
    return {
        "pixel_values": pixel_values,
        "pixel_mask": torch.ones(B, res, res, dtype=torch.long),
        "input_ids": ids,
        "attention_mask": torch.ones(B, text_len, dtype=torch.long),
        "token_type_ids": torch.zeros(B, text_len, dtype=torch.long),
        "target_scores": target,
    }

def pad_batch(batch, valid_hw, text_lens=None):
    """Turn a synthetic batch into what the HF ViLT processor yields for images of different sizes: sample b keeps
    the top-left valid_hw[b] = (h, w) pixels (multiples of the patch size), the rest is zero-padded with
    pixel_mask = 0; optionally questions are padded ([PAD] = 0, attention_mask = 0) beyond text_lens[b] tokens."""
    b = {k: v.clone() for k, v in batch.items()}
    for i, (h, w) in enumerate(valid_hw):
        b["pixel_mask"][i] = 0
        b["pixel_mask"][i, :h, :w] = 1
        b["pixel_values"][i] *= b["pixel_mask"][i][None].to(b["pixel_values"].dtype)
    if text_lens is not None:
        for i, n in enumerate(text_lens):
            b["attention_mask"][i, n:] = 0
            b["input_ids"][i, n:] = 0
    return b



# ----------------------------------------------------------------------------------------
# DAT module (adapter.py:124-163)
# ----------------------------------------------------------------------------------------
def adapter_single(h, inp, Wd, bd, Wu, bu):
    """adapter.py:125-131: input_tensor + up(relu(down(h)))."""
    return inp + F.linear(F.relu(F.linear(h, Wd, bd)), Wu, bu)


def adapter_gated(h, inp, A, B):
    """adapter.py:133-146 + get_agg_out 118-122: input + 0.5*A(h) + 0.5*B(h), scaling 1.0.
    A, B = (Wd, bd, Wu, bu) of adapter_0 and adapter_2 (or adapter_1 when no adapter_2)."""
    up0 = F.linear(F.relu(F.linear(h, A[0], A[1])), A[2], A[3])
    up1 = F.linear(F.relu(F.linear(h, B[0], B[1])), B[2], B[3])
    agg = 0.5 * up0
    agg = agg + 0.5 * up1
    return inp + agg * 1.0


def _adapter_params(P: Params, i: int, a: int):
    A = ADAPTER.format(i=i)
    return (P[A + f"adapter_{a}_down.weight"], P[A + f"adapter_{a}_down.bias"],
            P[A + f"adapter_{a}_up.weight"], P[A + f"adapter_{a}_up.bias"])


def adapter_forward(P: Params, i: int, h, mode: str):
    """mode: 'gating' (adapter_0 + adapter_2, adapter.py:133) or 'adapter_k' (single)."""
    if mode == "gating":
        return adapter_gated(h, h, _adapter_params(P, i, 0), _adapter_params(P, i, 2))
    a = int(mode.split("_")[1])
    return adapter_single(h, h, *_adapter_params(P, i, a))  # adaptered_output.py:77 passes (h, h)


# ----------------------------------------------------------------------------------------
# ViLT backbone (HF ViltModel; call site vilt.py:127)
# ----------------------------------------------------------------------------------------
def interp_pos_embed(P: Params, d: ViltDims, gh: int, gw: int):
    """HF ViltEmbeddings.visual_embed: bilinear(align_corners=True) resize of the 12x12 grid."""
    pos = P[ENC + "embeddings.position_embeddings"]
    spatial = pos[:, 1:, :].transpose(1, 2).reshape(1, d.hidden, d.grid, d.grid)
    if (gh, gw) != (d.grid, d.grid):
        spatial = F.interpolate(spatial, size=(gh, gw), mode="bilinear", align_corners=True)
    return spatial.flatten(2).transpose(1, 2)  # [1, gh*gw, H]


def patch_mask(batch, d: ViltDims) -> torch.Tensor:
    """HF ViltEmbeddings.visual_embed: pixel_mask -> patch-level mask by nearest interpolation to the patch grid
    (F.interpolate default mode: source index = dst * patch) -> bool [B, gh, gw]."""
    pm = batch["pixel_mask"]
    return pm[:, ::d.patch, ::d.patch].bool()


def vilt_embed(P: Params, d: ViltDims, batch) -> torch.Tensor:
    """HF ViltEmbeddings.forward: [text | CLS, patches] -> [B,S,H].

    All gh*gw patch tokens are kept, in raster order, the padded ones masked as attention keys (key_mask below).  HF
    instead (a) permutes the valid patches with torch.multinomial and (b) truncates every sample to the largest valid
    patch count of the batch, filling shorter samples with randomly drawn padding patches that are masked as well;
    masked tokens never reach a valid token's output, so the pooled [CLS] feature -- the only thing the FedDAT path
    reads (vilt.py:127) -- is the same up to fp32 re-association (pinned by tests/golden/g6_padded.npz).
    Position embeddings: per sample, the 12x12 grid is resized (bilinear, align_corners=True) to that sample's valid
    h x w patches, placed top-left and zero-padded, exactly as visual_embed does."""
    e = ENC + "embeddings."
    ids, tt = batch["input_ids"], batch["token_type_ids"]
    B, Lt = ids.shape
    te = P[e + "text_embeddings.word_embeddings.weight"][ids]
    te = te + P[e + "text_embeddings.token_type_embeddings.weight"][tt]
    te = te + P[e + "text_embeddings.position_embeddings.weight"][:Lt][None]
    te = F.layer_norm(te, (d.hidden,), P[e + "text_embeddings.LayerNorm.weight"],
                      P[e + "text_embeddings.LayerNorm.bias"], d.ln_eps)
    px = batch["pixel_values"]
    x = F.conv2d(px, P[e + "patch_embeddings.projection.weight"],
                 P[e + "patch_embeddings.projection.bias"], stride=d.patch)
    gh, gw = x.shape[2], x.shape[3]
    x = x.flatten(2).transpose(1, 2)
    pmask = patch_mask(batch, d) if "pixel_mask" in batch else torch.ones(B, gh, gw, dtype=torch.bool)
    if bool(pmask.all()):
        x = x + interp_pos_embed(P, d, gh, gw)
    else:
        x_h = pmask[:, :, 0].sum(1)          # valid rows / cols of the top-left aligned valid rectangle
        x_w = pmask[:, 0, :].sum(1)
        pos = []
        for h, w in zip(x_h.tolist(), x_w.tolist()):
            pe = interp_pos_embed(P, d, h, w).transpose(1, 2).reshape(1, d.hidden, h, w)
            pe = F.pad(pe, (0, gw - w, 0, gh - h))
            pos.append(pe.flatten(2).transpose(1, 2))
        x = x + torch.cat(pos, 0)
    cls = P[e + "cls_token"].expand(B, -1, -1) + P[e + "position_embeddings"][:, :1, :]
    ie = torch.cat([cls, x], dim=1)
    tok = P[e + "token_type_embeddings.weight"]
    te = te + tok[0]
    ie = ie + tok[1]
    return torch.cat([te, ie], dim=1)


def key_mask(batch, n_img_tokens: int, d: ViltDims = None) -> torch.Tensor:
    B = batch["attention_mask"].shape[0]
    img = torch.ones(B, n_img_tokens, dtype=torch.bool)
    if d is not None and "pixel_mask" in batch:
        img[:, 1:] = patch_mask(batch, d).flatten(1)
    return torch.cat([batch["attention_mask"].bool(), img], dim=1)


def vilt_layer_body(P: Params, d: ViltDims, i: int, h, kmask=None):
    """HF ViltLayer up to and including ViltOutput's dense+residual (adaptered_output.py:74-76):
    returns the adapter input h3."""
    L = LAYER.format(i=i)
    B, S, H = h.shape
    x = F.layer_norm(h, (H,), P[L + "layernorm_before.weight"], P[L + "layernorm_before.bias"], d.ln_eps)
    q = F.linear(x, P[L + "attention.attention.query.weight"], P[L + "attention.attention.query.bias"])
    k = F.linear(x, P[L + "attention.attention.key.weight"], P[L + "attention.attention.key.bias"])
    v = F.linear(x, P[L + "attention.attention.value.weight"], P[L + "attention.attention.value.bias"])
    sh = (B, S, d.heads, d.head_dim)
    q, k, v = (t.view(sh).transpose(1, 2) for t in (q, k, v))
    sc = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d.head_dim)
    if kmask is not None:
        sc = sc.masked_fill(~kmask[:, None, None, :], torch.finfo(sc.dtype).min)
    p = torch.softmax(sc, dim=-1)
    ctx = torch.matmul(p, v).permute(0, 2, 1, 3).reshape(B, S, H)
    h2 = F.linear(ctx, P[L + "attention.output.dense.weight"], P[L + "attention.output.dense.bias"]) + h
    x2 = F.layer_norm(h2, (H,), P[L + "layernorm_after.weight"], P[L + "layernorm_after.bias"], d.ln_eps)
    f = F.gelu(F.linear(x2, P[L + "intermediate.dense.weight"], P[L + "intermediate.dense.bias"]))
    h3 = F.linear(f, P[L + "output.layer.dense.weight"], P[L + "output.layer.dense.bias"]) + h2
    return h3


def vilt_pooled(P: Params, d: ViltDims, batch, mode: str) -> torch.Tensor:
    """ViltModel(**encodings).pooler_output with Adaptered_ViltOutput in every layer
    (vilt.py:127,356-361)."""
    h = vilt_embed(P, d, batch)
    km = key_mask(batch, h.shape[1] - batch["input_ids"].shape[1], d)
    if bool(km.all()):
        km = None
    for i in range(d.layers):
        h = adapter_forward(P, i, vilt_layer_body(P, d, i, h, km), mode)
    h = F.layer_norm(h, (d.hidden,), P[ENC + "layernorm.weight"], P[ENC + "layernorm.bias"], d.ln_eps)
    return torch.tanh(F.linear(h[:, 0], P[ENC + "pooler.dense.weight"], P[ENC + "pooler.dense.bias"]))


def task_head(P: Params, task: str, pooled):
    """vilt.py:202-209: fc0 -> LayerNorm(1536, eps 1e-5) -> GELU -> fc1."""
    t = f"task_layer.{task}."
    x = F.linear(pooled, P[t + "clf_fc0.weight"], P[t + "clf_fc0.bias"])
    x = F.layer_norm(x, (x.shape[-1],), P[t + "clf_norm0.weight"], P[t + "clf_norm0.bias"], 1e-5)
    return F.linear(F.gelu(x), P[t + "clf_fc1.weight"], P[t + "clf_fc1.bias"])


def vilt_forward(P: Params, d: ViltDims, batch, mode: str, task: str):
    """vilt.py:244-264 forward_single_image -> (pooled, logits)."""
    pooled = vilt_pooled(P, d, batch, mode)
    return pooled, task_head(P, task, pooled)


# ----------------------------------------------------------------------------------------
# losses (task_trainer.py:299-301,506-516; train_vqa_crossvqa.py:237)
# ----------------------------------------------------------------------------------------
def kl_loss(output, target, temp: float = 3.0):
    dim = -1 if output.shape[-1] > 3000 else 1
    p = F.log_softmax(output / temp, dim=dim)
    q = F.softmax(target / temp, dim=dim)
    return F.kl_div(p, q, reduction="batchmean") * temp ** 2


def dat_loss(logits, target, teacher):
    """(BCEWithLogits_mean * num_labels + KL_T3) / 2  (task_trainer.py:299-301)."""
    bce = F.binary_cross_entropy_with_logits(logits, target, reduction="mean") * target.shape[1]
    return (bce + kl_loss(logits, teacher.clone().detach())) / 2


# ----------------------------------------------------------------------------------------
# optimizer + schedule (task_trainer.py:52-59,477-504)
# ----------------------------------------------------------------------------------------
def poly_lr_lambda(t: int, warmup: int, total: int) -> float:
    """HF get_polynomial_decay_schedule_with_warmup(lr_end=0, power=1) as a multiplier of lr."""
    if t < warmup:
        return float(t) / float(max(1, warmup))
    if t > total:
        return 0.0
    return 1.0 - (t - warmup) / float(total - warmup)


def is_no_decay(name: str) -> bool:
    """task_trainer.py:478: names containing 'bias' or 'LayerNorm.weight' get weight_decay 0
    (clf_norm0.weight does NOT match and IS decayed)."""
    return ("bias" in name) or ("LayerNorm.weight" in name)


class AdamWState:
    """torch.optim.AdamW(lr, betas=(0.9,0.98), eps, weight_decay) restated; per-tensor step
    counts; a tensor without a grad in a sub-step is untouched (torch>=2.0 zero_grad semantics,
    SURVEY.md section 8a version note)."""

    def __init__(self, names: Sequence[str], lr: float, eps: float = 1e-8, wd: float = 1e-2,
                 betas=(0.9, 0.98)):
        self.lr, self.eps, self.wd, self.betas = lr, eps, wd, betas
        self.m: Dict[str, torch.Tensor] = {}
        self.v: Dict[str, torch.Tensor] = {}
        self.t: Dict[str, int] = {n: 0 for n in names}

    def step(self, P: Params, grads: Dict[str, torch.Tensor], lr_now: float):
        b1, b2 = self.betas
        for n, g in grads.items():
            if g is None:
                continue
            p = P[n]
            if n not in self.m:
                self.m[n] = torch.zeros_like(p)
                self.v[n] = torch.zeros_like(p)
            self.t[n] += 1
            t = self.t[n]
            wd = 0.0 if is_no_decay(n) else self.wd
            p.mul_(1 - lr_now * wd)
            self.m[n].lerp_(g, 1 - b1)
            self.v[n].mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1 = 1 - b1 ** t
            bc2 = 1 - b2 ** t
            denom = (self.v[n].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(self.m[n], denom, value=-(lr_now / bc1))


def trainable_names(P: Params, task: str, adapter: int) -> List[str]:
    return [k for k in P if (f"adapter_{adapter}_" in k) or k.startswith(f"task_layer.{task}.")]


# ----------------------------------------------------------------------------------------
# the local update (task_trainer.py:24-111, 266-330)
# ----------------------------------------------------------------------------------------
def copy_global_to_teacher(P: Params):
    """task_trainer.py:36-41: adapter_1 -> adapter_2 at the start of every local update."""
    for k in list(P.keys()):
        if "adapter_1" in k:
            P[k.replace("adapter_1", "adapter_2")].copy_(P[k])


def _grads(loss, P: Params, names: Sequence[str]):
    gs = torch.autograd.grad(loss, [P[n] for n in names], allow_unused=True)
    return dict(zip(names, gs))


class DatClient:
    """One client's local update state: optimizer + scheduler re-created per round
    (task_trainer.py:52-59)."""

    def __init__(self, P: Params, d: ViltDims, task: str, lr: float, steps_per_epoch: int,
                 num_epochs: int = 15, warmup_ratio: float = 0.1, eps: float = 1e-8, wd: float = 1e-2,
                 opt_adapters: Sequence[int] = (0, 1)):
        """opt_adapters: which adapters had requires_grad=True when train() built the optimizer
        (task_trainer.py:477-504 filters on p.requires_grad).  Round 0 after prepare_model: (0, 1).
        After TaskTrainer.eval left the server model in set_active_adapter('adapter_1') state
        (task_trainer.py:236-244 -> adapter.py:79-85) the following rounds get (1,)."""
        self.P, self.d, self.task = P, d, task
        self.lr = lr
        self.max_steps = steps_per_epoch * num_epochs          # train_vqa_crossvqa.py:238
        self.warmup = int(self.max_steps * warmup_ratio)       # task_trainer.py:55
        self.opt_adapters = tuple(opt_adapters)
        names = sorted(set(trainable_names(P, task, 0) + trainable_names(P, task, 1)))
        self.opt = AdamWState(names, lr, eps, wd)
        self.sched_t = 0
        copy_global_to_teacher(P)

    def lr_now(self) -> float:
        return self.lr * poly_lr_lambda(self.sched_t, self.warmup, self.max_steps)

    def train_step(self, batch, overflow=(False, False)):
        """task_trainer.py:280-330 (dat branch). Returns (loss_0, logits_all, logits_1, logits_0)
        where loss_0 is what the reference returns: the BCE*num_labels term of P2 alone (:319,330).
        overflow = (A, B): what the reference does under mixed_precision fp16 (accelerate_config.yaml:8) when the scaled
        backward of sub-step A (P1) / B (P2) produces an inf / NaN gradient -- torch.cuda.amp.GradScaler.step() does NOT call
        optimizer.step() (accelerator.backward / optimizer.step: task_trainer.py:302-308,323-328) and accelerate's scheduler
        wrapper does not tick (accelerate/scheduler.py: `if self.step_with_optimizer and ... optimizer.step_was_skipped: return`);
        nothing else changes -- on this fp32 CPU path the gradients themselves are of course finite, the flag only injects the
        skip.  (The halved loss scale has no effect on an fp32 trajectory.)"""
        P, d, task = self.P, self.d, self.task
        target = batch["target_scores"]
        with torch.no_grad():                                           # P0 :283-287
            _, logits_all = vilt_forward(P, d, batch, "gating", task)
        n1 = trainable_names(P, task, 1)                                 # P1 :290-308
        for n in n1:
            P[n].requires_grad_(True)
        _, logits_1 = vilt_forward(P, d, batch, "adapter_1", task)
        L1 = dat_loss(logits_1, target, logits_all)
        g1 = _grads(L1, P, n1)
        for n in n1:
            P[n].requires_grad_(False)
        if 1 not in self.opt_adapters:
            g1 = {n: g for n, g in g1.items() if "adapter_1_" not in n}
        if not overflow[0]:
            with torch.no_grad():
                self.opt.step(P, g1, self.lr_now())
            self.sched_t += 1
        n0 = trainable_names(P, task, 0)                                 # P2 :311-328
        for n in n0:
            P[n].requires_grad_(True)
        _, logits_0 = vilt_forward(P, d, batch, "gating", task)
        L0 = dat_loss(logits_0, target, logits_1.detach())
        g0 = _grads(L0, P, n0)
        for n in n0:
            P[n].requires_grad_(False)
        if 0 not in self.opt_adapters:
            g0 = {n: g for n, g in g0.items() if "adapter_0_" not in n}
        if not overflow[1]:
            with torch.no_grad():
                self.opt.step(P, g0, self.lr_now())
            self.sched_t += 1
        loss_0 = F.binary_cross_entropy_with_logits(logits_0.detach(), target, reduction="mean") * target.shape[1]
        self.last_L1, self.last_L0 = float(L1), float(L0)
        return loss_0, logits_all, logits_1.detach(), logits_0.detach()


# ----------------------------------------------------------------------------------------
# FedAvg (main.py:50-65) and the personal-parameter shuttle (main.py:440-450,473-478,493-503)
# ----------------------------------------------------------------------------------------
def comm_names(P: Params) -> List[str]:
    return [k for k in P if "adapter_1" in k]           # main.py:154-163


def personal_names(P: Params) -> List[str]:
    return [k for k in P if ("task" in k) or ("adapter_0" in k) or ("adapter_2" in k)]  # main.py:130,154


def get_average_net(server: Params, c_models: List[Params], nums: Sequence[float]) -> Params:
    total = sum(nums)
    for key in comm_names(server):
        if "clf" in key:
            continue
        temp = torch.zeros_like(server[key]).float()
        for net, num in zip(c_models, nums):
            temp += net[key] * num / total
        server[key].copy_(temp)
    return server


def fl_round(server: Params, personal: Dict[str, Params], d: ViltDims, tasks: Sequence[str],
             batches: Dict[str, list], lr: float, num_epochs: int = 15):
    """One communication round (main.py:453-510): sequential clients from identical server state."""
    c_models = []
    for task in tasks:
        P = {k: v.clone() for k, v in server.items()}                # main.py:472 deepcopy
        for n, v in personal[task].items():                          # main.py:473-478
            P[n].copy_(v)
        client = DatClient(P, d, task, lr, steps_per_epoch=len(batches[task]), num_epochs=num_epochs)
        for b in batches[task]:
            client.train_step(b)
        personal[task] = {n: P[n].clone() for n in personal_names(P)}  # main.py:493-497
        c_models.append({n: P[n].clone() for n in comm_names(P)})      # main.py:499-503
    get_average_net(server, c_models, [1] * len(tasks))
    return server, personal
