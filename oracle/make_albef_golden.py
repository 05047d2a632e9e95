"""tests/golden/g10_albef_*.npz: the ALBEF dual-adapter path (configs[3]) run on the REFERENCE's own modules (read-only import
from /root/reference, build container only) on name-seeded parameters and synthetic batches.

Reference code exercised (imported, not copied):
  src/modeling/models/vit.py        VisionTransformer, Block (adapter after the MLP residual)
  src/modeling/models/xbert.py      BertConfig, BertModel, BertLMHeadModel (BertOutput with adapter_layer_forward_bert)
  src/modeling/models/adapter.py    Adapter
  src/modeling/models/albef_model.py   ALBEF.forward (train), rank_answer
  src/modeling/albef.py             ALBEFContinualLearner (mode switches)
  src/train/visionlanguage_tasks/task_trainer.py   TaskTrainer.train_step / create_optimizer / kl_loss
Shims: timm stubs incl. a restated PatchEmbed (Conv2d 16/16 + flatten), transformers.modeling_utils helper
injections, BertPreTrainedModel.init_weights guard with the LM-head weight tied to the word embeddings by hand (what
transformers 4.16.2 -- the reference's pin -- does in init_weights), get_head_mask -> [None] * n, ALBEF / wrapper objects
built with __new__ (their __init__ load ./models/bert-base-uncased), .to('cuda') redirected to CPU.

  python oracle/make_albef_golden.py
"""
import os
import sys
import types
from functools import partial

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.nn as nn
import transformers  # noqa: F401
import accelerate  # noqa: F401

from oracle import albef_oracle as A


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class PatchEmbed(nn.Module):            # timm's PatchEmbed as published: Conv2d(patch, stride=patch) + flatten
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


def install_shims():
    class _LoraLinear(nn.Linear):
        def __init__(self, i, o, r=0, **kw):
            super().__init__(i, o)
    _stub("loralib", Linear=_LoraLinear)
    _stub("timm")
    _stub("timm.models")
    _stub("timm.models.vision_transformer", _cfg=lambda **k: {}, PatchEmbed=PatchEmbed)
    _stub("timm.models.registry", register_model=lambda f: f)
    _stub("timm.models.layers", trunc_normal_=nn.init.trunc_normal_, DropPath=nn.Identity)
    sys.path.insert(0, REF)
    for pkg, sub in (("src", "src"), ("src.modeling", "src/modeling"), ("src.modeling.models", "src/modeling/models"),
                     ("src.train", "src/train"), ("src.train.visionlanguage_tasks", "src/train/visionlanguage_tasks"),
                     ("src.utils", "src/utils")):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, sub)]
        sys.modules[pkg] = m
    import transformers.modeling_utils as mu
    from transformers import pytorch_utils as pu
    mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    mu.prune_linear_layer = pu.prune_linear_layer
    mu.find_pruneable_heads_and_indices = lambda *a, **k: (set(), None)
    _orig_to = torch.Tensor.to

    def _to(self, *a, **k):
        if a and isinstance(a[0], str) and a[0].startswith("cuda"):
            a = ("cpu",) + tuple(a[1:])
        return _orig_to(self, *a, **k)
    torch.Tensor.to = _to


install_shims()
import src.modeling.models.xbert as xb  # noqa: E402
from src.modeling.models.albef_model import ALBEF  # noqa: E402
from src.modeling.models.vit import VisionTransformer  # noqa: E402
from src.train.visionlanguage_tasks.task_trainer import TaskTrainer, kl_loss  # noqa: E402

xb.BertPreTrainedModel.init_weights = lambda self: None
xb.BertModel.get_head_mask = lambda self, hm, n, *a, **k: [None] * n


def _load_continual_learner_class():
    """src/modeling/albef.py imports BertTokenizer paths etc. at module level only; the class itself is plain."""
    _stub("src.modeling.continual_learner", EncoderWrapper=nn.Module, ContinualLearner=nn.Module)
    from src.modeling.albef import ALBEFContinualLearner
    return ALBEFContinualLearner


class _Wrapper(nn.Module):
    """Stands where ALBEFWrapper stands (attribute `.albef`); takes pre-tokenised tensors instead of strings."""

    def __init__(self, albef):
        super().__init__()
        self.albef = albef

    def forward(self, batch):
        ns = types.SimpleNamespace
        loss, logits = self.albef(image=batch["image"], question=ns(input_ids=batch["question_ids"], attention_mask=batch["question_mask"]),
                                  answer=ns(input_ids=batch["answer_ids"], attention_mask=batch["answer_mask"]), train=True,
                                  alpha=batch.get("alpha", 0), k=batch["k"], weights=batch["weights"])
        return [loss, logits]


def build_albef_module(d: A.AlbefDims, dropout: float = 0.0):
    ac = {"names": ["adapter_0", "adapter_1", "adapter_2"], "device": "cpu"}
    cfgd = dict(hidden_size=d.hidden, intermediate_size=d.inter, num_attention_heads=d.heads, num_hidden_layers=d.enc_layers,
                vocab_size=d.vocab, max_position_embeddings=d.max_pos, type_vocab_size=2, layer_norm_eps=1e-12,
                hidden_act="gelu", hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout, pad_token_id=d.pad_id,
                fusion_layer=d.fusion_layer, encoder_width=d.hidden)
    ce = xb.BertConfig(**cfgd)
    ce.adapter_config = ac
    cd = xb.BertConfig(**cfgd)
    cd.fusion_layer, cd.num_hidden_layers, cd.adapter_config = 0, d.dec_layers, ac
    m = ALBEF.__new__(ALBEF)
    nn.Module.__init__(m)
    m.tokenizer = types.SimpleNamespace(pad_token_id=d.pad_id)
    m.distill = False
    m.visual_encoder = VisionTransformer(img_size=d.image, patch_size=d.patch, embed_dim=d.hidden, depth=d.vit_depth,
                                         num_heads=d.heads, mlp_ratio=4, qkv_bias=True,
                                         norm_layer=partial(nn.LayerNorm, eps=1e-6), adapter_config=ac)
    m.text_encoder = xb.BertModel(config=ce, add_pooling_layer=False)
    m.text_decoder = xb.BertLMHeadModel(config=cd)
    # transformers 4.16.2 init_weights -> tie_weights: LM-head decoder weight IS the decoder's word-embedding matrix
    m.text_decoder.cls.predictions.decoder.weight = m.text_decoder.bert.embeddings.word_embeddings.weight
    return m


def build_reference_model(d: A.AlbefDims, dropout: float = 0.0, albef_module=None):
    m = albef_module if albef_module is not None else build_albef_module(d, dropout)
    CL = _load_continual_learner_class()
    cl = CL.__new__(CL)
    nn.Module.__init__(cl)
    cl.albef_model = _Wrapper(m)
    for p in cl.parameters():                         # main.py:138-139
        p.requires_grad = False
    for n, p in cl.named_parameters():                # main.py:157-159
        if "adapter" in n:
            p.requires_grad = True
    sd = cl.state_dict()
    shapes = A.param_shapes(d)
    missing = [k for k in shapes if k not in sd]
    assert not missing, missing[:5]
    with torch.no_grad():
        for k, shp in shapes.items():
            assert tuple(sd[k].shape) == tuple(shp), (k, sd[k].shape, shp)
            if albef_module is None or "adapter_" in k:        # (a loaded checkpoint keeps its backbone: golden_loader)
                sd[k].copy_(A.O.seeded_value(k, shp, 0.02, 0.02))
    extra = [k for k in sd if k not in shapes and "position_ids" not in k and "decoder.weight" not in k and "decoder.bias" not in k]
    assert not extra, extra[:5]
    dec = m.text_decoder
    assert dec.cls.predictions.decoder.weight.data_ptr() == dec.bert.embeddings.word_embeddings.weight.data_ptr()
    assert dec.cls.predictions.decoder.bias.data_ptr() == dec.cls.predictions.bias.data_ptr()
    cl.eval()
    return cl


class _Wrap:
    def __init__(self, m):
        self.module = m

    def __call__(self, *a, **k):
        return self.module(*a, **k)

    def named_parameters(self):
        return self.module.named_parameters()


class _Acc:
    device = torch.device("cpu")

    @staticmethod
    def backward(loss):
        loss.backward()


def ref_local_update(model, batches, lr, num_epochs=15, capture=None):
    from transformers import get_polynomial_decay_schedule_with_warmup
    sd = model.state_dict()
    for name in sd.keys():                            # task_trainer.py:36-45
        if "adapter_1" in name:
            sd[name.replace("adapter_1", "adapter_2")].data.copy_(sd[name].data.clone())
    for n, p in model.named_parameters():
        if "adapter_2" in n:
            p.requires_grad = False
    tr = TaskTrainer()
    tr.accelerator, tr.device, tr.task_key = _Acc(), torch.device("cpu"), "art"
    tr.args = types.SimpleNamespace(optimizer_mode="dat", encoder_name="albef_no_distill", debug=0)
    tr.batch2inputs_converter = lambda batch: dict(batch)
    tr.kl_criterion = kl_loss
    tr.lr, tr.adam_epsilon, tr.weight_decay, tr.warmup_ratio = lr, 1e-8, 1e-2, 0.1
    tr.max_steps = len(batches) * num_epochs
    opt = tr.create_optimizer(model, "dat")
    sch = get_polynomial_decay_schedule_with_warmup(opt, num_warmup_steps=int(tr.max_steps * tr.warmup_ratio),
                                                    num_training_steps=tr.max_steps, lr_end=0, power=1)
    model.zero_grad()
    w = _Wrap(model)
    losses = []
    for s, b in enumerate(batches):
        losses.append(float(tr.train_step(w, s, dict(b), opt, sch)))
        if capture is not None:
            capture(s, model)
    return losses


def np_(t):
    return t.detach().cpu().numpy().astype(np.float32)


def install_counter_dropout(model, seed):
    """Replace nn.Dropout.forward by the counter-based mask of oracle/albef_oracle.py, keyed on the MODULE's own name: every
    nn.Dropout of the reference model that fires in train mode must be one of the six kinds the oracle (and the HIP engine)
    place -- an unmapped one trips the assertion, a site the oracle places where the reference has none would make the
    fixture's numbers unreachable.  pass = index of the top-level forward inside its train_step (P0 / P1 / P2,
    task_trainer.py:283-315), step = train_steps done."""
    import re
    kinds = [("embeddings.dropout", "emb"), ("crossattention.self.dropout", "cross_probs"), ("attention.self.dropout", "self_probs"),
             ("crossattention.output.dropout", "cross_out"), ("attention.output.dropout", "self_out")]
    n_sites = 0
    for n, m in model.named_modules():
        if isinstance(m, nn.Dropout):
            kind = next((k for suf, k in kinds if n.endswith(suf)), None)
            if kind is None and re.search(r"layer\.\d+\.output\.dropout$", n):
                kind = "out"
            m._fd_name, m._fd_site = n, (A.dropout_site(n, kind) if kind else None)
            n_sites += kind is not None
    state = {"calls": 0, "fired": set()}
    model.albef_model.register_forward_pre_hook(lambda mod, inp: state.__setitem__("calls", state["calls"] + 1))

    def fwd(self, x):
        if not self.training or self.p == 0:
            return x
        assert self._fd_site is not None, "unmapped dropout module fired: " + self._fd_name
        call = state["calls"] - 1
        k0, k1 = A.dropout_keys(seed, call % 3, self._fd_site)
        keep = A.dropout_keep(x.numel(), self.p, k0, k1, call // 3).view(x.shape)
        scale = torch.tensor(1.0) / (torch.tensor(1.0) - torch.tensor(self.p, dtype=torch.float32))
        state["fired"].add(self._fd_name)
        return x * (keep.to(x.dtype) * scale)
    nn.Dropout.forward = fwd
    return state, n_sites


def set_mode(model, mode):
    if mode == "gating":
        model.activate_gating()
    else:
        model.deactivate_gating()
        model.set_active_adapter(mode)


SMALL = dict(vit_depth=2, enc_layers=3, fusion_layer=1, dec_layers=2, image=64, vocab=3072, max_pos=64)


def small_dims():
    return A.AlbefDims(**SMALL)


def golden_small(out):
    """Reduced depth / image / vocabulary, every code path: ragged questions + answers, k = [2, 1, 3], weights != 1."""
    d = small_dims()
    model = build_reference_model(d)
    rec = {}
    b0 = A.synthetic_batch(3, d, 500, q_len=12, a_len=5, k=[2, 1, 3], ragged=True)
    with torch.no_grad():
        for mode in ("gating", "adapter_1", "adapter_0"):
            set_mode(model, mode)
            loss, logits = model("art", dict(b0, train=True))
            rec[f"fwd.{mode}.loss"] = np_(loss)
            rec[f"fwd.{mode}.logits"] = np_(logits)
        set_mode(model, "gating")
        alb = model.albef_model.albef
        img = alb.visual_encoder(b0["image"])
        rec["fwd.gating.image_embeds"] = np_(img)
        # eval path: rank_answer over an 11-answer list, k = 4 (albef_model.py:171-228)
        ev = A.synthetic_batch(3, d, 501, q_len=12, a_len=5, k=[4, 4, 3], ragged=True)
        ans_ids, ans_mask = ev["answer_ids"], ev["answer_mask"]
        ns = types.SimpleNamespace
        ids, probs = alb(image=b0["image"], question=ns(input_ids=b0["question_ids"], attention_mask=b0["question_mask"]),
                         answer=ns(input_ids=ans_ids, attention_mask=ans_mask), train=False, k=4)
        rec["eval.topk_ids"], rec["eval.topk_probs"] = ids.numpy().astype(np.int64), np_(probs)
    for n, p in model.named_parameters():
        if "adapter" in n:
            p.requires_grad = True
    batches = [A.synthetic_batch(3, d, 510 + s, q_len=12, a_len=5, k=[2, 1, 3], ragged=True) for s in range(4)]
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rec["losses"] = np.array(ref_local_update(model, batches, lr=1e-4), np.float32)
    for k, v in model.state_dict().items():
        if "adapter_0" in k or "adapter_1" in k:
            dw = (v.detach() - init[k]).flatten()
            idx = torch.linspace(0, dw.numel() - 1, min(512, dw.numel())).long()
            rec["dnorm::" + k], rec["dmean::" + k], rec["dsamp::" + k] = np_(dw.norm()), np_(dw.abs().mean()), np_(dw[idx])
    np.savez_compressed(os.path.join(out, "g10_albef_small.npz"), **rec)
    print("G10 small losses", rec["losses"], "fwd", {m: float(rec[f'fwd.{m}.loss']) for m in ("gating", "adapter_1")})


def golden_full(out):
    """The real architecture (ViT-B/16 at 384 = 577 tokens, BERT-base 12 + 6 layers, vocab 30522), B = 2: forward in two
    modes and 2 train_steps; sampled updates."""
    d = A.AlbefDims()
    model = build_reference_model(d)
    rec = {}
    b0 = A.synthetic_batch(2, d, 600)
    with torch.no_grad():
        for mode in ("gating", "adapter_1"):
            set_mode(model, mode)
            loss, logits = model("art", dict(b0, train=True))
            rec[f"fwd.{mode}.loss"] = np_(loss)
            lg = logits.flatten()
            rec[f"fwd.{mode}.logits_samp"] = np_(lg[torch.linspace(0, lg.numel() - 1, 4096).long()])
            rec[f"fwd.{mode}.logits_norm"] = np_(lg.norm())
    for n, p in model.named_parameters():
        if "adapter" in n:
            p.requires_grad = True
    batches = [A.synthetic_batch(2, d, 610 + s) for s in range(2)]
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rec["losses"] = np.array(ref_local_update(model, batches, lr=1e-4, num_epochs=1), np.float32)
    for k, v in model.state_dict().items():
        if "adapter_0" in k or "adapter_1" in k:
            dw = (v.detach() - init[k]).flatten()
            idx = torch.linspace(0, dw.numel() - 1, min(256, dw.numel())).long()
            rec["dnorm::" + k], rec["dmean::" + k], rec["dsamp::" + k] = np_(dw.norm()), np_(dw.abs().mean()), np_(dw[idx])
    np.savez_compressed(os.path.join(out, "g10_albef_full.npz"), **rec)
    print("G10 full losses", rec["losses"])


def golden_round(out, steps=40):
    """G11: a local round at realistic length on the reference -- `steps` train_steps of the small configuration (every
    code path: ragged questions / answers, k = [2, 1, 3], weights != 1), one epoch, schedule past its warm-up.  Losses and,
    per adapter_0 / adapter_1 tensor, norm / mean / max / 512 samples of the update."""
    d = small_dims()
    model = build_reference_model(d)
    for n, p in model.named_parameters():
        if "adapter" in n:
            p.requires_grad = True
    batches = [A.synthetic_batch(3, d, 700 + s, q_len=12, a_len=5, k=[2, 1, 3], ragged=True) for s in range(steps)]
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rec = {"losses": np.array(ref_local_update(model, batches, lr=1e-4, num_epochs=1), np.float32),
           "steps": np.array(steps, np.int64)}
    for k, v in model.state_dict().items():
        if "adapter_0" in k or "adapter_1" in k:
            dw = (v.detach() - init[k]).flatten()
            idx = torch.linspace(0, dw.numel() - 1, min(512, dw.numel())).long()
            rec["dnorm::" + k], rec["dmean::" + k] = np_(dw.norm()), np_(dw.abs().mean())
            rec["dmax::" + k], rec["dsamp::" + k] = np_(dw.abs().max()), np_(dw[idx])
    np.savez_compressed(os.path.join(out, "g11_albef_round40.npz"), **rec)
    print("G11 losses first/last", rec["losses"][:3], rec["losses"][-3:],
          "mean |dW|", float(np.mean([rec[k] for k in rec if k.startswith("dmean::")])))


def golden_round_full(out, steps=40, batch=4, snaps=(20, 40), seed0=7700):
    """G11b: a local round at realistic length on the reference at FULL size (ViT-B/16 at 384 = 577 tokens, BERT-base 12 + 6
    layers, vocab 30522; B = 4, 25-token questions, one 4-token answer each = SURVEY 8d config 4's shapes; dropout 0 = its
    parity configuration): `steps` train_steps of ALBEFContinualLearner + TaskTrainer.train_step (albef_model.py:69-145,
    task_trainer.py:280-330), num_epochs = 15 like G8 (steps = 40: 600 scheduler ticks, warm-up 60 ticks = 30 batches, the last
    10 batches at the peak lr), the update of every adapter_0 /
    adapter_1 tensor after each n in `snaps` as norm / mean / max / 1024 samples (like G8b for ViLT)."""
    d = A.AlbefDims()
    model = build_reference_model(d)
    for n, p in model.named_parameters():
        if "adapter" in n:
            p.requires_grad = True
    batches = [A.synthetic_batch(batch, d, seed0 + s) for s in range(steps)]       # --seed0 N: an independent round, own file
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rec = {"steps": np.array(steps, np.int64), "batch": np.array(batch, np.int64), "snaps": np.array(snaps, np.int64),
           "seed0": np.array(seed0, np.int64)}

    def cap(step, m):
        n = step + 1
        print("G11b step", n, flush=True)
        if n not in snaps:
            return
        for k, v in m.state_dict().items():
            if "adapter_0" in k or "adapter_1" in k:
                dw = (v.detach() - init[k]).flatten()
                idx = torch.linspace(0, dw.numel() - 1, min(1024, dw.numel())).long()
                rec[f"s{n}::dnorm::" + k], rec[f"s{n}::dmean::" + k] = np_(dw.norm()), np_(dw.abs().mean())
                rec[f"s{n}::dmax::" + k], rec[f"s{n}::dsamp::" + k] = np_(dw.abs().max()), np_(dw[idx])
    rec["losses"] = np.array(ref_local_update(model, batches, lr=1e-4, num_epochs=15, capture=cap), np.float32)
    np.savez_compressed(os.path.join(out, f"g11b_albef_full_round{steps}" + ("" if seed0 == 7700 else f"_seed{seed0}") + ".npz"), **rec)
    print("G11b losses first/last", rec["losses"][:3], rec["losses"][-3:])


def golden_dropout(out, steps=3, p=0.1, seed=77):
    """G12: train_steps of the small configuration UNDER model.train() with the reference's dropout probabilities
    (hidden_dropout_prob = attention_probs_dropout_prob = 0.1, src/configs/model_configs.py:44-46; task_trainer.py:75) -- the
    random masks replaced by the counter-based ones (install_counter_dropout), everything else the reference's own code.
    Pins WHERE the oracle drops (embeddings, attention probabilities, BertSelfOutput of self- and cross-attention,
    BertOutput ahead of the adapter), that P0 / P1 / P2 draw independent masks, and the 1 / (1 - p) scaling."""
    d = small_dims()
    model = build_reference_model(d, dropout=p)
    state, n_sites = install_counter_dropout(model, seed)
    for n, q in model.named_parameters():
        if "adapter" in n:
            q.requires_grad = True
    model.train()
    batches = [A.synthetic_batch(3, d, 900 + s, q_len=12, a_len=5, k=[2, 1, 3], ragged=True) for s in range(steps)]
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rec = {"losses": np.array(ref_local_update(model, batches, lr=1e-4, num_epochs=1), np.float32),
           "steps": np.array(steps, np.int64), "p": np.array(p, np.float32), "seed": np.array(seed, np.int64)}
    assert state["calls"] == 3 * steps and len(state["fired"]) == n_sites, (state["calls"], len(state["fired"]), n_sites)
    for k, v in model.state_dict().items():
        if "adapter_0" in k or "adapter_1" in k:
            dw = (v.detach() - init[k]).flatten()
            idx = torch.linspace(0, dw.numel() - 1, min(512, dw.numel())).long()
            rec["dnorm::" + k], rec["dmean::" + k], rec["dsamp::" + k] = np_(dw.norm()), np_(dw.abs().mean()), np_(dw[idx])
    np.savez_compressed(os.path.join(out, "g12_albef_dropout.npz"), **rec)
    print("G12 losses", rec["losses"], "dropout sites fired", len(state["fired"]))


LOADER_DIMS = dict(vit_depth=1, enc_layers=7, fusion_layer=6, dec_layers=1, image=64, vocab=3072, max_pos=64)
LOADER_PRE_IMAGE = 48          # the synthetic ALBEF.pth was "pre-trained" at 48 x 48: a 3 x 3 position grid -> 4 x 4


def golden_loader(out):
    """G14: PRETRAINED-weight loading for ALBEF.  tests/ckpt_util.write_albef_checkpoint writes a file in the layout of the
    published ALBEF.pth (BertForMaskedLM text encoder with `bert.` keys, 12-layer numbering with fusion from layer 6, smaller
    pre-training position grid, momentum / projection / queue tensors).  The reference's OWN load_albef (src/modeling/albef.py:
    188-241: torch.load, interpolate_pos_embed, `bert.` strip, text-encoder layers >= 6 -> decoder layers, strict=False load)
    runs on it; only its constructor calls are redirected (ALBEF(...) -> the shimmed module of build_albef_module, tokenizer and
    ALBEFWrapper -> stand-ins: both load ./models/bert-base-uncased).  Stored per resulting state-dict key: L2 norm + 8 samples;
    plus loss / logits of a training forward in two modes with name-seeded adapters."""
    import logging
    import tempfile
    import src.modeling.albef as ref_albef
    from tests.ckpt_util import write_albef_checkpoint
    d = A.AlbefDims(**LOADER_DIMS)
    ref_albef.ALBEF = lambda config, text_encoder, text_decoder, tokenizer: build_albef_module(d)
    ref_albef.BertTokenizer = types.SimpleNamespace(from_pretrained=lambda *a, **k: None)
    ref_albef.ALBEFWrapper = lambda model, device: model
    with tempfile.TemporaryDirectory() as tmp:
        path = write_albef_checkpoint(os.path.join(tmp, "ALBEF.pth"), vit_depth=d.vit_depth, enc_layers=d.enc_layers,
                                      pre_image=LOADER_PRE_IMAGE, vocab=d.vocab, max_pos=d.max_pos)
        m = ref_albef.load_albef(logging.getLogger("g14"), {"text_encoder": None, "text_decoder": None, "distill": False},
                                 path, torch.device("cpu"), path)
    model = build_reference_model(d, albef_module=m)
    rec = {}
    for k, v in model.state_dict().items():
        if "adapter_" in k or "position_ids" in k:
            continue
        f = v.detach().float().flatten()
        rec["norm::" + k] = np_(f.norm())
        rec["samp::" + k] = np_(f[(torch.arange(8, dtype=torch.int64) * (f.numel() - 1)) // 7])
    b0 = A.synthetic_batch(3, d, 1400, q_len=12, a_len=5, k=[2, 1, 3], ragged=True)
    with torch.no_grad():
        for mode in ("gating", "adapter_1"):
            set_mode(model, mode)
            loss, logits = model("art", dict(b0, train=True))
            rec[f"fwd.{mode}.loss"], rec[f"fwd.{mode}.logits"] = np_(loss), np_(logits)
    np.savez_compressed(os.path.join(out, "g14_albef_pretrained.npz"), **rec)
    print("G14 keys", len(rec), "losses", float(rec["fwd.gating.loss"]), float(rec["fwd.adapter_1.loss"]))


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden")
    torch.manual_seed(0)
    if "--only-g14" in sys.argv:
        golden_loader(out)
        sys.exit(0)
    if "--only-g12" in sys.argv:
        golden_dropout(out)
        sys.exit(0)
    if "--only-g11" in sys.argv:
        golden_round(out)
        sys.exit(0)
    if "--only-g11b" in sys.argv:          # full size, 40 steps: ~1 CPU-hour
        golden_round_full(out, seed0=int(sys.argv[sys.argv.index("--seed0") + 1]) if "--seed0" in sys.argv else 7700)
        sys.exit(0)
    golden_small(out)
    golden_round(out)
    if "--small-only" not in sys.argv:
        golden_full(out)
