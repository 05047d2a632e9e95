"""Generate tests/golden/*.npz by running the REFERENCE's own modules (read-only import from
/root/reference) on deterministic, name-seeded parameters and synthetic batches.

Run in the build container only:  python oracle/make_golden.py
The reference never travels to the GPU box; only the small .npz fixtures written here do.

Reference code exercised (imported, not copied):
  src/modeling/models/adapter.py        Adapter
  src/modeling/adaptered_output.py      Adaptered_ViltOutput
  src/modeling/vilt.py                  ViltEncoderWrapper, ViltContinualLearner
  src/train/visionlanguage_tasks/task_trainer.py   TaskTrainer.train_step/create_optimizer, kl_loss
  src/train/main.py                     get_average_net  (function source exec'd: the module itself
                                        imports transformers.adapters, which does not exist here)
around HuggingFace transformers' ViltModel(ViltConfig()) (random init replaced by seeded fill).
Shims are the ones listed in SURVEY.md section 8c.
"""
import ast
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.nn as nn
import transformers  # noqa: F401  (must be imported before the stubs below)
import accelerate  # noqa: F401

from oracle import feddat_oracle as O


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_shims():
    class _LoraLinear(nn.Linear):
        def __init__(self, i, o, r=0, **kw):
            super().__init__(i, o)
    _stub("loralib", Linear=_LoraLinear)
    _stub("timm")
    _stub("timm.models")
    _stub("timm.models.vision_transformer", _cfg=lambda **k: {}, PatchEmbed=nn.Identity)
    _stub("timm.models.registry", register_model=lambda f: f)
    _stub("timm.models.layers", trunc_normal_=nn.init.trunc_normal_, DropPath=nn.Identity)
    sys.path.insert(0, REF)
    # bypass src/modeling/__init__.py (pulls in ALBEF -> HF 4.x internals)
    for pkg, sub in (("src", "src"), ("src.modeling", "src/modeling"), ("src.modeling.models", "src/modeling/models"),
                     ("src.train", "src/train"), ("src.train.visionlanguage_tasks", "src/train/visionlanguage_tasks"),
                     ("src.utils", "src/utils")):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, sub)]
        sys.modules[pkg] = m
    # adapter.py:144 hard-codes .to('cuda'): send it to CPU
    _orig_to = torch.Tensor.to

    def _to(self, *a, **k):
        if a and isinstance(a[0], str) and a[0].startswith("cuda"):
            a = ("cpu",) + tuple(a[1:])
        return _orig_to(self, *a, **k)
    torch.Tensor.to = _to


install_shims()
from src.modeling.models.adapter import Adapter  # noqa: E402
from src.modeling.vilt import ViltEncoderWrapper, ViltContinualLearner  # noqa: E402
from src.train.visionlanguage_tasks.task_trainer import TaskTrainer, kl_loss  # noqa: E402
from transformers import ViltConfig, ViltModel  # noqa: E402


def load_get_average_net():
    src = open(os.path.join(REF, "src/train/main.py")).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "get_average_net"][0]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "main.py:get_average_net", "exec"), ns)
    return ns["get_average_net"]


def build_reference_model(d: O.ViltDims, tasks, bias_std):
    cfg = ViltConfig(num_hidden_layers=d.layers, image_size=d.image_size)
    vilt = ViltModel(cfg)
    enc = ViltEncoderWrapper.__new__(ViltEncoderWrapper)  # its __init__ loads ./models/bert-base-uncased
    nn.Module.__init__(enc)
    enc.processor = None
    enc.vilt = vilt
    enc.device = torch.device("cpu")
    enc.max_text_length = cfg.max_position_embeddings
    enc.encoder_dim = cfg.hidden_size
    enc.expand_modality_type_embeddings()           # vilt.py:102-113
    enc.process_inputs = lambda images, texts: images  # pre-built tensor dict rides in `images`
    task_cfg = {t: {"num_labels": d.num_labels, "num_images": 1, "model_type": "classification"} for t in tasks}
    model = ViltContinualLearner(list(tasks), enc, cfg.hidden_size, task_cfg, torch.device("cpu"),
                                 {"names": ["adapter_0", "adapter_1", "adapter_2"], "device": "cpu"})
    for p in model.parameters():                    # main.py:138-139
        p.requires_grad = False
    # add_adapter() hard-codes range(12) (vilt.py:357); restate its body for d.layers
    from src.modeling.adaptered_output import Adaptered_ViltOutput
    for i in range(d.layers):
        model.vilt_encoder.vilt.encoder.layer[i].output = Adaptered_ViltOutput(
            model.vilt_encoder.vilt.encoder.layer[i].output, model.adapter_config)
    for n, p in model.named_parameters():           # main.py:157-159, 248-250
        if "adapter" in n or "task" in n:
            p.requires_grad = True
    model.comm_state_dict_names = [n for n in model.state_dict().keys() if "adapter_1" in n]
    sd = model.state_dict()
    shapes = O.param_shapes(d, tasks)
    missing = [k for k in shapes if k not in sd]
    assert not missing, missing[:5]
    with torch.no_grad():
        for k, shp in shapes.items():
            assert tuple(sd[k].shape) == tuple(shp), (k, sd[k].shape, shp)
            sd[k].copy_(O.seeded_value(k, shp, 0.02, bias_std))
    extra = [k for k in sd if k not in shapes and "position_ids" not in k and "token_type_ids" not in k]
    assert not extra, extra[:5]
    model.eval()
    return model


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


def make_trainer(task, lr, steps_per_epoch, num_epochs=15):
    tr = TaskTrainer()
    tr.accelerator = _Acc()
    tr.device = torch.device("cpu")
    tr.task_key = task
    tr.args = types.SimpleNamespace(optimizer_mode="dat", encoder_name="vilt", debug=0)
    tr.batch2inputs_converter = lambda batch: {"images": _enc_only(batch), "texts": None}
    tr.loss_criterion = nn.BCEWithLogitsLoss(reduction="mean")
    tr.lr, tr.adam_epsilon, tr.weight_decay = lr, 1e-8, 1e-2
    tr.max_steps = steps_per_epoch * num_epochs
    tr.warmup_ratio = 0.1
    return tr


def ref_local_update(model, task, batches, lr, num_epochs=15, capture=None):
    """task_trainer.py:36-59 prologue + train_step loop, driven on the reference objects."""
    from transformers import get_polynomial_decay_schedule_with_warmup
    sd = model.state_dict()
    for name in sd.keys():
        if "adapter_1" in name:
            sd[name.replace("adapter_1", "adapter_2")].data.copy_(sd[name].data.clone())
    for n, p in model.named_parameters():
        if "adapter_2" in n:
            p.requires_grad = False
    tr = make_trainer(task, lr, len(batches), num_epochs)
    opt = tr.create_optimizer(model, "dat")
    sch = get_polynomial_decay_schedule_with_warmup(opt, num_warmup_steps=int(tr.max_steps * tr.warmup_ratio),
                                                    num_training_steps=tr.max_steps, lr_end=0, power=1)
    model.zero_grad()
    w = _Wrap(model)
    losses = []
    for step, b in enumerate(batches):
        loss = tr.train_step(w, step, dict(b), opt, sch)
        losses.append(float(loss))
        if capture is not None:
            capture(step, model)
    return losses, opt


def _enc_only(b):
    return {k: v for k, v in b.items() if k != "target_scores"}


def np_(t):
    return t.detach().cpu().numpy().astype(np.float32)


SAMPLE_ABOVE = 40000
N_SAMPLES = 2048


def put(rec, key, t):
    """Small tensors are stored whole; large ones as L2 norm + N_SAMPLES strided samples
    (tests/golden_util.py:check reads both forms)."""
    t = t.detach().float()
    if t.numel() <= SAMPLE_ABOVE:
        rec[key] = np_(t)
    else:
        flat = t.flatten()
        idx = torch.linspace(0, flat.numel() - 1, N_SAMPLES).long()
        rec["samp::" + key] = np_(flat[idx])
        rec["norm::" + key] = np_(flat.norm())


G6_VALID = [(384, 384), (256, 384), (384, 224), (160, 320)]
G6_TEXT = [40, 31, 40, 12]


def golden_g6(out):
    """G6: padded images (pixel_mask with zeros, per-sample resized position embeddings) + padded questions, through
    the reference's ViltContinualLearner around HF ViltModel (visual_embed as installed): forward features for the
    gated and adapter_1 passes, then 2 train_steps."""
    d = O.ViltDims(layers=2)
    model = build_reference_model(d, ["art"], bias_std=0.02)
    batches = [O.pad_batch(O.synthetic_batch(4, 384, 6000 + s), G6_VALID, G6_TEXT) for s in range(2)]
    rec = {}
    with torch.no_grad():
        for mode in ("gating", "adapter_1"):
            if mode == "gating":
                model.activate_gating()
            else:
                model.deactivate_gating()
                model.set_active_adapter(mode)
            pooled, lg = model(task_key="art", images=_enc_only(batches[0]), texts=None)
            rec[f"fwd.{mode}.pooled"] = np_(pooled)
            rec[f"fwd.{mode}.logits"] = np_(lg)
    for n, p in model.named_parameters():
        if "adapter" in n:
            p.requires_grad = True
    losses, opt = ref_local_update(model, "art", batches, lr=1e-4)
    rec["losses"] = np.array(losses, np.float32)
    for k, v in model.state_dict().items():
        if ".layer.1." in k and ("adapter_0" in k or "adapter_1" in k):
            put(rec, "after2." + k, v)
    np.savez_compressed(os.path.join(out, "g6_padded.npz"), **rec)
    print("G6 losses", losses)


def golden_g8(out, steps=40):
    """(steps = 40: g8_round40.npz; steps = 80: g8_round80.npz, 1200 scheduler ticks, warm-up 120 ticks = 60 batches.)
    G8: one REALISTIC local round of the full 12-layer ViLT-B/32 on the reference: B=4, 384x384, len(loader)=40,
    num_epochs=15 -> N=600 scheduler ticks, warm-up 60 ticks = 30 batches, so the last 10 batches run at lr ~ 1e-4
    (task_trainer.py:53-59).  Stored per adapter_0 / adapter_1 / head tensor: the UPDATE dW = W_after - W_init as L2 norm,
    mean |dW| and 1024 strided samples (W_init is the name-seeded fill, regenerated by the tests), so that parity is
    asserted on the distance the weights moved, not on the weights."""
    d = O.ViltDims(layers=12)
    model = build_reference_model(d, ["art"], bias_std=0.02)
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batches = [O.synthetic_batch(4, 384, 8000 + s) for s in range(steps)]
    losses, _ = ref_local_update(model, "art", batches, lr=1e-4)
    rec = {"losses": np.array(losses, np.float32), "steps": np.array(steps)}
    for k, v in model.state_dict().items():
        if "adapter_0" in k or "adapter_1" in k or k.startswith("task_layer.art."):
            dw = (v.detach() - init[k]).flatten()
            idx = torch.linspace(0, dw.numel() - 1, min(1024, dw.numel())).long()
            rec["dnorm::" + k] = np_(dw.norm())
            rec["dmean::" + k] = np_(dw.abs().mean())
            rec["dmax::" + k] = np_(dw.abs().max())
            rec["dsamp::" + k] = np_(dw[idx])
    np.savez_compressed(os.path.join(out, f"g8_round{steps}.npz"), **rec)
    print("G8 losses", losses[:3], "...", losses[-3:])


def golden_g8b(out, steps=80, batch=32, snaps=(20, 40, 60, 80), seed0=8000, all_elements=False):
    """G8b: the same realistic local round as G8, at configs[1]/[2]'s OWN batch size (B = 32, 384x384, S = 185) on the
    reference: len(loader) = steps, num_epochs = 15 (task_trainer.py:53-59: 1200 ticks, warm-up 120 ticks = 60 batches
    at steps = 80).  The update dW = W_after_n - W_init is stored at every n in `snaps`, so one fixture serves all round
    lengths: per adapter_0 / adapter_1 / head tensor its L2 norm, mean |dW|, max |dW| and 1024 strided samples.
    seed0: batch seeds seed0 .. seed0 + steps - 1 (8000 = the original fixture; any other value goes into the file name).
    all_elements: additionally store EVERY element of every trainable tensor's update at the last snapshot, as float16 of
    dW * 256 (|dW| <= 1e-2: the encoding error is < 3e-6 absolute, 300x below the 1e-3 bar) in a second
    file g8b_round<N>_b<B>_all.npz, so that the GPU suite checks the tail of the distribution without stepping the oracle."""
    d = O.ViltDims(layers=12)
    model = build_reference_model(d, ["art"], bias_std=0.02)
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batches = [O.synthetic_batch(batch, 384, seed0 + s) for s in range(steps)]
    rec = {"steps": np.array(steps), "batch": np.array(batch), "snaps": np.array(snaps), "seed0": np.array(seed0)}
    full = {"steps": np.array(steps), "batch": np.array(batch), "seed0": np.array(seed0), "scale": np.array(256.0)}
    full_snaps = (snaps[-1],) if all_elements else ()
    full["snaps"] = np.array(full_snaps)

    def cap(step, m):
        n = step + 1
        print("G8b step", n, flush=True)
        if n not in snaps:
            return
        for k, v in m.state_dict().items():
            if "adapter_0" in k or "adapter_1" in k or k.startswith("task_layer.art."):
                dw = (v.detach() - init[k]).flatten()
                idx = torch.linspace(0, dw.numel() - 1, min(1024, dw.numel())).long()
                rec[f"s{n}::dnorm::" + k] = np_(dw.norm())
                rec[f"s{n}::dmean::" + k] = np_(dw.abs().mean())
                rec[f"s{n}::dmax::" + k] = np_(dw.abs().max())
                rec[f"s{n}::dsamp::" + k] = np_(dw[idx])
                if n in full_snaps:
                    full[f"s{n}::dall::" + k] = (dw * 256.0).to(torch.float16).numpy().reshape(tuple(v.shape))

    losses, _ = ref_local_update(model, "art", batches, lr=1e-4, capture=cap)
    rec["losses"] = np.array(losses, np.float32)
    tag = "" if seed0 == 8000 else f"_seed{seed0}"
    np.savez_compressed(os.path.join(out, f"g8b_round{steps}_b{batch}{tag}.npz"), **rec)
    if all_elements:
        np.savez_compressed(os.path.join(out, f"g8b_round{steps}_b{batch}{tag}_all.npz"), **full)
    print("G8b losses", losses[:3], "...", losses[-3:])


G15_OVERFLOW = {2: (True, False), 4: (False, True), 5: (True, True)}      # step -> (sub-step A, sub-step B) overflowed


def golden_g15(out):
    """G15: what the reference does when a scaled fp16 backward overflows -- pinned on accelerate's OWN optimizer / scheduler
    wrappers (accelerate.optimizer.AcceleratedOptimizer + accelerate.scheduler.AcceleratedScheduler around the reference's
    create_optimizer / HF poly schedule, torch.amp.GradScaler with its defaults), which is what accelerator.prepare() puts
    around them under mixed_precision fp16 (accelerate_config.yaml:8; task_trainer.py:63,302-308,323-328).  CPU fp32 arithmetic
    (the only kind this container has); the overflow is INJECTED: accelerator.backward = scaler.scale(loss).backward() as in
    accelerate, then one adapter gradient element is set to inf in the sub-steps listed in G15_OVERFLOW.  2-layer model,
    B = 4, 224 x 224, 7 steps.  Stored: losses, the scale and the scheduler index after every step, final trainable tensors."""
    from accelerate import Accelerator
    from accelerate.optimizer import AcceleratedOptimizer
    from accelerate.scheduler import AcceleratedScheduler
    from transformers import get_polynomial_decay_schedule_with_warmup
    Accelerator(cpu=True)                       # initialises accelerate's process state (no mixed precision on CPU)
    d = O.ViltDims(layers=2)
    model = build_reference_model(d, ["art"], bias_std=0.02)
    steps = 7
    batches = [O.synthetic_batch(4, 224, 1500 + s) for s in range(steps)]
    sd = model.state_dict()
    for name in sd.keys():
        if "adapter_1" in name:
            sd[name.replace("adapter_1", "adapter_2")].data.copy_(sd[name].data.clone())
    for n, p in model.named_parameters():
        if "adapter_2" in n:
            p.requires_grad = False
    tr = make_trainer("art", 1e-4, steps, 15)
    scaler = torch.amp.GradScaler("cpu")        # defaults: 65536, x2 / x0.5, growth interval 2000
    calls = {"n": 0}

    class Acc:
        device = torch.device("cpu")

        @staticmethod
        def backward(loss):
            scaler.scale(loss).backward()       # accelerate.Accelerator.backward under fp16
            step, sub = divmod(calls["n"], 2)
            calls["n"] += 1
            if G15_OVERFLOW.get(step, (False, False))[sub]:
                name = "adapter_1" if sub == 0 else "adapter_0"
                tgt = [p for n, p in model.named_parameters() if name + "_up.weight" in n and p.grad is not None][0]
                tgt.grad.view(-1)[7] = float("inf")
    tr.accelerator = Acc()
    opt = tr.create_optimizer(model, "dat")
    sch = get_polynomial_decay_schedule_with_warmup(opt, num_warmup_steps=int(tr.max_steps * tr.warmup_ratio),
                                                    num_training_steps=tr.max_steps, lr_end=0, power=1)
    aopt = AcceleratedOptimizer(opt, device_placement=False, scaler=scaler)
    asch = AcceleratedScheduler(sch, aopt, step_with_optimizer=True, split_batches=False)
    model.zero_grad()
    w = _Wrap(model)
    rec = {"losses": [], "scale": [], "sched_t": []}
    for step, b in enumerate(batches):
        rec["losses"].append(float(tr.train_step(w, step, dict(b), aopt, asch)))
        rec["scale"].append(scaler.get_scale())
        rec["sched_t"].append(sch.last_epoch)
    rec = {k: np.array(v, np.float32) for k, v in rec.items()}
    rec["overflow_steps"] = np.array(sorted(G15_OVERFLOW), np.int64)
    rec["overflow_ab"] = np.array([G15_OVERFLOW[k] for k in sorted(G15_OVERFLOW)], np.int64)
    for k, v in model.state_dict().items():
        if "adapter_0" in k or "adapter_1" in k or k.startswith("task_layer.art."):
            put(rec, "after." + k, v)
    np.savez_compressed(os.path.join(out, "g15_scaler_skip.npz"), **rec)
    print("G15 losses", rec["losses"], "scale", rec["scale"], "sched_t", rec["sched_t"])


def golden_g13(out):
    """G13: PRETRAINED-weight loading.  tests/ckpt_util.write_hf_vilt_checkpoint lays down a HuggingFace directory in the
    layout of dandelin/vilt-b32-mlm (ViltForMaskedLM keys, 2 layers, key-seeded fill); here it is loaded the way the reference
    does -- ViltModel.from_pretrained(dir) (vilt.py:401-405), ViltEncoderWrapper.expand_modality_type_embeddings
    (vilt.py:102-113, called from its __init__ :53), ViltContinualLearner + Adaptered_ViltOutput -- and the fixture keeps, per
    state-dict key of the resulting model, the tensor's L2 norm and 8 samples, plus pooled features / logits of two forward
    modes with name-seeded adapters and heads.  feddat_amd.weights must reproduce every entry from the same directory."""
    import tempfile
    from tests.ckpt_util import write_hf_vilt_checkpoint
    d = O.ViltDims(layers=2)
    with tempfile.TemporaryDirectory() as tmp:
        write_hf_vilt_checkpoint(tmp, layers=2)
        vilt, info = ViltModel.from_pretrained(tmp, output_loading_info=True)
        assert not info["missing_keys"], info["missing_keys"][:5]           # HF itself finds every ViltModel tensor in the file
        assert all(k.startswith("mlm_score") for k in info["unexpected_keys"]), info["unexpected_keys"][:5]
    enc = ViltEncoderWrapper.__new__(ViltEncoderWrapper)
    nn.Module.__init__(enc)
    enc.processor, enc.vilt, enc.device = None, vilt, torch.device("cpu")
    enc.max_text_length, enc.encoder_dim = vilt.config.max_position_embeddings, vilt.config.hidden_size
    enc.expand_modality_type_embeddings()
    enc.process_inputs = lambda images, texts: images
    task_cfg = {"art": {"num_labels": d.num_labels, "num_images": 1, "model_type": "classification"}}
    model = ViltContinualLearner(["art"], enc, vilt.config.hidden_size, task_cfg, torch.device("cpu"),
                                 {"names": ["adapter_0", "adapter_1", "adapter_2"], "device": "cpu"})
    from src.modeling.adaptered_output import Adaptered_ViltOutput
    for i in range(d.layers):
        model.vilt_encoder.vilt.encoder.layer[i].output = Adaptered_ViltOutput(
            model.vilt_encoder.vilt.encoder.layer[i].output, model.adapter_config)
    sd = model.state_dict()
    with torch.no_grad():
        for k, shp in O.param_shapes(d, ["art"]).items():
            if "adapter_" in k or k.startswith("task_layer."):
                sd[k].copy_(O.seeded_value(k, shp, 0.02, 0.02))
    model.eval()
    rec = {}
    for k, v in model.state_dict().items():
        if "adapter_" in k or k.startswith("task_layer.") or "position_ids" in k or "token_type_ids" in k:
            continue
        f = v.detach().float().flatten()
        rec["norm::" + k] = np_(f.norm())
        rec["samp::" + k] = np_(f[(torch.arange(8, dtype=torch.int64) * (f.numel() - 1)) // 7])
    batch = O.synthetic_batch(2, 384, 1300)
    with torch.no_grad():
        for mode in ("gating", "adapter_1"):
            if mode == "gating":
                model.activate_gating()
            else:
                model.deactivate_gating()
                model.set_active_adapter(mode)
            pooled, lg = model(task_key="art", images=_enc_only(batch), texts=None)
            rec[f"fwd.{mode}.pooled"], rec[f"fwd.{mode}.logits"] = np_(pooled), np_(lg)
    np.savez_compressed(os.path.join(out, "g13_vilt_pretrained.npz"), **rec)
    print("G13 keys", len(rec), "logits", rec["fwd.gating.logits"][0, :4])


def main():
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    torch.manual_seed(0)
    if "--only-g8" in sys.argv:          # [--steps N]: the same round at another length (g8_round<N>.npz; 80 = the longest
        steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 40      # len(loader) of configs[2])
        if "--batch" in sys.argv:        # --batch 32: the round at configs[1]'s own batch (g8b_round<N>_b<B>.npz,
            seed0 = int(sys.argv[sys.argv.index("--seed0") + 1]) if "--seed0" in sys.argv else 8000
            golden_g8b(out, steps, int(sys.argv[sys.argv.index("--batch") + 1]), seed0=seed0,      # updates stored every 20 steps)
                       all_elements="--all-elements" in sys.argv)
            return
        golden_g8(out, steps)
        return
    if "--only-g13" in sys.argv:
        golden_g13(out)
        return
    if "--only-g15" in sys.argv:
        golden_g15(out)
        return
    if "--only-g6" in sys.argv:          # the other fixtures are unchanged; regenerate just this one
        golden_g6(out)
        return

    # ---------------- G1: Adapter module (adapter.py:124-163) ----------------
    ad = Adapter(["adapter_0", "adapter_1", "adapter_2"], "cpu")
    g = torch.Generator().manual_seed(11)
    T = 128
    with torch.no_grad():
        for n, p in ad.named_parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.05 if "weight" in n else 0.02))
    x = torch.randn(2, T // 2, 768, generator=g)
    dy = torch.randn(2, T // 2, 768, generator=g)
    res = {"x": np_(x), "dy": np_(dy)}
    for n, p in ad.named_parameters():
        res["p." + n] = np_(p)
    for mode in ("adapter_1", "gating"):
        for p in ad.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        if mode == "gating":
            ad.activate_gating()
            ad.set_active_adapter("adapter_0")
        else:
            ad.deactivate_gating()
            ad.set_active_adapter("adapter_1")
        y = ad(xi, xi)
        y.backward(dy)
        res[f"{mode}.y"] = np_(y)
        res[f"{mode}.dx"] = np_(xi.grad)
        for n, p in ad.named_parameters():
            if p.grad is not None:
                res[f"{mode}.d.{n}"] = np_(p.grad)
    np.savez_compressed(os.path.join(out, "g1_adapter.npz"), **res)

    # ---------------- G2: loss (task_trainer.py:299-301,506-516) ----------------
    logits = (torch.randn(8, 100, generator=g) * 2).requires_grad_(True)
    teacher = torch.randn(8, 100, generator=g) * 2
    target = O.synthetic_batch(8, 32, 5)["target_scores"]
    bce = nn.BCEWithLogitsLoss(reduction="mean")(logits, target) * target.shape[1]
    kl = kl_loss(logits, teacher.clone().detach())
    L = (bce + kl) / 2
    L.backward()
    np.savez_compressed(os.path.join(out, "g2_loss.npz"), logits=np_(logits), teacher=np_(teacher),
                        target=np_(target), bce=np_(bce), kl=np_(kl), L=np_(L), dlogits=np_(logits.grad))

    # ---------------- G3: 2-layer ViLT, B=4, 224 and 384 (configs[0]) ----------------
    gavg = load_get_average_net()
    for res_px in (224, 384):
        d = O.ViltDims(layers=2)
        tasks = ["art", "gqa"]
        model = build_reference_model(d, tasks, bias_std=0.02)
        batches = [O.synthetic_batch(4, res_px, 1234 + s) for s in range(5)]
        rec = {}
        with torch.no_grad():
            for mode in ("gating", "adapter_1", "adapter_0"):
                if mode == "gating":
                    model.activate_gating()
                else:
                    model.deactivate_gating()
                    model.set_active_adapter(mode)
                pooled, lg = model(task_key="art", images=_enc_only(batches[0]), texts=None)
                rec[f"fwd.{mode}.pooled"] = np_(pooled)
                rec[f"fwd.{mode}.logits"] = np_(lg)
        # the forward captures above left the flags in the post-set_active_adapter('adapter_0')
        # state; restore prepare_model's round-0 state (main.py:157-159): all adapters trainable
        for n, p in model.named_parameters():
            if "adapter" in n:
                p.requires_grad = True
        snaps = {}

        def cap(step, m, snaps=snaps):
            if step + 1 in (1, 2, 5):
                sd = m.state_dict()
                snaps[step + 1] = {k: v.detach().clone() for k, v in sd.items()
                                   if ("adapter_0" in k or "adapter_1" in k or k.startswith("task_layer.art."))}
        losses, opt = ref_local_update(model, "art", batches, lr=1e-4, capture=cap)
        rec["losses"] = np.array(losses, np.float32)
        for n_steps, s in snaps.items():
            for k, v in s.items():
                if "adapter" in k and ".layer.0." not in k and n_steps != 5:
                    continue  # keep the fixture small: layer-0 adapters at 1,2; everything at 5
                put(rec, f"after{n_steps}.{k}", v)
        np.savez_compressed(os.path.join(out, f"g3_vilt2_{res_px}.npz"), **rec)
        print("G3", res_px, "losses", losses)

    # ---------------- G3q: optimizer membership follows requires_grad at train() time ----------
    # After TaskTrainer.eval (task_trainer.py:236-244) the server model is left in the
    # set_active_adapter('adapter_1') state, i.e. adapter_0.requires_grad == False; the next
    # round's create_optimizer (task_trainer.py:477-504) then leaves adapter_0 out.
    d = O.ViltDims(layers=2)
    model = build_reference_model(d, ["art", "gqa"], bias_std=0.02)
    model.deactivate_gating()
    model.set_active_adapter("adapter_1")
    batches = [O.synthetic_batch(4, 224, 1234 + s) for s in range(3)]
    losses, opt = ref_local_update(model, "art", batches, lr=1e-4)
    rec = {"losses": np.array(losses, np.float32)}
    for k, v in model.state_dict().items():
        if ".layer.1." in k and ("adapter_0" in k or "adapter_1" in k):
            put(rec, "after3." + k, v)
    np.savez_compressed(os.path.join(out, "g3q_flags.npz"), **rec)

    # ---------------- G5: FedAvg (main.py:50-65) + one round, 2 clients x 3 steps ----------------
    d = O.ViltDims(layers=2)
    tasks = ["art", "gqa"]
    server = build_reference_model(d, tasks, bias_std=0.02)
    import copy
    personal = {t: {n: v.clone() for n, v in server.state_dict().items()
                    if any(pn in n for pn in ("task", "adapter_0", "adapter_2"))} for t in tasks}
    c_models = []
    rec = {}
    for ci, t in enumerate(tasks):
        tm = copy.deepcopy(server)
        for n, v in personal[t].items():
            tm.state_dict()[n].data.copy_(v)
        batches = [O.synthetic_batch(4, 224, 777 + 10 * ci + s) for s in range(3)]
        losses, _ = ref_local_update(tm, t, batches, lr=1e-4)
        rec[f"losses.{t}"] = np.array(losses, np.float32)
        c_models.append({n: tm.state_dict()[n].data.clone() for n in server.comm_state_dict_names})
        for n, v in tm.state_dict().items():
            if "adapter_0" in n and ".layer.1." in n:
                rec[f"personal.{t}.{n}"] = np_(v)
    server = gavg(server, c_models, [1, 1], tasks, torch.device("cpu"))
    for n in server.comm_state_dict_names:
        rec["server." + n] = np_(server.state_dict()[n])
    np.savez_compressed(os.path.join(out, "g5_round.npz"), **rec)
    # pure FedAvg, K=5 synthetic client dicts, non-uniform nums
    K = 5
    keys = [k for k in server.comm_state_dict_names if ".layer.0." in k]
    cm = [{k: torch.randn(server.state_dict()[k].shape, generator=g) for k in keys} for _ in range(K)]
    srv = types.SimpleNamespace(comm_state_dict_names=keys,
                                state_dict=lambda sd={k: torch.zeros_like(cm[0][k]) for k in keys}: sd)
    nums = [1, 2, 3, 4, 5]
    gavg(srv, cm, nums, None, torch.device("cpu"))
    rec = {f"c{i}.{k}": np_(cm[i][k]) for i in range(K) for k in keys}
    rec.update({"avg." + k: np_(srv.state_dict()[k]) for k in keys})
    rec["nums"] = np.array(nums, np.float32)
    np.savez_compressed(os.path.join(out, "g5_fedavg.npz"), **rec)

    # ---------------- G4: full 12-layer, B=4, 384: logits x3, loss, sampled adapter tensors after 4 steps
    d = O.ViltDims(layers=12)
    model = build_reference_model(d, ["art"], bias_std=0.02)
    batches = [O.synthetic_batch(4, 384, 4321 + s) for s in range(4)]
    rec = {}
    with torch.no_grad():
        for mode in ("gating", "adapter_1"):
            if mode == "gating":
                model.activate_gating()
            else:
                model.deactivate_gating()
                model.set_active_adapter(mode)
            pooled, lg = model(task_key="art", images=_enc_only(batches[0]), texts=None)
            rec[f"fwd.{mode}.pooled"] = np_(pooled)
            rec[f"fwd.{mode}.logits"] = np_(lg)
    for n, p in model.named_parameters():   # back to prepare_model's round-0 flags
        if "adapter" in n:
            p.requires_grad = True
    losses, _ = ref_local_update(model, "art", batches, lr=1e-4)
    rec["losses"] = np.array(losses, np.float32)
    for k, v in model.state_dict().items():
        if "adapter_0" in k or "adapter_1" in k:
            flat = v.flatten()
            idx = torch.linspace(0, flat.numel() - 1, 256).long()
            rec["norm::" + k] = np_(flat.norm())
            rec["samp256::" + k] = np_(flat[idx])
    np.savez_compressed(os.path.join(out, "g4_vilt12_384.npz"), **rec)
    golden_g6(out)
    golden_g8(out)
    golden_g15(out)
    print("G4 losses", losses)
    for f in sorted(os.listdir(out)):
        print(f, os.path.getsize(os.path.join(out, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
