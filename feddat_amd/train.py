"""Client trainer + FL outer loop mirroring the reference (src/train/visionlanguage_tasks/task_trainer.py,
src/train/main.py) on the MI355X engine.

  TaskTrainer.train(model, ...)        -> (0., model)      task_trainer.py:24-111
  TaskTrainer.train_step(model, step, batch, optimizer, scheduler)  -> loss_0     task_trainer.py:266-330
  TaskTrainer.create_optimizer(model)  -> handle           task_trainer.py:477-504
  TaskTrainer.eval(model)              -> [score_gated, score_adapter0, score_adapter1]   task_trainer.py:211-246
  get_average_net(server, c_models, nums, ordered_tasks, device)   main.py:50-65   (feddat_amd.fedavg)
  main(argv)                           -> FL rounds x clients, same flags as main.py:262-323

Mapping of the outer loop: the reference visits the clients sequentially on one device from identical server
state (main.py:466-504).  Here one process owns one GPU and one or more clients; with torch.distributed initialised
(backend "nccl" == RCCL) the per-round average is ONE all-reduce of the flat adapter_1 buffer over xGMI.
"""
from __future__ import annotations

import argparse
import logging
import os
from typing import Dict, List, Optional, Sequence

import torch

from . import lib as L
from . import vilt_spec
from .fedavg import all_reduce_sum, allreduce_average, get_average_net  # noqa: F401  (re-exported: main.py:50)
from .modeling import ViltContinualLearner, convert_batch_to_vilt_input_dict, create_vilt_continual_learner_model


class OptimizerHandle:
    """What create_optimizer returns: the AdamW state lives on the device inside the engine (flat fp32 moments,
    step counters); the handle records which adapters are members (requires_grad at creation time) and hparams."""

    def __init__(self, adapters: Sequence[int], lr: float, eps: float, weight_decay: float):
        self.adapters, self.lr, self.eps, self.weight_decay = tuple(adapters), lr, eps, weight_decay

    def step(self):       # the fused train_step applies both AdamW sub-steps itself
        pass

    def zero_grad(self):
        pass


class SchedulerHandle:
    def __init__(self, num_warmup_steps: int, num_training_steps: int):
        self.num_warmup_steps, self.num_training_steps = num_warmup_steps, num_training_steps

    def step(self):
        pass


def get_polynomial_decay_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, lr_end=0, power=1):
    """HF signature used at task_trainer.py:53-59; lr_end=0 / power=1 is the only form the reference uses and the
    only one the device-side schedule implements."""
    if lr_end != 0 or power != 1:
        raise L.FeddatHipError("only lr_end=0, power=1 (task_trainer.py:57-58) is implemented on device")
    return SchedulerHandle(num_warmup_steps, num_training_steps)


def kl_loss(output: torch.Tensor, target: torch.Tensor, temp: float = 3.0) -> torch.Tensor:
    """task_trainer.py:506-516 (T^2 * KL_batchmean(log_softmax(out/T) || softmax(tgt/T))) via the fused loss kernel
    (the BCE half of its output is ignored here)."""
    B = output.shape[0]
    buf = torch.empty(4 + 2 * B, device=output.device)
    dl = torch.empty_like(output)
    L.dat_loss_fwd_bwd(output.contiguous(), target.contiguous(), torch.zeros_like(output), dl, buf, temp)
    return buf[1]


class TaskTrainer:
    """VQATrainerCross + TaskTrainer for the dat optimizer_mode (train_vqa_crossvqa.py:39-239, task_trainer.py)."""

    def __init__(self, args, task_key: str, train_batches: List[Dict[str, torch.Tensor]],
                 eval_batches: Optional[List[Dict[str, torch.Tensor]]] = None, logger=None):
        self.args = args
        self.task_key = task_key
        self.vqa_train_dataloader = train_batches
        self.vqa_test_dataloader = eval_batches or []
        self.local_epochs = args.local_epochs
        self.num_epochs = args.num_epochs                              # train_vqa_crossvqa.py:233
        self.lr = args.lr
        self.adam_epsilon, self.weight_decay, self.warmup_ratio = 1e-8, 1e-2, 0.1   # task_configs_fed.py:47-50
        self.max_steps = len(train_batches) * self.num_epochs          # train_vqa_crossvqa.py:238
        self.logger = logger or logging.getLogger("feddat_amd")
        self.use_graph = getattr(args, "hip_graph", True)
        self.batch2inputs_converter = convert_batch_to_vilt_input_dict     # train_vqa_crossvqa.py:70

    def encode_batch(self, model: ViltContinualLearner, batch: Dict) -> Dict[str, torch.Tensor]:
        """The reference's batches are {"images": [PIL / uint8 arrays], "raw_texts": [str], "target_scores": [B, C]}
        (vqa_dataset_crossvqa.py:377-422); the processor + tokenizer run here ONCE per batch on the device
        (task_trainer.py:281,285,293,313 call them inside each of the three forward passes).  Batches that already carry the
        encodings (synthetic benchmarks, the golden harness) pass through."""
        if "pixel_values" in batch:
            return batch
        enc = model.process_inputs(**self.batch2inputs_converter(batch))
        tgt = batch["target_scores"]
        enc["target_scores"] = tgt.to(model.device, torch.float32, non_blocking=True).contiguous()
        return enc

    def create_optimizer(self, model: ViltContinualLearner, mode: str = "dat") -> OptimizerHandle:
        return OptimizerHandle(model.optimizer_adapters(), self.lr, self.adam_epsilon, self.weight_decay)

    def train(self, model: ViltContinualLearner, *unused):
        eng = model.engine
        optimizer = self.create_optimizer(model, self.args.optimizer_mode)
        scheduler = get_polynomial_decay_schedule_with_warmup(
            optimizer, num_warmup_steps=int(self.max_steps * self.warmup_ratio), num_training_steps=self.max_steps,
            lr_end=0, power=1)
        eng.lr, eng.eps, eng.wd = optimizer.lr, optimizer.eps, optimizer.weight_decay
        # adapter_1 -> adapter_2 copy, freeze, fresh moments + schedule          task_trainer.py:36-59
        eng.begin_local_update(self.task_key, steps_per_epoch=len(self.vqa_train_dataloader),
                               num_epochs=self.num_epochs, warmup_ratio=self.warmup_ratio,
                               opt_adapters=optimizer.adapters)
        model.adapter_requires_grad[2] = False
        if self.use_graph:
            eng.ensure_captured()      # before the upload worker below starts: capture never overlaps a prefetch
        loss = None
        dev = model.device

        def upload(b):      # raw batches (images + questions): processor + tokenizer run here, on the prefetch stream
            b = self.encode_batch(model, b)
            return {k: (v.to(dev, non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
        for epoch in range(self.local_epochs):
            # host batches cross PCIe on a side stream while the previous step computes (device batches pass through)
            loader = self.vqa_train_dataloader
            if getattr(self.args, "prefetch", True):
                from .data import DevicePrefetcher
                loader = DevicePrefetcher(loader, upload, dev)
            for step, batch in enumerate(loader):
                if self.args.debug > 0 and step > self.args.debug:     # task_trainer.py:82-83
                    break
                loss = self.train_step(model, step, batch, optimizer, scheduler, hooks=None, epoch=epoch)
        self._finite_check(eng)    # one read-back per local update
        return 0.0, model

    def _finite_check(self, eng):
        """Non-finite trainable state after a local update is an error.  Standalone it raises here; under main() with several
        ranks it is only RECORDED (self.nonfinite) and main() agrees on it across ranks before the FedAvg collective, so that
        every rank raises together instead of one raising while its peers wait in the all-reduce."""
        st = eng.scaler_state() if hasattr(eng, "scaler_state") else None
        if st and st["dynamic"] and st["skipped_substeps"]:      # what accelerate logs as "Gradient overflow. Skipping step"
            self.logger.warning("%s: the loss scaler skipped %d optimizer sub-step(s) in %d batch(es) of this local update; scale now %g",
                                self.task_key, st["skipped_substeps"], st["skipped_batches"], st["scale"])
        if getattr(self, "defer_finite_check", False):
            self.nonfinite = eng.nonfinite_groups()
        else:
            eng.assert_finite()

    def train_step(self, model: ViltContinualLearner, step, batch, optimizer=None, scheduler=None, hooks=None,
                   epoch=None):
        """One DAT + MKD step; returns loss_0 (BCE * num_labels of the P2 pass) as a 0-d device tensor.  The mode
        switches the reference performs inside (activate_gating / set_active_adapter, task_trainer.py:284-312)
        leave the model in the same final state: gating on, adapter_0 active."""
        out = model.engine.train_step(self.encode_batch(model, batch), use_graph=self.use_graph)
        model.activate_gating()
        model.set_active_adapter("adapter_0")
        return out[0]

    @torch.no_grad()
    def eval_one_loader(self, model: ViltContinualLearner, loader) -> float:
        """VQA score (train_vqa_crossvqa.py:241-257; task_trainer.py:125-157): score of the arg-max answer, accumulated on
        the device (feddat_vqa_score_accumulate); ONE read-back per loader."""
        acc = torch.zeros(2, device=model.device)
        for batch in loader:
            batch = self.encode_batch(model, batch)
            _, logits = model(task_key=self.task_key, images=batch, texts=None)
            tgt = batch["target_scores"].to(logits.device, torch.float32, non_blocking=True).contiguous()
            L.vqa_score_accumulate(logits, tgt, acc)
        score, seen = acc.tolist()
        return 100.0 * score / max(seen, 1.0)

    def eval(self, model: ViltContinualLearner):
        loader = self.vqa_test_dataloader
        model.activate_gating()
        s = self.eval_one_loader(model, loader)
        model.deactivate_gating()
        model.set_active_adapter("adapter_0")
        s0 = self.eval_one_loader(model, loader)
        model.deactivate_gating()
        model.set_active_adapter("adapter_1")
        s1 = self.eval_one_loader(model, loader)
        return [s, s0, s1]


class AlbefTaskTrainer(TaskTrainer):
    """The same trainer driving the ALBEF model (encoder_name albef_no_distill: task_trainer.py:250-264,296-297,316-317;
    eval with answer ranking :159-204).  Batches are the tokenised dicts of feddat_amd.albef_spec.synthetic_batch."""

    def train(self, model, *unused):
        eng = model.engine
        optimizer = self.create_optimizer(model, self.args.optimizer_mode)
        eng.lr, eng.eps, eng.wd = optimizer.lr, optimizer.eps, optimizer.weight_decay
        eng.begin_local_update(steps_per_epoch=len(self.vqa_train_dataloader), num_epochs=self.num_epochs,
                               warmup_ratio=self.warmup_ratio, opt_adapters=optimizer.adapters,
                               dropout_epoch=getattr(self, "dropout_epoch", 0))
        model.adapter_requires_grad[2] = False
        for epoch in range(self.local_epochs):
            for step, batch in enumerate(self.vqa_train_dataloader):
                if self.args.debug > 0 and step > self.args.debug:
                    break
                self.train_step(model, step, batch, optimizer, None, hooks=None, epoch=epoch)
        self._finite_check(eng)
        return 0.0, model

    def train_step(self, model, step, batch, optimizer=None, scheduler=None, hooks=None, epoch=None):
        if isinstance(batch, (list, tuple)):           # the reference's collated list (albef.py:275-286)
            from .albef_modeling import convert_batch_to_albef_input_dict
            batch = convert_batch_to_albef_input_dict(batch)
        if "questions" in batch:                       # strings -> device tokenizer, once per batch (task_trainer.py:250-264)
            batch = model.process_inputs(dict(batch, train=True))
        out = model.engine.train_step(batch, use_graph=self.use_graph)
        model.activate_gating()
        model.set_active_adapter("adapter_0")
        return out[0]

    @torch.no_grad()
    def eval_one_loader(self, model, loader) -> float:
        """task_trainer.py:159-204: the top re-ranked answer must be one of the ground-truth answer indices."""
        hit, seen = 0, 0
        for batch in loader:
            ids, probs = model(self.task_key, dict(batch, train=False))
            pred = ids.gather(1, probs.argmax(1, keepdim=True)).squeeze(1).cpu()
            for p, gt in zip(pred.tolist(), batch["gts"]):
                hit += int(p in set(int(x) for x in gt))
                seen += 1
        return 100.0 * hit / max(seen, 1)


# -------------------------------------------------------------------------------------------------------------
def deal_clients(steps: Sequence[int], world: int) -> List[int]:
    """owner[k] = rank of client k.  K <= world: client k on rank k.  K > world: longest-processing-time dealing (clients by
    descending len(loader), ties by client index, each to the currently least-loaded rank, ties to the lowest rank)."""
    K = len(steps)
    if K <= world:
        return list(range(K))
    owner, load = [0] * K, [0] * world
    for k in sorted(range(K), key=lambda k: (-steps[k], k)):
        r = min(range(world), key=lambda r: (load[r], r))
        owner[k], load[r] = r, load[r] + steps[k]
    return owner


def round_efficiency(loads: Sequence[int]) -> float:
    """Upper bound of a round's scaling efficiency with heterogeneous clients: every rank waits at the all-reduce for the
    rank with the most steps, so N ranks deliver sum(loads) / max(loads) ranks' worth of work (SURVEY 8d config 3:
    len(loader) in {40..80} on 8 ranks -> 0.70)."""
    return float(sum(loads)) / (len(loads) * max(max(loads), 1))


def agree_on_exchange(make_comm, world: int, log):
    """RCCL communicator for the round's collective, or None on EVERY rank together.  A rank where creation fails (librccl
    not loadable, bootstrap timeout) must not raise while its peers enter feddat_fedavg_allreduce: all ranks all-reduce(MIN)
    an ok flag over the torch.distributed group and fall back to torch.distributed's all-reduce as one."""
    comm, err = None, None
    try:
        comm = make_comm()
    except L.FeddatHipError as e:
        err = e
    if world > 1:
        import torch.distributed as dist
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32)
        if dist.get_backend() == "nccl":
            ok = ok.cuda()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and comm is not None:
            comm.close()
            comm = None
    if comm is None:
        log.warning("C-ABI RCCL communicator unavailable%s: %s", f" ({err})" if err else " on a peer",
                    "torch.distributed all_reduce takes the exchange on all ranks" if world > 1 else
                    "single rank, the exchange is the identity")
    return comm


def build_parser() -> argparse.ArgumentParser:
    """Same flags as src/train/main.py:262-323 (unused ones are accepted and ignored) + synthetic-data knobs."""
    p = argparse.ArgumentParser()
    p.add_argument("--encoder_name", default="vilt", choices=["vilt", "albef_no_distill"])
    p.add_argument("--portion", default=1.0, type=float)
    p.add_argument("--optimizer_mode", default="dat", type=str)
    p.add_argument("--pretrained_model_name", default=None, type=str)
    p.add_argument("--climb_data_dir", type=str, default=None)
    p.add_argument("--debug", type=int, default=0)
    p.add_argument("--do_single", action="store_true")
    p.add_argument("--do_train", action="store_true")
    p.add_argument("--do_eval", action="store_true")
    p.add_argument("--do_test", action="store_true")
    p.add_argument("--adapter_config", default=None)
    p.add_argument("--adapter_reduction_factor", type=int, default=0)
    p.add_argument("--layers_to_freeze", type=int, default=0)
    p.add_argument("--output_dir", type=str, default="./out")
    p.add_argument("--do_wandb_logging", action="store_true")
    p.add_argument("--wandb_freq", type=int, default=100)
    p.add_argument("--comm_rounds", type=int, default=20)
    p.add_argument("--local_epochs", type=int, default=1)
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--num_epochs", type=int, default=15)
    p.add_argument("--val_batch_size", type=int, default=1)
    p.add_argument("--num_workers", type=int, default=2)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--ordered_cl_tasks", type=str, default="domain")
    p.add_argument("--lr", default=1e-4, type=float)
    p.add_argument("--splits", nargs="*", default=["train", "val"])
    p.add_argument("--checkpoint", type=str, default=None)
    p.add_argument("--model_path", type=str, default=None)
    # synthetic stand-ins for the private datasets / checkpoints (no network in this environment)
    p.add_argument("--synthetic_steps", type=str, default="8",
                   help="batches per client per round: one integer, or a comma list dealt to the clients in order "
                        "(heterogeneous len(loader), e.g. 40,50,60,70,80,45,55,65 for 8 clients)")
    p.add_argument("--synthetic_label_alpha", type=float, default=0.0,
                   help="> 0: client k of the synthetic ViLT data draws its answers from its own Dirichlet(alpha) label prior "
                        "(SURVEY.md 8d config 3 uses 0.5: heterogeneous clients); 0 = uniform labels")
    p.add_argument("--image_size", type=int, default=384)
    p.add_argument("--num_layers", type=int, default=12)
    p.add_argument("--albef_dropout", type=float, default=0.1,
                   help="hidden / attention-probability dropout of the ALBEF BERT towers under train_step (the reference's "
                        "model.train() with src/configs/model_configs.py:44-46); 0 = the deterministic parity configuration")
    p.add_argument("--albef_dims", type=str, default="",
                   help="ALBEF only: override depths for quick runs, e.g. vit_depth=2,enc_layers=3,fusion_layer=1,dec_layers=2")
    p.add_argument("--no_hip_graph", dest="hip_graph", action="store_false")
    p.add_argument("--mixed_precision", default=None, choices=["fp16", "bf16"],
                   help="16-bit MFMA operand format of the engine, named like accelerate's setting.  Default: fp16 for both encoders (the "
                        "reference's accelerate_config.yaml:8; dynamic loss scale with GradScaler semantics on the device) -- the format "
                        "that meets the 1e-3 round-length bar at 80 steps for ViLT and leaves two thirds of it after the reference's own "
                        "full-size 40-step ALBEF rounds (tests/golden/g11b*: 3.2e-4 / 2.6e-4; bf16 there: 7.6e-4 / 8.4e-4)")
    p.add_argument("--exchange", default="rccl_cabi", choices=["rccl_cabi", "torch"],
                   help="the round's FedAvg collective: rccl_cabi = feddat_fedavg_allreduce on a communicator made through "
                        "the C ABI (RCCL bound by dlopen; also taken with ONE rank, where it is the identity); torch = "
                        "torch.distributed all_reduce (always used when the process group is not RCCL, i.e. the gloo test rigs)")
    p.add_argument("--save_every", type=int, default=0,
                   help="write <output_dir>/round state every N rounds (0 = never, like the reference); "
                        "--checkpoint <dir> resumes from it")
    return p


TASK_SETS = {   # main.py:352-359
    "scene": ["clove_scene_a", "clove_scene_b", "clove_scene_c", "clove_scene_d", "clove_scene_e", "clove_scene_f"],
    "function": ["clove_function_a", "clove_function_b", "clove_function_c", "clove_function_d", "clove_function_e"],
    "domain": ["art", "abstract", "vizwiz", "toronto", "gqa"],
}


def main(argv=None):
    args = build_parser().parse_args(argv)
    if "dat" not in args.optimizer_mode:
        raise L.FeddatHipError("only --optimizer_mode dat is on the MI355X hot path (SURVEY.md section 2, row 13)")
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
    log = logging.getLogger("feddat_amd")
    if args.mixed_precision is None:       # one default, the same as bench.py's and the engines' own
        args.mixed_precision = "fp16"
    from . import weights
    # a checkpoint that was asked for must exist (local path; no network): never a silent random initialisation
    pretrained = weights.resolve(args.pretrained_model_name)
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # FEDDAT_FORCE_DEVICE / FEDDAT_DIST_BACKEND: several ranks on one GPU over gloo, so that the multi-rank control flow
    # (HIP pre-scale + all-reduce + write-back) can be tested on a single-GPU box; production = one GPU per rank, RCCL
    if os.environ.get("FEDDAT_FORCE_DEVICE") is not None:
        local = int(os.environ["FEDDAT_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        backend = os.environ.get("FEDDAT_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    tasks = TASK_SETS.get(args.ordered_cl_tasks, args.ordered_cl_tasks.split(","))
    steps_list = [int(x) for x in str(args.synthetic_steps).split(",")]
    steps_of = {t: steps_list[i % len(steps_list)] for i, t in enumerate(tasks)}
    # client -> GPU mapping.  One client per rank when there are at least as many ranks as clients (rank k <-> client k);
    # with more clients than ranks the clients are dealt by longest-processing-time (descending len(loader), each to the
    # least-loaded rank) instead of round-robin, which bounds the round's imbalance; a rank visits its clients in the
    # reference's client order (the local pre-sum below is order-sensitive in the last bit).
    owner = deal_clients([steps_of[t] for t in tasks], world)
    my_tasks = [t for t, o in zip(tasks, owner) if o == rank]
    loads = [sum(steps_of[t] for t, o in zip(tasks, owner) if o == r) for r in range(world)]
    if rank == 0:
        log.info("client -> rank %s; steps per rank %s; predicted round efficiency sum/(N*max) = %.3f (the all-reduce "
                 "barrier waits for the longest rank)", dict(zip(tasks, owner)), loads, round_efficiency(loads))
    if not my_tasks:
        # more GPUs than clients (the reference's default 'domain' set has 5): this rank holds no client; it still joins
        # every round's all-reduce with a zero contribution so that the average over the K real clients is unchanged
        log.warning("rank %d of %d holds no client (%d clients): idle, joins the all-reduce only", rank, world, len(tasks))
    dev = torch.device("cuda", local)
    albef = "albef" in args.encoder_name
    if albef:
        from . import albef_spec
        from .albef_modeling import create_albef_continual_learner_model
        dims = {k: int(v) for k, v in (kv.split("=") for kv in args.albef_dims.split(",") if kv)}
        if pretrained:       # load_albef (albef.py:205-241): ALBEF.pth['model'] + pos-embed interpolation + key surgery
            params = weights.load_albef_pretrained(pretrained, seed=args.seed, image=args.image_size, **dims)
            log.info("loaded ALBEF weights from %s", pretrained)
        else:                # no --pretrained_model_name: random weights of the real architecture (synthetic benchmarks)
            params = albef_spec.random_init(seed=args.seed, image=args.image_size, **dims)
        model = create_albef_continual_learner_model(params, dev, args.batch_size, args.batch_size, lr=args.lr,
                                                     image=args.image_size, dropout=args.albef_dropout,
                                                     seed=args.seed + 7919 * rank,
                                                     operands={"fp16": "f16", "bf16": "bf16"}[args.mixed_precision], **dims)
        Trainer = AlbefTaskTrainer
    else:
        if pretrained:       # load_vilt_encoder (vilt.py:387-420): HF directory / state dict + modality-embedding expansion
            params = weights.load_vilt_pretrained(pretrained, tasks, layers=args.num_layers, seed=args.seed)
            log.info("loaded ViLT weights from %s", pretrained)
        else:                # no --pretrained_model_name: random weights of the real architecture (synthetic benchmarks)
            params = vilt_spec.random_init(args.num_layers, tasks, seed=args.seed)
        model = create_vilt_continual_learner_model(params, tasks, dev, args.batch_size, args.image_size,
                                                    args.num_layers, args.lr,
                                                    operands={"fp16": "f16", "bf16": "bf16"}[args.mixed_precision])
        Trainer = TaskTrainer
    eng = model.engine
    # personal parameters per client (main.py:440-450): head + adapter_0 + adapter_2
    def personal(sd):
        return {n: v.clone() for n, v in sd.items() if ("task" in n or "adapter_0" in n or "adapter_2" in n)}
    personal_params = {t: personal(model.state_dict()) for t in my_tasks}
    def make_batch(seed, ti=0):
        if albef:
            return albef_spec.synthetic_batch(args.batch_size, seed, image=args.image_size, vocab=dims.get("vocab", 30522),
                                              device=dev)
        prior = vilt_spec.client_label_prior(ti, alpha=args.synthetic_label_alpha) if args.synthetic_label_alpha > 0 else None
        return vilt_spec.synthetic_batch(args.batch_size, args.image_size, seed, device=dev, label_prior=prior)
    data = {t: [make_batch(args.seed + 1000 * ti + s, ti) for s in range(steps_of[t])] for ti, t in enumerate(tasks)
            if t in my_tasks}
    server_flat = eng.comm_flat().clone()
    acc = torch.zeros_like(server_flat)
    # the exchange: one C-ABI collective per round (main.py:510 get_average_net -> feddat_fedavg_allreduce); the local
    # pre-sum below has already applied num / total, so the collective runs with num = total = 1 (a plain SUM)
    rccl, rccl_scratch = None, None
    # (one GPU per rank: whatever backend the torch.distributed group uses for its host-side rendezvous -- with "gloo" the C-ABI
    #  communicator is the ONLY RCCL communicator on the device; FEDDAT_FORCE_DEVICE = the several-ranks-on-one-GPU test rig,
    #  where RCCL refuses the duplicate device)
    if args.exchange == "rccl_cabi" and (world == 1 or os.environ.get("FEDDAT_FORCE_DEVICE") is None):
        from .fedavg import make_rccl_comm
        # one rank: the collective is the identity and nothing may depend on librccl being loadable (the communicator is
        # still made when it can be, so the single-GPU run exercises the same entry points)
        rccl = agree_on_exchange((lambda: make_rccl_comm(world, rank)) if world > 1 else
                                 (lambda: L.RcclComm(1, 0, lambda ident: ident)), world, log)
        if rccl is not None:
            rccl_scratch = torch.empty_like(acc)
            log.info("FedAvg exchange: feddat_fedavg_allreduce, %s", rccl.info())
    comm_names = model.comm_state_dict_names
    first_round = 0
    # requires_grad flags of the SERVER model.  Clients train on deepcopy(server) (main.py:472), so whatever train_step
    # toggles stays on the copy; only eval(server) (main.py:546) changes the server's flags -- it ends in the
    # adapter_1 state, after which fresh client optimizers no longer hold adapter_0 (reference behaviour, golden G3q).
    server_flags = dict(model.adapter_requires_grad)
    if args.checkpoint:                                   # resume: averaged adapter + this rank's personal tensors
        from . import checkpoint
        comm, pers, last, flags = checkpoint.load_federation(args.checkpoint, my_tasks)
        server_flags.update(flags)
        model.load_state_dict(comm)
        server_flat.copy_(eng.comm_flat())
        for t in my_tasks:
            personal_params[t].update({k: v.to(dev) for k, v in pers[t].items()})
        first_round = last + 1
    for comm_round in range(first_round, args.comm_rounds):
        nonfinite = []
        for k, task_key in enumerate(my_tasks):
            eng.comm_flat().copy_(server_flat)                      # main.py:472 deepcopy(server)
            eng.repack_adapter(1)
            model.load_state_dict(personal_params[task_key])        # main.py:473-478
            model.adapter_requires_grad = dict(server_flags)
            trainer = Trainer(args, task_key, data[task_key], data[task_key][:2], log)
            trainer.dropout_epoch = comm_round * len(tasks) + tasks.index(task_key)     # fresh dropout masks per round and client
            trainer.defer_finite_check = world > 1
            trainer.train(model, comm_round)
            nonfinite += [f"{task_key}:{g}" for g in getattr(trainer, "nonfinite", [])]
            personal_params[task_key] = personal(model.state_dict())     # main.py:493-497
            # local pre-sum in client order, then (if distributed) one all-reduce: main.py:50-65
            L.fedavg_accumulate(acc, eng.comm_flat(), 1.0, float(len(tasks)), k == 0)
        if not my_tasks:
            acc.zero_()
        if world > 1:       # every rank learns whether ANY rank's local update went non-finite, and all raise together
            bad = torch.tensor([1.0 if nonfinite else 0.0], device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if bad.item() > 0:
                raise L.FeddatHipError(f"round {comm_round}: non-finite trainable state after the local update on "
                                       + (f"this rank ({', '.join(nonfinite)})" if nonfinite else "another rank")
                                       + "; no rank enters the FedAvg all-reduce")
        if rccl is not None:
            rccl.fedavg_allreduce(acc, rccl_scratch, 1.0, 1.0)      # RCCL over xGMI on the 3.58 MB device buffer
        elif world > 1:
            all_reduce_sum(acc)
        server_flat.copy_(acc)
        if args.save_every and ((comm_round + 1) % args.save_every == 0 or comm_round == args.comm_rounds - 1):
            from . import checkpoint
            eng.comm_flat().copy_(server_flat)
            sd = model.state_dict()
            checkpoint.save_federation(args.output_dir, {}, personal_params, comm_round, write_server=False)
            if world > 1:
                dist.barrier()          # every rank's personal files are on disk before round.json appears
            if rank == 0:
                flags_after = dict(server_flags)
                if comm_round % 5 == 0 or comm_round == args.comm_rounds - 1:
                    flags_after.update({0: False, 1: True})         # the eval below leaves the server in this state
                checkpoint.save_federation(args.output_dir, {n: sd[n] for n in comm_names}, {}, comm_round,
                                           server_flags=flags_after)
        if (comm_round % 5 == 0 or comm_round == args.comm_rounds - 1) and not albef:   # main.py:520 (ALBEF: the synthetic
            for task_key in my_tasks:                                                     # stand-in has no answer list)
                eng.comm_flat().copy_(server_flat)
                model.load_state_dict(personal_params[task_key])
                model.after_load()
                model.adapter_requires_grad = dict(server_flags)
                scores = TaskTrainer(args, task_key, data[task_key], data[task_key][:2], log).eval(model)
                server_flags = dict(model.adapter_requires_grad)
                log.info("round %d %s test score server = %s", comm_round, task_key, scores)
        elif comm_round % 5 == 0 or comm_round == args.comm_rounds - 1:
            # ALBEF: the synthetic stand-in has no answer list to rank, so the periodic evaluation is skipped -- but its
            # side effect on the SERVER model is not: TaskTrainer.eval ends in set_active_adapter('adapter_1')
            # (task_trainer.py:236-244 -> adapter.py:79-85), so every client optimizer built from the next round on
            # (create_optimizer filters on requires_grad, task_trainer.py:477-504) no longer holds adapter_0.
            server_flags.update({0: False, 1: True})
    eng.comm_flat().copy_(server_flat)
    eng.repack_adapter(1)
    if rccl is not None:
        torch.cuda.synchronize()
        rccl.close()
    model.exchange_used = "feddat_fedavg_allreduce" if rccl is not None else ("torch.distributed" if world > 1 else "none")
    if world > 1:
        dist.destroy_process_group()
    return model


if __name__ == "__main__":
    main()
