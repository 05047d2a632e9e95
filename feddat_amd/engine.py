"""ViLT-B/32 dual-adapter (DAT) local-update engine on MI355X.

Host-side sequencing of the HIP kernels in libfeddat_hip.so for the reference's hot path
(src/train/visionlanguage_tasks/task_trainer.py:266-330 around src/modeling/vilt.py:244-264 and
src/modeling/models/adapter.py:124-163).  No arithmetic happens in Python/PyTorch here: torch only owns the
device buffers and the stream; every launch goes through the C ABI, on static buffers, so a whole train_step
can be captured into one hipGraph (torch.cuda.CUDAGraph stream capture) and replayed.

Restructuring relative to the reference (same results, fewer FLOPs -- DESIGN.md "step algebra"):
  * P0 (no-grad gated forward) and P2 (gated forward with grad) see identical backbone inputs and identical
    adapter_0/adapter_2 weights (P1 only updates adapter_1 and the task head), so the gated backbone forward is
    run ONCE and its pooled output feeds both the P0 logits (old head) and the P2 logits (updated head).
  * Embeddings and the body of layer 0 (everything below the first adapter) are shared by the gated and the
    adapter_1 pass; from the layer-0 adapter on, the two passes ride in ONE batch of 2*B*S rows through the
    frozen GEMMs / attention (rows [0,R) gated, rows [R,2R) adapter_1).
  * Nothing trainable lies below the layer-0 adapter, so the backward stops there.
"""
from __future__ import annotations

import functools
import math
from typing import Dict, List, Optional, Sequence

import torch

from . import lib as L

ENC = "vilt_encoder.vilt."
ADAPTER_TENSORS = ("down.weight", "down.bias", "up.weight", "up.bias")
HEAD_TENSORS = ("clf_fc0.weight", "clf_fc0.bias", "clf_norm0.weight", "clf_norm0.bias", "clf_fc1.weight",
                "clf_fc1.bias")


def _no_decay(name: str) -> bool:  # task_trainer.py:478
    return ("bias" in name) or ("LayerNorm.weight" in name)


def _bound(fn):
    """Run a public engine method with the calling thread bound to the library of the engine's operand format
    (lib.operands): engines of both formats can live in one process."""
    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        with L.operands(self.operands):
            return fn(self, *a, **kw)
    return wrapped


class FlatGroup:
    """A set of named fp32 tensors living back-to-back in one flat device buffer (+ grad, Adam m/v, segment table)."""

    def __init__(self, names_shapes: Sequence, device, with_opt: bool):
        self.names = [n for n, _ in names_shapes]
        self.shapes = {n: tuple(s) for n, s in names_shapes}
        self.offsets = {}
        off = 0
        for n, s in names_shapes:
            self.offsets[n] = off
            off += int(math.prod(s))
        self.numel = off
        self.p = torch.zeros(off, device=device)
        if with_opt:
            self.g = torch.zeros(off, device=device)
            self.m = torch.zeros(off, device=device)
            self.v = torch.zeros(off, device=device)
            offs = [self.offsets[n] for n in self.names] + [off]
            self.seg_off = torch.tensor(offs, dtype=torch.int64, device=device)
            self.seg_wd = torch.tensor([0.0 if _no_decay(n) else 1.0 for n in self.names], device=device)
            self.state = torch.zeros(2, dtype=torch.int32, device=device)  # {sched_t, adam_t}

    def view(self, name: str, buf: Optional[torch.Tensor] = None) -> torch.Tensor:
        buf = self.p if buf is None else buf
        o = self.offsets[name]
        return buf[o:o + int(math.prod(self.shapes[name]))].view(self.shapes[name])


class ViltDatEngine:
    def __init__(self, params: Dict[str, torch.Tensor], tasks: Sequence[str], device, batch: int, res: int,
                 text_len: int = 40, layers: int = 12, num_labels: int = 100, lr: float = 1e-4,
                 weight_decay: float = 1e-2, adam_eps: float = 1e-8, wgrad_splits: int = 16, fp8: bool = False,
                 fp8_ffn_chain: bool = True, gelu_codes: bool = True, operands: Optional[str] = None,
                 loss_scale: Optional[float] = None, fp8_mx_dqkv: bool = True, dynamic_loss_scale: Optional[bool] = None,
                 scale_growth_interval: int = 2000):
        """operands: "f16" (the default) or "bf16" (the default with fp8=True, configs[4]).  "f16": every 16-bit MFMA operand of the step -- frozen weights and their transposes, LayerNorm outputs,
        qkv, probabilities, ctx, gelu(u), the adapters' operand copies, and every gradient operand of the dX products and of
        the attention backward -- is IEEE half instead of bf16 (libfeddat_hip_f16.so: v_mfma_f32_16x16x32_f16, the same MFMA
        rate and the same bytes; 10 instead of 7 mantissa bits, i.e. the reference's own GPU arithmetic, fp16 autocast:
        accelerate_config.yaml:8).  The backward then carries a power-of-two `loss_scale` (default 2^14; 1 for bf16): the
        gradient entering the backbone (d pooler-input) is multiplied by it where it is produced, every kernel of the backward
        is linear in the gradient, and the factor leaves exactly where the adapter weight gradients are formed
        (feddat_wgrad_seg.scale); the task head's own gradients never see it.  The scaled gradients of this path span
        1e-4 .. 1e1 at the default, eleven binades inside either end of fp16's range.
        dynamic_loss_scale (default: on with "f16" operands): the reference's GradScaler (accelerate, mixed_precision fp16:
        accelerate_config.yaml:8; task_trainer.py:302-308,323-328) ON THE DEVICE, inside the captured step -- `loss_scale` is
        only the initial value (GradScaler's own is 65536); a non-finite adapter gradient (feddat_adapter_wgrad_reduce_checked)
        or loss (feddat_dat_loss_fwd_bwd_checked) skips that sub-step's optimizer AND scheduler step and halves the scale,
        `scale_growth_interval` clean sub-steps double it (feddat_dat_step_finish; DESIGN.md section 5b says where this
        differs from GradScaler: an overflow in sub-step A voids the whole batch).  With no overflow the step is bit-identical
        to the static scale.  A fresh scaler per local update, like the reference's fresh Accelerator per round (main.py:435).
        fp8=True (BASELINE.json configs[4]): four frozen products per layer run on the block-scaled fp8 MFMA with e4m3
        operands -- forward QKV and FFN1 (activations quantised per token row by the LayerNorm kernel that produces them) and
        the dX products FFN2^T and attention-output^T (the gradient rows quantised per row: by feddat_quant_rows_fp8 behind the
        adapter backward, by the LayerNorm backward itself); weights quantised once here per output channel of each product.
        Everything else -- adapters, attention, the products whose A operand comes out of a GEMM or attention epilogue --
        stays bf16 / fp32.
        gelu_codes=False: the backward of FFN2 reads the pre-GELU activation u in bf16 instead of the 8-bit gelu' codes
        (FEDDAT_EPI_GELU / _MUL_DGELU instead of _GELU_G8 / _MUL_G8; middle layers issued op by op).  25 % more bytes through
        the two heaviest epilogues (+0.14 ms/step at configs[1]); at B = 32 it takes the worst adapter element after an 80-step
        round from 1.31e-3 to 1.00e-3 and the worst update-norm error from 2.9 % to 1.6 % -- for callers
        that trade 2 % of throughput for that."""
        operands = operands or ("bf16" if fp8 else "f16")
        if operands not in L.OPERAND_DTYPE:
            raise L.FeddatHipError(f"operands must be 'bf16' or 'f16', got {operands!r}")
        if fp8 and operands != "bf16":
            raise L.FeddatHipError("fp8=True (configs[4]) pairs the e4m3 products with bf16 operands")
        self.operands = operands
        self.op_dtype = L.OPERAND_DTYPE[operands]
        self.loss_scale = float(loss_scale if loss_scale is not None else (16384.0 if operands == "f16" else 1.0))
        if self.loss_scale <= 0 or math.frexp(self.loss_scale)[0] != 0.5:
            raise L.FeddatHipError("loss_scale must be a power of two (it is removed exactly)")
        # fp8 (round 5): the attention backward writes dq | dk | dv as MX-scaled e4m3 (one E8M0 scale per (row, 32 columns):
        # feddat_attn_bwd_fp8mx) and QKV^T runs on the block-scaled fp8 MFMA with those scales (feddat_gemm_fp8mx_nt): the seventh
        # of the eight frozen products per layer, and half the bytes of the backward's largest write
        self.fp8_mx_dqkv = bool(fp8) and bool(fp8_mx_dqkv)
        self.dynamic_scale = bool(operands == "f16" if dynamic_loss_scale is None else dynamic_loss_scale)
        self.scale_growth, self.scale_backoff, self.scale_growth_interval = 2.0, 0.5, int(scale_growth_interval)
        self._init(params, tasks, device, batch, res, text_len, layers, num_labels, lr, weight_decay, adam_eps, wgrad_splits,
                   fp8, fp8_ffn_chain, gelu_codes)

    @_bound
    def _init(self, params, tasks, device, batch, res, text_len, layers, num_labels, lr, weight_decay, adam_eps, wgrad_splits,
              fp8, fp8_ffn_chain, gelu_codes):
        L.load()
        self.dev = torch.device(device)
        self.tasks = list(tasks)
        self.B, self.Lt, self.nl = batch, text_len, layers
        self.res = (res, res) if isinstance(res, int) else tuple(res)      # (height, width), multiples of 32
        self.H, self.I, self.heads, self.r, self.C = 768, 3072, 12, 48, num_labels
        self.P = 32
        self.gh, self.gw = self.res[0] // self.P, self.res[1] // self.P
        self.np = self.gh * self.gw
        self.S = text_len + 1 + self.np
        self.R = batch * self.S
        self.lr, self.wd, self.eps = lr, weight_decay, adam_eps
        self.fp8 = bool(fp8)
        self.fp8_ffn_chain = bool(fp8_ffn_chain)      # fp8: also FFN2 forward and FFN1^T (their A operands leave as e4m3)
        # fp8 attribution switches (tools/fp8_noise_attribution.py; production = both True): the backward's dX products on e4m3
        # gradient rows / the forward's products on e4m3 activations
        self.fp8_backward = True
        self.fp8_forward = True
        self.ksplit = wgrad_splits
        self.ln_eps = 1e-12
        dev = self.dev
        H, I = self.H, self.I

        def P(name):
            return params[name].to(dev, torch.float32).contiguous()

        def bf16_of(w):
            out = torch.empty(w.shape, dtype=self.op_dtype, device=dev)
            L.cvt_f32_bf16(w, out)
            return out

        def bf16_T(w):  # [R,C] fp32 -> [C,R] bf16
            out = torch.empty(w.shape[1], w.shape[0], dtype=self.op_dtype, device=dev)
            L.transpose_f32_bf16(w, out, w.shape[0], w.shape[1])
            return out

        def fp8_of(w):  # [N,K] fp32 -> e4m3 [N,K] + per-output-channel scale [N]
            w8 = torch.empty(w.shape, dtype=torch.uint8, device=dev)
            sc = torch.empty(w.shape[0], device=dev)
            L.quant_rows_fp8(w.contiguous(), w8, sc)
            return w8, sc

        # ---------------- frozen backbone (bf16 weights + their transposes for the dX products) --------------
        e = ENC + "embeddings."
        self.emb = {k: P(e + k) for k in (
            "text_embeddings.word_embeddings.weight", "text_embeddings.position_embeddings.weight",
            "text_embeddings.token_type_embeddings.weight", "text_embeddings.LayerNorm.weight",
            "text_embeddings.LayerNorm.bias", "patch_embeddings.projection.bias")}
        tok = P(e + "token_type_embeddings.weight")
        self.mod0, self.mod1 = tok[0].contiguous(), tok[1].contiguous()
        self.cls = P(e + "cls_token").reshape(H).contiguous()
        pos = P(e + "position_embeddings")[0]
        self.pos0 = pos[0].contiguous()
        # per-sample position grids: the 12 x 12 table resized to each sample's valid patch rectangle (pixel_mask)
        self.pos_grid = pos[1:].contiguous()
        self.g0 = int(round(math.sqrt(pos.shape[0] - 1)))
        self.pos_img = torch.empty(batch, self.np, H, device=dev)
        self.w_patch = bf16_of(P(e + "patch_embeddings.projection.weight").reshape(H, 3 * self.P * self.P))
        self.layers: List[dict] = []
        for i in range(layers):
            Lp = ENC + f"encoder.layer.{i}."
            wq, wk, wv = (P(Lp + f"attention.attention.{n}.weight") for n in ("query", "key", "value"))
            wqkv = torch.cat([wq, wk, wv], 0).contiguous()
            bqkv = torch.cat([P(Lp + f"attention.attention.{n}.bias") for n in ("query", "key", "value")]).contiguous()
            wo, w1, w2 = P(Lp + "attention.output.dense.weight"), P(Lp + "intermediate.dense.weight"), \
                P(Lp + "output.layer.dense.weight")
            extra = {}
            if self.fp8:
                extra["wqkv8"], extra["sqkv"] = fp8_of(wqkv)
                extra["w18"], extra["s1"] = fp8_of(w1)
                # dX products whose A operand (a gradient) is produced by a row kernel: FFN2^T and attention-output^T
                extra["w2T8"], extra["s2T"] = fp8_of(w2.t().contiguous())
                extra["woT8"], extra["soT"] = fp8_of(wo.t().contiguous())
                # the FFN chain: FFN1's epilogue leaves gelu(u) as e4m3 (fixed scale) for an fp8 FFN2; FFN2^T's leaves dU as
                # e4m3 rows that keep the incoming gradient's row scale x FEDDAT_F8_GRAD_HEADROOM (folded into W1^T's
                # channel scales here) for an fp8 FFN1^T
                extra["w28"], extra["s2"] = fp8_of(w2)
                extra["w1T8"], s1T = fp8_of(w1.t().contiguous())
                extra["s1T4"] = s1T * L.F8_GRAD_HEADROOM
                if self.fp8_mx_dqkv:      # QKV^T: [768, 2304], per output channel
                    extra["wqkvT8"], extra["sqkvT"] = fp8_of(wqkv.t().contiguous())
            self.layers.append(dict(
                extra, wqkv=bf16_of(wqkv), wqkvT=bf16_T(wqkv), bqkv=bqkv,
                wo=bf16_of(wo), woT=bf16_T(wo), bo=P(Lp + "attention.output.dense.bias"),
                w1=bf16_of(w1), w1T=bf16_T(w1), b1=P(Lp + "intermediate.dense.bias"),
                w2=bf16_of(w2), w2T=bf16_T(w2), b2=P(Lp + "output.layer.dense.bias"),
                ln1g=P(Lp + "layernorm_before.weight"), ln1b=P(Lp + "layernorm_before.bias"),
                ln2g=P(Lp + "layernorm_after.weight"), ln2b=P(Lp + "layernorm_after.bias")))
        self.lnf_g, self.lnf_b = P(ENC + "layernorm.weight"), P(ENC + "layernorm.bias")
        self.pool_w, self.pool_b = P(ENC + "pooler.dense.weight"), P(ENC + "pooler.dense.bias")

        # ---------------- trainable state: three adapters (flat per adapter) and one head per task ----------
        def adapter_names(a):
            return [(ENC + f"encoder.layer.{i}.output.adapter.adapter_{a}_{t}",
                     {"down.weight": (self.r, H), "down.bias": (self.r,), "up.weight": (H, self.r),
                      "up.bias": (H,)}[t]) for i in range(layers) for t in ADAPTER_TENSORS]
        self.ad = [FlatGroup(adapter_names(a), dev, with_opt=(a != 2)) for a in range(3)]
        self.ad_layer_numel = self.r * H + self.r + H * self.r + H
        head_shapes = {"clf_fc0.weight": (2 * H, H), "clf_fc0.bias": (2 * H,), "clf_norm0.weight": (2 * H,),
                       "clf_norm0.bias": (2 * H,), "clf_fc1.weight": (num_labels, 2 * H), "clf_fc1.bias": (num_labels,)}
        self.head = {t: FlatGroup([(f"task_layer.{t}.{n}", head_shapes[n]) for n in HEAD_TENSORS], dev, True)
                     for t in self.tasks}
        for grp in self.ad + list(self.head.values()):
            for n in grp.names:
                grp.view(n).copy_(params[n].to(dev, torch.float32))
        # bf16 operand copies of the adapters: [a][layer] -> dict(wd, wdT, wu, wuT, bd, bu)
        self._pack16 = {}
        self.ad16 = [[self._alloc_pack(a, i) for i in range(layers)] for a in range(3)]
        for a in range(3):
            self.repack_adapter(a)

        # ---------------- workspace (static: a whole step is graph-capturable) ----------------
        R, R2, B = self.R, 2 * self.R, batch

        def f32(*s):
            return torch.empty(*s, device=dev)

        def b16(*s):
            return torch.empty(*s, dtype=self.op_dtype, device=dev)
        self._px_shape = (B, 3, self.res[0], self.res[1])
        self.inp = dict(input_ids=torch.zeros(B, text_len, dtype=torch.int64, device=dev),
                        token_type_ids=torch.zeros(B, text_len, dtype=torch.int64, device=dev),
                        target=f32(B, num_labels),
                        attention_mask=torch.ones(B, text_len, dtype=torch.int64, device=dev),
                        # pixel_mask sampled at the patch origins (all that HF's visual_embed looks at): [B, gh, gw]
                        patch_mask=torch.ones(B, self.gh, self.gw, dtype=torch.int64, device=dev))
        # attention key masks of the [text | CLS | patches] sequence, derived on the device from the two HF masks
        # inside the step (no host sync, valid for every batch under one captured graph); rows [B, 2B) repeat [0, B)
        self.key_mask2 = torch.ones(2 * B, self.S, dtype=torch.uint8, device=dev)
        self.patches = b16(B * self.np, 3 * self.P * self.P)
        self.proj = f32(B * self.np, H)
        self.h0 = f32(R, H)
        self.x16 = b16(R2, H)          # LN output (GEMM operand), transient
        if self.fp8:                   # LN output as e4m3 + per-row scale (operand of the fp8 products)
            self.x8 = torch.empty(R2, H, dtype=torch.uint8, device=dev)
            self.xs = f32(R2)
            self.g8 = torch.empty(R2, H, dtype=torch.uint8, device=dev)     # gradient rows as e4m3 + per-row scale
            self.gsc = f32(R2)
            self.f8 = torch.empty(R2, I, dtype=torch.uint8, device=dev)     # gelu(u) as e4m3, fixed scale
            self.f8s = torch.full((R2,), L.F8_ACT_SCALE, dtype=torch.float32, device=dev)
            self.dU8 = torch.empty(R2, I, dtype=torch.uint8, device=dev)    # dU rows as e4m3, row scale = headroom x gsc
            if self.fp8_mx_dqkv:
                if self.S > 192:
                    raise L.FeddatHipError("fp8_mx_dqkv needs sequences of at most 192 tokens (feddat_attn_bwd_fp8mx)")
                self.dqkv8 = torch.empty(R2, 3 * H, dtype=torch.uint8, device=dev)        # dq | dk | dv as e4m3 ...
                self.dqkv_sc = torch.empty(R2, 3 * H // 32, dtype=torch.uint8, device=dev)   # ... + E8M0 per (row, 32 columns)
        self.f16 = b16(R2, I)          # gelu(u), transient
        # layer 0 (shared body, R rows): only h3 is kept
        self.l0 = dict(qkv=b16(R, 3 * H), ctx=b16(R, H), lse=f32(B, self.heads, self.S), h2=f32(R, H), h3=f32(R, H))
        # what FFN2^T needs of the pre-GELU u: 8-bit gelu'(u) codes where the persistent GEMM applies (FEDDAT_EPI_GELU_G8 /
        # _MUL_G8, M >= 1024: 25 % fewer bytes through the two HBM-bound epilogues), else u itself in bf16
        self.g8u = R2 >= 1024 and bool(gelu_codes)

        def u_buf():
            return torch.empty(R2, I, dtype=torch.uint8, device=dev) if self.g8u else b16(R2, I)
        self.act = [None] + [dict(h_in=f32(R2, H), st1=f32(R2, 2), qkv=b16(R2, 3 * H), ctx=b16(R2, H),
                                  lse=f32(2 * B, self.heads, self.S), h2=f32(R2, H), st2=f32(R2, 2),
                                  u=u_buf(), h3=f32(R2, H)) for _ in range(1, layers)]
        self.h_out = f32(R2, H)        # output of the last adapter (dense path: single-layer models only)
        # top layer: only token 0 of each sample feeds the pooler, so everything after its attention runs on 2B rows
        nb2 = 2 * B
        self.top = dict(h2=f32(nb2, H), st2=f32(nb2, 2), u=b16(nb2, I), h3=f32(nb2, H), x16=b16(nb2, H),
                        f16=b16(nb2, I), h_out=f32(nb2, H), dh3=f32(nb2, H), dh316=b16(nb2, H), dU=b16(nb2, I),
                        dx2=b16(nb2, H), dh2=f32(nb2, H), dh216=b16(nb2, H), dctx=f32(nb2, H))
        self.st0 = f32(R, 2)
        # head / pooler
        self.cls_ln = f32(2 * B, H)
        self.cls_st = f32(2 * B, 2)
        self.pooled = f32(2 * B, H)
        # task-head activations.  P0 (gated rows, old head) and P1 (adapter_1 rows, same old head) run as ONE 2B-row
        # pass ("all" = rows [0,B), "p1" = rows [B,2B) of the "both" buffers); P2 uses the updated head.
        both = dict(a0=f32(2 * B, 2 * H), n0=f32(2 * B, 2 * H), st=f32(2 * B, 2), g0=f32(2 * B, 2 * H),
                    logits=f32(2 * B, num_labels))
        self.hd = {"both": both, "all": {k: v[:B] for k, v in both.items()}, "p1": {k: v[B:] for k, v in both.items()},
                   "p2": dict(a0=f32(B, 2 * H), n0=f32(B, 2 * H), st=f32(B, 2), g0=f32(B, 2 * H),
                              logits=f32(B, num_labels))}
        self.dlogits = f32(B, num_labels)
        self.loss_buf = {k: f32(4 + 2 * B) for k in ("p1", "p2")}
        self.dg0, self.dn0, self.da0 = f32(B, 2 * H), f32(B, 2 * H), f32(B, 2 * H)
        self.dpooled = f32(2 * B, H)
        self.dpre = f32(2 * B, H)
        self.dcls_ln = f32(2 * B, H)
        self.dcls = f32(2 * B, H)
        # backward streams
        self.dh = [f32(R2, H), f32(R2, H)]       # ping-pong residual-gradient stream
        self.dh16 = b16(R2, H)
        self.dU = b16(R2, I)
        self.dx16 = b16(R2, H)
        self.dctx = b16(R2, H)
        self.dqkv = b16(R2, 3 * H)
        self.z = f32(R2, self.r)
        self.dz = f32(R2, self.r)
        # relu(W_down h3 + b_down) of every adapter slot, saved by the adapter forward of each layer for its backward
        # (fp32 [rows, 2, 48]: 384 B per token instead of re-reading the 3 KB row and repeating the down-projection)
        self.zsave = [f32(R2, 2, self.r) for _ in range(layers - 1)] + [f32(nb2 if layers > 1 else R2, 2, self.r)]
        # token-split partial sums of the adapter weight gradients: one slot per layer, folded into the flat gradient
        # buffers by ONE reduction at the end of the backward (feddat_adapter_wgrad_reduce) instead of one per layer
        self.wpart_stride = L.adapter_wgrad_workspace_elems(2)
        self.wpart_all = f32(layers * self.wpart_stride)
        self.wpart = self.wpart_all[:self.wpart_stride]
        self._segs_cache: Dict = {}
        self.graph = None
        self.ctx = L.Context(self.dev.index if self.dev.index is not None else torch.cuda.current_device())
        self._layer_structs: Dict = {}
        # True: one composite C-ABI call per middle layer (feddat_vilt_layer_fwd / _bwd); False: the same kernel sequence
        # issued op by op from here (what tools/step_breakdown.py brackets with events)
        self.use_layer_calls = bool(gelu_codes) or R2 < 1024      # (the composite layer calls take the code epilogues at M >= 1024)
        # True: the serial tail (token-0 LayerNorm + pooler, task head forward / backward, loss, optimizer bookkeeping) on the
        # fused kernels of csrc/head_tail.hip (20 launches); False: the round-3 sequence of 46 single-purpose launches (same
        # arithmetic up to fp32 summation order; tools/step_breakdown.py --unfused-tail, and the tests compare the two)
        self.fused_tail = True
        # True: the last layer's attention computes the ONE query per (sample, head) the pooler consumes (token 0) and its
        # rank-1 backward (feddat_attn_cls_fwd / _bwd); False: the dense kernels on all S queries (184 of 185 never read)
        self.cls_attention = True
        # True: with the token-0-only attention the last layer's QKV product computes K | V for every row and Q for the 2B token-0
        # rows only, and QKV^T contracts dK | dV densely + the token-0 rows' dQ as a skinny product (a third of both products:
        # -0.03 ms/step, ratio 0.996).  OFF by default: equally accurate, but not bit-identical on the 2B token-0 rows, and at 80
        # steps the AdamW trajectory is chaotic enough that this moves the draw of the round-length parity tests (DESIGN.md
        # section 5, "draws"): the default keeps round 5's arithmetic, whose draws on both reference rounds are pinned.
        self.top_q_cls = False
        self.sched = dict(warmup=1, total=2)
        self.opt_adapters = (0, 1)
        self.task = self.tasks[0]
        # dynamic loss scale (GradScaler on the device): {scale, 1 / scale}; {growth tracker, skipped sub-steps, batches with a
        # skip, -}; overflow flags {sub-step B = adapter_0 pass, sub-step A = adapter_1 pass}; the head's p | m | v before its
        # sub-step-A update (restored when A turns out to have overflowed in the backbone's backward)
        self.scaler_f = torch.tensor([self.loss_scale, 1.0 / self.loss_scale], dtype=torch.float32, device=dev)
        self.scaler_i = torch.zeros(4, dtype=torch.int32, device=dev)
        self.ovf_flags = torch.zeros(2, dtype=torch.int32, device=dev)
        self.head_bak = {t: torch.empty(3 * self.head[t].p.numel(), device=dev) for t in self.tasks} if self.dynamic_scale else {}

    # ------------------------------------------------------------------------------------------ adapters
    def _alloc_pack(self, a, i):
        H, r = self.H, self.r
        base = ENC + f"encoder.layer.{i}.output.adapter.adapter_{a}_"
        if a not in self._pack16:       # bf16 operand copies of all layers of adapter a: [layer][wd | wdT | wu | wuT]
            self._pack16[a] = torch.empty(self.nl, 4, r * H, dtype=self.op_dtype, device=self.dev)
        c = self._pack16[a][i]
        return dict(wd=c[0].view(r, H), wdT=c[1].view(H, r), wu=c[2].view(H, r), wuT=c[3].view(r, H),
                    bd=self.ad[a].view(base + "down.bias"), bu=self.ad[a].view(base + "up.bias"),
                    wd32=self.ad[a].view(base + "down.weight"), wu32=self.ad[a].view(base + "up.weight"))

    @_bound
    def repack_adapter(self, a: int):
        """fp32 masters -> bf16 MFMA operand copies (after every optimizer step / load / FedAvg): one launch for all
        layers when the flat fp32 layout is regular (it is: four tensors per layer, fixed order)."""
        packs = self.ad16[a]
        offs = [p["wd32"].data_ptr() for p in packs]
        stride = (offs[1] - offs[0]) // 4 if len(offs) > 1 else 0
        regular = all(offs[i] - offs[0] == 4 * stride * i for i in range(len(offs))) and all(
            p["wu32"].data_ptr() - p["wd32"].data_ptr() == packs[0]["wu32"].data_ptr() - packs[0]["wd32"].data_ptr()
            for p in packs)
        if regular:
            p0 = packs[0]
            L.adapter_pack_strided(p0["wd32"], p0["wu32"], stride, p0["wd"], p0["wdT"], p0["wu"], p0["wuT"],
                                   4 * self.r * self.H, len(packs))
        else:
            for p in packs:
                L.adapter_pack(p["wd32"], p["wu32"], p["wd"], p["wdT"], p["wu"], p["wuT"])

    @_bound
    def copy_global_to_teacher(self):
        """adapter_1 -> adapter_2 at the start of every local update (task_trainer.py:36-41)."""
        self.ad[2].p.copy_(self.ad[1].p)
        self.repack_adapter(2)

    def _segs(self, layer: int, first: bool, bwd: bool):
        """Two-segment descriptor: rows [0,R) gated (adapter_0 + adapter_2, 0.5 each), rows [R,2R) adapter_1."""
        key = (layer, first, bwd)
        if key not in self._segs_cache:
            a0, a1, a2 = (self.ad16[a][layer] for a in range(3))
            R = self.R
            self._segs_cache[key] = L.make_segs([
                dict(row_begin=0, row_end=R, train_slot=0 if bwd else -1, x_row_delta=0,
                     adapters=[dict(a0, scale=0.5), dict(a2, scale=0.5)]),
                dict(row_begin=R, row_end=2 * R, train_slot=0 if bwd else -1, x_row_delta=-R if first else 0,
                     adapters=[dict(a1, scale=1.0)]),
            ])
        return self._segs_cache[key]

    def _single_segs(self, layer: int, mode: str, rows: int):
        if mode == "gating":
            ads = [dict(self.ad16[0][layer], scale=0.5), dict(self.ad16[2][layer], scale=0.5)]
        else:
            ads = [dict(self.ad16[int(mode.split("_")[1])][layer], scale=1.0)]
        return L.make_segs([dict(row_begin=0, row_end=rows, adapters=ads)])

    # ------------------------------------------------------------------------------------------ inputs
    @_bound
    def set_batch(self, batch: Dict[str, torch.Tensor]):
        """Copy one batch (reference schema: HF ViLT encodings + target_scores) into the static input buffers."""
        px = batch["pixel_values"]
        if tuple(px.shape) != self._px_shape:
            raise L.FeddatHipError(f"engine built for pixel_values {self._px_shape}, got {tuple(px.shape)}")
        # the pixels are consumed right here, from the caller's tensor: patch extraction (im2col + bf16) is the only reader of
        # pixel_values, so it runs ahead of the captured step instead of a 57 MB device-to-device copy into a static buffer
        # followed by the same read inside the graph (stream-ordered with the replay that follows)
        if not px.is_cuda:
            px = px.to(self.dev, non_blocking=True)
        L.im2col_patches(px.to(torch.float32).contiguous(), self.patches, self.B, 3, self.res[0], self.res[1], self.P)
        ids, tts, am, tg, pm = (batch.get(k) for k in ("input_ids", "token_type_ids", "attention_mask", "target_scores",
                                                        "pixel_mask"))

        def dev_ok(t, dt, shape):
            return t is None or (t.is_cuda and t.dtype == dt and t.is_contiguous() and tuple(t.shape) == shape)
        B, Lt = self.B, self.Lt
        if (ids is not None and tts is not None and dev_ok(ids, torch.int64, (B, Lt)) and dev_ok(tts, torch.int64, (B, Lt))
                and dev_ok(am, torch.int64, (B, Lt)) and dev_ok(tg, torch.float32, (B, self.C))
                and dev_ok(pm, torch.int64, (B, self.res[0], self.res[1]))):
            # the usual case (device-resident batch in the reference's dtypes): one launch for all five
            L.vilt_stage_inputs(ids, tts, am, tg, pm, self.inp, B, Lt, self.C, self.res[0], self.res[1], self.P)
            return
        self.inp["input_ids"].copy_(batch["input_ids"], non_blocking=True)
        self.inp["token_type_ids"].copy_(batch["token_type_ids"], non_blocking=True)
        if "target_scores" in batch:
            self.inp["target"].copy_(batch["target_scores"], non_blocking=True)
        if batch.get("attention_mask") is not None:     # absent = all valid
            self.inp["attention_mask"].copy_(batch["attention_mask"], non_blocking=True)
        else:
            self.inp["attention_mask"].fill_(1)
        if batch.get("pixel_mask") is not None:
            self.inp["patch_mask"].copy_(batch["pixel_mask"][:, ::self.P, ::self.P], non_blocking=True)
        else:
            self.inp["patch_mask"].fill_(1)

    # ------------------------------------------------------------------------------------------ forward
    def _embed(self):
        B, H, S, Lt = self.B, self.H, self.S, self.Lt
        e = self.emb
        L.text_embed(self.inp["input_ids"], self.inp["token_type_ids"], e["text_embeddings.word_embeddings.weight"],
                     e["text_embeddings.position_embeddings.weight"], e["text_embeddings.token_type_embeddings.weight"],
                     e["text_embeddings.LayerNorm.weight"], e["text_embeddings.LayerNorm.bias"], self.ln_eps,
                     self.mod0, self.h0, B, Lt, S, H)
        # (self.patches was filled by set_batch: im2col of the caller's pixel_values)
        L.gemm_bf16_nt(self.patches, self.w_patch, L.EPI_F32, bias=e["patch_embeddings.projection.bias"],
                       out_f32=self.proj)
        if self.fused_tail:      # key mask + per-sample position grid + assembly in one launch (bit-identical)
            L.image_embed_assemble_masked(self.proj, self.cls, self.pos0, self.pos_grid, self.inp["patch_mask"],
                                          self.inp["attention_mask"], self.mod1, self.h0, self.key_mask2, B, Lt, self.gh,
                                          self.gw, self.g0, H, nrep=2)
            return
        L.vilt_key_mask(self.inp["attention_mask"], self.inp["patch_mask"], self.key_mask2, B, Lt, self.gh, self.gw, 1,
                        nrep=2)
        L.pos_embed_resize_masked(self.pos_grid, self.inp["patch_mask"], self.pos_img, self.g0, B, self.gh, self.gw, 1,
                                  H)
        L.image_embed_assemble(self.proj, self.cls, self.pos0, self.pos_img, self.mod1, self.h0, B, Lt, self.np, S, H,
                               pos_batch_stride=self.np * H)

    def _layer_body(self, i: int, h_in, rows: int, nb: int, qkv, ctx, lse, h2, h3, st1=None, st2=None, u=None,
                    mask=None, ln1_done=False):
        """LN -> QKV -> attention -> out-proj(+res) -> LN -> FFN1(gelu) -> FFN2(+res): HF ViltLayer with the
        Adaptered_ViltOutput dense+residual (adaptered_output.py:74-76); returns the adapter input in h3."""
        W, H = self.layers[i], self.H
        x16, f16 = self.x16[:rows], self.f16[:rows]
        g8 = u is not None and u.dtype == torch.uint8
        if self._fp8_rows(rows):           # fp8 MFMA for the two products fed by a LayerNorm
            x8, xs = self.x8[:rows], self.xs[:rows]
            L.layernorm_fwd_fp8(h_in, W["ln1g"], W["ln1b"], self.ln_eps, rows, H, x8, xs, stats=st1)
            L.gemm_fp8_nt(x8, xs, W["wqkv8"], W["sqkv"], L.EPI_BF16, bias=W["bqkv"], out_bf16=qkv)
            L.attn_fwd(qkv, ctx, lse, nb, self.S, self.heads, key_mask=mask)
            L.gemm_bf16_nt(ctx, W["wo"], L.EPI_RESID_F32, bias=W["bo"], resid=h_in, out_f32=h2)
            L.layernorm_fwd_fp8(h2, W["ln2g"], W["ln2b"], self.ln_eps, rows, H, x8, xs, stats=st2)
            if self.fp8_ffn_chain:      # FFN1 -> e4m3 gelu(u) (+ gelu' codes) -> fp8 FFN2
                codes = u if g8 else self.dU.view(torch.uint8)[:rows, :self.I]       # (no backward through this call: scratch)
                L.gemm_fp8_nt(x8, xs, W["w18"], W["s1"], L.EPI_GELU_G8_F8, bias=W["b1"], out_bf16=self.f8[:rows], out2_bf16=codes)
                L.gemm_fp8_nt_f32(self.f8[:rows], self.f8s[:rows], W["w28"], W["s2"], bias=W["b2"], resid=h2, out_f32=h3)
                return
            L.gemm_fp8_nt(x8, xs, W["w18"], W["s1"], L.EPI_GELU_G8 if g8 else L.EPI_GELU, bias=W["b1"], out_bf16=f16,
                          out2_bf16=u if u is not None else self.dU[:rows])
            L.gemm_bf16_nt(f16, W["w2"], L.EPI_RESID_F32, bias=W["b2"], resid=h2, out_f32=h3)
            return
        if not ln1_done:     # otherwise x16 / st1 were written by the previous layer's fused adapter + LN kernel
            L.layernorm_fwd(h_in, W["ln1g"], W["ln1b"], self.ln_eps, rows, H, y_bf16=x16, stats=st1)
        L.gemm_bf16_nt(x16, W["wqkv"], L.EPI_BF16, bias=W["bqkv"], out_bf16=qkv)
        L.attn_fwd(qkv, ctx, lse, nb, self.S, self.heads, key_mask=mask)
        L.gemm_bf16_nt(ctx, W["wo"], L.EPI_RESID_F32, bias=W["bo"], resid=h_in, out_f32=h2)
        L.layernorm_fwd(h2, W["ln2g"], W["ln2b"], self.ln_eps, rows, H, y_bf16=x16, stats=st2)
        L.gemm_bf16_nt(x16, W["w1"], L.EPI_GELU_G8 if g8 else L.EPI_GELU, bias=W["b1"], out_bf16=f16, out2_bf16=u)
        L.gemm_bf16_nt(f16, W["w2"], L.EPI_RESID_F32, bias=W["b2"], resid=h2, out_f32=h3)

    def _fp8_rows(self, rows: int, bwd: bool = False) -> bool:
        """fp8 products are used where feddat_gemm_fp8_nt applies (M >= 1024); smaller launches stay bf16."""
        return self.fp8 and rows >= 1024 and (self.fp8_backward if bwd else self.fp8_forward)

    @_bound
    def _forward_dual(self):
        """Shared embeddings + layer-0 body, then both passes (gated | adapter_1) batched through layers 1..L-1."""
        R, R2, B = self.R, 2 * self.R, self.B
        self._embed()
        l0 = self.l0
        m1 = self.key_mask2[:self.B]
        m2 = self.key_mask2
        self._layer_body(0, self.h0, R, B, l0["qkv"], l0["ctx"], l0["lse"], l0["h2"], l0["h3"], st1=self.st0,
                         st2=self.st0, mask=m1)
        def adapter_then_ln1(x, i, first):
            """Adapter of layer i fused with layer i+1's layernorm_before (its bf16 output and row statistics go
            where that layer's LN kernel would have put them)."""
            nx, Wn = self.act[i + 1], self.layers[i + 1]
            if self._fp8_rows(R2):      # the next layer quantises its own LayerNorm output (feddat_layernorm_fwd_fp8)
                L.adapter_fwd(x, nx["h_in"], self._segs(i, first, False), R2, z_save=self.zsave[i])
                return
            L.adapter_fwd_ln(x, nx["h_in"], self._segs(i, first, False), R2, Wn["ln1g"], Wn["ln1b"], self.ln_eps,
                             self.x16[:R2], nx["st1"], z_save=self.zsave[i])
        if self.nl > 1:
            adapter_then_ln1(l0["h3"], 0, True)
        else:
            L.adapter_fwd(l0["h3"], self.h_out, self._segs(0, True, False), R2, z_save=self.zsave[0])
        for i in range(1, self.nl - 1):
            # one C-ABI call per layer (feddat_vilt_layer_fwd): ViltLayer body + adapter + the next layer's layernorm_before
            if not self.use_layer_calls or self.fp8:
                a = self.act[i]
                self._layer_body(i, a["h_in"], R2, 2 * B, a["qkv"], a["ctx"], a["lse"], a["h2"], a["h3"], st1=a["st1"],
                                 st2=a["st2"], u=a["u"], mask=m2, ln1_done=True)
                adapter_then_ln1(a["h3"], i, False)
                continue
            Wn = self.layers[i + 1]
            W, A, _ = self._layer_struct(i)
            L.vilt_layer_fwd(self.ctx, W, A, 2 * B, self.S, self.heads, self._segs(i, False, False), key_mask=m2,
                             ln1_done=True, next_ln_g=Wn["ln1g"], next_ln_b=Wn["ln1b"])
        if self.nl > 1:
            self._top_layer_fwd(m2)
            self._pool(self.top["h_out"], 2 * B, x_stride=self.H)
        else:
            self._pool(self.h_out, 2 * B)

    def _sg(self, A, sa_i, sa_k, Bm, sb_k, sb_j, I, J, K, out, ksplit=1, bias_j=None, alpha=1.0):
        """Skinny exact-fp32 product; long contractions are split over the grid and reduced deterministically."""
        if ksplit <= 1:
            L.sgemm_f32(A, sa_i, sa_k, Bm, sb_k, sb_j, I, J, K, out, bias_j=bias_j, alpha=alpha)
            return
        part = self._scratch1(ksplit * I * J)
        L.sgemm_f32(A, sa_i, sa_k, Bm, sb_k, sb_j, I, J, K, part, ksplit=ksplit, bias_j=bias_j,
                    out_split_stride=I * J, alpha=alpha)
        L.reduce_partials(part, I * J, ksplit, I * J, out)

    def _scratch1(self, n):
        if not hasattr(self, "_scr") or self._scr.numel() < n:
            self._scr = torch.empty(n, device=self.dev)
        return self._scr

    def _cls_rows(self, t, nb: int):
        """Strided view of token 0 of every sample: [nb, width] with row stride S * width (no copy)."""
        w = t.shape[1]
        return t.view(nb, self.S * w)[:, :w]

    def _top_layer_fwd(self, mask):
        """Last layer.  LN1 / QKV / attention see every token (keys and values of all tokens feed token 0), but only
        token 0 of each sample reaches the pooler (HF ViltPooler takes hidden_states[:, 0]; vilt.py:127), so the
        attention-output projection, LN2, FFN and the adapter run on the 2B token-0 rows, read in place through
        strided GEMM operands."""
        i = self.nl - 1
        a, W, H, t = self.act[i], self.layers[i], self.H, self.top
        R2, nb = 2 * self.R, 2 * self.B
        x16 = self.x16[:R2]       # LN1 of this layer: written by the previous layer's fused adapter + LN kernel
        if self._fp8_rows(R2):
            L.layernorm_fwd_fp8(a["h_in"], W["ln1g"], W["ln1b"], self.ln_eps, R2, H, self.x8[:R2], self.xs[:R2],
                                stats=a["st1"])
            L.gemm_fp8_nt(self.x8[:R2], self.xs[:R2], W["wqkv8"], W["sqkv"], L.EPI_BF16, bias=W["bqkv"], out_bf16=a["qkv"])
        elif self.cls_attention and self.top_q_cls:
            # (round 6) token 0 is the only QUERY of this layer that anything reads (step algebra item 6): K | V for every row
            # (N = 1536: two exact rounds of tiles at configs[1]), Q for the 2B token-0 rows as one skinny product
            L.gemm_bf16_nt(x16, W["wqkv"][H:], L.EPI_BF16, bias=W["bqkv"][H:], out_bf16=a["qkv"][:, H:])
            L.gemm_bf16_nt(self._cls_rows(x16, nb), W["wqkv"][:H], L.EPI_BF16, bias=W["bqkv"][:H],
                           out_bf16=self._cls_rows(a["qkv"], nb)[:, :H], skinny_workspace=self._skinny_ws())
        else:
            L.gemm_bf16_nt(x16, W["wqkv"], L.EPI_BF16, bias=W["bqkv"], out_bf16=a["qkv"])
        (L.attn_cls_fwd if self.cls_attention else L.attn_fwd)(a["qkv"], a["ctx"], a["lse"], nb, self.S, self.heads,
                                                                key_mask=mask)
        L.gemm_bf16_nt(self._cls_rows(a["ctx"], nb), W["wo"], L.EPI_RESID_F32, bias=W["bo"],
                       resid=self._cls_rows(a["h_in"], nb), out_f32=t["h2"], skinny_workspace=self._skinny_ws())
        L.layernorm_fwd(t["h2"], W["ln2g"], W["ln2b"], self.ln_eps, nb, H, y_bf16=t["x16"], stats=t["st2"])
        L.gemm_bf16_nt(t["x16"], W["w1"], L.EPI_GELU, bias=W["b1"], out_bf16=t["f16"], out2_bf16=t["u"],
                       skinny_workspace=self._skinny_ws())
        L.gemm_bf16_nt(t["f16"], W["w2"], L.EPI_RESID_F32, bias=W["b2"], resid=t["h2"], out_f32=t["h3"],
                       skinny_workspace=self._skinny_ws())
        L.adapter_fwd(t["h3"], t["h_out"], self._top_segs(False), nb, z_save=self.zsave[i])

    def _skinny_ws(self):
        """fp32 split-K partials of the top layer's 2B-row GEMMs (largest: 2B x 3072 x 768)."""
        if getattr(self, "_skws", None) is None:
            nb, H, I = 2 * self.B, self.H, self.I
            n = max(L.gemm_skinny_workspace_elems(nb, I, H), L.gemm_skinny_workspace_elems(nb, H, I),
                    L.gemm_skinny_workspace_elems(nb, H, H), L.gemm_skinny_workspace_elems(nb, H, 3 * H)) if nb <= 64 else 0
            self._skws = torch.empty(max(n, 1), device=self.dev) if n else False
        return self._skws if self._skws is not False else None

    def _top_segs(self, bwd: bool):
        key = ("top", bwd)
        if key not in self._segs_cache:
            i, B = self.nl - 1, self.B
            a0, a1, a2 = (self.ad16[a][i] for a in range(3))
            self._segs_cache[key] = L.make_segs([
                dict(row_begin=0, row_end=B, train_slot=0 if bwd else -1,
                     adapters=[dict(a0, scale=0.5), dict(a2, scale=0.5)]),
                dict(row_begin=B, row_end=2 * B, train_slot=0 if bwd else -1, adapters=[dict(a1, scale=1.0)]),
            ])
        return self._segs_cache[key]

    def _pool(self, h_last, nb: int, x_stride: int = None):
        """ViltModel.layernorm on token 0 + ViltPooler (dense + tanh) -> self.pooled[:nb]."""
        H = self.H
        self._pool_src, self._pool_stride = h_last, (self.S * H if x_stride is None else x_stride)
        if self.fused_tail:      # LayerNorm (statistics in the block) -> dense -> tanh in one launch
            L.head_gemm(L.ht_job(h_last, self._pool_stride, 1, self.pool_w, 1, H, nb, H, H, self.pooled, bias_j=self.pool_b,
                                 pro=L.HT_PRO_LN, pro_a=self.lnf_g, pro_b=self.lnf_b, pro_eps=self.ln_eps,
                                 stats_out=self.cls_st, epi=L.HT_EPI_TANH))
            return
        L.layernorm_fwd(h_last, self.lnf_g, self.lnf_b, self.ln_eps, nb, H, x_stride=self._pool_stride,
                        y_f32=self.cls_ln, stats=self.cls_st)
        self._sg(self.cls_ln, H, 1, self.pool_w, 1, H, nb, H, H, self.pooled, ksplit=4, bias_j=self.pool_b)
        L.tanh_fwd(self.pooled[:nb])

    def _head_fwd(self, pooled, slot: str, task: str):
        """vilt.py:202-209: fc0 -> LayerNorm(1536, eps 1e-5) -> GELU -> fc1 on the rows of `pooled` (B, or 2B for the
        joint P0 + P1 pass)."""
        B, H, C = pooled.shape[0], self.H, self.C
        hp, s = self.head[task], self.hd[slot]
        pre = f"task_layer.{task}."
        if self.fused_tail:
            L.head_gemm(L.ht_job(pooled, H, 1, hp.view(pre + "clf_fc0.weight"), 1, H, B, 2 * H, H, s["a0"],
                                 bias_j=hp.view(pre + "clf_fc0.bias")))
            L.head_ln_gelu(s["a0"], hp.view(pre + "clf_norm0.weight"), hp.view(pre + "clf_norm0.bias"), 1e-5, s["n0"], s["st"],
                           s["g0"])
            L.head_gemm(L.ht_job(s["g0"], 2 * H, 1, hp.view(pre + "clf_fc1.weight"), 1, 2 * H, B, C, 2 * H, s["logits"],
                                 bias_j=hp.view(pre + "clf_fc1.bias")))
            return s["logits"]
        self._sg(pooled, H, 1, hp.view(pre + "clf_fc0.weight"), 1, H, B, 2 * H, H, s["a0"], ksplit=4,
                 bias_j=hp.view(pre + "clf_fc0.bias"))
        L.layernorm_fwd(s["a0"], hp.view(pre + "clf_norm0.weight"), hp.view(pre + "clf_norm0.bias"), 1e-5, B, 2 * H,
                        y_f32=s["n0"], stats=s["st"])
        L.gelu_fwd(s["n0"], s["g0"])
        self._sg(s["g0"], 2 * H, 1, hp.view(pre + "clf_fc1.weight"), 1, 2 * H, B, C, 2 * H, s["logits"], ksplit=16,
                 bias_j=hp.view(pre + "clf_fc1.bias"))
        return s["logits"]

    def _head_bwd(self, pooled, slot: str, task: str, dpooled_out):
        """Gradients of the task head (all six tensors, fp32) and d(pooled) for B rows."""
        B, H, C = self.B, self.H, self.C
        hp, s = self.head[task], self.hd[slot]
        pre = f"task_layer.{task}."

        def G(n):
            return hp.view(pre + n, hp.g)
        dl = self.dlogits
        if self.fused_tail:
            # {dW_fc1 = dl^T g0, db_fc1} next to {dn0 = (dl W_fc1) * gelu'(n0)}; LayerNorm backward (dx, dgamma, dbeta);
            # {dW_fc0 = da0^T pooled, db_fc0} next to {dpooled = da0 W_fc0}: three launches
            L.head_gemm(L.ht_job(dl, 1, C, s["g0"], 2 * H, 1, C, 2 * H, B, G("clf_fc1.weight"), mode=1, colsum=G("clf_fc1.bias")),
                        L.ht_job(dl, C, 1, hp.view(pre + "clf_fc1.weight"), 2 * H, 1, B, 2 * H, C, self.dn0,
                                 epi=L.HT_EPI_MUL_DGELU, aux=s["n0"], ld_aux=2 * H))
            L.head_ln_bwd_full(self.dn0, s["a0"], s["st"], hp.view(pre + "clf_norm0.weight"), self.da0, G("clf_norm0.weight"),
                               G("clf_norm0.bias"))
            L.head_gemm(L.ht_job(self.da0, 1, 2 * H, pooled, H, 1, 2 * H, H, B, G("clf_fc0.weight"), mode=1,
                                 colsum=G("clf_fc0.bias")),
                        L.ht_job(self.da0, 2 * H, 1, hp.view(pre + "clf_fc0.weight"), H, 1, B, H, 2 * H, dpooled_out))
            return
        L.sgemm_f32(dl, 1, C, s["g0"], 2 * H, 1, C, 2 * H, B, G("clf_fc1.weight"), colsum=G("clf_fc1.bias"))
        L.sgemm_f32(dl, C, 1, hp.view(pre + "clf_fc1.weight"), 2 * H, 1, B, 2 * H, C, self.dg0)
        L.gelu_bwd(s["n0"], self.dg0, self.dn0)
        L.layernorm_bwd_full(self.dn0, s["a0"], s["st"], hp.view(pre + "clf_norm0.weight"), B, 2 * H, self.da0,
                             G("clf_norm0.weight"), G("clf_norm0.bias"))
        L.sgemm_f32(self.da0, 1, 2 * H, pooled, H, 1, 2 * H, H, B, G("clf_fc0.weight"), colsum=G("clf_fc0.bias"))
        self._sg(self.da0, 2 * H, 1, hp.view(pre + "clf_fc0.weight"), H, 1, B, H, 2 * H, dpooled_out, ksplit=8)

    def _adamw(self, grp: FlatGroup):
        L.adamw_flat(grp.p, grp.g, grp.m, grp.v, grp.seg_off, self._wd_vec(grp), grp.state, self.lr,
                     self.sched["warmup"], self.sched["total"], 0.9, 0.98, self.eps)

    def _wd_vec(self, grp: FlatGroup):
        if not hasattr(grp, "_wdv") or grp._wdv_val != self.wd:
            grp._wdv = grp.seg_wd * self.wd
            grp._wdv_val = self.wd
        return grp._wdv

    # ------------------------------------------------------------------------------------------ backward
    @_bound
    def _backward_dual(self):
        """dpooled [2B,H] -> adapter_0 grads (rows [0,R)) and adapter_1 grads (rows [R,2R))."""
        R, R2, B, H = self.R, 2 * self.R, self.B, self.H
        nb = 2 * B
        # the gradient entering the frozen backbone carries the loss scale from here on (alpha of this product; 1 for bf16
        # operands): every kernel below is linear in it, and feddat_wgrad_seg.grad_unscale takes it out again
        if self.fused_tail:      # d(pooler input) = (dpooled * (1 - pooled^2)) W_pool in one launch
            L.head_gemm(L.ht_job(self.dpooled, H, 1, self.pool_w, H, 1, nb, H, H, self.dcls_ln, pro=L.HT_PRO_TANH_BWD,
                                 pro_a=self.pooled, **self._scale_in()))
        else:
            L.tanh_bwd(self.pooled, self.dpooled, self.dpre)
            self._sg(self.dpre, H, 1, self.pool_w, H, 1, nb, H, H, self.dcls_ln, ksplit=4, alpha=self.loss_scale)
        L.layernorm_bwd_dx(self._pool_src, self.cls_st, self.lnf_g, nb, H, dy_f32=self.dcls_ln,
                           x_stride=self._pool_stride, out_f32=self.dcls)
        cur, oth = self.dh
        m2 = self.key_mask2
        top = self.nl - 1 if self.nl > 1 else 0
        if self.nl > 1:
            self._top_layer_bwd(cur, oth, m2)      # leaves d(h_in of the top layer) in `oth`
            cur, oth = oth, cur
        else:
            L.scatter_cls_rows(self.dcls, cur, None, nb, self.S, H)
        for i in range(top - 1, 0, -1):
            # one C-ABI call per layer (feddat_vilt_layer_bwd): adapter backward + its weight gradients, FFN2^T (. gelu'),
            # FFN1^T, LN2 backward (+ residual), attention-out^T, attention backward, QKV^T, LN1 backward (+ residual)
            if self._fp8_rows(R2, bwd=True):      # configs[4]: FFN2^T and attention-output^T on the fp8 MFMA, e4m3 gradient rows
                a, W = self.act[i], self.layers[i]
                L.adapter_bwd_fp8(cur, oth, self.g8, self.gsc, self._segs(i, False, True), R2, z_saved=self.zsave[i],
                                  z_out=self.z, dz_out=self.dz)
                self._adapter_wgrads(i, a["h3"], 0, cur)
                if self.fp8_ffn_chain and self.g8u:
                    L.gemm_fp8_nt(self.g8, self.gsc, W["w2T8"], W["s2T"], L.EPI_MUL_G8_F8, aux=a["u"], out_bf16=self.dU8)
                    L.gemm_fp8_nt(self.dU8, self.gsc, W["w1T8"], W["s1T4"], L.EPI_BF16, out_bf16=self.dx16)
                else:
                    L.gemm_fp8_nt(self.g8, self.gsc, W["w2T8"], W["s2T"], L.EPI_MUL_G8 if self.g8u else L.EPI_MUL_DGELU,
                                  aux=a["u"], out_bf16=self.dU)
                    L.gemm_bf16_nt(self.dU, W["w1T"], L.EPI_BF16, out_bf16=self.dx16)
                L.layernorm_bwd_dx_fp8(a["h2"], a["st2"], W["ln2g"], R2, H, self.g8, self.gsc, dy_bf16=self.dx16, dres=oth,
                                       out_f32=cur)
                L.gemm_fp8_nt(self.g8, self.gsc, W["woT8"], W["soT"], L.EPI_BF16, out_bf16=self.dctx)
                if self.fp8_mx_dqkv:
                    L.attn_bwd_fp8mx(a["qkv"], a["ctx"], a["lse"], self.dctx, self.dqkv8, self.dqkv_sc, nb, self.S, self.heads,
                                     key_mask=m2)
                    L.gemm_fp8mx_nt(self.dqkv8, self.dqkv_sc, W["wqkvT8"], W["sqkvT"], out_bf16=self.dx16)
                else:
                    L.attn_bwd(a["qkv"], a["ctx"], a["lse"], self.dctx, self.dqkv, nb, self.S, self.heads, key_mask=m2)
                    L.gemm_bf16_nt(self.dqkv, W["wqkvT"], L.EPI_BF16, out_bf16=self.dx16)
                L.layernorm_bwd_dx(a["h_in"], a["st1"], W["ln1g"], R2, H, dy_bf16=self.dx16, dres=cur, out_f32=oth)
                cur, oth = oth, cur
                continue
            if not self.use_layer_calls or self.fp8:
                a, W = self.act[i], self.layers[i]
                L.adapter_bwd(None, cur, oth, self._segs(i, False, True), R2, dx_bf16=self.dh16, z_out=self.z,
                              dz_out=self.dz, z_saved=self.zsave[i])
                self._adapter_wgrads(i, a["h3"], 0, cur)
                L.gemm_bf16_nt(self.dh16, W["w2T"], L.EPI_MUL_G8 if self.g8u else L.EPI_MUL_DGELU, aux=a["u"], out_bf16=self.dU)
                L.gemm_bf16_nt(self.dU, W["w1T"], L.EPI_BF16, out_bf16=self.dx16)
                L.layernorm_bwd_dx(a["h2"], a["st2"], W["ln2g"], R2, H, dy_bf16=self.dx16, dres=oth, out_f32=cur,
                                   out_bf16=self.dh16)
                L.gemm_bf16_nt(self.dh16, W["woT"], L.EPI_BF16, out_bf16=self.dctx)
                L.attn_bwd(a["qkv"], a["ctx"], a["lse"], self.dctx, self.dqkv, nb, self.S, self.heads, key_mask=m2)
                L.gemm_bf16_nt(self.dqkv, W["wqkvT"], L.EPI_BF16, out_bf16=self.dx16)
                L.layernorm_bwd_dx(a["h_in"], a["st1"], W["ln1g"], R2, H, dy_bf16=self.dx16, dres=cur, out_f32=oth)
                cur, oth = oth, cur
                continue
            W, A, _ = self._layer_struct(i)
            G = self._grad_struct(cur, oth)
            L.vilt_layer_bwd(self.ctx, W, A, G, nb, self.S, self.heads, self._segs(i, False, True),
                             self._wgrad_segs(i, self.act[i]["h3"], 0, cur), self._wpart(i), key_mask=m2,
                             wgrad_reduce_now=False)
            cur, oth = oth, cur
        # layer 0: weight gradients only (nothing trainable below)
        L.adapter_bwd(None, cur, None, self._segs(0, True, True), R2, z_out=self.z, dz_out=self.dz,
                      z_saved=self.zsave[0])
        self._adapter_wgrads(0, self.l0["h3"], -R, cur)
        self._wgrad_reduce_all()

    def _top_layer_bwd(self, cur, oth, mask):
        """Backward of the last layer: the incoming gradient is non-zero only on the 2B token-0 rows, so the adapter,
        FFN, LN2 and attention-output backward run on those rows; from the attention backward on every token is live."""
        i = self.nl - 1
        a, W, H, t = self.act[i], self.layers[i], self.H, self.top
        R2, nb, B = 2 * self.R, 2 * self.B, self.B
        L.adapter_bwd(None, self.dcls, t["dh3"], self._top_segs(True), nb, dx_bf16=t["dh316"], z_out=self.z,
                      dz_out=self.dz, z_saved=self.zsave[i])
        key = ("wg-top", self.opt_adapters)
        if key not in self._segs_cache:
            n = self.ad_layer_numel
            segs = [dict(x=t["h3"][r0:], dy=self.dcls[r0:], z=self.z[r0:], dz=self.dz[r0:],
                         grad=self.ad[ad].g[i * n:(i + 1) * n], rows=B, scale=sc, **self._scale_out())
                    for ad, r0, sc in ((0, 0, 0.5), (1, B, 1.0)) if ad in self.opt_adapters]
            self._segs_cache[key] = L.make_wgrad_segs(segs) if segs else None
        if self._segs_cache[key] is not None:
            L.adapter_wgrad_partial(self._segs_cache[key], self._wpart(i))
        ws = self._skinny_ws()
        L.gemm_bf16_nt(t["dh316"], W["w2T"], L.EPI_MUL_DGELU, aux=t["u"], out_bf16=t["dU"], skinny_workspace=ws)
        L.gemm_bf16_nt(t["dU"], W["w1T"], L.EPI_BF16, out_bf16=t["dx2"], skinny_workspace=ws)
        L.layernorm_bwd_dx(t["h2"], t["st2"], W["ln2g"], nb, H, dy_bf16=t["dx2"], dres=t["dh3"], out_f32=t["dh2"],
                           out_bf16=t["dh216"])
        L.gemm_bf16_nt(t["dh216"], W["woT"], L.EPI_F32, out_f32=t["dctx"], skinny_workspace=ws)
        # the token-0 rows' residual gradient t["dh2"] reaches the LN1 backward below compact (dres_every = S); only the dense
        # attention path needs its own operand scattered into a dense buffer
        if self.cls_attention:       # rank-1 backward straight from the fp32 token-0 gradient rows
            L.attn_cls_bwd(a["qkv"], a["ctx"], a["lse"], t["dctx"], self.dqkv, nb, self.S, self.heads, key_mask=mask)
        else:
            L.scatter_cls_rows(t["dctx"], None, self.dctx, nb, self.S, H)
            L.attn_bwd(a["qkv"], a["ctx"], a["lse"], self.dctx, self.dqkv, nb, self.S, self.heads, key_mask=mask)
        if self.cls_attention and self.top_q_cls and nb <= 64:
            # dQ is non-zero on the token-0 rows only: the dense product contracts dK | dV alone (K = 1536: the zero third of
            # the contraction is not read or multiplied -- bit-identical on those rows), and the 2B token-0 rows are
            # overwritten by the full contraction as one skinny product
            L.gemm_bf16_nt(self.dqkv[:, H:], W["wqkvT"][:, H:], L.EPI_BF16, out_bf16=self.dx16)
            L.gemm_bf16_nt(self._cls_rows(self.dqkv, nb), W["wqkvT"], L.EPI_BF16, out_bf16=self._cls_rows(self.dx16, nb),
                           skinny_workspace=self._skinny_ws())
        else:
            L.gemm_bf16_nt(self.dqkv, W["wqkvT"], L.EPI_BF16, out_bf16=self.dx16)
        L.layernorm_bwd_dx(a["h_in"], a["st1"], W["ln1g"], R2, H, dy_bf16=self.dx16, dres=t["dh2"], dres_every=self.S, out_f32=oth)

    def _layer_struct(self, i: int):
        """ctypes views of layer i's frozen weights and static activation buffers for the composite entry points."""
        if i not in self._layer_structs:
            Wd, a = self.layers[i], self.act[i]
            R2 = 2 * self.R
            W = L._fill(L.ViltLayerWeights, wqkv=Wd["wqkv"], wo=Wd["wo"], w1=Wd["w1"], w2=Wd["w2"], wqkvT=Wd["wqkvT"],
                        woT=Wd["woT"], w1T=Wd["w1T"], w2T=Wd["w2T"], bqkv=Wd["bqkv"], bo=Wd["bo"], b1=Wd["b1"], b2=Wd["b2"],
                        ln1_g=Wd["ln1g"], ln1_b=Wd["ln1b"], ln2_g=Wd["ln2g"], ln2_b=Wd["ln2b"])
            W.ln_eps = self.ln_eps
            nxt = self.act[i + 1] if i + 1 < self.nl else None
            A = L._fill(L.ViltLayerActs, h_in=a["h_in"], st1=a["st1"], qkv=a["qkv"], ctx=a["ctx"], lse=a["lse"], h2=a["h2"],
                        st2=a["st2"], u=a["u"], h3=a["h3"], z_save=self.zsave[i], h_out=nxt["h_in"] if nxt else self.h_out,
                        x16=self.x16[:R2], f16=self.f16[:R2], st1_next=nxt["st1"] if nxt else None)
            self._layer_structs[i] = (W, A, None)
        return self._layer_structs[i]

    def _grad_struct(self, cur, oth):
        key = ("G", cur.data_ptr())
        if key not in self._layer_structs:      # dh3 and dh_in share `oth`: dh3 is dead before dh_in is written
            self._layer_structs[key] = L._fill(L.ViltLayerGrads, dh_out=cur, dh_in=oth, dh3=oth, dh16=self.dh16, dU=self.dU,
                                               dx16=self.dx16, dctx=self.dctx, dqkv=self.dqkv, z=self.z, dz=self.dz)
        return self._layer_structs[key]

    def _wgrad_segs(self, layer: int, x, x_delta_s: int, dy):
        key = ("wg", layer, x.data_ptr(), dy.data_ptr(), self.opt_adapters)
        if key not in self._segs_cache:
            R, n = self.R, self.ad_layer_numel
            segs = []
            for a, row0, xrow0, sc in ((0, 0, 0, 0.5), (1, R, R + x_delta_s, 1.0)):
                if a in self.opt_adapters:
                    segs.append(dict(x=x[xrow0:], dy=dy[row0:], z=self.z[row0:], dz=self.dz[row0:],
                                     grad=self.ad[a].g[layer * n:(layer + 1) * n], rows=R, scale=sc, **self._scale_out()))
            self._segs_cache[key] = L.make_wgrad_segs(segs) if segs else None
        return self._segs_cache[key]

    def _wpart(self, layer: int):
        return self.wpart_all[layer * self.wpart_stride:(layer + 1) * self.wpart_stride]

    def _wgrad_reduce_all(self):
        """Fold every layer's partial sums into the adapters' flat gradient buffers (one launch)."""
        ads = [a for a in (0, 1) if a in self.opt_adapters]
        if not ads:
            return
        key = ("wg-reduce", self.opt_adapters)
        if key not in self._segs_cache:
            n = self.ad_layer_numel
            ptrs = [self.ad[a].g[i * n:(i + 1) * n].data_ptr() for i in range(self.nl) for a in ads]
            self._segs_cache[key] = torch.tensor(ptrs, dtype=torch.int64, device=self.dev)
        if self._dyn():      # + GradScaler's inf check where the loss scale leaves the gradients: flags[a] for adapter a
            L.adapter_wgrad_reduce_checked(self._segs_cache[key], self.nl, len(ads), self.wpart_all, self.wpart_stride,
                                           self.ovf_flags[ads[0]:])
        else:
            L.adapter_wgrad_reduce(self._segs_cache[key], self.nl, len(ads), self.wpart_all, self.wpart_stride)

    def _dyn(self) -> bool:
        """Dynamic loss scale in effect (it rides on the fused tail's multi-group AdamW launch)."""
        return self.dynamic_scale and self.fused_tail

    def _scale_in(self):
        """How the loss scale enters the backbone's backward (factor of the pooler-backward product)."""
        return dict(alpha=1.0, alpha_dev=self.scaler_f[0:1]) if self._dyn() else dict(alpha=self.loss_scale)

    def _scale_out(self):
        """... and how it leaves, where the adapter weight gradients are formed."""
        return dict(grad_unscale=1.0, grad_unscale_dev=self.scaler_f[1:2]) if self._dyn() else \
            dict(grad_unscale=1.0 / self.loss_scale)

    def scaler_state(self) -> Dict[str, float]:
        """Host copy of the loss scaler (one device read-back): current scale, growth tracker, skipped sub-steps / batches."""
        f, i = self.scaler_f.tolist(), self.scaler_i.tolist()
        return dict(scale=f[0], growth_tracker=i[0], skipped_substeps=i[1], skipped_batches=i[2], dynamic=self._dyn())

    def _adapter_wgrads(self, layer: int, x, x_delta_s: int, dy):
        """dW_up = s dy^T z, db_up = s sum_t dy, dW_down = dz^T x, db_down = sum_t dz (autograd of adapter.py:125-146)
        for adapter_0 (rows [0,R)) and adapter_1 (rows [R,2R)): exact fp32 MFMA, split over tokens, deterministic
        reduction straight into the flat gradient buffers."""
        segs = self._wgrad_segs(layer, x, x_delta_s, dy)
        if segs is not None:
            L.adapter_wgrad_partial(segs, self._wpart(layer))

    # ------------------------------------------------------------------------------------------ train step
    @_bound
    def begin_local_update(self, task: str, steps_per_epoch: int, num_epochs: int = 15, warmup_ratio: float = 0.1,
                           opt_adapters: Sequence[int] = (0, 1)):
        """TaskTrainer.train prologue (task_trainer.py:36-59): teacher snapshot, fresh AdamW state and schedule."""
        self.task = task
        self.copy_global_to_teacher()
        total = steps_per_epoch * num_epochs
        self.sched = dict(total=total, warmup=int(total * warmup_ratio))
        self.opt_adapters = tuple(opt_adapters)
        for grp in (self.ad[0], self.ad[1], self.head[task]):
            grp.m.zero_()
            grp.v.zero_()
            grp.g.zero_()
        # scheduler index / Adam step count per group: adapter_1 is stepped at 2b, adapter_0 at 2b+1, head at both
        self.ad[1].state.copy_(torch.tensor([0, 0], dtype=torch.int32))
        self.ad[0].state.copy_(torch.tensor([1, 0], dtype=torch.int32))
        self.head[task].state.copy_(torch.tensor([0, 0], dtype=torch.int32))
        # a fresh GradScaler per local update (the reference builds a fresh Accelerator per round: main.py:435)
        self.scaler_f.copy_(torch.tensor([self.loss_scale, 1.0 / self.loss_scale], dtype=torch.float32))
        self.scaler_i.zero_()
        self.ovf_flags.zero_()
        # a captured step stays valid across local updates as long as everything it froze into kernel arguments or into its
        # launch list is unchanged (all mutable state -- weights, moments, counters -- lives in device buffers)
        sig = (task, total, self.sched["warmup"], self.opt_adapters, self.lr, self.wd, self.eps, self.use_layer_calls, self.fp8,
               self.fp8_ffn_chain, self.fused_tail, self.cls_attention, self.operands, self.loss_scale, self.fp8_mx_dqkv,
               self._dyn(), self.scale_growth_interval, self.top_q_cls)        # host-side switches that change the launch list are part of the signature
        if getattr(self, "_graph_sig", None) != sig:
            self.graph = None
            self._graph_sig = sig

    def _adamw_group(self, grp: FlatGroup, d_sched: int = 0, d_adam: int = 0, **kw):
        return L.adamw_group(grp.p, grp.g, grp.m, grp.v, grp.seg_off, self._wd_vec(grp), grp.state, d_sched, d_adam, **kw)

    def _adamw_many(self, groups):
        L.adamw_multi(groups, self.lr, self.sched["warmup"], self.sched["total"], 0.9, 0.98, self.eps)

    def _loss(self, logits, teacher, slot):
        if self._dyn():      # + non-finite loss -> the sub-step's overflow flag (p1 = sub-step A, p2 = B)
            flag = self.ovf_flags[1:2] if slot == "p1" else self.ovf_flags[0:1]
            L.dat_loss_fwd_bwd_checked(logits, teacher, self.inp["target"], self.dlogits, self.loss_buf[slot], flag)
        elif self.fused_tail:
            L.dat_loss_fwd_bwd_single(logits, teacher, self.inp["target"], self.dlogits, self.loss_buf[slot])
        else:
            L.dat_loss_fwd_bwd(logits, teacher, self.inp["target"], self.dlogits, self.loss_buf[slot])

    @_bound
    def _step_kernels(self):
        B, task = self.B, self.task
        hp = self.head[task]
        self._forward_dual()
        pooled_g, pooled_s = self.pooled[:B], self.pooled[B:]
        # P0: logits of the gated pass with the current head (no grad)          task_trainer.py:283-287
        # P1: adapter_1 pass, KL to P0                                           task_trainer.py:290-308
        # same head weights for both -> one 2B-row pass over [pooled_g; pooled_s]
        logits_both = self._head_fwd(self.pooled[:2 * B], "both", task)
        logits_all, logits_1 = logits_both[:B], logits_both[B:]
        self._loss(logits_1, logits_all, "p1")
        self._head_bwd(pooled_s, "p1", task, self.dpooled[B:])
        dyn = self._dyn()
        fB, fA = self.ovf_flags[0:1], self.ovf_flags[1:2]
        if dyn:      # sub-step A's head update: skipped on a non-finite loss; the old p | m | v are kept for the restore below
            self._adamw_many([self._adamw_group(hp, skip_if=(fA,), bak=self.head_bak[task], bak_mode=1)])
        elif self.fused_tail:
            self._adamw_many([self._adamw_group(hp)])       # sub-step 2b; its counters are ticked once, at the end of the step
        else:
            self._adamw(hp)
            L.step_tick(hp.state, 1, 1)
        # P2: gated pass again -- same pooled features, UPDATED head, KL to logits_1     task_trainer.py:311-328
        logits_0 = self._head_fwd(pooled_g, "p2", task)
        self._loss(logits_0, logits_1, "p2")
        self._head_bwd(pooled_g, "p2", task, self.dpooled[:B])
        # one backward for both passes, then the deferred adapter_1 step (lr index 2b) and the P2 steps (2b+1)
        self._backward_dual()
        if self.fused_tail:
            # adapter_1 (2b), head (2b + 1: reads its counters one ahead), adapter_0 (2b + 1) in ONE launch, then the bf16
            # operand copies, then ONE tick for all counters
            if dyn:
                # GradScaler's skips as device predicates: A overflowed (flag A) -> nothing of this batch is applied: adapter_1
                # and adapter_0 stay, the head returns to its state before sub-step A; only B overflowed -> A stands, the head's
                # second update and adapter_0's are skipped.  feddat_dat_step_finish ticks the counters by what was applied
                # (a skipped optimizer step skips its scheduler tick), updates the scale and clears the flags.
                groups = ([self._adamw_group(self.ad[1], skip_if=(fA,))] if 1 in self.opt_adapters else []) + \
                    [self._adamw_group(hp, 1, 1, skip_if=(fB,), bak=self.head_bak[task], bak_mode=2, restore_if=fA)] + \
                    ([self._adamw_group(self.ad[0], skip_if=(fA, fB))] if 0 in self.opt_adapters else [])
            else:
                groups = ([self._adamw_group(self.ad[1])] if 1 in self.opt_adapters else []) + [self._adamw_group(hp, 1, 1)] + \
                    ([self._adamw_group(self.ad[0])] if 0 in self.opt_adapters else [])
            self._adamw_many(groups)
            for a in (1, 0):
                if a in self.opt_adapters:
                    self.repack_adapter(a)
            if dyn:
                L.dat_step_finish(hp.state, self.ad[1].state, self.ad[0].state, self.ovf_flags, self.scaler_f, self.scaler_i,
                                  self.scale_growth, self.scale_backoff, self.scale_growth_interval)
            else:
                L.step_tick_multi([hp.state, self.ad[1].state, self.ad[0].state], [2, 2, 2], [2, 1, 1])
            return
        if 1 in self.opt_adapters:
            self._adamw(self.ad[1])
            self.repack_adapter(1)
        L.step_tick(self.ad[1].state, 2, 1)
        self._adamw(hp)
        L.step_tick(hp.state, 1, 1)
        if 0 in self.opt_adapters:
            self._adamw(self.ad[0])
            self.repack_adapter(0)
        L.step_tick(self.ad[0].state, 2, 1)

    @_bound
    def train_step(self, batch: Optional[Dict[str, torch.Tensor]] = None, use_graph: bool = False):
        """One DAT+MKD step (task_trainer.py:280-330).  Returns the device tensor holding what the reference
        returns: loss_0 = BCE * num_labels of the P2 pass (loss_buf['p2'][0]); [1] = KL, [2] = L_0."""
        if batch is not None:
            self.set_batch(batch)
        if not use_graph:
            self._step_kernels()
        else:
            if self.graph is None:
                self._capture()
            self.graph.replay()
        return self.loss_buf["p2"]

    @_bound
    def ensure_captured(self):
        """Capture the step graph now if it is not there yet (TaskTrainer.train calls this before it starts the upload
        worker, so no capture ever overlaps a prefetch)."""
        if self.graph is None:
            self._capture()

    @_bound
    def _capture(self):
        """Capture the whole step into one hipGraph (all launches are on static buffers; the LR schedule and Adam
        step counts live on the device).  The optimizer state is saved/restored around the warm-up + capture
        run so that capturing does not advance training."""
        groups = [self.ad[0], self.ad[1], self.head[self.task]]
        saved = [(g.p.clone(), g.m.clone(), g.v.clone(), g.state.clone()) for g in groups]
        saved_scaler = (self.scaler_f.clone(), self.scaler_i.clone(), self.ovf_flags.clone())
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._step_kernels()      # warm-up (sets function attributes, allocates lazily created scratch)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        # thread_local: other host threads (feddat_amd.data.DevicePrefetcher's upload worker) may allocate and copy on their
        # own streams while this thread captures; the default global mode would turn their hipMalloc / hipMemcpy into
        # hipErrorStreamCaptureUnsupported
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            self._step_kernels()
        torch.cuda.synchronize()
        for g, (p, m, v, st) in zip(groups, saved):
            g.p.copy_(p)
            g.m.copy_(m)
            g.v.copy_(v)
            g.state.copy_(st)
        self.scaler_f.copy_(saved_scaler[0])
        self.scaler_i.copy_(saved_scaler[1])
        self.ovf_flags.copy_(saved_scaler[2])
        for a in (0, 1):
            self.repack_adapter(a)
        torch.cuda.synchronize()
        self.graph = graph

    def assert_finite(self):
        """Last line of defence.  With the dynamic loss scale (the default for fp16 operands) an overflowed sub-step is skipped
        on the device like GradScaler does (task_trainer.py:302 via accelerate) and this never fires; with a STATIC scale
        (dynamic_loss_scale=False, or the unfused tail) a gradient operand that left fp16's range turns the update non-finite.
        One host read-back of the trainable state, meant to be called once per local update (TaskTrainer.train does; train.main
        agrees on the outcome across ranks BEFORE the FedAvg collective); raises with what to change."""
        bad = self.nonfinite_groups()
        if bad:
            raise L.FeddatHipError(
                f"non-finite values in {', '.join(bad)} after the local update: with operands={self.operands!r} the backward "
                f"carries a {'dynamic' if self._dyn() else 'static'} loss scale (initial value {self.loss_scale:g}); "
                + ("the scaler skips overflowed steps, so the non-finite values entered through the inputs or the weights"
                   if self._dyn() else
                   "this model's gradients leave fp16's range at that scale -- construct the engine with "
                   "dynamic_loss_scale=True, a smaller power of two (loss_scale=...) or operands='bf16'"))

    def nonfinite_groups(self):
        """Names of the trainable groups holding an inf / NaN (one host read-back each); [] = all finite."""
        return [name for name, grp in (("adapter_0", self.ad[0]), ("adapter_1", self.ad[1]), ("head", self.head[self.task]))
                if not bool(torch.isfinite(grp.p).all())]

    # ------------------------------------------------------------------------------------------ inference
    @_bound
    @torch.no_grad()
    def forward(self, batch: Dict[str, torch.Tensor], mode: str, task: Optional[str] = None):
        """model(task_key, images, texts) -> (pooled, logits) in adapter mode `mode` ('gating' | 'adapter_k')
        (vilt.py:244-264 with the adapter switches of vilt.py:363-373).  Single pass over B*S rows."""
        task = task or self.task
        self.set_batch(batch)
        R, B = self.R, self.B
        self._embed()
        l0 = self.l0
        m1 = self.key_mask2[:self.B]
        h = self.h0
        for i in range(self.nl):
            self._layer_body(i, h, R, B, l0["qkv"], l0["ctx"], l0["lse"], l0["h2"], l0["h3"], st1=self.st0,
                             st2=self.st0, mask=m1)
            h = self.dh[i & 1][:R]
            L.adapter_fwd(l0["h3"], h, self._single_segs(i, mode, R), R)
        self._pool(h, B)
        logits = self._head_fwd(self.pooled[:B], "all", task)
        return self.pooled[:B].clone(), logits.clone()

    # ------------------------------------------------------------------------------------------ state dict
    def state_dict(self) -> Dict[str, torch.Tensor]:
        """Trainable tensors under the reference's state-dict keys (views into the flat buffers)."""
        out = {}
        for grp in self.ad + list(self.head.values()):
            for n in grp.names:
                out[n] = grp.view(n)
        return out

    @_bound
    def load_tensors(self, tensors: Dict[str, torch.Tensor]):
        sd = self.state_dict()
        touched = set()
        for n, v in tensors.items():
            sd[n].copy_(v.to(self.dev, torch.float32))
            for a in range(3):
                if f"adapter_{a}_" in n:
                    touched.add(a)
        for a in touched:
            self.repack_adapter(a)

    def comm_flat(self) -> torch.Tensor:
        """The FedAvg payload: all adapter_1 tensors back-to-back in state-dict order (main.py:154-163,499-503)."""
        return self.ad[1].p
