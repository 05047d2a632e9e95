"""ALBEF (ViT-B/16 + BERT-base encoder + 6-layer decoder) dual-adapter (DAT + MKD) local-update engine on MI355X --
BASELINE.json configs[3].

Host-side sequencing of the HIP kernels in libfeddat_hip.so for the reference's two-stream path: ALBEF.forward(train=True)
(src/modeling/models/albef_model.py:69-145) inside the dat branch of TaskTrainer.train_step
(src/train/visionlanguage_tasks/task_trainer.py:280-330, ALBEF wiring :296-297,316-317, vocabulary-axis KL :506-516), with
the adapters of vit.py:99-110 (after the MLP residual) and xbert.py:438-445 / adapter.py:97-116 (BertOutput, same LayerNorm
before and after the adapter).  As in the ViLT engine, no arithmetic happens in Python / PyTorch: torch owns device
buffers and the stream, every launch goes through the C ABI.

Step algebra: only adapter tensors are trainable (main.py:138-159: the LM head `.cls.` is personal but frozen), so the
no-grad gated pass P0 and the gated pass P2 of one train_step are the same computation (P1 updates adapter_1 only, the
gated passes read adapter_0 / adapter_2) -> the gated forward runs ONCE; its logits serve as the teacher of P1 and as the
student of P2.  Nothing trainable lies below the first ViT block's adapter, so the backward stops there.

Dropout.  The reference trains under model.train() (task_trainer.py:75) with hidden_dropout_prob =
attention_probs_dropout_prob = 0.1 (src/configs/model_configs.py:44-46) in the two BERT towers -- after the embedding
LayerNorm (xbert.py:216), on the attention probabilities (:333), on BertSelfOutput's dense output (:360) and on BertOutput's
dense output ahead of the adapter (:440); the ViT has none (vit.py:120).  `dropout=p` reproduces that with counter-based
masks (seed, train step, pass, site, element -> bit; include/feddat_hip.h) that the backward regenerates.  With p > 0 the P0
and P2 forwards draw DIFFERENT masks (three independent forwards in the reference), so the "gated forward once" identity
only holds for the image encoder: the text towers then run three times (P0 no-grad, P1, P2).  dropout=0 (the default) is
the deterministic configuration every parity fixture except g12 is captured in; feddat_amd.train.main passes the
reference's 0.1 (--albef_dropout).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch

from . import lib as L
from .engine import FlatGroup, _bound

PRE = "albef_model.albef."
ADAPTER_TENSORS = ("down.weight", "down.bias", "up.weight", "up.bias")


class AlbefDatEngine:
    def __init__(self, params: Dict[str, torch.Tensor], device, batch: int, n_answers: int, q_len: int = 25, a_len: int = 4,
                 vit_depth: int = 12, enc_layers: int = 12, fusion_layer: int = 6, dec_layers: int = 6, image: int = 384,
                 vocab: int = 30522, lr: float = 1e-4, weight_decay: float = 1e-2, adam_eps: float = 1e-8, pad_id: int = 0,
                 max_pos: int = 512, dropout: float = 0.0, seed: int = 0, stack_text: bool = False,
                 operands: str = "f16", loss_scale: Optional[float] = None, dynamic_loss_scale: Optional[bool] = None,
                 scale_growth_interval: int = 2000):
        """operands="f16": every 16-bit MFMA operand in IEEE half (libfeddat_hip_f16.so) with a power-of-two loss scale on
        dL/dlogits (feddat_lm_loss_fwd_bwd's grad_scale, default 2^14) that leaves through feddat_wgrad_seg.grad_unscale -- the
        ViLT engine's scheme (engine.ViltDatEngine), with the same device-side dynamic scaler (dynamic_loss_scale, default on for
        "f16").  The default operand format is fp16 -- in this class, in train.main (--encoder_name albef_no_distill without
        --mixed_precision) and in bench.py --workload albef alike, the same as the ViLT engine's and the reference's own setting
        (accelerate_config.yaml:8): against the reference's own full-size 40-step rounds (tests/golden/g11b_albef_full_round40.npz
        and ..._seed8800.npz) fp16 operands land at 3.2e-4 / 2.6e-4 on the worst adapter element (mean ratio 0.005 / 0.003), bf16
        operands at 7.6e-4 / 8.4e-4 (0.024 / 0.028): both inside north_star's 1e-3, bf16 with a fifth of it to spare, for 1.5 % of
        the step (tests/test_sizes_gpu.py::test_albef_full_size_round_of_40_steps_vs_reference_golden asserts all four)."""
        if operands not in L.OPERAND_DTYPE:
            raise L.FeddatHipError(f"operands must be 'bf16' or 'f16', got {operands!r}")
        self.operands, self.op_dtype = operands, L.OPERAND_DTYPE[operands]
        # fp16 operands: the loss scale is dynamic by default -- GradScaler on the device exactly as in ViltDatEngine (DESIGN.md
        # section 5b).  The LM head is frozen here, so the two sub-steps share no trainable tensor: flag B (adapter_0's pass)
        # skips adapter_0's step alone; flag A (adapter_1's pass) voids the batch like in the ViLT engine (one finish kernel).
        self.dynamic_scale = bool(operands == "f16" if dynamic_loss_scale is None else dynamic_loss_scale)
        self.scale_growth_interval = int(scale_growth_interval)
        self.loss_scale = float(loss_scale if loss_scale is not None else (16384.0 if operands == "f16" else 1.0))
        if self.loss_scale <= 0 or math.frexp(self.loss_scale)[0] != 0.5:
            raise L.FeddatHipError("loss_scale must be a power of two (it is removed exactly)")
        with L.operands(operands):
            self._init(params, device, batch, n_answers, q_len, a_len, vit_depth, enc_layers, fusion_layer, dec_layers, image, vocab,
                       lr, weight_decay, adam_eps, pad_id, max_pos, dropout, seed, stack_text)

    def _init(self, params, device, batch, n_answers, q_len, a_len, vit_depth, enc_layers, fusion_layer, dec_layers, image, vocab, lr,
              weight_decay, adam_eps, pad_id, max_pos, dropout, seed, stack_text):
        L.load()
        if not 0.0 <= dropout < 1.0:
            raise L.FeddatHipError("dropout must be in [0, 1)")
        self.dropout, self.seed = float(dropout), int(seed)
        self.dev = dev = torch.device(device)
        self.B, self.N, self.Lq, self.La = batch, n_answers, q_len, a_len
        self.vd, self.el, self.fl, self.dl = vit_depth, enc_layers, fusion_layer, dec_layers
        self.H, self.I, self.heads, self.r, self.P = 768, 3072, 12, 48, 16
        self.img = image
        self.gp = image // self.P
        self.Ni = self.gp * self.gp + 1
        self.V, self.Vp = vocab, ((vocab + 127) // 128) * 128
        self.pad_id = pad_id
        self.lr, self.wd, self.eps = lr, weight_decay, adam_eps
        H, I = self.H, self.I
        self.Mi, self.Mq, self.Ma, self.R = batch * self.Ni, batch * q_len, n_answers * a_len, n_answers * (a_len - 1)

        def Pm(name):
            return params[PRE + name].to(dev, torch.float32).contiguous()

        def bf16_of(w):
            out = torch.empty(w.shape, dtype=self.op_dtype, device=dev)
            L.cvt_f32_bf16(w.contiguous(), out)
            return out

        def bf16_T(w):
            out = torch.empty(w.shape[1], w.shape[0], dtype=self.op_dtype, device=dev)
            L.transpose_f32_bf16(w.contiguous(), out, w.shape[0], w.shape[1])
            return out

        def lin(prefix):          # forward operand, transposed operand (dX), bias
            w = Pm(prefix + ".weight")
            return dict(w=bf16_of(w), wT=bf16_T(w), b=Pm(prefix + ".bias"))

        # ---------------- frozen ViT-B/16 ----------------
        v = "visual_encoder."
        self.vit = dict(cls=Pm(v + "cls_token").reshape(H), pos=Pm(v + "pos_embed")[0].contiguous(),
                        wp=bf16_of(Pm(v + "patch_embed.proj.weight").reshape(H, 3 * self.P * self.P)),
                        bp=Pm(v + "patch_embed.proj.bias"), ng=Pm(v + "norm.weight"), nb=Pm(v + "norm.bias"), blocks=[])
        for i in range(vit_depth):
            b = f"{v}blocks.{i}."
            self.vit["blocks"].append(dict(qkv=lin(b + "attn.qkv"), proj=lin(b + "attn.proj"), fc1=lin(b + "mlp.fc1"),
                                           fc2=lin(b + "mlp.fc2"), n1g=Pm(b + "norm1.weight"), n1b=Pm(b + "norm1.bias"),
                                           n2g=Pm(b + "norm2.weight"), n2b=Pm(b + "norm2.bias")))
        self.zero_h = torch.zeros(H, device=dev)

        # ---------------- frozen BERT towers ----------------
        def attn_block(prefix, cross):
            wq, wk, wv = (Pm(f"{prefix}self.{n}.weight") for n in ("query", "key", "value"))
            bq, bk, bv = (Pm(f"{prefix}self.{n}.bias") for n in ("query", "key", "value"))
            d = dict(o=lin(prefix + "output.dense"), lng=Pm(prefix + "output.LayerNorm.weight"),
                     lnb=Pm(prefix + "output.LayerNorm.bias"))
            if cross:          # Q from the text stream, K | V fused from the other stream
                d.update(q=dict(w=bf16_of(wq), wT=bf16_T(wq), b=bq),
                         kv=dict(w=bf16_of(torch.cat([wk, wv], 0)), wT=bf16_T(torch.cat([wk, wv], 0)), b=torch.cat([bk, bv])))
            else:
                w = torch.cat([wq, wk, wv], 0)
                d.update(qkv=dict(w=bf16_of(w), wT=bf16_T(w), b=torch.cat([bq, bk, bv])))
            return d

        def tower(prefix, layers, fusion):
            e = prefix + "embeddings."
            t = dict(word=Pm(e + "word_embeddings.weight"), pos=Pm(e + "position_embeddings.weight"),
                     typ=Pm(e + "token_type_embeddings.weight"), lng=Pm(e + "LayerNorm.weight"), lnb=Pm(e + "LayerNorm.bias"),
                     layers=[])
            for i in range(layers):
                Lp = f"{prefix}encoder.layer.{i}."
                t["layers"].append(dict(att=attn_block(Lp + "attention.", False),
                                        cross=attn_block(Lp + "crossattention.", True) if i >= fusion else None,
                                        fc1=lin(Lp + "intermediate.dense"), fc2=lin(Lp + "output.dense"),
                                        lng=Pm(Lp + "output.LayerNorm.weight"), lnb=Pm(Lp + "output.LayerNorm.bias")))
            # The cross-attention K | V projections of ALL fusion layers read the same encoder-side states (image_embeds for
            # the text encoder, the repeated question states for the decoder): ONE product with the layers' weights stacked
            # along N in the forward ([rows, n_cross * 1536], each layer's K | V a column slice), and ONE product over the
            # stacked dK | dV columns (K = n_cross * 1536) for the gradient of those states in the backward -- instead of six
            # 18 464-row launches of N = 1536 and six read-modify-write passes over the fp32 gradient.
            cross = [i for i, Ly in enumerate(t["layers"]) if Ly["cross"] is not None]
            t["cross_layers"] = cross
            if cross:
                t["kv_all"] = dict(w=torch.cat([t["layers"][i]["cross"]["kv"]["w"] for i in cross], 0).contiguous(),
                                   b=torch.cat([t["layers"][i]["cross"]["kv"]["b"] for i in cross], 0).contiguous(),
                                   wT=torch.cat([t["layers"][i]["cross"]["kv"]["wT"] for i in cross], 1).contiguous())
                for i in cross:          # the per-layer copies are not used any more
                    t["layers"][i]["cross"]["kv"] = None
            return t
        self.enc = tower("text_encoder.", enc_layers, fusion_layer)
        self.dec = tower("text_decoder.bert.", dec_layers, 0)
        c = "text_decoder.cls.predictions."
        wemb = torch.zeros(self.Vp, H, device=dev)
        wemb[:vocab] = self.dec["word"]                      # tied LM-head weight, vocabulary padded to a multiple of 128
        bpad = torch.zeros(self.Vp, device=dev)
        bpad[:vocab] = Pm(c + "bias")
        self.head = dict(t=lin(c + "transform.dense"), lng=Pm(c + "transform.LayerNorm.weight"),
                         lnb=Pm(c + "transform.LayerNorm.bias"), w=bf16_of(wemb), wT=bf16_T(wemb), b=bpad)
        del wemb

        # ---------------- trainable state: three adapters over the 30 modules, flat per adapter ----------------
        self.modules: List[str] = [f"visual_encoder.blocks.{i}.adapter." for i in range(vit_depth)] + \
            [f"text_encoder.encoder.layer.{i}.output.adapter." for i in range(enc_layers)] + \
            [f"text_decoder.bert.encoder.layer.{i}.output.adapter." for i in range(dec_layers)]
        shp = {"down.weight": (self.r, H), "down.bias": (self.r,), "up.weight": (H, self.r), "up.bias": (H,)}
        self.ad = [FlatGroup([(PRE + m + f"adapter_{a}_{t}", shp[t]) for m in self.modules for t in ADAPTER_TENSORS], dev,
                             with_opt=(a != 2)) for a in range(3)]
        for grp in self.ad:
            for n in grp.names:
                grp.view(n).copy_(params[n].to(dev, torch.float32))
        self.ad_numel = self.r * H + self.r + H * self.r + H
        nm = len(self.modules)
        self._pack16 = [torch.empty(nm, 4, self.r * H, dtype=self.op_dtype, device=dev) for _ in range(3)]
        for a in range(3):
            self.repack_adapter(a)
        self.sched = dict(warmup=1, total=2)
        self.opt_adapters = (0, 1)
        self.graph = None
        self._segs_cache: Dict = {}
        # weight-gradient partial sums: one slot per adapter module and pass; ONE batched reduction per backward pass folds them
        # into the flat gradient buffers (feddat_adapter_wgrad_partial / _reduce) instead of one reduce launch per module
        self.wpart_stride = L.adapter_wgrad_workspace_elems(1)
        self.wpart = {m: torch.empty(len(self.modules) * self.wpart_stride, device=dev) for m in ("gating", "adapter_1")}
        self._wg_done = {m: [] for m in ("gating", "adapter_1", "both")}
        # stack_text (round 4, OFF by default -- measured slower): with dropout = 0 the text towers of the gated and the
        # adapter_1 pass are the same launches on different rows, so they can run ONCE on 2x the rows (rows [0, M) gated,
        # [M, 2M) adapter_1 -- the ViLT engine's 2R-row batching; adapters, weight gradients and the two losses take
        # two-segment descriptors): 366 -> 183 text-side GEMM launches per step, results equal to the two-pass form
        # (tests/test_albef_gpu.py::test_stacked_text_towers_equal_the_two_separate_passes).  Under hipGraph replay it LOSES:
        # 34.6 against 32.6 ms / step at B = 32 in round 4, 31.2 against 30.4 in round 5 once its 1600-row products left the
        # persistent GEMM kernels (36 tiles = 36 CUs) for the small-tile one (same box, tools/albef_stack_ab.py) -- the two passes' small kernels already
        # overlap each other on two streams, and a stacked launch of twice the rows costs more than one of the pair.  It
        # stays as a switch because it halves the host launches of the eager (no-graph) step.
        self.batch_text = bool(stack_text) and self.dropout <= 0
        if self.batch_text:
            self.wpart_stride2 = L.adapter_wgrad_workspace_elems(2)
            self.wpart["both"] = torch.empty(len(self.modules) * self.wpart_stride2, device=dev)
        self.side = None           # second stream of train_step (created lazily on the engine's device)
        self.scaler_f = torch.tensor([self.loss_scale, 1.0 / self.loss_scale], dtype=torch.float32, device=self.dev)
        self.scaler_i = torch.zeros(4, dtype=torch.int32, device=self.dev)
        self.ovf_flags = torch.zeros(2, dtype=torch.int32, device=self.dev)      # {B: adapter_0's pass, A: adapter_1's pass}
        self._no_head = torch.zeros(2, dtype=torch.int32, device=self.dev)       # feddat_dat_step_finish's head counters: none here
        self.drop_ctr = torch.zeros(2, dtype=torch.int32, device=dev)      # [0] = train_steps since begin_local_update
        self._alloc()
        if self.dropout > 0:       # teacher logits of P0: the gated text pass is re-run (other masks) for P2
            self.logits_all = torch.empty(self.R, self.Vp, device=dev)

    # ------------------------------------------------------------------------------------------ buffers
    def _alloc(self):
        dev, H, I = self.dev, self.H, self.I

        def f32(*s):
            return torch.empty(*s, device=dev)

        def b16(*s):
            return torch.empty(*s, dtype=self.op_dtype, device=dev)
        B, N, Lq, La = self.B, self.N, self.Lq, self.La
        self.inp = dict(image=f32(B, 3, self.img, self.img), question_ids=torch.zeros(B, Lq, dtype=torch.int64, device=dev),
                        question_mask=torch.ones(B, Lq, dtype=torch.int64, device=dev),
                        answer_ids=torch.zeros(N, La, dtype=torch.int64, device=dev),
                        answer_mask=torch.ones(N, La, dtype=torch.int64, device=dev), weights=f32(N))
        self.zero_tt_q = torch.zeros(B, Lq, dtype=torch.int64, device=dev)
        self.zero_tt_a = torch.zeros(N, La, dtype=torch.int64, device=dev)
        self.qmask8 = torch.ones(B, Lq, dtype=torch.uint8, device=dev)
        self.amask8 = torch.ones(N, La, dtype=torch.uint8, device=dev)
        self.qmask8_rep = torch.ones(N, Lq, dtype=torch.uint8, device=dev)
        self.rep_idx = torch.zeros(N * Lq, dtype=torch.int32, device=dev)       # (answer, token) -> question-state row
        self.seg_off = torch.zeros(B + 1, dtype=torch.int32, device=dev)        # answers of question b: [off[b], off[b+1])
        self.sel_idx = torch.zeros(self.R, dtype=torch.int32, device=dev)       # LM rows: (answer, t < La-1) -> decoder row
        self.unsel_idx = torch.full((self.Ma,), -1, dtype=torch.int32, device=dev)
        self.labels = torch.zeros(self.R, dtype=torch.int64, device=dev)
        self.row_w = f32(self.R)
        self.row_kl = torch.ones(self.R, device=dev)      # per-row factor on the MKD term (0 on rows that only padding created)

        def vit_set():
            Mi = self.Mi

            def u_img():     # what fc2^T needs of the pre-GELU u: 8-bit gelu' codes where FEDDAT_EPI_GELU_G8 applies, else bf16 u
                return torch.empty(Mi, I, dtype=torch.uint8, device=dev) if Mi >= 1024 else b16(Mi, I)
            return dict(patches=b16(B * (self.Ni - 1), 3 * self.P * self.P), proj=f32(B * (self.Ni - 1), H),
                        h0=f32(Mi, H), st0=f32(Mi, 2), x16=b16(Mi, H), f16=b16(Mi, I),
                        blocks=[dict(h_in=f32(Mi, H) if i else None, st1=f32(Mi, 2), qkv=b16(Mi, 3 * H), ctx=b16(Mi, H),
                                     lse=f32(self.B, self.heads, self.Ni), h2=f32(Mi, H), st2=f32(Mi, 2), u=u_img(),
                                     h3=f32(Mi, H), zs=f32(Mi, 2, self.r)) for i in range(self.vd)],
                        out=f32(Mi, H), stf=f32(Mi, 2), emb16=b16(Mi, H))

        def bert_set(M, nb, Sq, layers, cross_from, kv_rows):
            nc = max(layers - cross_from, 0)
            kvc_all = b16(kv_rows, nc * 2 * H) if nc else None      # K | V of every cross-attention layer, one product

            def layer(i):
                d = dict(qkv=b16(M, 3 * H), ctx=b16(M, H), lse=f32(nb, self.heads, Sq), t1=f32(M, H), st_a=f32(M, 2),
                         a=f32(M, H), a16=b16(M, H), u=b16(M, I), s1=f32(M, H), st_x=f32(M, 2), x=f32(M, H),
                         zs=f32(M, 2, self.r), s2=f32(M, H), st_o=f32(M, 2), out=f32(M, H), out16=b16(M, H))
                if i >= cross_from:
                    j = i - cross_from
                    d.update(qc=b16(M, H), kvc=kvc_all[:, j * 2 * H:(j + 1) * 2 * H], ctx2=b16(M, H),
                             lse2=f32(nb, self.heads, Sq), t2=f32(M, H), st_c=f32(M, 2), c=f32(M, H), c16=b16(M, H))
                return d
            return dict(h=f32(M, H), h16=b16(M, H), f16=b16(M, I), tA=f32(M, H), td=f32(M, H), kvc_all=kvc_all,
                        layers=[layer(i) for i in range(layers)])

        def act_set():
            return dict(vit=vit_set(), enc=bert_set(self.Mq, B, Lq, self.el, self.fl, self.Mi),
                        dec=bert_set(self.Ma, N, La, self.dl, 0, N * Lq), enc_rep16=b16(N * Lq, H),
                        hsel=f32(self.R, H), hsel16=b16(self.R, H), tu=f32(self.R, H), tg=f32(self.R, H), tst=f32(self.R, 2),
                        ty16=b16(self.R, H), logits=f32(self.R, self.Vp), loss=f32(4 + 2 * self.R))
        self.acts = {"gating": act_set(), "adapter_1": act_set()}
        # Everything in the image encoder AHEAD of block 0's adapter (patch embedding, block 0's attention and MLP) is frozen
        # and sees the same image in both passes: train_step computes it once (_vit_fwd(..., part="prefix")), both activation
        # sets point at the same buffers (read-only in the backward: block 0's adapter input h3 for its weight gradient).
        vg, v1 = self.acts["gating"]["vit"], self.acts["adapter_1"]["vit"]
        for k in ("patches", "proj", "h0", "st0"):
            v1[k] = vg[k]
        for k in ("st1", "qkv", "ctx", "lse", "h2", "st2", "u", "h3"):
            v1["blocks"][0][k] = vg["blocks"][0][k]
        # image_embeds of the two passes back to back: one K / V source for the batched text encoder's cross-attention
        self.emb16_both = b16(2 * self.Mi, H)
        self.acts["gating"]["vit"]["emb16"] = self.emb16_both[:self.Mi]
        self.acts["adapter_1"]["vit"]["emb16"] = self.emb16_both[self.Mi:]
        if self.batch_text:
            R2 = 2 * self.R
            self.acts["both"] = dict(enc=bert_set(2 * self.Mq, 2 * B, Lq, self.el, self.fl, 2 * self.Mi),
                                     dec=bert_set(2 * self.Ma, 2 * N, La, self.dl, 0, 2 * N * Lq), enc_rep16=b16(2 * N * Lq, H),
                                     hsel=f32(R2, H), hsel16=b16(R2, H), tu=f32(R2, H), tg=f32(R2, H), tst=f32(R2, 2),
                                     ty16=b16(R2, H), logits=f32(R2, self.Vp))
        # backward scratch: one set per pass (their text-side backward passes run on two streams)
        Mmax = max(self.Mi, self.Mq, self.Ma, N * Lq, self.R)

        def scratch():
            return dict(dlogits=b16(self.R, self.Vp), d1=f32(Mmax, H), d2=f32(Mmax, H), d3=f32(Mmax, H), d4=f32(Mmax, H),
                      b1=b16(Mmax, H), b2=b16(Mmax, H), bI=b16(Mmax, I), b3=b16(Mmax, 3 * H),
                      bkv=b16(max(self.Mi, N * Lq), max(self.el - self.fl, self.dl) * 2 * H),
                      z=f32(Mmax, self.r), dz=f32(Mmax, self.r), dsum=f32(max(B, N), self.heads, max(self.Ni, Lq, La)),
                      d_img=f32(self.Mi, H), d_qs=f32(self.Mq, H), d_rep=f32(N * Lq, H), d_dec=f32(self.Ma, H))
        self.gs = {"gating": scratch(), "adapter_1": scratch()}
        if self.batch_text:
            M2 = 2 * max(self.Mq, self.Ma, N * Lq, self.R)
            self.gs["both"] = dict(dlogits=b16(2 * self.R, self.Vp), d1=f32(M2, H), d2=f32(M2, H), d3=f32(M2, H), d4=f32(M2, H),
                                   b1=b16(M2, H), b2=b16(M2, H), bI=b16(M2, I), b3=b16(M2, 3 * H),
                                   bkv=b16(2 * max(self.Mi, N * Lq), max(self.el - self.fl, self.dl) * 2 * H),
                                   z=f32(M2, self.r), dz=f32(M2, self.r),
                                   dsum=f32(2 * max(B, N), self.heads, max(self.Ni, Lq, La)), d_img=f32(2 * self.Mi, H),
                                   d_qs=f32(2 * self.Mq, H), d_rep=f32(2 * N * Lq, H), d_dec=f32(2 * self.Ma, H))
        # text-side inputs / index maps, per form: "single" = one pass (B questions, N answers), "both" = the two passes stacked
        self.tx = {"single": dict(nq=B, na=N, Mq=self.Mq, Ma=self.Ma, R=self.R, q_ids=self.inp["question_ids"], q_tt=self.zero_tt_q,
                                  a_ids=self.inp["answer_ids"], a_tt=self.zero_tt_a, qmask8=self.qmask8, amask8=self.amask8,
                                  qmask8_rep=self.qmask8_rep, rep_idx=self.rep_idx, seg_off=self.seg_off, sel_idx=self.sel_idx,
                                  unsel_idx=self.unsel_idx)}
        if self.batch_text:
            i64 = lambda *s_: torch.zeros(*s_, dtype=torch.int64, device=dev)      # noqa: E731
            u8 = lambda *s_: torch.ones(*s_, dtype=torch.uint8, device=dev)        # noqa: E731
            i32 = lambda *s_: torch.zeros(*s_, dtype=torch.int32, device=dev)      # noqa: E731
            self.tx["both"] = dict(nq=2 * B, na=2 * N, Mq=2 * self.Mq, Ma=2 * self.Ma, R=2 * self.R, q_ids=i64(2 * B, Lq),
                                   q_tt=i64(2 * B, Lq), a_ids=i64(2 * N, La), a_tt=i64(2 * N, La), qmask8=u8(2 * B, Lq),
                                   amask8=u8(2 * N, La), qmask8_rep=u8(2 * N, Lq), rep_idx=i32(2 * N * Lq), seg_off=i32(2 * B + 1),
                                   sel_idx=i32(2 * self.R), unsel_idx=i32(2 * self.Ma))

    # ------------------------------------------------------------------------------------------ adapters
    def _pack(self, a: int, m: int):
        c = self._pack16[a][m]
        H, r = self.H, self.r
        base = PRE + self.modules[m] + f"adapter_{a}_"
        return dict(wd=c[0].view(r, H), wdT=c[1].view(H, r), wu=c[2].view(H, r), wuT=c[3].view(r, H),
                    bd=self.ad[a].view(base + "down.bias"), bu=self.ad[a].view(base + "up.bias"),
                    wd32=self.ad[a].view(base + "down.weight"), wu32=self.ad[a].view(base + "up.weight"))

    @_bound
    def repack_adapter(self, a: int):
        p0 = self._pack(a, 0)
        L.adapter_pack_strided(p0["wd32"], p0["wu32"], self.ad_numel, p0["wd"], p0["wdT"], p0["wu"], p0["wuT"],
                               4 * self.r * self.H, len(self.modules))

    def _segs(self, m: int, mode: str, rows: int, bwd: bool):
        key = (m, mode, rows, bwd)
        if key not in self._segs_cache and mode == "both":       # rows [0, rows / 2) gated, [rows / 2, rows) adapter_1
            h = rows // 2
            ts = 0 if bwd else -1
            self._segs_cache[key] = L.make_segs([
                dict(row_begin=0, row_end=h, train_slot=ts, adapters=[dict(self._pack(0, m), scale=0.5), dict(self._pack(2, m), scale=0.5)]),
                dict(row_begin=h, row_end=rows, train_slot=ts, adapters=[dict(self._pack(1, m), scale=1.0)])])
        if key not in self._segs_cache:
            if mode == "gating":
                ads = [dict(self._pack(0, m), scale=0.5), dict(self._pack(2, m), scale=0.5)]
            else:
                ads = [dict(self._pack(int(mode.split("_")[1]), m), scale=1.0)]
            self._segs_cache[key] = L.make_segs([dict(row_begin=0, row_end=rows, train_slot=0 if bwd else -1, adapters=ads)])
        return self._segs_cache[key]

    def _wgrad(self, m: int, mode: str, x, dy, rows: int):
        if mode == "both":       # adapter_0 from the gated half (scale 0.5), adapter_1 from the other; both are optimised here
            key = ("wg2", m, x.data_ptr(), dy.data_ptr(), rows)
            if key not in self._segs_cache:
                n, h, g = self.ad_numel, rows // 2, self.gs["both"]
                self._segs_cache[key] = L.make_wgrad_segs([
                    dict(x=x, dy=dy, z=g["z"], dz=g["dz"], grad=self.ad[0].g[m * n:(m + 1) * n], rows=h, scale=0.5,
                         **self._scale_out()),
                    dict(x=x[h:], dy=dy[h:], z=g["z"][h:], dz=g["dz"][h:], grad=self.ad[1].g[m * n:(m + 1) * n], rows=h, scale=1.0,
                         **self._scale_out())])
            ws = self.wpart_stride2
            L.adapter_wgrad_partial(self._segs_cache[key], self.wpart["both"][m * ws:(m + 1) * ws])
            self._wg_done["both"].append(m)
            return
        a = 0 if mode == "gating" else int(mode.split("_")[1])
        if a not in self.opt_adapters:
            return
        key = ("wg", m, a, x.data_ptr(), dy.data_ptr(), rows)
        if key not in self._segs_cache:
            n = self.ad_numel
            self._segs_cache[key] = L.make_wgrad_segs([dict(x=x, dy=dy, z=self.gs[mode]["z"], dz=self.gs[mode]["dz"],
                                                            grad=self.ad[a].g[m * n:(m + 1) * n], rows=rows,
                                                            scale=0.5 if mode == "gating" else 1.0, **self._scale_out())])
        ws = self.wpart_stride
        L.adapter_wgrad_partial(self._segs_cache[key], self.wpart[mode][m * ws:(m + 1) * ws])
        self._wg_done[mode].append(m)

    def _wgrad_reduce(self, mode: str):
        """Fold the partial sums of every module this backward pass produced into the adapter's gradient buffer (one launch)."""
        done, self._wg_done[mode] = self._wg_done[mode], []
        if not done:
            return
        if mode == "both":
            ms = tuple(sorted(done))
            key = ("wg-reduce2", ms)
            if key not in self._segs_cache:
                n = self.ad_numel
                assert ms == tuple(range(ms[0], ms[0] + len(ms)))
                ptrs = [self.ad[a].g[m * n:(m + 1) * n].data_ptr() for m in ms for a in (0, 1)]
                self._segs_cache[key] = torch.tensor(ptrs, dtype=torch.int64, device=self.dev)
            ws = self.wpart_stride2
            if self.dynamic_scale:      # segments (adapter_0, adapter_1) = flags (B, A)
                L.adapter_wgrad_reduce_checked(self._segs_cache[key], len(ms), 2, self.wpart["both"][ms[0] * ws:], ws, self.ovf_flags)
            else:
                L.adapter_wgrad_reduce(self._segs_cache[key], len(ms), 2, self.wpart["both"][ms[0] * ws:], ws)
            return
        a = 0 if mode == "gating" else int(mode.split("_")[1])
        ms = tuple(sorted(done))
        key = ("wg-reduce", mode, ms)
        if key not in self._segs_cache:
            n, ws = self.ad_numel, self.wpart_stride
            contiguous = ms == tuple(range(ms[0], ms[0] + len(ms)))
            ptrs = torch.tensor([self.ad[a].g[m * n:(m + 1) * n].data_ptr() for m in ms], dtype=torch.int64, device=self.dev)
            self._segs_cache[key] = (ptrs, contiguous)
        ptrs, contiguous = self._segs_cache[key]
        ws = self.wpart_stride
        red = (lambda *a_: L.adapter_wgrad_reduce_checked(*a_, self.ovf_flags[a:a + 1])) if self.dynamic_scale else L.adapter_wgrad_reduce
        if contiguous:       # the usual case (all 30 modules): slots m0 .. m0 + len - 1 are one strided batch
            red(ptrs, len(ms), 1, self.wpart[mode][ms[0] * ws:], ws)
        else:
            for j, m in enumerate(ms):
                red(ptrs[j:j + 1], 1, 1, self.wpart[mode][m * ws:], ws)

    def _scale_out(self):
        """How the loss scale leaves, where the adapter weight gradients are formed (engine.ViltDatEngine._scale_out)."""
        return dict(grad_unscale=1.0, grad_unscale_dev=self.scaler_f[1:2]) if self.dynamic_scale else \
            dict(grad_unscale=1.0 / self.loss_scale)

    def _scale_in(self, mode: str):
        """... and how it enters, on dL/dlogits of the pass `mode` (+ that pass's overflow flag for a non-finite loss)."""
        if not self.dynamic_scale:
            return dict(grad_scale=self.loss_scale)
        a = 0 if mode == "gating" else 1
        return dict(grad_scale=1.0, grad_scale_dev=self.scaler_f[0:1], nonfinite=self.ovf_flags[a:a + 1])

    def scaler_state(self) -> Dict[str, float]:
        """Host copy of the loss scaler (one device read-back): current scale, growth tracker, skipped sub-steps / batches."""
        f, i = self.scaler_f.tolist(), self.scaler_i.tolist()
        return dict(scale=f[0], growth_tracker=i[0], skipped_substeps=i[1], skipped_batches=i[2], dynamic=self.dynamic_scale)

    @_bound
    def copy_global_to_teacher(self):
        self.ad[2].p.copy_(self.ad[1].p)
        self.repack_adapter(2)

    _KINDS = {"emb": 0, "self_probs": 1, "self_out": 2, "cross_probs": 3, "cross_out": 4, "out": 5}

    def _drop(self, pass_id, tower: int, layer: int, kind: str):
        """(p, key0, key1, step counter) of one dropout site of pass `pass_id` (0 / 1 / 2 = P0 / P1 / P2), or None when the
        pass runs without dropout (pass_id None: eval / plain forwards, or dropout = 0).  Site numbering: (tower * 64 + layer) * 8 + kind
        (tower 0 = text encoder, 1 = decoder)."""
        if pass_id is None or self.dropout <= 0:
            return None
        key = (pass_id, tower, layer, kind)
        if key not in self._segs_cache:
            k0, k1 = L.dropout_keys(self.seed, pass_id, (tower * 64 + layer) * 8 + self._KINDS[kind])
            self._segs_cache[key] = (self.dropout, k0, k1, self.drop_ctr)
        return self._segs_cache[key]

    # ------------------------------------------------------------------------------------------ inputs
    @_bound
    def set_batch(self, batch: Dict):
        """Reference batch after tokenisation (albef.py:52-60): image [B,3,R,R] f32; question_ids / question_mask [B, lq];
        answer_ids / answer_mask [n, la]; weights [n]; k = answers per question (host list, sum = n).
        The reference pads to the LONGEST question / answer of the batch and has as many answers as the batch brings
        (vqa_dataset_crossvqa.py:377-422); the engine's buffers are one static frame [B, Lq], [N, La] (one captured graph).
        Any lq <= Lq, la <= La, n <= N is accepted and embedded in the frame so that the RESULT is the reference's on the
        unpadded batch: padded key positions are masked out of every attention; padded answer POSITIONS and padded ANSWERS get
        label -100, weight 0 and factor 0 on their MKD rows (row_kl), and the MKD's batchmean divides by n, not N -- the
        reference's KL runs over every row of logits[:, :-1] of ITS batch, i.e. [n, la - 1, V] (task_trainer.py:506-516)."""
        B, N, Lq, La = self.B, self.N, self.Lq, self.La
        img = batch["image"]
        if tuple(img.shape) != tuple(self.inp["image"].shape):
            raise L.FeddatHipError(f"engine built for image {tuple(self.inp['image'].shape)}, got {tuple(img.shape)}")
        self.inp["image"].copy_(img, non_blocking=True)
        qi, qm, ai, am, wt = (batch[k] for k in ("question_ids", "question_mask", "answer_ids", "answer_mask", "weights"))
        lq, n, la = qi.shape[1], ai.shape[0], ai.shape[1]
        if qi.shape[0] != B or tuple(qm.shape) != tuple(qi.shape) or tuple(am.shape) != tuple(ai.shape) or wt.shape[0] != n:
            raise L.FeddatHipError("question / answer tensors of one batch disagree in shape")
        if lq > Lq or la > La or n > N or n < 1 or la < 2:
            raise L.FeddatHipError(f"batch ({lq} question tokens, {n} answers of {la} tokens) exceeds the engine's frame "
                                   f"({Lq}, {N}, {La})")
        for k, t in (("question_ids", qi), ("answer_ids", ai)):      # an out-of-range id would fault in the embedding gather;
            if not t.is_cuda and int(t.max()) >= self.V:             # host batches are checked (device ones: no sync)
                raise L.FeddatHipError("token id outside the vocabulary")
        full = (lq, n, la) == (Lq, N, La)
        if full:
            for k, src in (("question_ids", qi), ("question_mask", qm), ("answer_ids", ai), ("answer_mask", am), ("weights", wt)):
                self.inp[k].copy_(src, non_blocking=True)
        else:
            self.inp["question_ids"].fill_(self.pad_id)
            self.inp["question_mask"].zero_()
            self.inp["question_ids"][:, :lq].copy_(qi, non_blocking=True)
            self.inp["question_mask"][:, :lq].copy_(qm, non_blocking=True)
            self.inp["answer_ids"].fill_(self.pad_id)
            self.inp["answer_mask"].zero_()
            self.inp["answer_ids"][:n, :la].copy_(ai, non_blocking=True)
            self.inp["answer_mask"][:n, :la].copy_(am, non_blocking=True)
            if n < N:       # padded answers: one live [CLS] key so that no attention row is empty; they carry no loss
                self.inp["answer_ids"][n:, 0] = self.inp["answer_ids"][0, 0]
                self.inp["answer_mask"][n:, 0] = 1
            self.inp["weights"].zero_()
            self.inp["weights"][:n].copy_(wt, non_blocking=True)
        ks = [int(x) for x in batch["k"]]
        if len(ks) != B or sum(ks) != n:
            raise L.FeddatHipError("k must list the answers per question and sum to the number of answers")
        ks[-1] += N - n                              # padded answers ride with the last question (zero gradient rows)
        shape_key = (tuple(ks), n, la)
        if getattr(self, "_k", None) != shape_key:   # index maps depend on the batch's shape only (host-built once per pattern)
            self._k = shape_key
            qof = torch.repeat_interleave(torch.arange(B), torch.tensor(ks))
            self.qof = qof.to(self.dev)
            self.rep_idx.copy_((qof[:, None] * Lq + torch.arange(Lq)[None]).reshape(-1).int())
            self.seg_off.copy_(torch.tensor([0] + list(torch.tensor(ks).cumsum(0)), dtype=torch.int32))
            sel = (torch.arange(N)[:, None] * La + torch.arange(La - 1)[None]).reshape(-1)
            self.sel_idx.copy_(sel.int())
            un = torch.full((self.Ma,), -1, dtype=torch.int32)
            un[sel] = torch.arange(self.R, dtype=torch.int32)
            self.unsel_idx.copy_(un)
            rk = torch.zeros(N, La - 1)
            rk[:n, :la - 1] = float(N) / float(n)
            self.row_kl.copy_(rk.reshape(-1))
            if self.batch_text:       # the same maps for the stacked passes: the second pass's rows sit behind the first's
                tb = self.tx["both"]
                rep, off = self.rep_idx.cpu(), self.seg_off.cpu()
                tb["rep_idx"].copy_(torch.cat([rep, rep + self.Mq]))
                tb["seg_off"].copy_(torch.cat([off[:B], off + N]))
                tb["sel_idx"].copy_(torch.cat([sel, sel + self.Ma]).int())
                tb["unsel_idx"].copy_(torch.cat([un, torch.where(un >= 0, un + self.R, un)]))
        # masks, labels and per-row weights of this batch (tiny integer work on the device tensors)
        self.qmask8.copy_(self.inp["question_mask"])
        self.amask8.copy_(self.inp["answer_mask"])
        self.qmask8_rep.copy_(self.qmask8[self.qof])
        if self.batch_text:
            tb = self.tx["both"]
            for dst, src in ((tb["q_ids"], self.inp["question_ids"]), (tb["a_ids"], self.inp["answer_ids"]), (tb["qmask8"], self.qmask8),
                             (tb["amask8"], self.amask8), (tb["qmask8_rep"], self.qmask8_rep)):
                dst.view(2, *src.shape).copy_(src.unsqueeze(0).expand(2, *src.shape))
        ids = self.inp["answer_ids"]
        lab = ids[:, 1:].masked_fill(ids[:, 1:] == self.pad_id, -100)
        if not full:
            lab[n:] = -100
            lab[:, la - 1:] = -100
        self.labels.copy_(lab.reshape(-1))
        self.row_w.copy_((self.inp["weights"] / self.B)[:, None].expand(self.N, self.La - 1).reshape(-1))

    # ------------------------------------------------------------------------------------------ forward
    def _vit_fwd(self, S, mode: str, part: str = "all"):
        """part: "prefix" = patch embedding + block 0 up to (not including) its adapter -- the same for every adapter mode;
        "suffix" = from block 0's adapter on; "all" = both."""
        B, H, Ni, vt = self.B, self.H, self.Ni, self.vit
        V = S["vit"]
        if part != "suffix":
            L.im2col_patches(self.inp["image"], V["patches"], B, 3, self.img, self.img, self.P)
            L.gemm_bf16_nt(V["patches"], vt["wp"], L.EPI_F32, bias=vt["bp"], out_f32=V["proj"])
            # x = cat(cls, patches) + pos_embed  (vit.py:179-184): row 0 = cls + pos[0], rows 1.. = proj + pos[1..]
            L.image_embed_assemble(V["proj"], vt["cls"], vt["pos"][0], vt["pos"][1:], self.zero_h, V["h0"], B, 0, Ni - 1, Ni, H)
            b0 = vt["blocks"][0]
            L.layernorm_fwd(V["h0"], b0["n1g"], b0["n1b"], 1e-6, self.Mi, H, y_bf16=V["x16"], stats=V["blocks"][0]["st1"])
        h = V["h0"]
        for i, W in enumerate(vt["blocks"]):
            A = V["blocks"][i]
            if i > 0 or part != "suffix":
                L.gemm_bf16_nt(V["x16"], W["qkv"]["w"], L.EPI_BF16, bias=W["qkv"]["b"], out_bf16=A["qkv"])
                q, k, v = A["qkv"][:, :H], A["qkv"][:, H:2 * H], A["qkv"][:, 2 * H:]
                L.attn2_fwd(q, k, v, A["ctx"], A["lse"], B, Ni, Ni, self.heads)
                L.gemm_bf16_nt(A["ctx"], W["proj"]["w"], L.EPI_RESID_F32, bias=W["proj"]["b"], resid=h, out_f32=A["h2"])
                L.layernorm_fwd(A["h2"], W["n2g"], W["n2b"], 1e-6, self.Mi, H, y_bf16=V["x16"], stats=A["st2"])
                L.gemm_bf16_nt(V["x16"], W["fc1"]["w"], L.EPI_GELU_G8 if A["u"].dtype == torch.uint8 else L.EPI_GELU,
                               bias=W["fc1"]["b"], out_bf16=V["f16"], out2_bf16=A["u"])
                L.gemm_bf16_nt(V["f16"], W["fc2"]["w"], L.EPI_RESID_F32, bias=W["fc2"]["b"], resid=A["h2"], out_f32=A["h3"])
            if part == "prefix":
                return
            last = i == self.vd - 1
            nxt = V["out"] if last else V["blocks"][i + 1]["h_in"]
            g_, b_ = (vt["ng"], vt["nb"]) if last else (vt["blocks"][i + 1]["n1g"], vt["blocks"][i + 1]["n1b"])
            # adapter (vit.py:107) fused with the NEXT LayerNorm: the next block's norm1, or the encoder's final norm whose
            # bf16 output is image_embeds, the K / V source of the text encoder's cross-attention
            L.adapter_fwd_ln(A["h3"], nxt, self._segs(i, mode, self.Mi, False), self.Mi, g_, b_, 1e-6,
                             V["emb16"] if last else V["x16"], V["stf"] if last else V["blocks"][i + 1]["st1"],
                             z_save=A["zs"])
            h = nxt

    def _embed(self, T, ids, tts, nb, Lt, out_f32, out_b16, drop=None):
        L.text_embed(ids, tts, T["word"], T["pos"], T["typ"], T["lng"], T["lnb"], 1e-12, self.zero_h, out_f32, nb, Lt, Lt,
                     self.H)
        if drop is not None:                 # embeddings = self.dropout(LayerNorm(...))   xbert.py:216
            L.dropout(out_f32, drop, out_f32=out_f32, out_bf16=out_b16)
        else:
            L.cvt_f32_bf16(out_f32, out_b16)

    def _dense_resid(self, x16, lin, resid, out, tmp, drop):
        """out = dropout(x W^T + b) + resid: one GEMM with the residual epilogue, or -- with dropout between the dense
        layer and the residual add (xbert.py:360,440) -- the plain product followed by the fused mask + add."""
        if drop is None:
            L.gemm_bf16_nt(x16, lin["w"], L.EPI_RESID_F32, bias=lin["b"], resid=resid, out_f32=out)
        else:
            L.gemm_bf16_nt(x16, lin["w"], L.EPI_F32, bias=lin["b"], out_f32=tmp)
            L.dropout(tmp, drop, resid=resid, out_f32=out)

    def _attn_out(self, W, ctx, resid, M, t_out, st, y, y16, tmp=None, drop=None):
        """BertSelfOutput: LN(dropout(dense(ctx)) + input) (xbert.py:356-362)."""
        self._dense_resid(ctx, W["o"], resid, t_out, tmp, drop)
        L.layernorm_fwd(t_out, W["lng"], W["lnb"], 1e-12, M, self.H, y_bf16=y16, y_f32=y, stats=st)

    def _bert_fwd(self, T, S, m0: int, mode: str, M, nb, Sq, self_mask, causal, enc16, enc_rows, Skv, enc_mask,
                  pass_id=None, tower: int = 0):
        H = self.H
        h, h16 = S["h"], S["h16"]
        if T["cross_layers"]:         # K | V of every cross-attention layer of the tower in one product (see tower())
            L.gemm_bf16_nt(enc16, T["kv_all"]["w"], L.EPI_BF16, bias=T["kv_all"]["b"], out_bf16=S["kvc_all"])
        for i, W in enumerate(T["layers"]):
            A = S["layers"][i]
            dk = lambda kind: self._drop(pass_id, tower, i, kind)      # noqa: E731
            L.gemm_bf16_nt(h16, W["att"]["qkv"]["w"], L.EPI_BF16, bias=W["att"]["qkv"]["b"], out_bf16=A["qkv"])
            L.attn2_fwd(A["qkv"][:, :H], A["qkv"][:, H:2 * H], A["qkv"][:, 2 * H:], A["ctx"], A["lse"], nb, Sq, Sq,
                        self.heads, key_mask=self_mask, causal=causal, drop=dk("self_probs"))
            self._attn_out(W["att"], A["ctx"], h, M, A["t1"], A["st_a"], A["a"], A["a16"], S["td"], dk("self_out"))
            c, c16 = A["a"], A["a16"]
            if W["cross"] is not None:
                Wc = W["cross"]
                L.gemm_bf16_nt(A["a16"], Wc["q"]["w"], L.EPI_BF16, bias=Wc["q"]["b"], out_bf16=A["qc"])
                L.attn2_fwd(A["qc"], A["kvc"][:, :H], A["kvc"][:, H:], A["ctx2"], A["lse2"], nb, Sq, Skv, self.heads,
                            key_mask=enc_mask, q_rows=Sq, kv_rows=enc_rows, drop=dk("cross_probs"))
                self._attn_out(Wc, A["ctx2"], A["a"], M, A["t2"], A["st_c"], A["c"], A["c16"], S["td"], dk("cross_out"))
                c, c16 = A["c"], A["c16"]
            L.gemm_bf16_nt(c16, W["fc1"]["w"], L.EPI_GELU, bias=W["fc1"]["b"], out_bf16=S["f16"], out2_bf16=A["u"])
            # BertOutput with the adapter (xbert.py:438-445 -> adapter.py:97-116): s1 = dropout(dense) + inp; x = LN(s1);
            # y + inp = s1 + (x + A(x)) - x; out = LN(y + inp), same LayerNorm twice
            self._dense_resid(S["f16"], W["fc2"], c, A["s1"], S["td"], dk("out"))
            L.layernorm_fwd(A["s1"], W["lng"], W["lnb"], 1e-12, M, H, y_f32=A["x"], stats=A["st_x"])
            L.adapter_fwd(A["x"], S["tA"], self._segs(m0 + i, mode, M, False), M, z_save=A["zs"])
            L.axpby3(A["s1"], 1.0, S["tA"], 1.0, A["x"], -1.0, out_f32=A["s2"])
            L.layernorm_fwd(A["s2"], W["lng"], W["lnb"], 1e-12, M, H, y_f32=A["out"], y_bf16=A["out16"], stats=A["st_o"])
            h, h16 = A["out"], A["out16"]
        return h, h16

    def _forward(self, mode: str, pass_id=None):
        """ALBEF.forward(train=True) up to logits[:, :-1] (albef_model.py:69-145), activations kept in self.acts[mode].
        pass_id (0 / 1 / 2): the train_step pass whose dropout masks apply; None = no dropout."""
        self._vit_fwd(self.acts[mode], mode)
        return self._forward_text(mode, pass_id)

    def _forward_text(self, mode: str, pass_id=None):
        """Everything behind the image encoder: text encoder with cross-attention, answer decoder, LM head.  mode "both":
        the gated and the adapter_1 pass stacked (rows of the gated pass first) over the stacked image_embeds."""
        S = self.acts[mode]
        t = self.tx["both" if mode == "both" else "single"]
        emb16 = self.emb16_both if mode == "both" else S["vit"]["emb16"]
        Lq, La, H = self.Lq, self.La, self.H
        E, D = S["enc"], S["dec"]
        self._embed(self.enc, t["q_ids"], t["q_tt"], t["nq"], Lq, E["h"], E["h16"], self._drop(pass_id, 0, 0, "emb"))
        qs, _ = self._bert_fwd(self.enc, E, self.vd, mode, t["Mq"], t["nq"], Lq, t["qmask8"], False, emb16, self.Ni,
                               self.Ni, None, pass_id, 0)
        # one question's states for each of its k answers (albef_model.py:93-98)
        L.gather_rows(qs, t["rep_idx"], dst_bf16=S["enc_rep16"])
        self._embed(self.dec, t["a_ids"], t["a_tt"], t["na"], La, D["h"], D["h16"], self._drop(pass_id, 1, 0, "emb"))
        out, _ = self._bert_fwd(self.dec, D, self.vd + self.el, mode, t["Ma"], t["na"], La, t["amask8"], True, S["enc_rep16"], Lq,
                                Lq, t["qmask8_rep"], pass_id, 1)
        # BertOnlyMLMHead on the positions that predict a next token (logits[:, :-1])
        hd = self.head
        L.gather_rows(out, t["sel_idx"], dst_f32=S["hsel"], dst_bf16=S["hsel16"])
        L.gemm_bf16_nt(S["hsel16"], hd["t"]["w"], L.EPI_F32, bias=hd["t"]["b"], out_f32=S["tu"])
        L.gelu_fwd(S["tu"], S["tg"])
        L.layernorm_fwd(S["tg"], hd["lng"], hd["lnb"], 1e-12, t["R"], H, y_bf16=S["ty16"], stats=S["tst"])
        L.gemm_bf16_nt(S["ty16"], hd["w"], L.EPI_F32, bias=hd["b"], out_f32=S["logits"])
        return S["logits"]

    # ------------------------------------------------------------------------------------------ backward
    def _bert_bwd(self, T, S, m0: int, mode: str, M, nb, Sq, self_mask, causal, enc16, enc_rows, Skv, enc_mask, d_out,
                  d_enc, pass_id=None, tower: int = 0):
        """d_out: fp32 [M,768] gradient of the tower's output (consumed); d_enc: fp32 gradient of the encoder-side states
        (written: one product over all cross-attention layers' dK | dV); returns the gradient wrt the tower's embedding output
        (unused: embeddings frozen)."""
        H, g = self.H, self.gs[mode]
        cross0 = T["cross_layers"][0] if T["cross_layers"] else 0
        for i in range(len(T["layers"]) - 1, -1, -1):
            W, A = T["layers"][i], S["layers"][i]
            dk = lambda kind: self._drop(pass_id, tower, i, kind)      # noqa: E731

            def ln_bwd_to_operand(x, st, gam, dy, dres, out, kind):
                """LayerNorm backward whose fp32 result also becomes the bf16 dY operand of the dense layer ahead of it --
                through that layer's dropout mask when the pass has one (d dense = mask / (1 - p) . d out)."""
                d = dk(kind)
                if d is None:
                    L.layernorm_bwd_dx(x, st, gam, M, H, dy_f32=dy, dres=dres, out_f32=out, out_bf16=g["b1"][:M])
                else:
                    L.layernorm_bwd_dx(x, st, gam, M, H, dy_f32=dy, dres=dres, out_f32=out)
                    L.dropout(out, d, out_bf16=g["b1"][:M])
            ds2, dxf, dx, ds1 = g["d1"][:M], g["d2"][:M], g["d3"][:M], g["d4"][:M]
            L.layernorm_bwd_dx(A["s2"], A["st_o"], W["lng"], M, H, dy_f32=d_out, out_f32=ds2)
            L.adapter_bwd(None, ds2, dxf, self._segs(m0 + i, mode, M, True), M, z_out=g["z"], dz_out=g["dz"],
                          z_saved=A["zs"])
            self._wgrad(m0 + i, mode, A["x"], ds2, M)
            L.axpby3(dxf, 1.0, ds2, -1.0, out_f32=dx)                      # d x = Wd^T dz (the adapter's residual is r, not x)
            ln_bwd_to_operand(A["s1"], A["st_x"], W["lng"], dx, ds2, ds1, "out")
            L.gemm_bf16_nt(g["b1"][:M], W["fc2"]["wT"], L.EPI_MUL_DGELU, aux=A["u"], out_bf16=g["bI"][:M])
            dc = d_out                                                      # reuse: d_out is dead
            L.gemm_bf16_nt(g["bI"][:M], W["fc1"]["wT"], L.EPI_RESID_F32, resid=ds1, out_f32=dc)
            if W["cross"] is not None:
                Wc = W["cross"]
                dt2 = ds2
                ln_bwd_to_operand(A["t2"], A["st_c"], Wc["lng"], dc, None, dt2, "cross_out")
                L.gemm_bf16_nt(g["b1"][:M], Wc["o"]["wT"], L.EPI_BF16, out_bf16=g["b2"][:M])
                kv_rows, j = A["kvc"].shape[0], i - cross0
                dq, dkv = g["b1"][:M], g["bkv"][:kv_rows, j * 2 * H:(j + 1) * 2 * H]
                L.attn2_bwd(A["qc"], A["kvc"][:, :H], A["kvc"][:, H:], A["ctx2"], A["lse2"], g["b2"][:M], g["dsum"], dq,
                            dkv[:, :H], dkv[:, H:], nb, Sq, Skv, self.heads, key_mask=enc_mask, q_rows=Sq, kv_rows=enc_rows,
                            drop=dk("cross_probs"))
                da = dc
                L.gemm_bf16_nt(dq, Wc["q"]["wT"], L.EPI_RESID_F32, resid=dt2, out_f32=da)
            else:
                da = dc
            dt1 = ds2
            ln_bwd_to_operand(A["t1"], A["st_a"], W["att"]["lng"], da, None, dt1, "self_out")
            L.gemm_bf16_nt(g["b1"][:M], W["att"]["o"]["wT"], L.EPI_BF16, out_bf16=g["b2"][:M])
            dqkv = g["b3"][:M]
            L.attn2_bwd(A["qkv"][:, :H], A["qkv"][:, H:2 * H], A["qkv"][:, 2 * H:], A["ctx"], A["lse"], g["b2"][:M],
                        g["dsum"], dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:], nb, Sq, Sq, self.heads, key_mask=self_mask,
                        causal=causal, drop=dk("self_probs"))
            L.gemm_bf16_nt(dqkv, W["att"]["qkv"]["wT"], L.EPI_RESID_F32, resid=dt1, out_f32=d_out)
        if T["cross_layers"]:         # d(encoder-side states) = sum over the layers of [dK | dV]_l Wkv_l: one product, K = n 1536
            nc = len(T["cross_layers"])
            L.gemm_bf16_nt(g["bkv"][:enc16.shape[0], :nc * 2 * H], T["kv_all"]["wT"], L.EPI_F32, out_f32=d_enc)
        return d_out

    def _vit_bwd(self, S, mode: str, d_img):
        """d_img: fp32 [Mi,768] gradient wrt image_embeds (the final norm's output)."""
        H, g, vt, Mi = self.H, self.gs[mode], self.vit, self.Mi
        V = S["vit"]
        cur, oth = g["d1"][:Mi], g["d2"][:Mi]
        L.layernorm_bwd_dx(V["out"], V["stf"], vt["ng"], Mi, H, dy_f32=d_img, out_f32=cur)
        for i in range(self.vd - 1, -1, -1):
            W, A = vt["blocks"][i], V["blocks"][i]
            if i == 0:       # nothing trainable below the first adapter: weight gradients only
                L.adapter_bwd(None, cur, None, self._segs(0, mode, Mi, True), Mi, z_out=g["z"], dz_out=g["dz"],
                              z_saved=A["zs"])
                self._wgrad(0, mode, A["h3"], cur, Mi)
                break
            L.adapter_bwd(None, cur, oth, self._segs(i, mode, Mi, True), Mi, dx_bf16=g["b1"][:Mi], z_out=g["z"],
                          dz_out=g["dz"], z_saved=A["zs"])
            self._wgrad(i, mode, A["h3"], cur, Mi)
            L.gemm_bf16_nt(g["b1"][:Mi], W["fc2"]["wT"], L.EPI_MUL_G8 if A["u"].dtype == torch.uint8 else L.EPI_MUL_DGELU,
                           aux=A["u"], out_bf16=g["bI"][:Mi])
            L.gemm_bf16_nt(g["bI"][:Mi], W["fc1"]["wT"], L.EPI_BF16, out_bf16=g["b2"][:Mi])
            L.layernorm_bwd_dx(A["h2"], A["st2"], W["n2g"], Mi, H, dy_bf16=g["b2"][:Mi], dres=oth, out_f32=cur,
                               out_bf16=g["b1"][:Mi])
            L.gemm_bf16_nt(g["b1"][:Mi], W["proj"]["wT"], L.EPI_BF16, out_bf16=g["b2"][:Mi])
            dqkv = g["b3"][:Mi]
            L.attn2_bwd(A["qkv"][:, :H], A["qkv"][:, H:2 * H], A["qkv"][:, 2 * H:], A["ctx"], A["lse"], g["b2"][:Mi],
                        g["dsum"], dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:], self.B, self.Ni, self.Ni, self.heads)
            L.gemm_bf16_nt(dqkv, W["qkv"]["wT"], L.EPI_BF16, out_bf16=g["b2"][:Mi])
            L.layernorm_bwd_dx(A["h_in"], A["st1"], W["n1g"], Mi, H, dy_bf16=g["b2"][:Mi], dres=cur, out_f32=oth)
            cur, oth = oth, cur

    def _backward(self, mode: str, teacher_logits, pass_id=None):
        """L = (loss + kl) / 2 of the pass `mode` (task_trainer.py:300-302 / 320-323) -> gradients of its trainable adapter."""
        self._backward_text(mode, teacher_logits, pass_id)
        self._vit_bwd(self.acts[mode], mode, self.gs[mode]["d_img"])
        self._wgrad_reduce(mode)

    def _backward_text(self, mode: str, teacher_logits, pass_id=None):
        """Loss, LM head, decoder and text encoder backward of the pass `mode`; leaves d(image_embeds) in its scratch set.
        mode "both": the two passes stacked, each half's MKD teacher the OTHER half's logits (teacher_logits unused)."""
        S, g, hd, H = self.acts[mode], self.gs[mode], self.head, self.H
        t = self.tx["both" if mode == "both" else "single"]
        R, R1 = t["R"], self.R
        if mode == "both":       # L_0 = (loss_0 + KL(logits_0 || logits_1)) / 2 on rows [0, R), L_1 likewise on rows [R, 2R)
            lg = S["logits"]
            for half, own in ((0, "gating"), (1, "adapter_1")):
                L.lm_loss_fwd_bwd(lg[half * R1:(half + 1) * R1], lg[(1 - half) * R1:(2 - half) * R1], self.labels, self.row_w, self.V,
                                  3.0, 9.0 / self.N, g["dlogits"][half * R1:(half + 1) * R1], self.acts[own]["loss"],
                                  row_kl=self.row_kl, **self._scale_in(own))
        else:
            L.lm_loss_fwd_bwd(S["logits"], teacher_logits, self.labels, self.row_w, self.V, 3.0, 9.0 / self.N, g["dlogits"],
                              S["loss"], row_kl=self.row_kl, **self._scale_in(mode))
        # LM head backward (frozen): logits = LN(gelu(dense(h))) W_emb^T
        L.gemm_bf16_nt(g["dlogits"], hd["wT"], L.EPI_BF16, out_bf16=g["b1"][:R])
        L.layernorm_bwd_dx(S["tg"], S["tst"], hd["lng"], R, H, dy_bf16=g["b1"][:R], out_f32=g["d1"][:R])
        L.gelu_bwd(S["tu"], g["d1"][:R], g["d2"][:R])
        L.cvt_f32_bf16(g["d2"][:R], g["b1"][:R])
        L.gemm_bf16_nt(g["b1"][:R], hd["t"]["wT"], L.EPI_F32, out_f32=g["d1"][:R])
        L.gather_rows(g["d1"][:R], t["unsel_idx"], dst_f32=g["d_dec"])          # zero rows at the last position of each answer
        emb16 = self.emb16_both if mode == "both" else S["vit"]["emb16"]
        self._bert_bwd(self.dec, S["dec"], self.vd + self.el, mode, t["Ma"], t["na"], self.La, t["amask8"], True,
                       S["enc_rep16"], self.Lq, self.Lq, t["qmask8_rep"], g["d_dec"], g["d_rep"], pass_id, 1)
        # question states were repeated per answer: sum the answers of each question back (rows = [N, Lq * H])
        L.segment_sum_rows(g["d_rep"].view(t["na"], self.Lq * H), t["seg_off"], g["d_qs"].view(t["nq"], self.Lq * H))
        self._bert_bwd(self.enc, S["enc"], self.vd, mode, t["Mq"], t["nq"], self.Lq, t["qmask8"], False, emb16,
                       self.Ni, self.Ni, None, g["d_qs"], g["d_img"], pass_id, 0)

    # ------------------------------------------------------------------------------------------ train step
    @_bound
    def begin_local_update(self, steps_per_epoch: int, num_epochs: int = 15, warmup_ratio: float = 0.1,
                           opt_adapters: Sequence[int] = (0, 1), dropout_epoch: int = 0):
        """TaskTrainer.train prologue (task_trainer.py:36-59): teacher snapshot, fresh AdamW state and schedule.
        dropout_epoch: which local update this is, federation-wide (train.main passes round * n_clients + client index).  The
        masks are a function of (seed, site, step counter): the counter starts at dropout_epoch * 2^16, so every round and
        every client draws its own masks -- like the reference's nn.Dropout, which keeps consuming the global RNG across
        rounds and clients -- and a resumed run reproduces them without any saved RNG state."""
        self.copy_global_to_teacher()
        total = steps_per_epoch * num_epochs
        self.sched = dict(total=total, warmup=int(total * warmup_ratio))
        self.opt_adapters = tuple(opt_adapters)
        for grp in (self.ad[0], self.ad[1]):
            grp.m.zero_()
            grp.v.zero_()
            grp.g.zero_()
        self.ad[1].state.copy_(torch.tensor([0, 0], dtype=torch.int32))       # adapter_1 is stepped at tick 2b
        self.ad[0].state.copy_(torch.tensor([1, 0], dtype=torch.int32))       # adapter_0 at tick 2b + 1
        self.drop_ctr.copy_(torch.tensor([(int(dropout_epoch) << 16) & 0x7FFFFFFF, 0], dtype=torch.int32))
        # a fresh GradScaler per local update (the reference builds a fresh Accelerator per round: main.py:435)
        self.scaler_f.copy_(torch.tensor([self.loss_scale, 1.0 / self.loss_scale], dtype=torch.float32))
        self.scaler_i.zero_()
        self.ovf_flags.zero_()
        # a captured step stays valid across local updates as long as everything it froze into kernel arguments or into its
        # launch list is unchanged (all mutable state -- weights, moments, counters -- lives in device buffers)
        sig = (total, self.sched["warmup"], self.opt_adapters, self.lr, self.wd, self.eps, self.dropout, self.batch_text,
               self.operands, self.loss_scale, self.dynamic_scale, self.scale_growth_interval)
        if getattr(self, "_graph_sig", None) != sig:
            self.graph = None
            self._graph_sig = sig

    def _adamw(self, grp: FlatGroup):
        if not hasattr(grp, "_wdv"):
            grp._wdv = grp.seg_wd * self.wd
        L.adamw_flat(grp.p, grp.g, grp.m, grp.v, grp.seg_off, grp._wdv, grp.state, self.lr, self.sched["warmup"],
                     self.sched["total"], 0.9, 0.98, self.eps)

    @_bound
    def _step_kernels(self):
        # The two passes are independent up to the loss and from the loss down to their adapters, so they run side by
        # side on two streams (separate activation, scratch and weight-gradient workspaces per pass; inside a captured
        # step the fork / join become graph edges): the text towers' launches are small (25-token questions, 4-token
        # answers: 24-156 workgroups) and overlap each other, and the other pass's kernels fill the CUs that the last,
        # partial round of an image-encoder GEMM leaves idle (388 tiles on 256 CUs at N = 768).
        cur = torch.cuda.current_stream()
        if self.side is None:
            self.side = torch.cuda.Stream(device=self.dev)
        side = self.side
        drop = self.dropout > 0
        if self.batch_text and self.opt_adapters == (0, 1):
            # dropout = 0: the image encoders of the two passes side by side, then ONE text-side forward / backward over both
            # passes' rows, then the two image-encoder backward passes side by side
            self._vit_fwd(self.acts["gating"], "gating", "prefix")
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self._vit_fwd(self.acts["gating"], "gating", "suffix")
            self._vit_fwd(self.acts["adapter_1"], "adapter_1", "suffix")
            cur.wait_stream(side)
            self._forward_text("both")
            self._backward_text("both", None)
            d_img = self.gs["both"]["d_img"]
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self._vit_bwd(self.acts["gating"], "gating", d_img[:self.Mi])
                self._wgrad_reduce("gating")
            self._vit_bwd(self.acts["adapter_1"], "adapter_1", d_img[self.Mi:])
            self._wgrad_reduce("adapter_1")
            self._wgrad_reduce("both")
            cur.wait_stream(side)
            self._optimizer_tail(False)
            return
        # the image encoder ahead of block 0's adapter: once for both passes (frozen, adapter-free, no dropout)
        self._vit_fwd(self.acts["gating"], "gating", "prefix")
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self._vit_fwd(self.acts["gating"], "gating", "suffix")
            if drop:
                # P0 (task_trainer.py:283-287): no-grad gated forward under its own masks.  The image encoder has no dropout,
                # so its gated forward is shared with P2; the text towers run again for P2 below (other masks).
                self.logits_all.copy_(self._forward_text("gating", 0))
                logits_teacher = self.logits_all
            else:
                logits_teacher = self._forward_text("gating")     # P0 == P2 forward (task_trainer.py:283-287,311-315)
        self._vit_fwd(self.acts["adapter_1"], "adapter_1", "suffix")
        logits_1 = self._forward_text("adapter_1", 1 if drop else None)       # P1 (task_trainer.py:290-295)
        cur.wait_stream(side)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            logits_g = self._forward_text("gating", 2) if drop else logits_teacher
            if 0 in self.opt_adapters:
                self._backward("gating", logits_1, 2 if drop else None)   # L_0 = (loss_0 + KL(logits_0 || logits_1)) / 2
            else:        # adapter_0 left the optimizer (server flags after the first eval): only the loss values are needed
                L.lm_loss_fwd_bwd(logits_g, logits_1, self.labels, self.row_w, self.V, 3.0, 9.0 / self.N, None,
                                  self.acts["gating"]["loss"], row_kl=self.row_kl)
        self._backward("adapter_1", logits_teacher, 1 if drop else None)  # L_1 = (loss_1 + KL(logits_1 || logits_all)) / 2
        cur.wait_stream(side)
        self._optimizer_tail(drop)

    def _optimizer_tail(self, drop: bool):
        if self.dynamic_scale:
            # GradScaler's skips as device predicates (engine.ViltDatEngine._step_kernels; here without a head): adapter_1 stays on
            # flag A, adapter_0 on either flag; feddat_dat_step_finish ticks the counters by what was applied, updates the scale and
            # clears the flags
            fB, fA = self.ovf_flags[0:1], self.ovf_flags[1:2]

            def grp(a, skip):
                G = self.ad[a]
                if not hasattr(G, "_wdv"):
                    G._wdv = G.seg_wd * self.wd
                return L.adamw_group(G.p, G.g, G.m, G.v, G.seg_off, G._wdv, G.state, skip_if=skip)
            groups = ([grp(1, (fA,))] if 1 in self.opt_adapters else []) + ([grp(0, (fA, fB))] if 0 in self.opt_adapters else [])
            if groups:
                L.adamw_multi(groups, self.lr, self.sched["warmup"], self.sched["total"], 0.9, 0.98, self.eps)
            for a in (1, 0):
                if a in self.opt_adapters:
                    self.repack_adapter(a)
            L.dat_step_finish(self._no_head, self.ad[1].state, self.ad[0].state, self.ovf_flags, self.scaler_f, self.scaler_i,
                              2.0, 0.5, self.scale_growth_interval)
            if drop:
                L.step_tick(self.drop_ctr, 1, 0)
            return
        if 1 in self.opt_adapters:
            self._adamw(self.ad[1])
            self.repack_adapter(1)
        L.step_tick(self.ad[1].state, 2, 1)
        if 0 in self.opt_adapters:
            self._adamw(self.ad[0])
            self.repack_adapter(0)
        L.step_tick(self.ad[0].state, 2, 1)
        if drop:
            L.step_tick(self.drop_ctr, 1, 0)                 # the next train_step draws fresh masks (also under graph replay)

    @_bound
    def train_step(self, batch: Optional[Dict] = None, use_graph: bool = False):
        """One DAT + MKD step; returns the device buffer {loss_0, kl_0, L_0} of the P2 pass (the reference returns loss_0).
        use_graph: replay the ~3000 launches of a step as one hipGraph (the BERT towers' launches are tiny: eager mode is
        host-bound)."""
        if batch is not None:
            self.set_batch(batch)
        if not use_graph:
            self._step_kernels()
        else:
            if self.graph is None:
                self._capture()
            self.graph.replay()
        return self.acts["gating"]["loss"]

    @_bound
    def _capture(self):
        """Capture the whole step into one hipGraph (static buffers; schedule and Adam counters live on the device); the
        optimizer state is saved / restored around the warm-up + capture run so that capturing does not advance training."""
        groups = [self.ad[0], self.ad[1]]
        saved = [(g.p.clone(), g.m.clone(), g.v.clone(), g.state.clone()) for g in groups]
        saved_ctr = self.drop_ctr.clone()
        saved_scaler = (self.scaler_f.clone(), self.scaler_i.clone(), self.ovf_flags.clone())
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._step_kernels()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            self._step_kernels()
        torch.cuda.synchronize()
        for g, (p, m, v, st) in zip(groups, saved):
            g.p.copy_(p)
            g.m.copy_(m)
            g.v.copy_(v)
            g.state.copy_(st)
        self.drop_ctr.copy_(saved_ctr)
        self.scaler_f.copy_(saved_scaler[0])
        self.scaler_i.copy_(saved_scaler[1])
        self.ovf_flags.copy_(saved_scaler[2])
        for a in (0, 1):
            self.repack_adapter(a)
        torch.cuda.synchronize()
        self.graph = graph

    # ------------------------------------------------------------------------------------------ inference
    @_bound
    @torch.no_grad()
    def forward_train_logits(self, batch: Dict, mode: str):
        """-> (loss, logits [N, La-1, V]) of ALBEF.forward(train=True) in adapter mode `mode` ('gating' | 'adapter_k')."""
        self.set_batch(batch)
        key = "gating" if mode == "gating" else "adapter_1"          # activation set whose buffers the pass uses
        S = self.acts[key]
        self.acts[mode] = S
        try:
            logits = self._forward(mode)
        finally:
            if mode != key:
                del self.acts[mode]
        L.lm_loss_fwd_bwd(logits, None, self.labels, self.row_w, self.V, 3.0, 0.0, None, S["loss"])
        _, n, la = self._k             # the batch's own answer count / length inside the engine's frame
        return S["loss"][0].clone(), logits[:, :self.V].reshape(self.N, self.La - 1, self.V)[:n, :la - 1].clone()

    @_bound
    @torch.no_grad()
    def rank_answer(self, batch: Dict, answer_ids: torch.Tensor, answer_mask: torch.Tensor, k: int, mode: str = "gating"):
        """ALBEF.forward(train=False) -> rank_answer (albef_model.py:147-156,171-228; eval loop task_trainer.py:159-204):
        first-token shortlist of k candidates per question out of the answer list, re-ranked by sequence likelihood.
        The engine must have been built with n_answers = B * k and a_len = answer_ids.shape[1].
        Returns (topk_ids [B,k] int64, topk_probs [B,k]).  The two decoder passes, the first-token softmax over the
        vocabulary (feddat_softmax_gather_rows) and both top-k selections (feddat_topk_rows) run on the HIP kernels; what is left
        to torch is index bookkeeping (index_select / gather of the shortlisted answers)."""
        B, N, La, V = self.B, self.N, self.La, self.V
        if N != B * k or answer_ids.shape[1] != La:
            raise L.FeddatHipError("rank_answer: engine must be built with n_answers = B * k and a_len = answer length")
        answer_ids, answer_mask = answer_ids.to(self.dev), answer_mask.to(self.dev)
        key = "gating" if mode == "gating" else "adapter_1"
        S = self.acts[key]
        self.acts[mode] = S
        try:
            # pass 1: [BOS] only (later positions are padding; the causal mask keeps position 0 blind to them)
            start = torch.full((N, La), self.pad_id, dtype=torch.int64, device=self.dev)
            start[:, 0] = answer_ids[0, 0]
            m0 = torch.zeros(N, La, dtype=torch.int64, device=self.dev)
            m0[:, 0] = 1
            self.set_batch(dict(batch, answer_ids=start, answer_mask=m0, weights=torch.ones(N, device=self.dev), k=[k] * B))
            lg = self._forward(mode)          # [N (La - 1), Vp] fp32; position 0 of one slot per question = row b k (La - 1)
            prob_first = torch.empty(B, answer_ids.shape[0], device=self.dev)
            L.softmax_gather_rows(lg, B, k * (La - 1) * lg.stride(0), V, answer_ids[:, 1], prob_first)
            topk_probs, topk_ids = L.topk_rows(prob_first, k)
            # pass 2: the shortlisted answers, per-answer next-token loss
            ids = answer_ids.index_select(0, topk_ids.reshape(-1))
            atts = answer_mask.index_select(0, topk_ids.reshape(-1))
            self.set_batch(dict(batch, answer_ids=ids, answer_mask=atts, weights=torch.full((N,), float(B), device=self.dev),
                                k=[k] * B))
            lg = self._forward(mode)
            L.lm_loss_fwd_bwd(lg, None, self.labels, self.row_w, V, 3.0, 0.0, None, S["loss"])
            answer_loss = S["loss"][4:4 + 2 * self.R:2].view(N, La - 1).sum(1)       # row_terms[2 r] = ce of row r (weight 1)
        finally:
            if mode != key:
                del self.acts[mode]
        # softmax(log(topk_probs) - answer_loss) over the k candidates, sorted (albef_model.py:223-226)
        probs, rerank = L.topk_rows(topk_probs, k, minus=answer_loss.view(B, k).contiguous(), log_first=True, softmax=True)
        return torch.gather(topk_ids, 1, rerank), probs

    @_bound
    def image_embeds(self, mode_key: str = "gating"):
        """fp32 image_embeds of the last forward in that activation set (final-norm output recomputed from its input)."""
        V = self.acts[mode_key]["vit"]
        out = torch.empty(self.Mi, self.H, device=self.dev)
        L.layernorm_fwd(V["out"], self.vit["ng"], self.vit["nb"], 1e-6, self.Mi, self.H, y_f32=out)
        return out.view(self.B, self.Ni, self.H)

    # ------------------------------------------------------------------------------------------ state dict
    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {n: grp.view(n) for grp in self.ad for n in grp.names}

    @_bound
    def load_tensors(self, tensors: Dict[str, torch.Tensor]):
        sd = self.state_dict()
        touched = set()
        for n, v in tensors.items():
            sd[n].copy_(v.to(self.dev, torch.float32))
            touched.update(a for a in range(3) if f"adapter_{a}_" in n)
        for a in touched:
            self.repack_adapter(a)

    def assert_finite(self):
        """As ViltDatEngine.assert_finite -- the last line of defence: with the dynamic loss scale (default for operands='f16') an
        overflowed sub-step is skipped on the device and this never fires; with a static scale (dynamic_loss_scale=False) one host
        read-back of the trainable adapters per local update turns an overflow into an error that names the knob."""
        bad = self.nonfinite_groups()
        if bad:
            raise L.FeddatHipError(
                f"non-finite values in {', '.join(bad)} after the local update: with operands={self.operands!r} the backward carries a "
                f"{'dynamic' if self.dynamic_scale else 'static'} loss scale (initial value {self.loss_scale:g}); construct the engine "
                "with dynamic_loss_scale=True, a smaller power of two (loss_scale=...) or operands='bf16' (this engine's default)")

    def nonfinite_groups(self):
        """Names of the trainable groups holding an inf / NaN (one host read-back each); [] = all finite (train.main agrees on
        this across ranks before the FedAvg collective)."""
        return [f"adapter_{a}" for a in (0, 1) if not bool(torch.isfinite(self.ad[a].p).all())]

    def comm_flat(self) -> torch.Tensor:
        """The FedAvg payload: all adapter_1 tensors of the 30 modules back-to-back (2 236 320 floats = 8.95 MB)."""
        return self.ad[1].p
