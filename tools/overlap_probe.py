#!/usr/bin/env python
"""Probe: is there room to hide work in the serial tail of a train_step?  The step's graph is captured once as it is and
once with the embeddings + layer-0 body (0.2 ms of large kernels that depend on no trainable tensor -- a real
implementation would run them for the NEXT batch) repeated on a side stream that forks after the forward pass and joins
before the backward pass.  If the second graph is not slower, the tail can hide that work."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddat_amd import engine, lib as L, vilt_spec  # noqa: E402

dev = torch.device("cuda", 0)
params = vilt_spec.random_init(12, ["c0"], seed=0, device="cpu")


class Probe(engine.ViltDatEngine):
    side_work = False
    join_late = False
    cap16 = 0          # persistent GEMM grids of the side work capped at 16 x cap16 workgroups (0 = all CUs)

    def _step_kernels(self):
        if not self.side_work:
            return super()._step_kernels()
        B, task = self.B, self.task
        hp = self.head[task]
        self._forward_dual()
        cur = torch.cuda.current_stream()
        if not hasattr(self, "_side"):
            self._side = torch.cuda.Stream(device=self.dev)
        side = self._side
        side.wait_stream(cur)
        with torch.cuda.stream(side):              # stand-in for the next batch's embeddings + layer-0 body
            L.set_debug_flags(self.cap16 << 28)
            self._embed()
            l0 = self.l0
            self._layer_body(0, self.h0, self.R, B, l0["qkv"], l0["ctx"], l0["lse"], l0["h2"], l0["h3"], st1=self.st0,
                             st2=self.st0, mask=self.key_mask2[:B])
            L.set_debug_flags(0)
        pooled_g, pooled_s = self.pooled[:B], self.pooled[B:]
        logits_both = self._head_fwd(self.pooled[:2 * B], "both", task)
        logits_all, logits_1 = logits_both[:B], logits_both[B:]
        L.dat_loss_fwd_bwd(logits_1, logits_all, self.inp["target"], self.dlogits, self.loss_buf["p1"])
        self._head_bwd(pooled_s, "p1", task, self.dpooled[B:])
        self._adamw(hp)
        L.step_tick(hp.state, 1, 1)
        logits_0 = self._head_fwd(pooled_g, "p2", task)
        L.dat_loss_fwd_bwd(logits_0, logits_1, self.inp["target"], self.dlogits, self.loss_buf["p2"])
        self._head_bwd(pooled_g, "p2", task, self.dpooled[:B])
        if not self.join_late:
            cur.wait_stream(side)
        self._backward_dual()
        if self.join_late:
            cur.wait_stream(side)
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


def run(side, late, cap16=0):
    eng = Probe(params, ["c0"], dev, batch=32, res=384, layers=12)
    eng.side_work, eng.join_late, eng.cap16 = side, late, cap16
    eng.begin_local_update("c0", steps_per_epoch=400)
    b = vilt_spec.synthetic_batch(32, 384, 7, device=dev)
    for _ in range(5):
        eng.train_step(b, use_graph=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        eng.train_step(None, use_graph=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 100 * 1e3


for name, side, late, cap in (("as built", False, False, 0),
                              ("+ layer-0 body on a side stream, joined before the backward", True, False, 0),
                              ("+ the same, its GEMM grids capped at 224 workgroups", True, False, 14),
                              ("+ the same, capped at 192", True, False, 12),
                              ("+ the same, capped at 128", True, False, 8),
                              ("+ the same, capped at 64", True, False, 4),
                              ("+ layer-0 body on a side stream, joined after the backward", True, True, 0),
                              ("+ the same, capped at 128", True, True, 8),
                              ("as built", False, False, 0)):
    print(f"{name:70s} {run(side, late, cap):.3f} ms/step", flush=True)
