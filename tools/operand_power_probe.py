#!/usr/bin/env python
"""Why is the fp16-operand step ~2 % slower than the bf16 one when every kernel is the same code at the same MFMA rate?
(profiles/r05_operand_power_probe.txt: the GEMM launches are 3-4 % longer inside the replayed graph, equal when bracketed eagerly.)

Probe: the SAME fp16 library and kernels, fed operands whose low mantissa bits are zero -- frozen weights rounded to bf16
values before the engine converts them to fp16 -- next to the plain fp16 and bf16 engines, all as hipGraph replays, interleaved
on one box.  If the step time follows the number of live mantissa bits and not the opcode, the difference is switching
activity in the MFMA datapath (clocks under a power cap), not instruction issue.
    python tools/operand_power_probe.py [--steps 60] [--rounds 3]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    from feddat_amd import engine, vilt_spec
    dev = torch.device("cuda", 0)
    params = vilt_spec.random_init(12, ["c0"], seed=0)
    coarse = {k: (v.to(torch.bfloat16).float() if ("encoder.layer" in k and k.endswith("weight") and "adapter" not in k
                                                   and "layernorm" not in k) else v) for k, v in params.items()}
    batches = [vilt_spec.synthetic_batch(32, 384, 1234 + i, device=dev) for i in range(4)]
    engs = {"bf16": engine.ViltDatEngine(params, ["c0"], dev, batch=32, res=384, layers=12, operands="bf16"),
            "f16": engine.ViltDatEngine(params, ["c0"], dev, batch=32, res=384, layers=12, operands="f16"),
            "f16, weights with bf16's 7 mantissa bits": engine.ViltDatEngine(coarse, ["c0"], dev, batch=32, res=384, layers=12,
                                                                            operands="f16")}
    for e in engs.values():
        e.begin_local_update("c0", steps_per_epoch=400)
        for i in range(5):
            e.train_step(batches[i % 4], use_graph=True)
    res = {k: [] for k in engs}
    for _ in range(args.rounds):
        for k, e in engs.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                e.train_step(batches[i % 4], use_graph=True)
            torch.cuda.synchronize()
            res[k].append((time.perf_counter() - t0) / args.steps * 1e3)
    for k, v in res.items():
        print(f"{k:45s} ms/step " + " ".join(f"{x:.3f}" for x in v) + f"   (min {min(v):.3f})")


if __name__ == "__main__":
    main()
