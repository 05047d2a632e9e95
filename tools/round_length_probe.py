#!/usr/bin/env python
"""How does the update error grow with the length of a round?  N train_steps of the 12-layer model (B = 4, 384 x 384) on the
HIP engine and on the CPU oracle (which reproduces the reference to 8e-7); max / mean |ddW| over all trainable tensors at
checkpoints.  python tools/round_length_probe.py [steps=80]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import feddat_oracle as O  # noqa: E402  (tools/ may use the oracle as a checker)
from feddat_amd import engine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
torch.set_num_threads(32)
d = O.ViltDims(layers=12)
P = O.make_params(d, ["art"], bias_std=0.02)
P0 = {k: v.clone() for k, v in P.items()}
eng = engine.ViltDatEngine(P, ["art"], "cuda", batch=4, res=384, layers=12)
eng.begin_local_update("art", steps_per_epoch=steps)
client = O.DatClient(P, d, "art", lr=1e-4, steps_per_epoch=steps)
names = None
for s in range(steps):
    b = O.synthetic_batch(4, 384, 8000 + s)
    client.train_step(b)
    eng.train_step({k: v.to("cuda") for k, v in b.items()}, use_graph=True)
    if (s + 1) % 10 == 0:
        sd = eng.state_dict()
        if names is None:
            names = [k for k in sd if ("adapter_0" in k or "adapter_1" in k or "task_layer" in k) and k in P]
        wmax, wratio = 0.0, 0.0
        for k in names:
            dref, dgot = P[k] - P0[k], sd[k].cpu() - P0[k]
            if float(dref.abs().max()) == 0:
                continue
            e = (dgot - dref).abs()
            wmax = max(wmax, float(e.max()))
            wratio = max(wratio, float(e.mean()) / float(dref.abs().mean()))
        print(f"after {s + 1:3d} steps: worst max |ddW| {wmax:.2e}   worst mean ratio {wratio:.3f}", flush=True)
sd = eng.state_dict()
rows = []
for k in names:
    dref, dgot = P[k] - P0[k], sd[k].cpu() - P0[k]
    if float(dref.abs().max()) == 0:
        continue
    e = (dgot - dref).abs()
    rows.append((float(e.max()), float(e.mean()) / float(dref.abs().mean()), float((e > 5e-4).float().mean()), k, tuple(dref.shape)))
rows.sort(reverse=True)
print("worst tensors (max |ddW|, mean ratio, share of elements off by > 5e-4):")
for r in rows[:12]:
    print(f"  {r[0]:.2e}  {r[1]:.3f}  {r[2]:.2e}  {r[3]}  {r[4]}")
