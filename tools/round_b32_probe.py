#!/usr/bin/env python
"""80-step round at B = 32 (configs[1] / configs[2]) on the engine, against the reference's own run
(tests/golden/g8b_round80_b32.npz), for engine configurations given on the command line:
    python tools/round_b32_probe.py bf16 f16 f16:codes=0 f16:scale=1024 bf16:codes=0
Prints the worst adapter / head tensor's max |ddW|, mean ratio and update-norm error after 20 / 40 / 60 / 80 steps and the
step time of the hipGraph replay -- the table of DESIGN.md section 5, ~10 s per configuration (no live oracle)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import feddat_oracle as O  # noqa: E402  (tools/ are measurement scripts, not product code)
from tests.test_round_b32_gpu import SNAPS, _table, _vs_golden  # noqa: E402


def run(cfg, g, batches):
    from feddat_amd import engine
    name, _, opts = cfg.partition(":")
    kw = dict(operands=name)
    for o in filter(None, opts.split(",")):
        k, v = o.split("=")
        if k == "codes":
            kw["gelu_codes"] = bool(int(v))
        elif k == "scale":
            kw["loss_scale"] = float(v)
    d = O.ViltDims(layers=12)
    P = O.make_params(d, ["art"], bias_std=0.02)
    P0 = {k: v.clone() for k, v in P.items()}
    eng = engine.ViltDatEngine(P, ["art"], "cuda", batch=32, res=384, layers=12, **kw)
    eng.begin_local_update("art", steps_per_epoch=80)
    keys = [k.split("::", 2)[2] for k in g if k.startswith("s80::dsamp::")]
    snaps, losses = {}, []
    for s, b in enumerate(batches):
        losses.append(eng.train_step(b, use_graph=True)[0].clone())
        if s + 1 in SNAPS:
            sd = eng.state_dict()
            snaps[s + 1] = {k: (sd[k].cpu() - P0[k]) for k in keys}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches[:40]:
        eng.train_step(b, use_graph=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 40 * 1e3
    losses = np.array([float(x) for x in losses])
    rel = np.abs(losses - g["losses"]) / np.maximum(g["losses"], 1.0)
    r = dict(g=g, keys=keys, snaps=snaps)
    line = f"{cfg:22s} {ms:6.3f} ms/step (incl. set_batch) | loss rel {rel.max():.1e}"
    for n in SNAPS:
        t = _table(_vs_golden(r, n))
        line += (f"\n    {n:2d}: adapters max {t['adapters']['max']:.2e} ratio {t['adapters']['ratio']:.4f} norm "
                 f"{t['adapters']['norm']:.4f} | head max {t['head']['max']:.2e} ratio {t['head']['ratio']:.4f}")
    print(line, flush=True)
    del eng
    torch.cuda.empty_cache()


if __name__ == "__main__":
    from tests.golden_util import load
    g = load(os.path.join(ROOT, "tests", "golden"), "g8b_round80_b32.npz")
    batches = [{k: v.to("cuda") for k, v in O.synthetic_batch(32, 384, 8000 + s).items()} for s in range(80)]
    for cfg in sys.argv[1:] or ["bf16", "f16"]:
        run(cfg, g, batches)
