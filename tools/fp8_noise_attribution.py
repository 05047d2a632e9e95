#!/usr/bin/env python
"""configs[4] (fp8 frozen products, B = 64): where does the update noise come from -- the e4m3 ACTIVATIONS of the forward products or
the e4m3 GRADIENT rows of the backward's dX products?  The reference's own 40-step round (tests/golden/g8b_round40_b64.npz) replayed
with the engine's attribution switches: everything fp8 (production), forward only, backward only, neither (= the bf16 engine)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import feddat_oracle as O  # noqa: E402
from tests.golden_util import load  # noqa: E402
from tests.test_round_b32_gpu import _table, _vs_golden  # noqa: E402
from feddat_amd import engine  # noqa: E402

DEV = "cuda"
g = load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"), "g8b_round40_b64.npz")
steps, B = int(g["steps"]), int(g["batch"])
d = O.ViltDims(layers=12)
keys = [k.split("::", 2)[2] for k in g if k.startswith("s40::dsamp::")]
for name, fwd, bwd in (("fp8 forward + backward (production)", True, True), ("fp8 forward only", True, False),
                       ("fp8 backward only", False, True), ("neither (bf16 operands)", False, False)):
    P = O.make_params(d, ["art"], bias_std=0.02)
    P0 = {k: v.clone() for k, v in P.items()}
    eng = engine.ViltDatEngine(P, ["art"], DEV, batch=B, res=384, layers=12, fp8=True)
    eng.fp8_forward, eng.fp8_backward = fwd, bwd
    eng.begin_local_update("art", steps_per_epoch=steps)
    snaps = {}
    for s in range(steps):
        eng.train_step({k: v.to(DEV) for k, v in O.synthetic_batch(B, 384, 8000 + s).items()}, use_graph=True)
        if s + 1 in (20, 40):
            sd = eng.state_dict()
            snaps[s + 1] = {k: (sd[k].cpu() - P0[k]) for k in keys}
    r = dict(g=g, keys=keys, snaps=snaps)
    for n in (20, 40):
        t = _table(_vs_golden(r, n))
        print(f"{name:38s} {n} steps | adapters: max |ddW| {t['adapters']['max']:.2e}, mean ratio {t['adapters']['ratio']:.4f}, norm "
              f"{t['adapters']['norm']:.4f} | head: max {t['head']['max']:.2e}, ratio {t['head']['ratio']:.4f}", flush=True)
    del eng
    torch.cuda.empty_cache()
