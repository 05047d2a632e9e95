"""Same-box A/B of the ALBEF engine's stack_text switch (text towers of the two passes as one 2x-row launch list vs two
streams): ms per hipGraph-replayed train_step at B = 32."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddat_amd import albef_engine, albef_spec
B = 32
params = albef_spec.random_init(seed=0, image=384)
batches = [albef_spec.synthetic_batch(B, 1234 + i, image=384, device="cuda") for i in range(2)]
res = {}
from feddat_amd import lib as L
# round 5: 1600-row products on the small-tile kernel (36 tiles of the persistent kernels = 36 CUs); debug flag 1 = the old routing
for name, bt, flags in (("stacked", True, 0), ("separate", False, 0), ("stacked, persistent kernels for M = 1600 (r04 routing)", True, 1),
                        ("stacked", True, 0), ("separate", False, 0)):
    L.set_debug_flags(flags)
    eng = albef_engine.AlbefDatEngine(params, "cuda", batch=B, n_answers=B, image=384, stack_text=bt)
    eng.begin_local_update(steps_per_epoch=100)
    for i in range(5):
        eng.train_step(batches[i % 2], use_graph=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(30):
        eng.train_step(batches[i % 2], use_graph=True)
    torch.cuda.synchronize()
    print(name, f"{(time.perf_counter() - t) / 30 * 1e3:.3f} ms/step", flush=True)
    del eng
L.set_debug_flags(0)
