#!/usr/bin/env python
"""Hash of the trained state (adapter_0, adapter_1, head) of the full ViLT-B/32 engine after 12 hipGraph-replayed train_steps at
configs[1]'s size, default settings.  Two builds with the same hash run the same arithmetic: at 80 steps the AdamW trajectory is
chaotic enough that a last-bit change moves the round-length parity draws ("draws"), so a kernel change meant
to be arithmetic-neutral is checked with this before the 80-step tests are trusted.  python tools/state_hash.py [repo root]
round 6 HEAD (= round 5's arithmetic): cb900273526616cd"""
import sys, os, hashlib, torch
root = sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from feddat_amd import engine, vilt_spec
dev = torch.device("cuda", 0)
params = vilt_spec.random_init(12, ["c0"], seed=0)
batches = [vilt_spec.synthetic_batch(32, 384, 1234 + i, device=dev) for i in range(4)]
e = engine.ViltDatEngine(params, ["c0"], dev, batch=32, res=384, layers=12)
if hasattr(e, "top_q_cls"):
    e.top_q_cls = False
e.begin_local_update("c0", steps_per_epoch=80)
for i in range(12):
    e.train_step(batches[i % 4], use_graph=True)
torch.cuda.synchronize()
h = hashlib.sha256()
for a in (0, 1):
    h.update(e.ad[a].p.cpu().numpy().tobytes())
h.update(e.head["c0"].p.cpu().numpy().tobytes())
print(os.path.basename(root), h.hexdigest()[:16], float(e.ad[1].p.double().abs().sum()), float(e.loss_buf["p2"][0]))
