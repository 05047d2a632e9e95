#!/usr/bin/env python
"""Probe: do two half-batch ViLT steps on two streams beat one full-batch step?  Two independent engines (B = 16 each,
own weights / activations / graphs) replayed concurrently on two streams against one engine at B = 32.  If the
MFMA-bound GEMMs of one half overlap the HBM-bound LayerNorm / adapter / attention kernels of the other, the pair is
faster than the sum."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddat_amd import engine, vilt_spec  # noqa: E402

dev = torch.device("cuda", 0)
res = 384
params = vilt_spec.random_init(12, ["c0"], seed=0, device="cpu")


def make(B):
    eng = engine.ViltDatEngine(params, ["c0"], dev, batch=B, res=res, layers=12)
    eng.begin_local_update("c0", steps_per_epoch=200)
    b = vilt_spec.synthetic_batch(B, res, 7, device=dev)
    for _ in range(3):
        eng.train_step(b, use_graph=True)
    torch.cuda.synchronize()
    return eng, b


def timeit(fn, n=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


full, bf = make(32)
t_full = timeit(lambda: full.train_step(None, use_graph=True))
print(f"one engine  B=32          : {t_full:.3f} ms/step  {32 / t_full * 1e3:.0f} samples/s")
h1, b1 = make(16)
t_half = timeit(lambda: h1.train_step(None, use_graph=True))
print(f"one engine  B=16          : {t_half:.3f} ms/step  {16 / t_half * 1e3:.0f} samples/s")
h2, b2 = make(16)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def pair():
    with torch.cuda.stream(s1):
        h1.train_step(None, use_graph=True)
    with torch.cuda.stream(s2):
        h2.train_step(None, use_graph=True)


t_pair = timeit(pair)
print(f"two engines B=16 + 16, two streams: {t_pair:.3f} ms per pair  {32 / t_pair * 1e3:.0f} samples/s")
