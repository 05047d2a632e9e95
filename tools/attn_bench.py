#!/usr/bin/env python
"""ViLT attention (K2) at configs[1]'s shape: 64 sample-passes x 12 heads x 185 tokens, forward and backward, 20 launches
inside one hipGraph.  REPS=1 for PMC passes (tools/pmc_attn.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddat_amd import lib as L  # noqa: E402

dev = "cuda:0"
reps = int(os.environ.get("REPS", "20"))
B, S, heads = 64, 185, 12
H = heads * 64
qkv = torch.randn(B * S, 3 * H, device=dev).bfloat16()
ctx = torch.zeros(B * S, H, dtype=torch.bfloat16, device=dev)
lse = torch.zeros(B, heads, S, device=dev)
dctx = torch.randn(B * S, H, device=dev).bfloat16()
dqkv = torch.zeros_like(qkv)
km = torch.ones(B, S, dtype=torch.uint8, device=dev)


def fwd():
    L.attn_fwd(qkv, ctx, lse, B, S, heads, key_mask=km)


def bwd():
    L.attn_bwd(qkv, ctx, lse, dctx, dqkv, B, S, heads, key_mask=km)


for name, fn, units in (("fwd", fwd, 2), ("bwd", bwd, 5)):
    fn()
    torch.cuda.synchronize()
    if reps == 1:
        continue
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = units * 2 * B * heads * S * S * 64
    print(f"{name} {us:8.1f} us   {fl / us / 1e6:6.1f} TF/s ({units} matmul units)")
