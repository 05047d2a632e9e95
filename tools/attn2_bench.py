#!/usr/bin/env python
"""attn2 forward / backward at the ViT-B/16 shape of the ALBEF step (B x 12 heads x 577 tokens) and at the cross-attention
shape, timed as 20 launches inside one hipGraph.  REPS=1 for PMC passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddat_amd import lib as L  # noqa: E402

dev = "cuda:0"
reps = int(os.environ.get("REPS", "20"))
for B, Sq, Skv, heads in [(32, 577, 577, 12), (32, 25, 577, 12)]:
    H = heads * 64
    qkv = torch.randn(B * Skv, 3 * H, device=dev).bfloat16()
    q = qkv[:B * Sq, :H] if Sq == Skv else torch.randn(B * Sq, H, device=dev).bfloat16()
    k, v = qkv[:, H:2 * H], qkv[:, 2 * H:]
    do = torch.randn(B * Sq, H, device=dev).bfloat16()
    ctx = torch.zeros(B * Sq, H, dtype=torch.bfloat16, device=dev)
    lse = torch.zeros(B, heads, Sq, device=dev)
    ws = torch.empty(B, heads, Sq, device=dev)
    dq = torch.zeros(B * Sq, H, dtype=torch.bfloat16, device=dev)
    dkv = torch.zeros(B * Skv, 2 * H, dtype=torch.bfloat16, device=dev)

    def fwd():
        L.attn2_fwd(q, k, v, ctx, lse, B, Sq, Skv, heads)

    def bwd():
        L.attn2_bwd(q, k, v, ctx, lse, do, ws, dq, dkv[:, :H], dkv[:, H:], B, Sq, Skv, heads)

    for name, fn, units in (("fwd", fwd, 2), ("bwd", bwd, 7)):
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
        fl = units * 2 * B * heads * Sq * Skv * 64
        print(f"B={B} Sq={Sq} Skv={Skv}: {name} {us:8.1f} us   {fl / us / 1e6:6.1f} TF/s (executed units: {units})")
