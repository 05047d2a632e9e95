#!/usr/bin/env python
"""Where does the time of the fused ViLT attention backward go?  Timing-only ablations (debug flags bits 20..22: 1 = no
global stores, 2 = no phase-A arithmetic, 4 = no phase B) at configs[1]'s shape: 64 sample-passes x 12 heads x 185 tokens."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from feddat_amd import lib as L
L.use_ablation_build()      # timing-only probes live in libfeddat_hip_ablate.so (python -m feddat_amd.build --ablate)
dev = "cuda:0"
B, S, heads = 64, 185, 12
H = heads * 64
qkv = torch.randn(B * S, 3 * H, device=dev).bfloat16()
ctx = torch.zeros(B * S, H, dtype=torch.bfloat16, device=dev)
lse = torch.zeros(B, heads, S, device=dev)
dctx = torch.randn(B * S, H, device=dev).bfloat16()
dqkv = torch.zeros_like(qkv)
km = torch.ones(B, S, dtype=torch.uint8, device=dev)
L.attn_fwd(qkv, ctx, lse, B, S, heads, key_mask=km)
big = torch.empty(512 * 1024 * 1024 // 4, device=dev)


def warm(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def cold(fn, n=8):
    ts = []
    for _ in range(n):
        big.add_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts)[len(ts) // 2]


bwd = lambda: L.attn_bwd(qkv, ctx, lse, dctx, dqkv, B, S, heads, key_mask=km)
names = {0: "as built", 1: "no stores", 2: "no phase-A arithmetic", 4: "no phase B", 6: "loads + barriers + dK/dV stores only",
         7: "loads + barriers only"}
for flag in (0, 1, 2, 4, 6, 7):
    L.set_debug_flags(flag << 20)
    print(f"{names[flag]:42s} warm {warm(bwd):6.1f} us   cold {cold(bwd):6.1f} us", flush=True)
L.set_debug_flags(8 << 20)
print(f"{'one block per pair (r02 launch, bit 23)':42s} warm {warm(bwd):6.1f} us   cold {cold(bwd):6.1f} us", flush=True)
L.set_debug_flags(0)
