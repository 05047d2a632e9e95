#!/usr/bin/env python
"""Where does the time of the weight-stationary adapter kernels go?  Timing-only ablations through
feddat_set_debug_flags bits 24..26 (1 = no global stores, 2 = no compute, 4 = no DMA beyond the first tile) at the step's
size (T = 11 840: gated | adapter_1 segments).  Results under a flag are wrong by construction."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from feddat_amd import lib as L
L.use_ablation_build()      # timing-only probes live in libfeddat_hip_ablate.so (python -m feddat_amd.build --ablate)
dev = "cuda"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 11840
R = T // 2


def mk():
    wd = torch.randn(48, 768, device=dev) * 0.02
    wu = torch.randn(768, 48, device=dev) * 0.02
    p = dict(wd=torch.empty(48, 768, dtype=torch.bfloat16, device=dev), wdT=torch.empty(768, 48, dtype=torch.bfloat16, device=dev),
             wu=torch.empty(768, 48, dtype=torch.bfloat16, device=dev), wuT=torch.empty(48, 768, dtype=torch.bfloat16, device=dev),
             bd=torch.zeros(48, device=dev), bu=torch.zeros(768, device=dev))
    L.adapter_pack(wd, wu, p["wd"], p["wdT"], p["wu"], p["wuT"])
    return p


a0, a1, a2 = mk(), mk(), mk()
x = torch.randn(T, 768, device=dev)
dy = torch.randn(T, 768, device=dev)
out = torch.empty_like(x)
dx = torch.empty_like(x)
dx16 = torch.empty(T, 768, dtype=torch.bfloat16, device=dev)
y16 = torch.empty(T, 768, dtype=torch.bfloat16, device=dev)
st = torch.empty(T, 2, device=dev)
gam, bet = torch.ones(768, device=dev), torch.zeros(768, device=dev)
z, dz, zs = torch.empty(T, 48, device=dev), torch.empty(T, 48, device=dev), torch.zeros(T, 2, 48, device=dev)
sa = L.make_segs([dict(row_begin=0, row_end=R, adapters=[dict(a0, scale=0.5), dict(a2, scale=0.5)], train_slot=0),
                  dict(row_begin=R, row_end=T, adapters=[dict(a1, scale=1.0)], train_slot=0)])
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


ops = {"fwd_ln": lambda: L.adapter_fwd_ln(x, out, sa, T, gam, bet, 1e-12, y16, st, z_save=zs),
       "bwd_zs": lambda: L.adapter_bwd(None, dy, dx, sa, T, dx_bf16=dx16, z_out=z, dz_out=dz, z_saved=zs)}
names = {0: "as built", 1: "no stores", 2: "no compute", 4: "no DMA (first tile only)", 3: "no stores, no compute",
         5: "no stores, no DMA", 6: "no compute, no DMA", 7: "nothing (launch + prologue + barriers)"}
for flag in (0, 1, 2, 4, 3, 5, 6, 7):
    L.set_debug_flags(flag << 24)
    print(f"{names[flag]:40s} " + "   ".join(f"{k}: warm {warm(f):5.1f} us cold {cold(f):5.1f} us" for k, f in ops.items()), flush=True)
L.set_debug_flags(0)
