#!/usr/bin/env python
"""K4 micro-benchmark at the step's size (T = 11 840 rows: gated | adapter_1 segments): back-to-back (MALL-warm), cold
(L2 + MALL thrashed) and right-behind-a-GEMM timings, optionally with the resident blocks per CU capped (extra dynamic
LDS through feddat_set_debug_flags bits 16..23)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from feddat_amd import lib as L
L.use_ablation_build()      # timing-only probes live in libfeddat_hip_ablate.so (python -m feddat_amd.build --ablate)
dev = 'cuda'
T = 11840; R = T // 2
def mk():
    wd = torch.randn(48, 768, device=dev) * 0.02; wu = torch.randn(768, 48, device=dev) * 0.02
    p = dict(wd=torch.empty(48, 768, dtype=torch.bfloat16, device=dev), wdT=torch.empty(768, 48, dtype=torch.bfloat16, device=dev),
             wu=torch.empty(768, 48, dtype=torch.bfloat16, device=dev), wuT=torch.empty(48, 768, dtype=torch.bfloat16, device=dev),
             bd=torch.zeros(48, device=dev), bu=torch.zeros(768, device=dev))
    L.adapter_pack(wd, wu, p["wd"], p["wdT"], p["wu"], p["wuT"])
    return p
a0, a1, a2 = mk(), mk(), mk()
x = torch.randn(T, 768, device=dev); dy = torch.randn(T, 768, device=dev)
out = torch.empty_like(x); dx = torch.empty_like(x); dx16 = torch.empty(T, 768, dtype=torch.bfloat16, device=dev)
y16 = torch.empty(T, 768, dtype=torch.bfloat16, device=dev); st = torch.empty(T, 2, device=dev)
gam = torch.ones(768, device=dev); bet = torch.zeros(768, device=dev)
z = torch.empty(T, 48, device=dev); dz = torch.empty(T, 48, device=dev); zs = torch.zeros(T, 2, 48, device=dev)
segs = [dict(row_begin=0, row_end=R, adapters=[dict(a0, scale=0.5), dict(a2, scale=0.5)], train_slot=0),
        dict(row_begin=R, row_end=T, adapters=[dict(a1, scale=1.0)], train_slot=0)]
sa = L.make_segs(segs)
big = torch.empty(512 * 1024 * 1024 // 4, device=dev)
A = torch.randn(11840, 3072, device=dev).to(torch.bfloat16); Bw = (torch.randn(768, 3072, device=dev) * 0.02).to(torch.bfloat16)
o32 = torch.empty(11840, 768, device=dev); res = torch.randn(11840, 768, device=dev); bias = torch.zeros(768, device=dev)
def warm(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
def ctx(fn, pre, n=8):
    ts = []
    for _ in range(n):
        pre()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts)[len(ts) // 2]
def gemm_pre():
    for _ in range(3): L.gemm_bf16_nt(A, Bw, 1, bias=bias, resid=res, out_f32=o32)
ops = {
    "fwd_ln": (lambda xx: L.adapter_fwd_ln(xx, out, sa, T, gam, bet, 1e-12, y16, st, z_save=zs), (2 * 4 + 2) * T * 768),
    "fwd": (lambda xx: L.adapter_fwd(xx, out, sa, T), 2 * 4 * T * 768),
    "bwd": (lambda xx: L.adapter_bwd(xx, dy, dx, sa, T, dx_bf16=dx16, z_out=z, dz_out=dz), (3 * 4 + 2) * T * 768),
    "bwd_zs": (lambda xx: L.adapter_bwd(None, xx, dx, sa, T, dx_bf16=dx16, z_out=z, dz_out=dz, z_saved=zs), (2 * 4 + 2) * T * 768),
}
if os.environ.get("QUICK"):       # a handful of dispatches per op, for counter collection
    for name, (fn, nbytes) in ops.items():
        for _ in range(6):
            fn(x)
    torch.cuda.synchronize()
    sys.exit(0)
for extra in [int(a) for a in (sys.argv[1:] or ["0"])]:
    L.set_debug_flags(extra << 16)
    for name, (fn, nbytes) in ops.items():
        w = warm(lambda: fn(x))
        c = ctx(lambda: fn(x), lambda: big.add_(1.0))
        g = ctx(lambda: fn(o32), gemm_pre)
        print(f"extra_lds={extra:3d}K {name:7s} warm {w:6.1f} us ({nbytes / w / 1e6:.2f} TB/s)  cold {c:6.1f} us ({nbytes / c / 1e6:.2f})  "
              f"behind GEMM (x = its output) {g:6.1f} us ({nbytes / g / 1e6:.2f})", flush=True)
L.set_debug_flags(0)
