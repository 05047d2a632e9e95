#!/usr/bin/env python
"""Would a DEFERRED epilogue pay?  For the step's GEMM shapes and epilogues: the production launch, the k-loop alone
(debug flag 8: no epilogue) and the timing probe of the v3 kernel (flag 512: no epilogue at the tile boundary; every
k-tile issues its 1/12 share of the conversions, activation math, aux / residual loads and -- as inline asm, invisible to
hipcc's vmcnt bookkeeping -- 8-byte row-per-lane stores between the MFMAs of its second half).  Results under a flag are
wrong by construction; back-to-back launches, operands MALL-warm."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from feddat_amd import lib as L
L.use_ablation_build()      # timing-only probes live in libfeddat_hip_ablate.so (python -m feddat_amd.build --ablate)
dev = "cuda"


def t(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


M = int(sys.argv[1]) if len(sys.argv) > 1 else 11840
names = {0: "bf16", 1: "resid_f32", 2: "gelu+u", 3: "mul_dgelu"}
for (N, K, epi) in [(2304, 768, 0), (768, 768, 1), (3072, 768, 2), (768, 3072, 1), (3072, 768, 3), (768, 3072, 0), (768, 768, 0),
                    (768, 2304, 0)]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    kw = {}
    if epi in (0, 2, 3):
        kw["out_bf16"] = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    if epi == 2:
        kw["out2_bf16"] = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    if epi == 3:
        kw["aux"] = torch.randn(M, N, device=dev).to(torch.bfloat16)
    if epi == 1:
        kw["resid"] = torch.randn(M, N, device=dev)
        kw["out_f32"] = torch.empty(M, N, device=dev)
    if epi != 3:
        kw["bias"] = bias
    f = 2.0 * M * N * K
    res = {}
    for tag, flag in (("production", 0), ("v3 forced", 2 | 32), ("k-loop only (v3)", 2 | 32 | 8), ("deferred probe", 512)):
        L.set_debug_flags(flag)
        res[tag] = t(lambda: L.gemm_bf16_nt(A, W, epi, **kw))
    L.set_debug_flags(0)
    print(f"({M},{N},{K}) {names[epi]:10s} " + "   ".join(f"{k}: {v:6.1f} us ({f / v / 1e6:5.0f} TF/s)" for k, v in res.items()), flush=True)
