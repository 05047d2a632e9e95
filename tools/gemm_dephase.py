#!/usr/bin/env python
"""Timing probe (ablation build): do the persistent GEMMs gain when their blocks run OUT OF PHASE, i.e. when the CUs'
epilogues (HBM write bursts) stop coinciding?  Debug bits 24..26 = q: block b starts ((b / 8) % 4) x q x 2 us late."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddat_amd import lib as L  # noqa: E402
L.use_ablation_build()
dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 11840
SHAPES = [(3072, 768, L.EPI_GELU_G8), (3072, 768, L.EPI_GELU), (768, 3072, L.EPI_RESID_F32), (2304, 768, L.EPI_BF16),
          (768, 768, L.EPI_RESID_F32), (768, 3072, L.EPI_BF16), (3072, 768, L.EPI_MUL_G8)]
for N, K, epi in SHAPES:
    A = torch.randn(M, K, device=dev).bfloat16()
    B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    o2 = torch.empty(M, N, dtype=torch.uint8 if epi == L.EPI_GELU_G8 else torch.bfloat16, device=dev)
    aux8 = torch.randint(0, 255, (M, N), dtype=torch.uint8, device=dev)
    o32 = torch.empty(M, N, device=dev)
    resid = torch.randn(M, N, device=dev)
    bias = torch.randn(N, device=dev)

    def run():
        if epi in (L.EPI_GELU_G8, L.EPI_GELU):
            L.gemm_bf16_nt(A, B, epi, bias=bias, out_bf16=o, out2_bf16=o2)
        elif epi == L.EPI_RESID_F32:
            L.gemm_bf16_nt(A, B, epi, bias=bias, resid=resid, out_f32=o32)
        elif epi == L.EPI_MUL_G8:
            L.gemm_bf16_nt(A, B, epi, aux=aux8, out_bf16=o)
        else:
            L.gemm_bf16_nt(A, B, epi, bias=bias, out_bf16=o)
    row = []
    for q in (0, 1, 2, 3, 4, 6):
        L.set_debug_flags(q << 24)
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(40):
            run()
        e1.record()
        torch.cuda.synchronize()
        row.append(f"q={q}: {e0.elapsed_time(e1) / 40 * 1e3:6.1f} us")
    L.set_debug_flags(8)
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(40):
        run()
    e1.record()
    torch.cuda.synchronize()
    L.set_debug_flags(0)
    print(f"M={M} N={N} K={K} epi={epi}: " + "  ".join(row) + f"  | no epilogue: {e0.elapsed_time(e1) / 40 * 1e3:6.1f} us", flush=True)
