#!/usr/bin/env python
"""160- / 224-row tiles of the one-wave-per-SIMD GEMM kernel (RT = 5 / 7) against the 192 / 256-row choice (debug flag 1 << 27 = off):
the launches where the tile count fills the 256 CUs' rounds better."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddat_amd import lib as L  # noqa: E402

dev = "cuda:0"
SHAPES = [(18464, 768, 768, L.EPI_RESID_F32), (18464, 768, 3072, L.EPI_RESID_F32), (18464, 768, 3072, L.EPI_BF16),
          (18464, 768, 2304, L.EPI_BF16), (18464, 768, 768, L.EPI_BF16), (18464, 2304, 768, L.EPI_BF16),
          (5920, 2304, 768, L.EPI_BF16), (5920, 768, 3072, L.EPI_RESID_F32), (11840, 768, 768, L.EPI_RESID_F32),
          (11840, 2304, 768, L.EPI_BF16), (18464, 9216, 768, L.EPI_BF16), (18464, 768, 9216, L.EPI_F32)]
for M, N, K, epi in SHAPES:
    A = torch.randn(M, K, device=dev).bfloat16()
    B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    o32 = torch.empty(M, N, device=dev)
    resid = torch.randn(M, N, device=dev)
    bias = torch.randn(N, device=dev)
    outs = []
    for flags in (1 << 27, 0):
        L.set_debug_flags(flags)

        def run():
            if epi == L.EPI_RESID_F32:
                L.gemm_bf16_nt(A, B, epi, bias=bias, resid=resid, out_f32=o32)
            elif epi == L.EPI_F32:
                L.gemm_bf16_nt(A, B, epi, bias=bias, out_f32=o32)
            else:
                L.gemm_bf16_nt(A, B, epi, bias=bias, out_bf16=o)
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(40):
            run()
        e1.record()
        torch.cuda.synchronize()
        outs.append(e0.elapsed_time(e1) / 40 * 1e3)
    L.set_debug_flags(0)
    print(f"M={M} N={N} K={K} epi={epi}: without 160/224-row tiles {outs[0]:6.1f} us   with {outs[1]:6.1f} us   ratio {outs[1] / outs[0]:.3f}", flush=True)
