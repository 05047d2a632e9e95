#!/usr/bin/env python
"""Few-row products (ALBEF text / answer streams): the small-tile ring kernel against the 128 x 128 kernel (debug flag 128)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddat_amd import lib as L  # noqa: E402

dev = "cuda:0"
shapes = [(800, 768, 768), (800, 2304, 768), (800, 3072, 768), (800, 768, 3072), (800, 768, 2304), (1600, 768, 768),
          (1600, 3072, 768), (1600, 768, 3072), (128, 768, 768), (128, 3072, 768), (128, 768, 3072), (96, 30592, 768)]
for M, N, K in shapes:
    A = torch.randn(M, K, device=dev).bfloat16()
    B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    res = []
    for flag in (0, 128):
        if M >= 1024 and flag == 0:
            pass
        L.set_debug_flags(flag)
        for _ in range(3):
            L.gemm_bf16_nt(A, B, L.EPI_BF16, out_bf16=o)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()          # 50 launches in one graph: no host launch cost between them
        with torch.cuda.graph(graph):
            for _ in range(50):
                L.gemm_bf16_nt(A, B, L.EPI_BF16, out_bf16=o)
        graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 50 * 1e3)
    L.set_debug_flags(0)
    print(f"{M:6d} x {N:6d} x {K:5d}: ring {res[0]:7.1f} us   128x128 {res[1]:7.1f} us   {2 * M * N * K / res[0] / 1e6:7.1f} TF/s")
