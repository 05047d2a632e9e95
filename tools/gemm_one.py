#!/usr/bin/env python
"""One K1 launch shape for counter collection: python tools/gemm_one.py M N K epi flags [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddat_amd import lib as L  # noqa: E402
L.use_ablation_build()      # timing-only probes live in libfeddat_hip_ablate.so (python -m feddat_amd.build --ablate)

M, N, K, epi, flags = (int(v, 0) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
dev = "cuda:0"
A = torch.randn(M, K, device=dev).bfloat16()
B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
o2 = torch.empty_like(o)
o32 = torch.empty(M, N, device=dev)
aux = torch.randn(M, N, device=dev).bfloat16()
resid = torch.randn(M, N, device=dev)
bias = torch.randn(N, device=dev)
L.set_debug_flags(flags)
for _ in range(reps):
    if epi == 0:
        L.gemm_bf16_nt(A, B, 0, bias=bias, out_bf16=o)
    elif epi == 1:
        L.gemm_bf16_nt(A, B, 1, bias=bias, resid=resid, out_f32=o32)
    elif epi == 2:
        L.gemm_bf16_nt(A, B, 2, bias=bias, out_bf16=o, out2_bf16=o2)
    elif epi == 3:
        L.gemm_bf16_nt(A, B, 3, aux=aux, out_bf16=o)
    else:
        L.gemm_bf16_nt(A, B, 4, bias=bias, out_f32=o32)
torch.cuda.synchronize()
