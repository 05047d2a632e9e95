#!/usr/bin/env python
"""Which scale byte does feddat_gemm_fp8mx_nt's MFMA apply to which data?  A = 1.0 (e4m3 0x38) in ONE 32-column block b of every
row, zeros elsewhere; B = 1.0; scales unit except block b' = 2^3.  out = 32 * (8 if the hardware pairs scale block b' with data
block b else 1).  Then a row test: scale of row r = 2^(r % 4) on the data block -> which row's scale each output row saw."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddat_amd import lib as L  # noqa: E402

dev = "cuda"
M, N = 1024, 192
for K in (128, 256):
    nb = K // 32
    W8 = torch.full((N, K), 0x38, dtype=torch.uint8, device=dev)
    sw = torch.ones(N, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    print(f"K = {K}: rows = data block b, columns = scaled block b'")
    for b in range(nb):
        row = []
        for b2 in range(nb):
            A8 = torch.zeros(M, K, dtype=torch.uint8, device=dev)
            A8[:, 32 * b:32 * b + 32] = 0x38
            sc = torch.full((M, nb), 127, dtype=torch.uint8, device=dev)
            sc[:, b2] = 130
            L.gemm_fp8mx_nt(A8, sc, W8, sw, out_bf16=out)
            torch.cuda.synchronize()
            vals = out.float().unique().tolist()
            row.append("/".join(f"{v:g}" for v in vals[:4]))
        print("  b =", b, row)
K = 128
A8 = torch.zeros(M, K, dtype=torch.uint8, device=dev)
A8[:, 32:64] = 0x38
sc = torch.full((M, 4), 127, dtype=torch.uint8, device=dev)
sc[:, 1] = (127 + torch.arange(M, device=dev) % 4).to(torch.uint8)
W8 = torch.full((N, K), 0x38, dtype=torch.uint8, device=dev)
L.gemm_fp8mx_nt(A8, sc, W8, torch.ones(N, device=dev), out_bf16=out)
torch.cuda.synchronize()
print("row test (expect 32, 64, 128, 256 repeating):", out[:12, 0].float().tolist(), "cols equal:", bool((out == out[:, :1]).all()))
