#!/usr/bin/env python
"""Is the persistent GEMM's rate set by its instruction schedule or by the chip's power cap?  The same launches (production
kernel and the DUAL form, selection flags 1 | 2) on random operands, on operands whose low mantissa bits are cleared, and on
zeros: identical instruction streams and bytes, different switching activity in the MFMA datapath
(MI355X_MICROARCH.md "DVFS give-back")."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddat_amd import lib as L  # noqa: E402

dev = "cuda:0"
FMT = sys.argv[1] if len(sys.argv) > 1 else "f16"
DT = L.OPERAND_DTYPE[FMT]
M = 11840


def timed(run, n=60):
    for _ in range(10):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def data(kind, rows, K, scale):
    x = torch.randn(rows, K, device=dev) * scale
    if kind == "zeros":
        x.zero_()
    x = x.to(DT)
    if kind == "3bit":        # keep sign, exponent and the top 3 mantissa bits
        bits = x.view(torch.int16)
        keep = -128 if DT == torch.float16 else -16          # fp16: 10 mantissa bits -> clear the low 7; bf16: 7 -> clear the low 4
        x = (bits & keep).view(DT)
    return x


with L.operands(FMT):
    for N, K, epi in ((3072, 768, L.EPI_BF16), (768, 3072, L.EPI_BF16)):
        for kind in ("random", "3bit", "zeros"):
            A, B = data(kind, M, K, 1.0), data(kind, N, K, 0.05)
            o = torch.empty(M, N, dtype=DT, device=dev)
            row = []
            for name, flags in (("prod", 0), ("dual", 3)):
                L.set_debug_flags(flags)
                t = timed(lambda: L.gemm_bf16_nt(A, B, epi, out_bf16=o))
                row.append(f"{name} {t:6.1f} us = {2.0 * M * N * K / t / 1e6:5.0f} TF/s")
            L.set_debug_flags(0)
            print(f"N={N} K={K} operands {kind:6s}: " + " | ".join(row), flush=True)
