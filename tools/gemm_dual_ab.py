#!/usr/bin/env python
"""Round 6: the DUAL form of the persistent GEMM ("v4": two independent 128 x 192 workgroups per CU, gemm_nt_v3_kernel<E, 4, 0, true>)
against the production choice (v2 / v3) per shape and epilogue: time (warm = back-to-back launches; cold = a 512 MB write
between launches, what the launch sees inside the step) and bit-identity of every output.  Selection flags 1 | 2 together route every
persistent launch to it, 1 | 2 | 64 only launches with >= 2 rounds of the doubled grid.  fp16-operand library (the engine's default)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddat_amd import lib as L  # noqa: E402

dev = "cuda:0"
FMT = sys.argv[1] if len(sys.argv) > 1 else "f16"
DT = L.OPERAND_DTYPE[FMT]
M = 11840
SHAPES = [(M, 3072, 768, L.EPI_GELU_G8), (M, 3072, 768, L.EPI_MUL_G8), (M, 3072, 768, L.EPI_BF16), (M, 2304, 768, L.EPI_BF16),
          (M, 768, 3072, L.EPI_RESID_F32), (M, 768, 3072, L.EPI_BF16), (M, 768, 2304, L.EPI_BF16), (M, 768, 768, L.EPI_RESID_F32),
          (M, 768, 768, L.EPI_BF16), (5920, 3072, 768, L.EPI_GELU), (5920, 768, 3072, L.EPI_RESID_F32)]
big = torch.empty(512 * 1024 * 1024 // 4, device=dev)


def timed(run, n=30):
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def cold(run, n=8):
    ts = []
    for _ in range(n):
        big.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


with L.operands(FMT):
    print("dual form: workgroups per CU granted by the runtime:", L.gemm_dual_blocks_per_cu(), flush=True)
    for Mr, N, K, epi in SHAPES:
        A = torch.randn(Mr, K, device=dev).to(DT)
        B = (torch.randn(N, K, device=dev) * 0.05).to(DT)
        bias = torch.randn(N, device=dev)
        resid = torch.randn(Mr, N, device=dev)
        codes_in = torch.randint(0, 255, (Mr, N), dtype=torch.uint8, device=dev)
        res = {}
        for name, flags in (("prod", 0), ("dual", 3)):
            L.set_debug_flags(flags)
            o16 = torch.zeros(Mr, N, dtype=DT, device=dev)
            o2 = torch.zeros(Mr, N, dtype=DT, device=dev)
            o8 = torch.zeros(Mr, N, dtype=torch.uint8, device=dev)
            o32 = torch.zeros(Mr, N, device=dev)

            def run():
                if epi == L.EPI_RESID_F32:
                    L.gemm_bf16_nt(A, B, epi, bias=bias, resid=resid, out_f32=o32)
                elif epi == L.EPI_GELU_G8:
                    L.gemm_bf16_nt(A, B, epi, bias=bias, out_bf16=o16, out2_bf16=o8)
                elif epi == L.EPI_MUL_G8:
                    L.gemm_bf16_nt(A, B, epi, aux=codes_in, out_bf16=o16)
                elif epi == L.EPI_GELU:
                    L.gemm_bf16_nt(A, B, epi, bias=bias, out_bf16=o16, out2_bf16=o2)
                else:
                    L.gemm_bf16_nt(A, B, epi, bias=bias, out_bf16=o16)
            res[name] = (timed(run), cold(run), [t.clone() for t in (o16, o2, o8, o32)])
        L.set_debug_flags(0)
        same = all(torch.equal(a, b) for a, b in zip(res["prod"][2], res["dual"][2]))
        fl = 2.0 * Mr * N * K
        print(f"M={Mr} N={N} K={K} epi={epi}: prod warm {res['prod'][0]:6.1f} cold {res['prod'][1]:6.1f} us | dual warm "
              f"{res['dual'][0]:6.1f} ({fl / res['dual'][0] / 1e6:5.0f} TF/s) cold {res['dual'][1]:6.1f} us | ratio warm "
              f"{res['dual'][0] / res['prod'][0]:.3f} cold {res['dual'][1] / res['prod'][1]:.3f} | bit-identical {same}", flush=True)
