#!/usr/bin/env python
"""configs[4]: the two frozen products per layer that stay bf16 -- measured, not estimated.
  (a) attention-output projection (forward): A = ctx, N = 768, K = 768, fp32 + residual epilogue.  fp8 needs an e4m3 copy of ctx
      (fixed scale: ctx rows are convex combinations of V rows) written by the attention kernel NEXT to the bf16 ctx the
      backward's D = rowsum(dO . O) reads: + M x 768 bytes of writes per layer.
  (b) QKV^T (backward): A = dqkv, N = 768, K = 2304, bf16 epilogue.  fp8 needs per-row scales of dqkv, whose rows are written
      64 columns at a time by 36 different attention-backward blocks: a separate amax + quantise pass over [M, 2304].
Timed: the bf16 product, the fp8 product (block-scaled K = 128 MFMA) on ready-made e4m3 operands = the UPPER bound of the gain,
and the quantise pass (feddat_quant_rows_fp8 reads fp32; a bf16-input form would move 3/5 of its bytes: both figures printed).
    python tools/fp8_remaining_probe.py [--batch 64]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddat_amd import lib as L  # noqa: E402

dev = "cuda"


def t(fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    M = 2 * args.batch * 185
    print(f"M = {M} rows (B = {args.batch}, both passes), isolated launches, operands MALL-warm (the bf16 figure in the step is 5-10 % higher)")
    for name, N, K, resid in (("(a) attention-output + residual", 768, 768, True), ("(b) QKV^T", 768, 2304, False)):
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) * 0.05
        A16, W16 = A.to(torch.bfloat16), W.to(torch.bfloat16)
        A8, sa = torch.empty(M, K, dtype=torch.uint8, device=dev), torch.empty(M, device=dev)
        W8, sw = torch.empty(N, K, dtype=torch.uint8, device=dev), torch.empty(N, device=dev)
        L.quant_rows_fp8(A, A8, sa)
        L.quant_rows_fp8(W, W8, sw)
        bias = torch.randn(N, device=dev)
        if resid:
            r, o32 = torch.randn(M, N, device=dev), torch.empty(M, N, device=dev)
            tb = t(lambda: L.gemm_bf16_nt(A16, W16, L.EPI_RESID_F32, bias=bias, resid=r, out_f32=o32))
            t8 = t(lambda: L.gemm_fp8_nt_f32(A8, sa, W8, sw, bias=bias, resid=r, out_f32=o32))
        else:
            o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            tb = t(lambda: L.gemm_bf16_nt(A16, W16, L.EPI_BF16, bias=bias, out_bf16=o))
            t8 = t(lambda: L.gemm_fp8_nt(A8, sa, W8, sw, L.EPI_BF16, bias=bias, out_bf16=o))
        tq = t(lambda: L.quant_rows_fp8(A, A8, sa))
        bytes_q32, bytes_q16 = M * K * 5, M * K * 3
        print(f"{name:34s} bf16 {tb:6.1f} us | fp8 product alone {t8:6.1f} us (upper bound of the gain {tb - t8:5.1f} us) | "
              f"quantise pass [M, {K}] fp32 in: {tq:5.1f} us ({bytes_q32 / tq / 1e6:.1f} TB/s); bf16 in would move "
              f"{bytes_q16 / 1e6:.0f} MB: >= {bytes_q16 / 6.5e6:5.1f} us at 6.5 TB/s | extra e4m3 copy of A: {M * K / 1e6:.0f} MB "
              f">= {M * K / 6.5e6:4.1f} us", flush=True)


if __name__ == "__main__":
    main()
