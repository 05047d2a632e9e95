#!/usr/bin/env python
"""Do the persistent GEMMs gain when HALF of the compute units walk 256-row tiles and the other half 192-row tiles, so that the
two halves' epilogues (chip-wide HBM write bursts) stop coinciding -- at no cost in work?
tools/gemm_dephase.py delayed blocks and found the contention but paid for the delay; here nothing waits: the row range is
split in two launches on two streams, each capped at 128 workgroups (production selection bits of feddat_set_debug_flags: 64 / 32
force the tile height, bits 28..31 cap the persistent grid), captured in one graph.
    python tools/gemm_stagger_probe.py [--operands f16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddat_amd import lib as L  # noqa: E402

dev = "cuda:0"
M = 11840
CAP128 = 8 << 28


def make(N, K, epi, dt):
    A = torch.randn(M, K, device=dev).to(dt)
    B = (torch.randn(N, K, device=dev) * 0.05).to(dt)
    t = dict(A=A, B=B, o=torch.empty(M, N, dtype=dt, device=dev), bias=torch.randn(N, device=dev))
    if epi == L.EPI_GELU_G8:
        t["o2"] = torch.empty(M, N, dtype=torch.uint8, device=dev)
    if epi == L.EPI_MUL_G8:
        t["aux"] = torch.randint(0, 255, (M, N), dtype=torch.uint8, device=dev)
    if epi == L.EPI_RESID_F32:
        t["resid"], t["o32"] = torch.randn(M, N, device=dev), torch.empty(M, N, device=dev)
    return t


def launch(t, epi, r0, r1):
    A = t["A"][r0:r1]
    if epi == L.EPI_GELU_G8:
        L.gemm_bf16_nt(A, t["B"], epi, bias=t["bias"], out_bf16=t["o"][r0:r1], out2_bf16=t["o2"][r0:r1])
    elif epi == L.EPI_MUL_G8:
        L.gemm_bf16_nt(A, t["B"], epi, aux=t["aux"][r0:r1], out_bf16=t["o"][r0:r1])
    elif epi == L.EPI_RESID_F32:
        L.gemm_bf16_nt(A, t["B"], epi, bias=t["bias"], resid=t["resid"][r0:r1], out_f32=t["o32"][r0:r1])
    else:
        L.gemm_bf16_nt(A, t["B"], epi, bias=t["bias"], out_bf16=t["o"][r0:r1])


def timed_graph(fn, reps=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    for _ in range(2):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--operands", default="f16")
    args = ap.parse_args()
    dt = L.OPERAND_DTYPE[args.operands]
    side = [torch.cuda.Stream(), torch.cuda.Stream()]
    with L.operands(args.operands):
        for name, N, K, epi, splits in (("FFN1 + GELU + codes", 3072, 768, L.EPI_GELU_G8, (6144, 5888, 6400, 5632)),
                                        ("FFN2^T . code", 3072, 768, L.EPI_MUL_G8, (6144, 5888, 6400, 5632)),
                                        ("QKV", 2304, 768, L.EPI_BF16, (6144, 5632, 6656, 5120)),
                                        ("FFN2 + residual", 768, 3072, L.EPI_RESID_F32, (6144, 7168, 5120)),
                                        ("FFN1^T", 768, 3072, L.EPI_BF16, (6144, 7168, 5120))):
            t = make(N, K, epi, dt)

            def base():
                L.set_debug_flags(0)
                launch(t, epi, 0, M)
            row = [f"one launch {timed_graph(base):6.1f} us"]
            for m1 in splits:
                for fa, fb, tag in ((64, 32, "256|192"), (32, 64, "192|256")):
                    def pair():
                        cur = torch.cuda.current_stream()
                        for s in side:
                            s.wait_stream(cur)
                        with torch.cuda.stream(side[0]):
                            L.set_debug_flags(fa | CAP128)
                            launch(t, epi, 0, m1)
                        with torch.cuda.stream(side[1]):
                            L.set_debug_flags(fb | CAP128)
                            launch(t, epi, m1, M)
                        for s in side:
                            cur.wait_stream(s)
                    row.append(f"{tag} rows {m1}+{M - m1}: {timed_graph(pair):6.1f}")
                    L.set_debug_flags(0)
            print(f"{name:22s} " + " | ".join(row), flush=True)
    L.set_debug_flags(0)


if __name__ == "__main__":
    main()
