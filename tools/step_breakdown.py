#!/usr/bin/env python
"""Per-op time split of one train_step, measured IN the step with HIP events (no profiler): every C-ABI wrapper the
engine calls is bracketed by events on the launch stream while the whole eager step is queued behind a spin kernel, so
each kernel runs with the caches as its predecessors left them.  python tools/step_breakdown.py [--batch 32] [--steps 3]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--res", type=int, default=384)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--detail", action="store_true", help="split GEMMs by shape/epilogue")
    args = ap.parse_args()
    from feddat_amd import engine, lib as L, vilt_spec
    dev = torch.device("cuda", 0)
    params = vilt_spec.random_init(12, ["c0"], seed=0)
    eng = engine.ViltDatEngine(params, ["c0"], dev, batch=args.batch, res=args.res, layers=12)
    batches = [vilt_spec.synthetic_batch(args.batch, args.res, 1234 + i, device=dev) for i in range(2)]
    eng.begin_local_update("c0", steps_per_epoch=100)
    rec = []
    skip = {"load", "make_segs", "make_wgrad_segs", "adapter_wgrad_workspace_elems", "gemm_skinny_workspace_elems"}
    names = [n for n in dir(L) if callable(getattr(L, n)) and not n.startswith("_") and n not in skip
             and getattr(getattr(L, n), "__module__", "") == L.__name__ and n[0].islower()]
    orig = {n: getattr(L, n) for n in names}

    def wrap(n):
        f = orig[n]

        def g(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = f(*a, **k)
            e1.record()
            key = n
            if n == "gemm_bf16_nt" and args.detail:
                A, B, epi = a[0], a[1], a[2]
                key = f"gemm M={k.get('M') or A.shape[0]} N={B.shape[0]} K={A.shape[1]} epi={epi}"
            rec.append((key, e0, e1))
            return r
        return g
    for n in names:
        setattr(L, n, wrap(n))
    eng.train_step(batches[0])
    rec.clear()
    tot_ev = []
    for i in range(args.steps):
        torch.cuda.synchronize()
        torch.cuda._sleep(int(0.03 * 2.4e9))
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        eng.train_step(batches[i % 2])
        s1.record()
        tot_ev.append((s0, s1))
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for key, e0, e1 in rec:
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
    step_ms = sum(a.elapsed_time(b) for a, b in tot_ev) / args.steps
    print(f"eager step behind a plug: {step_ms:.3f} ms (events add ~1-2 us per launch)")
    tot = 0.0
    for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{key:48s} {n // args.steps:4d} launches  {ms / args.steps:8.3f} ms/step  {ms / n * 1e3:8.2f} us avg")
        tot += ms / args.steps
    print(f"sum of bracketed ops: {tot:.3f} ms/step")


if __name__ == "__main__":
    main()
