#!/usr/bin/env python
"""Per-op time split of one train_step, measured IN the step with HIP events (no profiler): every C-ABI wrapper the
engine calls is bracketed by an event pair on the launch stream while the whole eager step is queued behind a spin kernel
(so the GPU never waits for the host and each kernel runs with the caches as its predecessors left them).  The cost of an
empty bracket (two event markers back to back) is measured in the same queue and subtracted.
  python tools/step_breakdown.py [--batch 32] [--steps 3] [--detail]
bench.py imports measure() for its in-step roofline figure."""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import torch  # noqa: E402

_SKIP = {"load", "make_segs", "make_wgrad_segs", "adapter_wgrad_workspace_elems", "gemm_skinny_workspace_elems",
         "make_rccl_comm", "ht_job", "adamw_group", "use_ablation_build", "set_debug_flags", "operands", "current_operands"}


def measure(eng, L, batches, steps=3, detail=True, plug_s=0.03):
    """-> (agg: key -> [launches per step, ms per step (bracket overhead removed)], eager step ms, empty bracket us).
    GEMM keys (detail): ('gemm', M, N, K, epi, skinny); general attention: ('attn2_fwd' | 'attn2_bwd', B, Sq, Skv, heads, causal)."""
    names = [n for n in dir(L) if callable(getattr(L, n)) and not n.startswith("_") and n not in _SKIP
             and getattr(getattr(L, n), "__module__", "") == L.__name__ and n[0].islower()]
    orig = {n: getattr(L, n) for n in names}
    rec = []

    def wrap(n):
        f = orig[n]

        def g(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = f(*a, **k)
            e1.record()
            key = n
            if n == "gemm_bf16_nt" and detail:
                A, B, epi = a[0], a[1], a[2]
                M = k.get("M") or A.shape[0]
                key = ("gemm", M, B.shape[0], A.shape[1], epi, bool(k.get("skinny_workspace") is not None and M <= 64))
            elif n == "attn2_fwd" and detail:        # (name, B, Sq, Skv, heads, causal)
                key = (n, a[5], a[6], a[7], a[8], bool(k.get("causal", False)))
            elif n == "attn2_bwd" and detail:
                key = (n, a[10], a[11], a[12], a[13], bool(k.get("causal", False)))
            rec.append((key, e0, e1))
            return r
        return g
    for n in names:
        setattr(L, n, wrap(n))
    empties, tot_ev = [], []
    layer_calls, eng.use_layer_calls = getattr(eng, "use_layer_calls", False), False      # op-by-op: every kernel bracketed
    try:
        eng.train_step(batches[0])
        rec.clear()
        for i in range(steps):
            torch.cuda.synchronize()
            torch.cuda._sleep(int(plug_s * 2.4e9))
            for _ in range(8):          # empty brackets: the marker cost to subtract
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                e1.record()
                empties.append((e0, e1))
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            eng.train_step(batches[i % len(batches)])
            s1.record()
            tot_ev.append((s0, s1))
        torch.cuda.synchronize()
    finally:
        eng.use_layer_calls = layer_calls
        for n in names:
            setattr(L, n, orig[n])
    empty_ms = sorted(a.elapsed_time(b) for a, b in empties)[len(empties) // 2]
    agg = collections.OrderedDict()
    for key, e0, e1 in rec:
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += max(e0.elapsed_time(e1) - empty_ms, 0.0)
    for a in agg.values():
        a[0] //= steps
        a[1] /= steps
    step_ms = sum(a.elapsed_time(b) for a, b in tot_ev) / steps
    return agg, step_ms, empty_ms * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--res", type=int, default=384)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--detail", action="store_true", help="split GEMMs by shape/epilogue")
    ap.add_argument("--operands", default="f16", choices=["bf16", "f16"], help="16-bit operand format of the engine")
    ap.add_argument("--debug-flags", type=lambda v: int(v, 0), default=0, help="feddat_set_debug_flags value (ablations)")
    args = ap.parse_args()
    from feddat_amd import engine, lib as L, vilt_spec
    if args.debug_flags & ~(1 | 2 | 32 | 64 | 128 | 256 | (1 << 23) | (0xf << 28)):
        if args.operands != "bf16":      # the ablation build exists for bf16 operands only: say so instead of timing un-ablated kernels
            raise SystemExit("timing-only ablation flags need --operands bf16 (libfeddat_hip_ablate.so is a bf16-operand build)")
        L.use_ablation_build()      # timing-only probes: libfeddat_hip_ablate.so (python -m feddat_amd.build --ablate)
    with L.operands(args.operands):     # the debug flags are per library: set them in the one the engine below runs in
        L.set_debug_flags(args.debug_flags)
    dev = torch.device("cuda", 0)
    params = vilt_spec.random_init(12, ["c0"], seed=0)
    eng = engine.ViltDatEngine(params, ["c0"], dev, batch=args.batch, res=args.res, layers=12, operands=args.operands)
    batches = [vilt_spec.synthetic_batch(args.batch, args.res, 1234 + i, device=dev) for i in range(2)]
    eng.begin_local_update("c0", steps_per_epoch=100)
    agg, step_ms, empty_us = measure(eng, L, batches, args.steps, args.detail)
    print(f"eager step behind a plug: {step_ms:.3f} ms; empty event bracket {empty_us:.2f} us (subtracted per op)")
    tot = 0.0
    for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{str(key):48s} {n:4d} launches  {ms:8.3f} ms/step  {ms / max(n, 1) * 1e3:8.2f} us avg")
        tot += ms
    print(f"sum of bracketed ops: {tot:.3f} ms/step")


if __name__ == "__main__":
    main()
