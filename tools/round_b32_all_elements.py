#!/usr/bin/env python
"""The 80-step round at B = 32 on ALL elements: the CPU oracle steps the round once (snapshots after 20 / 40 / 60 / 80 steps; ~8
minutes on the GPU box's host cores), then each engine configuration replays the same batches (10 s each) and is compared on
every element of every trainable tensor -- max |ddW|, the number of elements off by more than 1e-3 / 5e-4, mean ratio.
    python tools/round_b32_all_elements.py f16 f16:codes=0 bf16
    FEDDAT_ROUND_SEED0=9000 python tools/round_b32_all_elements.py f16 bf16      # the same on other batches (default 8000 = the fixture's)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import feddat_oracle as O  # noqa: E402  (measurement script)

SNAPS = (20, 40, 60, 80)


def main():
    from feddat_amd import engine
    cfgs = sys.argv[1:] or ["f16", "bf16"]
    d = O.ViltDims(layers=12)
    P = O.make_params(d, ["art"], bias_std=0.02)
    P0 = {k: v.clone() for k, v in P.items()}
    names = [k for k in P if ("adapter_0" in k or "adapter_1" in k or k.startswith("task_layer.art."))]
    seed0 = int(os.environ.get("FEDDAT_ROUND_SEED0", "8000"))
    print("batches", seed0, "..", seed0 + 79)
    host = [O.synthetic_batch(32, 384, seed0 + s) for s in range(80)]
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    client = O.DatClient(P, d, "art", lr=1e-4, steps_per_epoch=80)
    ref = {}
    for s, b in enumerate(host):
        client.train_step(b)
        if s + 1 in SNAPS:
            ref[s + 1] = {k: (P[k] - P0[k]).clone() for k in names}
            print("oracle at step", s + 1, flush=True)
    dev = [{k: v.to("cuda") for k, v in b.items()} for b in host]
    for cfg in cfgs:
        name, _, opts = cfg.partition(":")
        kw = dict(operands=name)
        for o in filter(None, opts.split(",")):
            k, v = o.split("=")
            if k == "codes":
                kw["gelu_codes"] = bool(int(v))
            elif k == "scale":
                kw["loss_scale"] = float(v)
        eng = engine.ViltDatEngine({k: v.clone() for k, v in P0.items()}, ["art"], "cuda", batch=32, res=384, layers=12, **kw)
        eng.begin_local_update("art", steps_per_epoch=80)
        print(cfg)
        for s, b in enumerate(dev):
            eng.train_step(b, use_graph=True)
            if s + 1 in SNAPS:
                sd = eng.state_dict()
                row = {}
                for grp, sel in (("adapter_1 (communicated)", [k for k in names if "adapter_1" in k]),
                                 ("adapter_0 (personal)", [k for k in names if "adapter_0" in k]),
                                 ("head", [k for k in names if "adapter" not in k])):
                    mx, n1, n5, tot, ratio = 0.0, 0, 0, 0, 0.0
                    for k in sel:
                        e = (sd[k].cpu() - P0[k] - ref[s + 1][k]).abs()
                        mx = max(mx, float(e.max()))
                        n1 += int((e > 1e-3).sum())
                        n5 += int((e > 5e-4).sum())
                        tot += e.numel()
                        ratio = max(ratio, float(e.mean()) / float(ref[s + 1][k].abs().mean()))
                    row[grp] = f"max {mx:.2e}, > 1e-3: {n1}, > 5e-4: {n5} of {tot}, worst mean ratio {ratio:.4f}"
                print(f"   {s + 1:2d} steps | " + " | ".join(f"{g}: {v}" for g, v in row.items()), flush=True)
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
