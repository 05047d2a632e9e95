#!/usr/bin/env python
"""Where do the fp16 gradient operands of the default engine sit in fp16's range?  One eager train_step at configs[1]'s size (after
20 graph steps so that the weights are not at their initial values), then the 16-bit gradient operand buffers as the backward left
them (the deepest layers' values -- the smallest of the pass -- and the top layer's -- the largest): max |x|, the share of non-zero
values that are subnormal (< 6.1e-5: fewer than 10 mantissa bits), the share that are exactly zero.
    python tools/f16_grad_range.py [--scale 16384]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, nargs="*", default=[16384.0, 1024.0, 65536.0 * 16])
    args = ap.parse_args()
    from feddat_amd import engine, vilt_spec
    dev = torch.device("cuda", 0)
    params = vilt_spec.random_init(12, ["c0"], seed=0)
    batches = [vilt_spec.synthetic_batch(32, 384, 1234 + i, device=dev) for i in range(4)]
    for sc in args.scale:
        eng = engine.ViltDatEngine(params, ["c0"], dev, batch=32, res=384, layers=12, operands="f16", loss_scale=sc)
        eng.begin_local_update("c0", steps_per_epoch=80)
        for i in range(20):
            eng.train_step(batches[i % 4], use_graph=True)
        eng.train_step(batches[0], use_graph=False)
        torch.cuda.synchronize()
        print(f"loss scale 2^{int(torch.log2(torch.tensor(sc)))}:")
        bufs = {"top layer dh3 (adapter bwd -> FFN2^T)": eng.top["dh316"], "top layer dU": eng.top["dU"], "top layer dx2": eng.top["dx2"],
                "dh16 (layer 1: adapter / LayerNorm bwd -> dX products)": eng.dh16, "dU (layer 1)": eng.dU, "dx16 (layer 1)": eng.dx16,
                "dctx (layer 1)": eng.dctx, "dqkv (layer 1)": eng.dqkv}
        for name, t in bufs.items():
            x = t.float().abs()
            nz = x[x > 0]
            print(f"   {name:58s} max {float(x.max()):9.3e}  median {float(nz.median()):9.3e}  subnormal {float((nz < 6.1e-5).float().mean()):7.4f} of non-zero"
                  f"  zero {float((x == 0).float().mean()):7.4f}  inf/nan {int((~torch.isfinite(t.float())).sum())}")
        eng.assert_finite()
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
