"""How much of the B = 32 round-length error is the bf16 rounding of the FROZEN GEMM weights: give the oracle and the engine
the same bf16-representable backbone (every frozen matrix the engine holds in bf16 rounded once, on both sides) and compare
all elements after 20 / 40 / 60 / 80 steps."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import feddat_oracle as O
from feddat_amd import engine
d = O.ViltDims(layers=12)
P = O.make_params(d, ["art"], bias_std=0.02)
n = 0
for k in P:
    if (k.endswith(".weight") and ("attention.attention" in k or "attention.output.dense" in k or "intermediate.dense" in k
                                   or "output.layer.dense" in k)) or k.endswith("patch_embeddings.projection.weight"):
        P[k] = P[k].to(torch.bfloat16).float()
        n += 1
print("rounded", n, "frozen matrices")
P0 = {k: v.clone() for k, v in P.items()}
eng = engine.ViltDatEngine(P, ["art"], "cuda", batch=32, res=384, layers=12)
eng.begin_local_update("art", steps_per_epoch=80)
client = O.DatClient(P, d, "art", lr=1e-4, steps_per_epoch=80)
torch.set_num_threads(min(torch.get_num_threads(), 32))
names = [k for k in eng.state_dict() if "adapter_2" not in k]
for s in range(80):
    b = O.synthetic_batch(32, 384, 8000 + s)
    client.train_step(b)
    eng.train_step({k: v.cuda() for k, v in b.items()}, use_graph=True)
    if s + 1 in (20, 40, 60, 80):
        sd = eng.state_dict()
        row = {"adapters": [0, 0, 0], "head": [0, 0, 0]}
        for k in names:
            d_ref, d_got = P[k] - P0[k], sd[k].cpu() - P0[k]
            err = (d_got - d_ref).abs()
            g = "head" if k.startswith("task_layer") else "adapters"
            r = row[g]
            r[0] = max(r[0], float(err.max())); r[1] = max(r[1], float(err.mean()) / max(float(d_ref.abs().mean()), 1e-12))
            r[2] = max(r[2], abs(float(d_got.norm()) - float(d_ref.norm())) / max(float(d_ref.norm()), 1e-12))
        print(f"same bf16 backbone, n={s+1}: " + "  ".join(f"{g}: max {r[0]:.2e} ratio {r[1]:.4f} norm {r[2]:.4f}" for g, r in row.items()), flush=True)
