"""North-star parity bar (adapter-weight max-abs-diff < 1e-3 after one FL round) at ROUND length AND at the batch size the
metric is quoted on: configs[1] / configs[2] are B = 32, 384 x 384, S = 185, rounds of 40-80 steps.

Reference run: tests/golden/g8b_round80_b32.npz -- oracle/make_golden.py --only-g8 --steps 80 --batch 32: the reference's own
ViltContinualLearner + TaskTrainer.train_step (task_trainer.py:53-59,280-330), 12 layers, len(loader) = 80 -> 1200 scheduler
ticks, 120 warm-up ticks = 60 batches; the update dW = W_after_n - W_init of every adapter_0 / adapter_1 / head tensor stored
after n = 20 / 40 / 60 / 80 steps (L2 norm, mean |dW|, max |dW|, 1024 strided samples), so one fixture serves every round
length.  The engine replays the same 80 batches as one hipGraph per step (production path).

  * test_b32_round_vs_reference_golden[n]: the update after n steps against the reference's samples.
  * test_b32_round_vs_live_oracle: the same run compared on ALL elements with the CPU oracle stepping batch for batch
    (the oracle is pinned to the reference's 40- and 80-step rounds at B = 4: tests/test_oracle_golden.py, and to this
    fixture's first snapshot below).
  * (round 6) test_b32_round_80_steps_all_elements_vs_reference_fixture: all 4.4 M elements at 80 steps against the reference's
    own all-element fixture -- the tail of the distribution, in the driver's suite, no CPU stepping;
    test_b32_round_second_seed_vs_reference_golden: a second, independent 80-step round of the reference.
The B = 4 rounds of tests/test_round40_gpu.py stay as the stress case: per-element gradients are noisiest there.

Round 5: both operand formats.  "f16" (the engine's default: fp16 MFMA operands + 2^14 loss scale, the reference's own
mixed_precision) is asserted against north_star's bound itself -- every adapter AND head tensor, every snapshot, max |ddW| < 1e-3
(measured 8.1e-4 at 80 steps, head 1.8e-4) --; "bf16" keeps its measured bounds, plus a strict-xfail copy of the north-star
assertion at 80 steps so that the gap of that format stays visible."""
import os

import numpy as np
import pytest
import torch

from oracle import feddat_oracle as O
from tests.golden_util import load

pytestmark = pytest.mark.gpu
DEV = "cuda"
SNAPS = (20, 40, 60, 80)
ORACLE_STEPS = int(os.environ.get("FEDDAT_B32_ORACLE_STEPS", "20"))     # live-oracle prefix (6 CPU-seconds per step on the GPU box;
# 80 = the whole round: profiles/r05_round_b32_all_elements.txt holds that table)


def _dev(b):
    return {k: v.to(DEV) for k, v in b.items()}


def _samples(flat, n=1024):
    return flat[torch.linspace(0, flat.numel() - 1, min(n, flat.numel())).long()]


@pytest.fixture(scope="module", params=["f16", "bf16"])
def b32_round(golden_dir, request):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from feddat_amd import engine
    fmt = request.param
    oracle_steps = ORACLE_STEPS if fmt == "f16" else 0      # the live oracle steps next to the default format only
    g = load(golden_dir, "g8b_round80_b32.npz")
    steps, B = int(g["steps"]), int(g["batch"])
    assert (steps, B) == (80, 32)
    d = O.ViltDims(layers=12)
    P = O.make_params(d, ["art"], bias_std=0.02)
    P0 = {k: v.clone() for k, v in P.items()}
    eng = engine.ViltDatEngine(P, ["art"], DEV, batch=B, res=384, layers=12, operands=fmt)
    assert eng.op_dtype == {"f16": torch.float16, "bf16": torch.bfloat16}[fmt]
    eng.begin_local_update("art", steps_per_epoch=steps)
    client = O.DatClient(P, d, "art", lr=1e-4, steps_per_epoch=steps)
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    keys = [k.split("::", 2)[2] for k in g if k.startswith("s80::dsamp::")]
    losses, snaps, osnaps = [], {}, {}
    for s in range(steps):
        b = O.synthetic_batch(B, 384, 8000 + s)
        if s < oracle_steps:
            client.train_step(b)
        losses.append(float(eng.train_step(_dev(b), use_graph=True)[0]))
        if s + 1 in SNAPS:
            sd = eng.state_dict()
            snaps[s + 1] = {k: (sd[k].cpu() - P0[k]) for k in keys}
            if s + 1 <= oracle_steps:
                osnaps[s + 1] = {k: (P[k] - P0[k]).clone() for k in keys}
    return dict(g=g, keys=keys, losses=np.array(losses), snaps=snaps, osnaps=osnaps, fmt=fmt)


def _vs_golden(r, n):
    g, rows = r["g"], {}
    for k in r["keys"]:
        dw = r["snaps"][n][k].flatten()
        ref = torch.from_numpy(g[f"s{n}::dsamp::{k}"])
        err = (_samples(dw) - ref).abs()
        rows[k] = dict(max=float(err.max()), ratio=float(err.mean()) / max(float(ref.abs().mean()), 1e-12),
                       n_gt_1e3=int((err > 1e-3).sum()), n=err.numel(),
                       norm=abs(float(dw.norm()) - float(g[f"s{n}::dnorm::{k}"])) / float(g[f"s{n}::dnorm::{k}"]),
                       moved=float(g[f"s{n}::dmax::{k}"]))
    return rows


def _group(k):
    return "head" if k.startswith("task_layer.") else "adapters"


def _group3(k):
    return "head" if k.startswith("task_layer.") else "adapter_1" if "adapter_1" in k else "adapter_0"


def _table(rows):
    out = {}
    for grp in ("adapters", "head"):
        sel = [r for k, r in rows.items() if _group(k) == grp]
        out[grp] = {m: max(r[m] for r in sel) for m in ("max", "ratio", "norm", "moved")}
    return out


# Measured on MI355X (r04; hipGraph replay; the same to two digits with the fused or the single-purpose tail, token-0 or dense
# attention in the last layer -- four variants; worst tensor, max |ddW| on the reference's samples):
#     steps                      20         40         60         80
#     adapters               0.9-2.2e-4  2.8-4.2e-4  6.5-8.6e-4  1.29-1.32e-3      (the round moves them by 0.35 / 1.5 / 3.1 / 3.9e-3)
#     head (clf_norm0.bias)    1.32e-3     2.74e-3     2.93e-3     3.09e-3         (moved 0.76 / 3.4 / 6.9 / 9.2e-3)
# north_star's "adapter-weight max-abs-diff < 1e-3 after one FL round" is MET at B = 32 for rounds of up to 60 steps and
# missed by 30 % at 80 (1.2e-3 over all elements vs the live oracle).  The head's LayerNorm gain / bias have a few elements
# whose batch gradient (a sum over 32 rows) is smaller than the systematic error the bf16 frozen weights put on it: AdamW
# normalises every element to +-lr steps, so such an element walks the other way at full speed -- 1.32e-3 after 20 steps is
# exactly 2 sum(lr) of the head's 40 updates, and identical across all four kernel variants (it is the operand precision
# of the backbone, not a kernel).  Their bound is therefore stated against 2 sum(lr) and is not the north-star's.
BOUNDS = {20: dict(adapters=1.0e-3, head=1.7e-3), 40: dict(adapters=1.0e-3, head=3.5e-3),
          60: dict(adapters=1.2e-3, head=3.8e-3), 80: dict(adapters=1.7e-3, head=4.0e-3)}
# fp16 operands (r05, same fixture, same replay): the north-star bound on EVERY trainable tensor at every round length
#     adapters               2.1e-5      8.4e-5      6.4e-4      8.1e-4     mean ratio 0.002 / 0.002 / 0.009 / 0.009, norm <= 0.5 %
#     head                   3.7e-5      5.3e-5      1.1e-4      1.8e-4     (the bf16 weights' systematic walk of clf_norm0.bias is gone)
NORTH_STAR = 1.0e-3
BOUNDS_F16 = {n: dict(adapters=b, head=min(b, 5e-4)) for n, b in ((20, 2e-4), (40, 4e-4), (60, NORTH_STAR), (80, NORTH_STAR))}
MOVED = {20: (3e-4, 7e-4), 40: (1.4e-3, 3e-3), 60: (3e-3, 6e-3), 80: (3.5e-3, 9e-3)}      # the reference's own max |dW| per group


@pytest.mark.parametrize("n", SNAPS)
def test_b32_round_vs_reference_golden(b32_round, n):
    """Per tensor, on the UPDATE after n steps, against the reference's own run: max |dW_hip - dW_ref| over the reference's
    samples below the cell's bound (adapters: north_star's 1e-3 at 20 and 40 steps, measured-with-margin beyond), mean error
    <= 0.05 mean |dW_ref| and update norm within 1.5 % through 60 steps (0.09 / 4 % at 80), at most 0.3 % (0.6 % at 80) of a
    tensor's samples off by more than 1e-3; and the fixture really moves the weights by what the table says."""
    rows = _vs_golden(b32_round, n)
    t = _table(rows)
    f16 = b32_round["fmt"] == "f16"
    bounds = (BOUNDS_F16 if f16 else BOUNDS)[n]
    print(f"B=32, {b32_round['fmt']}, {n:2d} steps vs the reference | adapters: max |ddW| {t['adapters']['max']:.2e}, mean ratio "
          f"{t['adapters']['ratio']:.4f}, norm {t['adapters']['norm']:.5f}, moved {t['adapters']['moved']:.2e} | head: max |ddW| "
          f"{t['head']['max']:.2e}, mean ratio {t['head']['ratio']:.4f}, norm {t['head']['norm']:.5f}, moved {t['head']['moved']:.2e}")
    assert t["adapters"]["moved"] > MOVED[n][0] and t["head"]["moved"] > MOVED[n][1]
    for k, r in rows.items():
        assert r["max"] < bounds[_group(k)], (n, k, r)
        if f16:      # north_star: no sample of any trainable tensor off by 1e-3; bulk within 2 % (3 % at 80) and norm within 1 %
            assert r["n_gt_1e3"] == 0 and r["ratio"] < (0.02 if n <= 60 else 0.03) and r["norm"] < 0.01, (n, k, r)
            continue
        # bulk of the tensor: mean error / mean |dW_ref| and the update norm (measured worst adapter tensor 0.011 / 0.015 /
        # 0.030 / 0.063 and 0.4 / 0.4 / 0.8 / 2.9 % at 20 / 40 / 60 / 80 steps; head 0.007-0.009 and < 0.1 % throughout)
        assert r["ratio"] < (0.05 if n <= 60 else 0.09), (n, k, r)
        assert r["norm"] < (0.015 if n <= 60 else 0.04), (n, k, r)
        assert r["n_gt_1e3"] <= max(1, int((3e-3 if n <= 60 else 6e-3) * r["n"])), (n, k, r)
    if n == 80:
        rel = np.abs(b32_round["losses"] - b32_round["g"]["losses"]) / np.maximum(b32_round["g"]["losses"], 1.0)
        print("loss trajectory: worst rel diff", rel.max(), "final", b32_round["losses"][-1], b32_round["g"]["losses"][-1])
        assert rel[:10].max() < 3e-3 and rel.max() < 3e-2


def test_north_star_bound_at_80_steps_every_format(b32_round, request):
    """north_star verbatim: adapter-weight max-abs-diff < 1e-3 after one FL round (the longest round of configs[2], 80 steps,
    at its own batch size).  Holds for the default fp16 operands; for bf16 operands it is a STRICT xfail (1.3e-3: that format's
    gap, kept visible -- if it ever passes the marker trips)."""
    if b32_round["fmt"] == "bf16":
        request.applymarker(pytest.mark.xfail(strict=True, reason="bf16 operands: 1.3e-3 at 80 steps (frozen-weight rounding)"))
    t = _table(_vs_golden(b32_round, 80))
    assert t["adapters"]["max"] < NORTH_STAR, t


def test_b32_round_vs_live_oracle(b32_round):
    """ALL elements (4.4 M), against the CPU oracle stepping the same batches (first ORACLE_STEPS steps of the round; 20 by default:
    ~2 CPU-minutes on the GPU box -- FEDDAT_B32_ORACLE_STEPS=80 runs the whole round; tools/round_b32_all_elements.py prints the
    same table for several engine configurations); the oracle's own update at its last snapshot is pinned to the reference's
    samples first (< 1e-4: two fp32 implementations).  Measured, fp16 operands (profiles/r05_round_b32_all_elements.txt):
        steps                                   20        40        60        80
        adapter_1 (the communicated adapter)  0.98e-4   1.96e-4   2.66e-4   4.77e-4     no element above 5e-4 at any length
        adapter_0 (personal, gated pass)      0.35e-4   1.15e-4   7.20e-4   1.09e-3     80 steps: ONE element of 894 528 above 1e-3
        head                                  0.37e-4   0.90e-4   1.14e-4   1.82e-4
    (bf16 operands at 80 steps: adapter_1 1.46e-3 with 132 elements above 1e-3, head 3.06e-3 with 89.)  Asserted: the adapter the
    FL round aggregates (adapter_1) and the head below north_star's 1e-3 on every element at every length, with margin (6e-4 / 5e-4);
    adapter_0 below 1e-3 through 60 steps, and at 80 steps at most 3 of its 894 528 elements above it, none above 1.3e-3."""
    r = b32_round
    if r["fmt"] != "f16":
        pytest.skip("the live oracle steps next to the default operand format")
    assert r["osnaps"], "no oracle snapshot inside the prefix"
    g, bad = r["g"], []
    for n in sorted(r["osnaps"]):
        worst = {grp: dict(max=0.0, ratio=0.0, over=0) for grp in ("adapter_1", "adapter_0", "head")}
        pin_worst = 0.0
        for k in r["keys"]:
            d_ref, d_got = r["osnaps"][n][k], r["snaps"][n][k]
            pin = float((_samples(d_ref.flatten()) - torch.from_numpy(g[f"s{n}::dsamp::{k}"])).abs().max())
            err = (d_got - d_ref).abs()
            ratio = float(err.mean()) / max(float(d_ref.abs().mean()), 1e-12)
            w = worst[_group3(k)]
            w["max"], w["ratio"], pin_worst = max(w["max"], float(err.max())), max(w["ratio"], ratio), max(pin_worst, pin)
            w["over"] += int((err > NORTH_STAR).sum())
            if pin >= 1e-4 or ratio >= 0.03:
                bad.append((n, k, "oracle vs reference", pin, "ratio", ratio))
        print(f"B=32, {n:2d} steps vs the live oracle, ALL elements | " + " | ".join(
            f"{grp}: max |ddW| {w['max']:.2e}, > 1e-3: {w['over']}, mean ratio {w['ratio']:.4f}" for grp, w in worst.items()) +
            f" | oracle vs reference samples {pin_worst:.1e}")
        if worst["adapter_1"]["max"] >= 6e-4 or worst["head"]["max"] >= 5e-4:
            bad.append((n, "adapter_1 / head over all elements", worst))
        if (n <= 60 and worst["adapter_0"]["max"] >= NORTH_STAR) or worst["adapter_0"]["over"] > 3 or worst["adapter_0"]["max"] >= 1.3e-3:
            bad.append((n, "adapter_0 over all elements", worst["adapter_0"]))
    assert not bad, bad[:4]


def test_b32_round_80_steps_all_elements_vs_reference_fixture(b32_round, golden_dir):
    """Round 6: the TAIL of the distribution in the driver's suite without stepping the oracle -- EVERY element of every trainable
    tensor after the longest round (80 steps, B = 32) against the reference's own run (tests/golden/g8b_round80_b32_all.npz from
    oracle/make_golden.py --only-g8 --steps 80 --batch 32 --all-elements: float16 of dW * 256, encoding error < 3e-6).  Asserted
    for the default fp16 operands: the communicated adapter_1 below 6e-4 and the head below 5e-4 on every element; the personal
    adapter_0 with at most 3 of its 894 528 elements above 1e-3 and none above 1.3e-3 (measured r05: 4.8e-4 / 1.8e-4 / one
    element at 1.09e-3).  bf16 operands: printed only (132 / 4 / 89 elements above 1e-3)."""
    r = b32_round
    f = load(golden_dir, "g8b_round80_b32_all.npz")
    assert int(f["steps"]) == 80 and int(f["batch"]) == 32 and int(f["seed0"]) == 8000
    scale = float(f["scale"])
    worst = {grp: dict(max=0.0, over_1e3=0, over_5e4=0, n=0) for grp in ("adapter_1", "adapter_0", "head")}
    for k in r["keys"]:
        ref = torch.from_numpy(f["s80::dall::" + k].astype(np.float32)) / scale
        err = (r["snaps"][80][k] - ref).abs()
        w = worst[_group3(k)]
        w["max"], w["n"] = max(w["max"], float(err.max())), w["n"] + err.numel()
        w["over_1e3"] += int((err > NORTH_STAR).sum())
        w["over_5e4"] += int((err > 5e-4).sum())
    print(f"B=32, 80 steps, {r['fmt']}, ALL elements vs the reference's all-element fixture | " + " | ".join(
        f"{grp}: max |ddW| {w['max']:.2e}, > 1e-3: {w['over_1e3']}, > 5e-4: {w['over_5e4']} of {w['n']}" for grp, w in worst.items()))
    assert worst["adapter_1"]["n"] == worst["adapter_0"]["n"] == 894528
    if r["fmt"] != "f16":
        return
    assert worst["adapter_1"]["max"] < 6e-4 and worst["head"]["max"] < 5e-4, worst
    assert worst["adapter_0"]["over_1e3"] <= 3 and worst["adapter_0"]["max"] < 1.3e-3, worst


@pytest.mark.parametrize("seed0", [9000, 10000])
def test_b32_round_second_seed_vs_reference_golden(golden_dir, seed0):
    """Independent rounds (the same model, 80 OTHER batches: seeds seed0..seed0+79; tests/golden/g8b_round80_b32_seed<seed0>.npz
    from oracle/make_golden.py --only-g8 --steps 80 --batch 32 --seed0 <seed0>), default engine: north_star's bound on the
    reference's samples of every trainable tensor at 40 and 80 steps -- the 80-step figure of the first fixture is one draw; these
    are a second and a third."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import os as _os
    if not _os.path.exists(_os.path.join(golden_dir, f"g8b_round80_b32_seed{seed0}.npz")):
        pytest.skip("fixture not generated")
    from feddat_amd import engine
    g = load(golden_dir, f"g8b_round80_b32_seed{seed0}.npz")
    assert int(g["seed0"]) == seed0
    d = O.ViltDims(layers=12)
    P = O.make_params(d, ["art"], bias_std=0.02)
    P0 = {k: v.clone() for k, v in P.items()}
    eng = engine.ViltDatEngine(P, ["art"], DEV, batch=32, res=384, layers=12)
    eng.begin_local_update("art", steps_per_epoch=80)
    keys = [k.split("::", 2)[2] for k in g if k.startswith("s80::dsamp::")]
    snaps = {}
    for s in range(80):
        eng.train_step(_dev(O.synthetic_batch(32, 384, seed0 + s)), use_graph=True)
        if s + 1 in (40, 80):
            sd = eng.state_dict()
            snaps[s + 1] = {k: (sd[k].cpu() - P0[k]) for k in keys}
    assert eng.scaler_state()["skipped_substeps"] == 0
    r = dict(g=g, keys=keys, snaps=snaps)
    for n in (40, 80):
        rows = _vs_golden(r, n)
        t = _table(rows)
        print(f"seed0 {seed0}, B=32, {n} steps vs the reference | adapters: max |ddW| {t['adapters']['max']:.2e}, mean ratio "
              f"{t['adapters']['ratio']:.4f} | head: max |ddW| {t['head']['max']:.2e}")
        for k, row in rows.items():
            assert row["max"] < BOUNDS_F16[n][_group(k)] and row["n_gt_1e3"] == 0 and row["ratio"] < 0.03, (n, k, row)


def test_b32_round_80_steps_with_bf16_u_instead_of_gelu_codes(golden_dir):
    """ViltDatEngine(gelu_codes=False): FFN2's backward reads the pre-GELU activation in bf16 instead of the 8-bit gelu' codes.
    At B = 32 the run is NOT chaotic (perturbing the head's summation order or the last layer's attention kernel changes no
    digit of the table above), so this is a clean measurement of what the codes cost at round length: worst adapter element
    1.31e-3 -> 1.00e-3, worst update-norm error 2.9 % -> 1.6 %, for +0.14 ms/step."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from feddat_amd import engine
    g = load(golden_dir, "g8b_round80_b32.npz")
    d = O.ViltDims(layers=12)
    P = O.make_params(d, ["art"], bias_std=0.02)
    P0 = {k: v.clone() for k, v in P.items()}
    eng = engine.ViltDatEngine(P, ["art"], DEV, batch=32, res=384, layers=12, gelu_codes=False, operands="bf16")
    assert not eng.g8u and eng.act[1]["u"].dtype == torch.bfloat16
    eng.begin_local_update("art", steps_per_epoch=80)
    for s in range(80):
        eng.train_step(_dev(O.synthetic_batch(32, 384, 8000 + s)), use_graph=True)
    sd = eng.state_dict()
    keys = [k.split("::", 2)[2] for k in g if k.startswith("s80::dsamp::")]
    r = dict(g=g, keys=keys, snaps={80: {k: (sd[k].cpu() - P0[k]) for k in keys}})
    t = _table(_vs_golden(r, 80))
    print(f"B=32, 80 steps, bf16 u: adapters max |ddW| {t['adapters']['max']:.2e}, mean ratio {t['adapters']['ratio']:.4f}, norm "
          f"{t['adapters']['norm']:.4f}")
    assert t["adapters"]["max"] < 1.2e-3 and t["adapters"]["ratio"] < 0.065 and t["adapters"]["norm"] < 0.022
