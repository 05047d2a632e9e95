"""Pins oracle/feddat_oracle.py (the CPU restatement) against fixtures captured from the
reference's own modules (oracle/make_golden.py). CPU only."""
import numpy as np
import pytest
import torch

from oracle import feddat_oracle as O
from tests.golden_util import load, max_abs_diff_vs_golden

torch.set_num_threads(8)

# AdamW turns fp32 re-association noise on near-zero gradients into O(lr) differences (the
# normalised update m/(sqrt(v)+eps) is +-1 whatever |g| is); HF permutes the image patches
# randomly on every forward, so the reference itself is only reproducible to that level.
TOL_W = 3e-5      # a few elements may differ by ~lr_t (<= 1e-4 * t / warmup)


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_g1_adapter_single_and_gated(golden_dir):
    g = load(golden_dir, "g1_adapter.npz")
    x, dy = T(g["x"]), T(g["dy"])
    par = {a: tuple(T(g[f"p.adapter_{a}_{n}"]).requires_grad_(True)
                    for n in ("down.weight", "down.bias", "up.weight", "up.bias")) for a in range(3)}
    xi = x.clone().requires_grad_(True)
    y = O.adapter_single(xi, xi, *par[1])
    y.backward(dy)
    assert (y - T(g["adapter_1.y"])).abs().max() < 1e-5
    assert (xi.grad - T(g["adapter_1.dx"])).abs().max() < 1e-5
    for t, n in zip(par[1], ("down.weight", "down.bias", "up.weight", "up.bias")):
        assert (t.grad - T(g[f"adapter_1.d.adapter_1_{n}"])).abs().max() < 2e-4
    for a in par:
        for t in par[a]:
            t.grad = None
    xi = x.clone().requires_grad_(True)
    y = O.adapter_gated(xi, xi, par[0], par[2])
    y.backward(dy)
    assert (y - T(g["gating.y"])).abs().max() < 1e-5
    assert (xi.grad - T(g["gating.dx"])).abs().max() < 1e-5
    for t, n in zip(par[0], ("down.weight", "down.bias", "up.weight", "up.bias")):
        assert (t.grad - T(g[f"gating.d.adapter_0_{n}"])).abs().max() < 2e-4


def test_g2_loss(golden_dir):
    g = load(golden_dir, "g2_loss.npz")
    lg = T(g["logits"]).requires_grad_(True)
    L = O.dat_loss(lg, T(g["target"]), T(g["teacher"]))
    L.backward()
    assert abs(float(L.detach()) - float(g["L"])) < 1e-4
    assert abs(float(O.kl_loss(lg, T(g["teacher"]))) - float(g["kl"])) < 1e-5
    assert (lg.grad - T(g["dlogits"])).abs().max() < 1e-6


@pytest.mark.parametrize("res", [224, 384])
def test_g3_two_layer_forward_and_steps(golden_dir, res):
    g = load(golden_dir, f"g3_vilt2_{res}.npz")
    d = O.ViltDims(layers=2)
    P = O.make_params(d, ["art", "gqa"], bias_std=0.02)
    batches = [O.synthetic_batch(4, res, 1234 + s) for s in range(5)]
    with torch.no_grad():
        for mode in ("gating", "adapter_1", "adapter_0"):
            pooled, lg = O.vilt_forward(P, d, batches[0], mode, "art")
            # HF permutes patches (fp32 re-association only): SURVEY.md 8a, measured 9.5e-7
            assert (pooled - T(g[f"fwd.{mode}.pooled"])).abs().max() < 5e-6
            assert (lg - T(g[f"fwd.{mode}.logits"])).abs().max() < 5e-5
    client = O.DatClient(P, d, "art", lr=1e-4, steps_per_epoch=5)
    losses = []
    for s, b in enumerate(batches):
        losses.append(float(client.train_step(b)[0]))
        if s + 1 in (1, 2, 5):
            keys = [k[len(f"after{s+1}."):] for k in g if k.startswith(f"after{s+1}.")]
            keys += [k.split("::", 1)[1][len(f"after{s+1}."):] for k in g
                     if k.startswith("samp::after%d." % (s + 1))]
            assert keys
            for k in keys:
                assert max_abs_diff_vs_golden(g, f"after{s+1}.{k}", P[k]) < TOL_W, k
    assert np.allclose(losses, g["losses"], rtol=2e-5, atol=1e-4), (losses, g["losses"])


def test_g3q_optimizer_membership_follows_flags(golden_dir):
    """Reference quirk: after eval() the server model is left with adapter_0.requires_grad=False,
    so later rounds' optimizers never update adapter_0 (task_trainer.py:236-244)."""
    g = load(golden_dir, "g3q_flags.npz")
    d = O.ViltDims(layers=2)
    P = O.make_params(d, ["art", "gqa"], bias_std=0.02)
    P0 = {k: v.clone() for k, v in P.items()}
    batches = [O.synthetic_batch(4, 224, 1234 + s) for s in range(3)]
    client = O.DatClient(P, d, "art", lr=1e-4, steps_per_epoch=3, opt_adapters=(1,))
    losses = [float(client.train_step(b)[0]) for b in batches]
    assert np.allclose(losses, g["losses"], rtol=2e-5, atol=1e-4)
    for k in [k[len("after3."):] for k in g if k.startswith("after3.")]:
        assert max_abs_diff_vs_golden(g, "after3." + k, P[k]) < TOL_W, k
        if "adapter_0" in k:
            assert torch.equal(P[k], P0[k])


def test_g15_gradscaler_skip_semantics(golden_dir):
    """What the reference does when a scaled fp16 backward overflows, pinned on accelerate's own AcceleratedOptimizer /
    AcceleratedScheduler + torch.amp.GradScaler around the reference's train_step (oracle/make_golden.py: golden_g15, overflow
    injected into sub-step A of step 2, B of step 4, both of step 5): the oracle's train_step(overflow=(A, B)) -- skip that
    sub-step's optimizer AND scheduler step, nothing else -- reproduces the weights and the scheduler index."""
    g = load(golden_dir, "g15_scaler_skip.npz")
    d = O.ViltDims(layers=2)
    P = O.make_params(d, ["art"], bias_std=0.02)
    steps = len(g["losses"])
    ovf = {int(s): tuple(bool(x) for x in ab) for s, ab in zip(g["overflow_steps"], g["overflow_ab"])}
    client = O.DatClient(P, d, "art", lr=1e-4, steps_per_epoch=steps)
    losses, sched = [], []
    for s in range(steps):
        losses.append(float(client.train_step(O.synthetic_batch(4, 224, 1500 + s), overflow=ovf.get(s, (False, False)))[0]))
        sched.append(client.sched_t)
    assert sched == [int(x) for x in g["sched_t"]], (sched, g["sched_t"])
    assert np.allclose(losses, g["losses"], rtol=2e-5, atol=1e-4), (losses, g["losses"])
    keys = [k[len("after."):] for k in g if k.startswith("after.")] + \
           [k.split("::", 1)[1][len("after."):] for k in g if k.startswith("samp::after.")]
    assert len(keys) > 20
    for k in keys:
        assert max_abs_diff_vs_golden(g, "after." + k, P[k]) < TOL_W, k
    # GradScaler's own bookkeeping, for the record of what the device-side scaler mirrors: x0.5 per overflowed sub-step
    assert [float(x) for x in g["scale"]] == [65536.0, 65536.0, 32768.0, 32768.0, 16384.0, 4096.0, 4096.0]


def test_g5_fedavg_and_round(golden_dir):
    g = load(golden_dir, "g5_fedavg.npz")
    keys = [k[4:] for k in g if k.startswith("avg.")]
    cms = [{k: T(g[f"c{i}.{k}"]) for k in keys} for i in range(5)]
    server = {k: torch.zeros_like(cms[0][k]) for k in keys}
    O.get_average_net(server, cms, list(g["nums"]))
    for k in keys:
        assert torch.equal(server[k], T(g["avg." + k])), k   # same op order -> bit-exact

    r = load(golden_dir, "g5_round.npz")
    d = O.ViltDims(layers=2)
    tasks = ["art", "gqa"]
    server = O.make_params(d, tasks, bias_std=0.02)
    personal = {t: {n: server[n].clone() for n in O.personal_names(server)} for t in tasks}
    batches = {t: [O.synthetic_batch(4, 224, 777 + 10 * ci + s) for s in range(3)] for ci, t in enumerate(tasks)}
    server, personal = O.fl_round(server, personal, d, tasks, batches, lr=1e-4)
    for k in [k[7:] for k in r if k.startswith("server.")]:
        assert (server[k] - T(r["server." + k])).abs().max() < TOL_W, k
    for t in tasks:
        for k in [k for k in r if k.startswith(f"personal.{t}.")]:
            n = k[len(f"personal.{t}."):]
            assert (personal[t][n] - T(r[k])).abs().max() < TOL_W, k


def test_g4_full_12_layer(golden_dir):
    g = load(golden_dir, "g4_vilt12_384.npz")
    d = O.ViltDims(layers=12)
    P = O.make_params(d, ["art"], bias_std=0.02)
    batches = [O.synthetic_batch(4, 384, 4321 + s) for s in range(4)]
    with torch.no_grad():
        for mode in ("gating", "adapter_1"):
            pooled, lg = O.vilt_forward(P, d, batches[0], mode, "art")
            assert (pooled - T(g[f"fwd.{mode}.pooled"])).abs().max() < 1e-5
            assert (lg - T(g[f"fwd.{mode}.logits"])).abs().max() < 1e-4
    client = O.DatClient(P, d, "art", lr=1e-4, steps_per_epoch=4)
    losses = [float(client.train_step(b)[0]) for b in batches]
    assert np.allclose(losses, g["losses"], rtol=5e-5, atol=2e-4), (losses, g["losses"])
    for k in [k.split("::", 1)[1] for k in g if k.startswith("samp256::")]:
        assert max_abs_diff_vs_golden(g, k, P[k]) < TOL_W, k


G6_VALID = [(384, 384), (256, 384), (384, 224), (160, 320)]
G6_TEXT = [40, 31, 40, 12]


def test_g6_padded_images_and_questions(golden_dir):
    """pixel_mask with zeros: per-sample resized position embeddings + masked patch / text keys.  The reference (HF
    visual_embed) drops / re-draws masked patch tokens at random; the oracle keeps all of them, masked -- same
    pooled feature up to fp32 re-association."""
    g = load(golden_dir, "g6_padded.npz")
    d = O.ViltDims(layers=2)
    P = O.make_params(d, ["art"], bias_std=0.02)
    batches = [O.pad_batch(O.synthetic_batch(4, 384, 6000 + s), G6_VALID, G6_TEXT) for s in range(2)]
    with torch.no_grad():
        for mode in ("gating", "adapter_1"):
            pooled, lg = O.vilt_forward(P, d, batches[0], mode, "art")
            assert (pooled - T(g[f"fwd.{mode}.pooled"])).abs().max() < 5e-6
            assert (lg - T(g[f"fwd.{mode}.logits"])).abs().max() < 5e-5
    client = O.DatClient(P, d, "art", lr=1e-4, steps_per_epoch=2)
    losses = [float(client.train_step(b)[0]) for b in batches]
    assert np.allclose(losses, g["losses"], rtol=2e-5, atol=1e-4), (losses, g["losses"])
    for k in [k[len("after2."):] for k in g if k.startswith("after2.")]:
        assert max_abs_diff_vs_golden(g, "after2." + k, P[k]) < TOL_W, k


def test_g8_oracle_reproduces_reference_round_of_40_steps(golden_dir):
    """G8: a realistic local round (12 layers, B=4, 384x384, 40 steps, schedule past warm-up) -- the oracle against the
    reference's own run, on the weight UPDATES (they reach 2-3e-3): this pins the oracle at round length."""
    from tests.golden_util import delta_vs_golden
    g = load(golden_dir, "g8_round40.npz")
    d = O.ViltDims(layers=12)
    P = O.make_params(d, ["art"], bias_std=0.02)
    init = {k: v.clone() for k, v in P.items()}
    c = O.DatClient(P, d, "art", lr=1e-4, steps_per_epoch=int(g["steps"]))
    losses = [float(c.train_step(O.synthetic_batch(4, 384, 8000 + s))[0]) for s in range(int(g["steps"]))]
    assert np.abs(np.array(losses) - g["losses"]).max() < 1e-3
    for k in [k.split("::", 1)[1] for k in g if k.startswith("dsamp::")]:
        mx, mean, ref_mean, dnorm = delta_vs_golden(g, k, P[k] - init[k])
        assert mx < 3e-5 and mean < 0.01 * ref_mean, (k, mx, mean, ref_mean)


def test_g8_oracle_reproduces_reference_round_of_80_steps(golden_dir):
    """The longest round configs[2] names: len(loader) = 80 (1200 scheduler ticks, warm-up 60 batches, 20 batches at full
    lr; the weights move up to ~5e-3).  The oracle against the reference's own run -- this is what allows the GPU test
    (tests/test_round40_gpu.py::test_round_of_80_steps_*) to take the live oracle as the full-tensor reference."""
    from tests.golden_util import delta_vs_golden
    g = load(golden_dir, "g8_round80.npz")
    steps = int(g["steps"])
    assert steps == 80
    d = O.ViltDims(layers=12)
    P = O.make_params(d, ["art"], bias_std=0.02)
    init = {k: v.clone() for k, v in P.items()}
    c = O.DatClient(P, d, "art", lr=1e-4, steps_per_epoch=steps)
    losses = [float(c.train_step(O.synthetic_batch(4, 384, 8000 + s))[0]) for s in range(steps)]
    assert np.abs(np.array(losses) - g["losses"]).max() < 1e-3
    for k in [k.split("::", 1)[1] for k in g if k.startswith("dsamp::")]:
        mx, mean, ref_mean, dnorm = delta_vs_golden(g, k, P[k] - init[k])
        assert mx < 6e-5 and mean < 0.01 * ref_mean, (k, mx, mean, ref_mean)
