"""The dynamic loss scale of the fp16-operand engine: torch.cuda.amp.GradScaler as accelerate drives it for the reference's
mixed_precision fp16 (src/accelerate_config.yaml:8; accelerator.backward / optimizer.step / scheduler.step in
task_trainer.py:302-308,323-328), but ON THE DEVICE and inside the captured step (feddat_adapter_wgrad_reduce_checked,
feddat_dat_loss_fwd_bwd_checked, feddat_adamw_group.skip_if / bak / restore_if, feddat_dat_step_finish).

  * nothing overflows -> bit-identical with the static scale, eager and hipGraph;
  * an overflow in sub-step B alone -> exactly GradScaler: B's optimizer + scheduler step skipped, everything else stands
    (the oracle's train_step(overflow=(False, True)), which tests/test_oracle_golden.py::test_g15 pins on accelerate's own
    wrappers around the reference);
  * an overflow in sub-step A -> the whole batch is void (B's forward has already used A's head update);
    the head returns BIT-EXACTLY to its state before the step; = the oracle's overflow=(True, True);
  * a scale far too large for fp16 backs off by itself, no update is ever non-finite, and the run continues on the oracle's
    trajectory with the same skips; clean sub-steps grow the scale again."""
import pytest
import torch

from oracle import feddat_oracle as O
from tests.golden_util import assert_update_parity

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _needs_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _dev(b):
    return {k: v.to(DEV) for k, v in b.items()}


def _engine(layers=2, batch=3, **kw):
    from feddat_amd import engine
    d = O.ViltDims(layers=layers)
    P = O.make_params(d, ["art"], bias_std=0.02)
    eng = engine.ViltDatEngine(P, ["art"], DEV, batch=batch, res=224, layers=layers, operands="f16", **kw)
    return d, P, eng


def _names(P):
    return O.trainable_names(P, "art", 0) + [n for n in O.trainable_names(P, "art", 1) if "adapter_1" in n]


@pytest.mark.parametrize("use_graph", [False, True])
def test_no_overflow_is_bit_identical_to_the_static_scale(use_graph):
    steps = 4
    out = {}
    for dyn in (False, True):
        d, P, eng = _engine(dynamic_loss_scale=dyn)
        assert eng.scaler_state()["dynamic"] == dyn
        eng.begin_local_update("art", steps_per_epoch=steps)
        for s in range(steps):
            eng.train_step(_dev(O.synthetic_batch(3, 224, 300 + s)), use_graph=use_graph)
        torch.cuda.synchronize()
        out[dyn] = ({k: v.clone() for k, v in eng.state_dict().items()}, eng.scaler_state(),
                    [g.state.tolist() for g in (eng.head["art"], eng.ad[1], eng.ad[0])])
    for k in out[False][0]:
        assert torch.equal(out[False][0][k], out[True][0][k]), k
    assert out[False][2] == out[True][2] == [[2 * steps, 2 * steps], [2 * steps, steps], [2 * steps + 1, steps]]
    st = out[True][1]
    assert st["scale"] == 16384.0 and st["growth_tracker"] == 2 * steps and st["skipped_substeps"] == 0


@pytest.mark.parametrize("which,use_graph", [("B", False), ("B", True), ("A", False), ("A", True), ("AB", False)])
def test_injected_overflow_follows_gradscaler(which, use_graph):
    """The overflow flag of a sub-step is forced before step 2 (the flags are OR-ed into by the kernels and cleared by the end of
    the step, so a preset flag is an injected overflow); 5 steps; every trainable tensor against the oracle with the same skip."""
    steps, at = 5, 2
    d, P, eng = _engine()
    P0 = {k: v.clone() for k, v in P.items()}
    client = O.DatClient(P, d, "art", lr=1e-4, steps_per_epoch=steps)
    eng.begin_local_update("art", steps_per_epoch=steps)
    names = _names(P)
    hp = eng.head["art"]
    for s in range(steps):
        b = O.synthetic_batch(3, 224, 300 + s)
        ovf = (False, False)
        if s == at:
            if "B" in which:
                eng.ovf_flags[0] = 1
            if "A" in which:
                eng.ovf_flags[1] = 1
            # engine semantics: an overflow in A voids the batch
            ovf = (True, True) if "A" in which else (False, True)
            before = (hp.p.clone(), hp.m.clone(), hp.v.clone(), eng.ad[0].p.clone(), eng.ad[1].p.clone())
        ref_loss = float(client.train_step(b, overflow=ovf)[0])
        loss = float(eng.train_step(_dev(b), use_graph=use_graph)[0])
        assert abs(loss - ref_loss) < 1e-3 * abs(ref_loss) + 1e-3, (s, loss, ref_loss)
        if s == at and "A" in which:      # nothing of this batch was applied; the head is back bit-exactly
            for t0, t1 in zip(before, (hp.p, hp.m, hp.v, eng.ad[0].p, eng.ad[1].p)):
                assert torch.equal(t0, t1)
        if s == at and which == "B":      # A stands (adapter_1 and the head moved), adapter_0 did not
            assert torch.equal(before[3], eng.ad[0].p) and not torch.equal(before[4], eng.ad[1].p)
            assert not torch.equal(before[0], hp.p)
        assert_update_parity(names, eng.state_dict(), P, P0, 1e-3, 0.06, f"{which} step {s + 1}")
    st = eng.scaler_state()
    applied = 2 * steps - (1 if which == "B" else 2)
    assert st["scale"] == 8192.0 and st["skipped_batches"] == 1 and st["skipped_substeps"] == 2 * steps - applied
    assert hp.state.tolist() == [applied, applied] and client.sched_t == applied
    assert eng.ad[1].state.tolist() == [applied, steps - (0 if which == "B" else 1)]
    assert eng.ad[0].state.tolist() == [applied + 1, steps - 1]
    assert eng.ovf_flags.tolist() == [0, 0]


def test_a_scale_too_large_for_fp16_backs_off_and_training_continues():
    """Initial scale 2^32: the backbone's gradient operands overflow fp16 until the scaler has halved its way down.  No update is
    ever non-finite, the skipped batches are exactly those the device flags report, and the run lands on the oracle's trajectory
    with the same skips (the oracle's fp32 path has no scale; only the skips matter)."""
    steps = 18
    d, P, eng = _engine(loss_scale=2.0 ** 32)
    P0 = {k: v.clone() for k, v in P.items()}
    client = O.DatClient(P, d, "art", lr=1e-4, steps_per_epoch=steps)
    eng.begin_local_update("art", steps_per_epoch=steps)
    names = _names(P)
    skipped, prev, scales = [], 0, []
    for s in range(steps):
        b = O.synthetic_batch(3, 224, 300 + s)
        eng.train_step(_dev(b))
        st = eng.scaler_state()
        delta = st["skipped_substeps"] - prev
        prev = st["skipped_substeps"]
        assert delta in (0, 1, 2)
        skipped.append(delta)
        scales.append(st["scale"])
        client.train_step(b, overflow=(delta == 2, delta >= 1))
        assert not eng.nonfinite_groups()
    print("skipped sub-steps per batch", skipped, "scale", scales)
    # (measured on this 2-layer model: twelve halvings down to 2^20, where single batches still overflow now and then -- a
    #  GradScaler sits one binade under the edge by construction)
    assert skipped[0] == 2 and sum(1 for x in skipped if x) >= 4 and sum(1 for x in skipped if x == 0) >= 3
    assert scales[-1] < 2.0 ** 29 and all(a >= b for a, b in zip(scales, scales[1:]))
    eng.assert_finite()
    assert_update_parity(names, eng.state_dict(), P, P0, 1e-3, 0.06, "after back-off")
    assert eng.head["art"].state.tolist()[0] == client.sched_t


def test_clean_substeps_grow_the_scale():
    d, P, eng = _engine(scale_growth_interval=4, loss_scale=1024.0)
    eng.begin_local_update("art", steps_per_epoch=6)
    seen = []
    for s in range(6):
        eng.train_step(_dev(O.synthetic_batch(3, 224, 300 + s)), use_graph=True)
        seen.append(eng.scaler_state()["scale"])
    assert seen == [1024.0, 2048.0, 2048.0, 4096.0, 4096.0, 8192.0], seen
    eng.assert_finite()
    eng.begin_local_update("art", steps_per_epoch=6)      # a fresh scaler per local update (main.py:435: fresh Accelerator)
    assert eng.scaler_state()["scale"] == 1024.0 and eng.scaler_state()["growth_tracker"] == 0


def test_a_nonfinite_loss_skips_the_step():
    """inf in the images -> NaN logits -> NaN losses: both sub-steps are skipped by the loss check alone (the weight gradients
    are NaN as well), no parameter changes, and the next clean batch trains normally."""
    d, P, eng = _engine()
    eng.begin_local_update("art", steps_per_epoch=3)
    b = _dev(O.synthetic_batch(3, 224, 300))
    bad = dict(b)
    bad["pixel_values"] = b["pixel_values"].clone()
    bad["pixel_values"][0, 0, :8, :8] = float("inf")
    before = {k: v.clone() for k, v in eng.state_dict().items()}
    eng.train_step(bad)
    for k, v in eng.state_dict().items():
        assert torch.equal(before[k], v), k
    assert eng.scaler_state()["skipped_substeps"] == 2 and eng.head["art"].state.tolist() == [0, 0]
    eng.train_step(b)
    assert not eng.nonfinite_groups() and eng.head["art"].state.tolist() == [2, 2]
    assert not torch.equal(before["task_layer.art.clf_fc1.weight"], eng.state_dict()["task_layer.art.clf_fc1.weight"])


# ---------------------------------------------------------------------------------------------------------------- ALBEF
def _albef(**kw):
    from feddat_amd import albef_engine
    from oracle import albef_oracle as A
    from tests.test_albef_gpu import SMALL, _small_engine
    d = A.AlbefDims(**SMALL)
    P = A.make_params(d)
    eng = _small_engine(albef_engine, P, 3, 6, 12, 5, operands="f16", **kw)
    batches = [A.synthetic_batch(3, d, 510 + s, q_len=12, a_len=5, k=[2, 1, 3], ragged=True) for s in range(5)]
    return eng, [{k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in b.items()} for b in batches]


@pytest.mark.parametrize("use_graph", [False, True])
def test_albef_dynamic_scale_is_bit_identical_without_overflow_and_skips_like_the_vilt_engine(use_graph):
    """AlbefDatEngine(operands="f16"): the same device-side scaler (feddat_lm_loss_fwd_bwd_dyn, the checked reduce, predicated
    AdamW, feddat_dat_step_finish without a head).  No overflow -> bit-identical with the static scale; flag B -> adapter_0's
    sub-step alone is skipped; flag A -> nothing of the batch is applied; counters and scale follow."""
    out = {}
    for dyn in (False, True):
        eng, batches = _albef(dynamic_loss_scale=dyn)
        eng.begin_local_update(steps_per_epoch=5)
        for b in batches[:3]:
            eng.train_step(b, use_graph=use_graph)
        torch.cuda.synchronize()
        out[dyn] = {k: v.clone() for k, v in eng.state_dict().items()}
    for k in out[False]:
        assert torch.equal(out[False][k], out[True][k]), k
    eng, batches = _albef()
    assert eng.scaler_state()["dynamic"]
    eng.begin_local_update(steps_per_epoch=5)
    eng.train_step(batches[0], use_graph=use_graph)
    a0, a1 = eng.ad[0].p.clone(), eng.ad[1].p.clone()
    eng.ovf_flags[0] = 1                                     # sub-step B (adapter_0's pass) "overflowed"
    eng.train_step(batches[1], use_graph=use_graph)
    assert torch.equal(a0, eng.ad[0].p) and not torch.equal(a1, eng.ad[1].p)
    assert eng.ad[1].state.tolist() == [3, 2] and eng.ad[0].state.tolist() == [4, 1] and eng.scaler_state()["scale"] == 8192.0
    a0, a1 = eng.ad[0].p.clone(), eng.ad[1].p.clone()
    eng.ovf_flags[1] = 1                                     # sub-step A: the batch is void
    eng.train_step(batches[2], use_graph=use_graph)
    assert torch.equal(a0, eng.ad[0].p) and torch.equal(a1, eng.ad[1].p)
    assert eng.ad[1].state.tolist() == [3, 2] and eng.ad[0].state.tolist() == [4, 1] and eng.scaler_state()["scale"] == 4096.0
    eng.train_step(batches[3], use_graph=use_graph)
    assert not torch.equal(a0, eng.ad[0].p) and not torch.equal(a1, eng.ad[1].p)
    st = eng.scaler_state()
    assert st["skipped_substeps"] == 3 and st["skipped_batches"] == 2 and eng.ovf_flags.tolist() == [0, 0]
    eng.assert_finite()


def test_albef_a_scale_too_large_backs_off():
    eng, batches = _albef(loss_scale=2.0 ** 24)       # (this model's gradients overflow fp16 down to a scale of ~2^17)
    eng.begin_local_update(steps_per_epoch=16)
    skipped = []
    prev = 0
    for s in range(16):
        eng.train_step(batches[s % len(batches)])
        st = eng.scaler_state()
        skipped.append(st["skipped_substeps"] - prev)
        prev = st["skipped_substeps"]
        assert not eng.nonfinite_groups()
    print("ALBEF: skipped sub-steps per batch", skipped, "final scale", st["scale"])
    assert skipped[0] == 2 and sum(1 for x in skipped if x == 0) >= 3 and st["scale"] < 2.0 ** 22
