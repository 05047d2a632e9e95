"""Pins oracle/albef_oracle.py (CPU restatement of the ALBEF dual-adapter path) against fixtures captured from the
reference's own modules (oracle/make_albef_golden.py).  CPU only."""
import numpy as np
import torch

from oracle import albef_oracle as A
from tests.golden_util import load

torch.set_num_threads(8)
SMALL = dict(vit_depth=2, enc_layers=3, fusion_layer=1, dec_layers=2, image=64, vocab=3072, max_pos=64)


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_g10_small_forward_modes_and_rank_answer(golden_dir):
    g = load(golden_dir, "g10_albef_small.npz")
    d = A.AlbefDims(**SMALL)
    P = A.make_params(d)
    b0 = A.synthetic_batch(3, d, 500, q_len=12, a_len=5, k=[2, 1, 3], ragged=True)
    with torch.no_grad():
        img = A.vit_forward(P, d, b0["image"], "gating")
        assert (img - T(g["fwd.gating.image_embeds"])).abs().max() < 2e-5
        for mode in ("gating", "adapter_1", "adapter_0"):
            loss, logits = A.albef_train_forward(P, d, b0, mode)
            assert abs(float(loss) - float(g[f"fwd.{mode}.loss"])) < 1e-4 * float(loss)
            assert (logits - T(g[f"fwd.{mode}.logits"])).abs().max() < 1e-4
        ev = A.synthetic_batch(3, d, 501, q_len=12, a_len=5, k=[4, 4, 3], ragged=True)
        eb = dict(b0, answer_list_ids=ev["answer_ids"], answer_list_mask=ev["answer_mask"])
        ids, probs = A.albef_eval_forward(P, d, eb, 4, "gating")
        assert np.array_equal(ids.numpy(), g["eval.topk_ids"])
        assert (probs - T(g["eval.topk_probs"])).abs().max() < 1e-5


def test_g10_small_local_update_vs_reference(golden_dir):
    """4 train_steps (P0 / P1 / P2 with the vocabulary-axis KL, two AdamW + scheduler ticks per batch): losses and the
    UPDATE of every adapter_0 / adapter_1 tensor of all three towers."""
    from tests.golden_util import delta_vs_golden
    g = load(golden_dir, "g10_albef_small.npz")
    d = A.AlbefDims(**SMALL)
    P = A.make_params(d)
    init = {k: v.clone() for k, v in P.items()}
    c = A.AlbefDatClient(P, d, lr=1e-4, steps_per_epoch=4)
    losses = [float(c.train_step(A.synthetic_batch(3, d, 510 + s, q_len=12, a_len=5, k=[2, 1, 3], ragged=True)))
              for s in range(4)]
    assert np.abs(np.array(losses) - g["losses"]).max() < 2e-4 * np.abs(g["losses"]).max()
    keys = [k.split("::", 1)[1] for k in g if k.startswith("dsamp::")]
    assert len(keys) == 2 * 4 * (SMALL["vit_depth"] + SMALL["enc_layers"] + SMALL["dec_layers"])
    for k in keys:
        dw = (P[k] - init[k]).flatten()
        idx = torch.linspace(0, dw.numel() - 1, min(512, dw.numel())).long()
        err = (dw[idx] - T(g["dsamp::" + k])).abs()
        assert float(err.max()) < 3e-5 and float(err.mean()) <= 0.02 * float(g["dmean::" + k]) + 1e-9, k


def test_g11_round_of_40_steps_vs_reference(golden_dir):
    """A round at realistic length (40 train_steps, schedule past its warm-up) against the reference's own run (G11)."""
    g = load(golden_dir, "g11_albef_round40.npz")
    steps = int(g["steps"])
    d = A.AlbefDims(**SMALL)
    P = A.make_params(d)
    init = {k: v.clone() for k, v in P.items()}
    c = A.AlbefDatClient(P, d, lr=1e-4, steps_per_epoch=steps, num_epochs=1)
    losses = [float(c.train_step(A.synthetic_batch(3, d, 700 + s, q_len=12, a_len=5, k=[2, 1, 3], ragged=True)))
              for s in range(steps)]
    assert np.abs(np.array(losses) - g["losses"]).max() < 5e-4 * np.abs(g["losses"]).max()
    keys = [k.split("::", 1)[1] for k in g if k.startswith("dsamp::")]
    for k in keys:
        dw = (P[k] - init[k]).flatten()
        idx = torch.linspace(0, dw.numel() - 1, min(512, dw.numel())).long()
        err = (dw[idx] - T(g["dsamp::" + k])).abs()
        assert float(err.max()) < 1e-4 and float(err.mean()) <= 0.02 * float(g["dmean::" + k]) + 1e-9, (k, float(err.max()))
        assert abs(float(dw.norm()) - float(g["dnorm::" + k])) <= 0.02 * float(g["dnorm::" + k]), k
