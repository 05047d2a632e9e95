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


def test_g12_train_mode_dropout_vs_reference(golden_dir):
    """The reference under model.train() with its dropout probabilities (0.1 at the embeddings, the attention probabilities,
    BertSelfOutput of self- and cross-attention, BertOutput ahead of the adapter: xbert.py:216,333,360,440), its nn.Dropout
    masks replaced by the counter-based function both sides share (oracle/make_albef_golden.py install_counter_dropout).
    The oracle must place its 25 sites where the reference's modules fire, draw independent masks for P0 / P1 / P2 and scale
    by 1 / (1 - p): losses and the update of every adapter tensor after 3 train_steps."""
    g = load(golden_dir, "g12_albef_dropout.npz")
    steps, p, seed = int(g["steps"]), float(g["p"]), int(g["seed"])
    d = A.AlbefDims(**SMALL)
    P = A.make_params(d)
    init = {k: v.clone() for k, v in P.items()}
    c = A.AlbefDatClient(P, d, lr=1e-4, steps_per_epoch=steps, num_epochs=1, dropout=p, seed=seed)
    losses = [float(c.train_step(A.synthetic_batch(3, d, 900 + s, q_len=12, a_len=5, k=[2, 1, 3], ragged=True)))
              for s in range(steps)]
    assert np.abs(np.array(losses) - g["losses"]).max() < 2e-4 * np.abs(g["losses"]).max(), (losses, g["losses"])
    # and the masks matter: the same round without dropout gives visibly different losses
    P2 = A.make_params(d)
    c2 = A.AlbefDatClient(P2, d, lr=1e-4, steps_per_epoch=steps, num_epochs=1)
    l2 = [float(c2.train_step(A.synthetic_batch(3, d, 900 + s, q_len=12, a_len=5, k=[2, 1, 3], ragged=True))) for s in range(steps)]
    assert np.abs(np.array(l2) - g["losses"]).max() > 1e-2 * np.abs(g["losses"]).max()
    for k in [k.split("::", 1)[1] for k in g if k.startswith("dsamp::")]:
        dw = (P[k] - init[k]).flatten()
        idx = torch.linspace(0, dw.numel() - 1, min(512, dw.numel())).long()
        err = (dw[idx] - T(g["dsamp::" + k])).abs()
        assert float(err.max()) < 3e-5 and float(err.mean()) <= 0.02 * float(g["dmean::" + k]) + 1e-9, (k, float(err.max()))


def test_dropout_mask_statistics():
    """Keep rate, independence across sites / passes / steps of the counter-based mask."""
    n, p = 1 << 20, 0.1
    k = A.dropout_keys(77, 1, A.dropout_site("text_encoder.encoder.layer.3.attention.", "self_probs"))
    m = A.dropout_keep(n, p, k[0], k[1], 5).float()
    assert abs(float(m.mean()) - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5
    for other in (A.dropout_keys(77, 2, A.dropout_site("text_encoder.encoder.layer.3.attention.", "self_probs")),
                  A.dropout_keys(77, 1, A.dropout_site("text_encoder.encoder.layer.4.attention.", "self_probs")),
                  A.dropout_keys(78, 1, A.dropout_site("text_encoder.encoder.layer.3.attention.", "self_probs"))):
        m2 = A.dropout_keep(n, p, other[0], other[1], 5).float()
        corr = float(((m - m.mean()) * (m2 - m2.mean())).mean() / (m.std() * m2.std()))
        assert abs(corr) < 6e-3, corr
    m3 = A.dropout_keep(n, p, k[0], k[1], 6).float()                 # next train_step
    assert abs(float(((m - m.mean()) * (m3 - m3.mean())).mean() / (m.std() * m3.std()))) < 6e-3
    # neighbouring elements are uncorrelated too (lag-1 autocorrelation)
    assert abs(float(((m[1:] - m.mean()) * (m[:-1] - m.mean())).mean() / m.var())) < 6e-3
