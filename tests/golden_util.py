"""Helpers for reading the fixtures written by oracle/make_golden.py."""
import os

import numpy as np
import torch

N_SAMPLES = 2048


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def max_abs_diff_vs_golden(rec, key, t: torch.Tensor) -> float:
    """Compare tensor `t` with the golden entry `key` (whole tensor or norm+samples form)."""
    t = t.detach().float().cpu()
    if key in rec:
        return float((t - torch.from_numpy(rec[key]).reshape(t.shape)).abs().max())
    flat = t.flatten()
    for tag, n in (("samp::", N_SAMPLES), ("samp256::", 256)):
        if tag + key in rec:
            idx = torch.linspace(0, flat.numel() - 1, n).long()
            d = float((flat[idx] - torch.from_numpy(rec[tag + key])).abs().max())
            dn = abs(float(flat.norm()) - float(rec["norm::" + key]))
            return max(d, dn / max(1.0, flat.numel() ** 0.5))
    raise KeyError(key)


def delta_vs_golden(rec, key, dw: torch.Tensor):
    """Fixtures that store the UPDATE dW = W_after - W_init (g8): returns (max |dW - dW_ref| over the stored samples,
    mean |dW - dW_ref| over the samples, the reference's mean |dW| over the whole tensor, | ||dW|| - ||dW_ref|| |)."""
    flat = dw.detach().float().cpu().flatten()
    ref = torch.from_numpy(rec["dsamp::" + key])
    idx = torch.linspace(0, flat.numel() - 1, min(1024, flat.numel())).long()
    d = (flat[idx] - ref).abs()
    return float(d.max()), float(d.mean()), float(rec["dmean::" + key]), abs(float(flat.norm()) - float(rec["dnorm::" + key]))


def assert_update_parity(names, got, ref, init, max_tol=1e-3, rel_mean=0.1, what=""):
    """Parity of the weight UPDATE, not of the weight: for every tensor, with dW = W - W_init,
         |dW_got - dW_ref|.max()  <  max_tol                     (BASELINE.json north_star: 1e-3)
         |dW_got - dW_ref|.mean() <= rel_mean * |dW_ref|.mean()  (has teeth: the tensors move only ~1e-4..1e-3 in the few
                                                                  steps a test runs, far less than 1e-3)
       and a tensor the reference moved must have moved here too.  Returns (worst max diff, worst mean ratio)."""
    worst_max, worst_ratio = 0.0, 0.0
    for n in names:
        w0 = init[n].detach().float().cpu()
        d_ref = ref[n].detach().float().cpu() - w0
        d_got = got[n].detach().float().cpu() - w0
        err = (d_got - d_ref).abs()
        move = float(d_ref.abs().mean())
        assert float(err.max()) < max_tol, (what, n, "max", float(err.max()))
        assert float(err.mean()) <= rel_mean * move + 1e-10, (what, n, "mean", float(err.mean()), "moved", move)
        if move > 0:
            assert float(d_got.abs().max()) > 0, (what, n, "did not move")
            worst_ratio = max(worst_ratio, float(err.mean()) / move)
        worst_max = max(worst_max, float(err.max()))
    return worst_max, worst_ratio


def golden_tensor(rec, key):
    """Whole-tensor golden entry or None (large tensors are stored as norm + samples only)."""
    return torch.from_numpy(rec[key]) if key in rec else None


def sampled_update_parity(rec, prefix, got, init, n_samples, max_tol=1e-3, rel_mean=0.1):
    """assert_update_parity for fixtures that hold W_after as `samp::`/`samp256::` strided samples: the update is formed on
    the sampled elements (W_init is regenerated from the name-seeded fill)."""
    worst_max, worst_ratio = 0.0, 0.0
    tag = {2048: "samp::", 256: "samp256::"}[n_samples]
    for gk in [k for k in rec if k.startswith(tag + prefix)]:
        key = gk[len(tag) + len(prefix):]
        flat = got[key].detach().float().cpu().flatten()
        idx = torch.linspace(0, flat.numel() - 1, n_samples).long()
        w0 = init[key].detach().float().cpu().flatten()[idx]
        d_ref = torch.from_numpy(rec[gk]) - w0
        err = ((flat[idx] - w0) - d_ref).abs()
        move = float(d_ref.abs().mean())
        assert float(err.max()) < max_tol, (key, "max", float(err.max()))
        assert float(err.mean()) <= rel_mean * move + 1e-10, (key, "mean", float(err.mean()), "moved", move)
        if move > 0:
            worst_ratio = max(worst_ratio, float(err.mean()) / move)
        worst_max = max(worst_max, float(err.max()))
    return worst_max, worst_ratio
