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
