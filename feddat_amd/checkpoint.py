"""Persisted adapter / personal-parameter checkpoints.

The reference's FL path never writes a checkpoint (main.py only creates directories, main.py:511-512); its only
"format" is the in-memory client -> server payload {state_dict key: fp32 tensor} for keys containing 'adapter_1'
(main.py:499-503).  These helpers persist exactly those dictionaries -- reference key names and shapes -- as
safetensors files so that rounds can be resumed: <dir>/server_adapter.safetensors (communicated tensors) and
<dir>/personal_<task>.safetensors (task head + adapter_0 + adapter_2, main.py:154,440-450)."""
from __future__ import annotations

import os
from typing import Dict

import torch
from safetensors.torch import load_file, save_file


def save_round(model, out_dir: str, task_key: str):
    os.makedirs(out_dir, exist_ok=True)
    sd = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in model.state_dict().items()}
    comm = {k: v for k, v in sd.items() if k in set(model.comm_state_dict_names)}
    personal = {k: v for k, v in sd.items()
                if (k.startswith(f"task_layer.{task_key}.") or "adapter_0" in k or "adapter_2" in k)}
    save_file(comm, os.path.join(out_dir, "server_adapter.safetensors"))
    save_file(personal, os.path.join(out_dir, f"personal_{task_key}.safetensors"))
    return sorted(comm), sorted(personal)


def load_round(model, out_dir: str, task_key: str) -> Dict[str, torch.Tensor]:
    sd = dict(load_file(os.path.join(out_dir, "server_adapter.safetensors")))
    p = os.path.join(out_dir, f"personal_{task_key}.safetensors")
    if os.path.exists(p):
        sd.update(load_file(p))
    model.load_state_dict(sd)
    return sd
