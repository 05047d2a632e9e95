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


def save_federation(out_dir: str, comm: Dict[str, torch.Tensor], personal: Dict[str, Dict[str, torch.Tensor]],
                    comm_round: int, write_server: bool = True, server_flags=None) -> None:
    """State of a whole federation after `comm_round`: the averaged adapter_1 tensors (written by one rank) and each
    local client's personal tensors (written by the rank that owns the client).  round.json is written last, so a
    directory with a round.json is complete."""
    import json
    os.makedirs(out_dir, exist_ok=True)
    cpu = lambda d: {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in d.items()}
    for task_key, sd in personal.items():
        save_file(cpu(sd), os.path.join(out_dir, f"personal_{task_key}.safetensors"))
    if write_server:
        save_file(cpu(comm), os.path.join(out_dir, "server_adapter.safetensors"))
        with open(os.path.join(out_dir, "round.json"), "w") as f:
            json.dump({"comm_round": int(comm_round),
                       "server_adapter_requires_grad": {str(k): bool(v) for k, v in (server_flags or {}).items()}}, f)


def load_federation(out_dir: str, tasks):
    """-> (comm dict, {task: personal dict}, last finished round, server requires_grad flags {adapter index: bool}).
    Raises FileNotFoundError on an incomplete dir."""
    import json
    with open(os.path.join(out_dir, "round.json")) as f:
        meta = json.load(f)
    comm_round = int(meta["comm_round"])
    flags = {int(k): bool(v) for k, v in meta.get("server_adapter_requires_grad", {}).items()}
    comm = dict(load_file(os.path.join(out_dir, "server_adapter.safetensors")))
    personal = {t: dict(load_file(os.path.join(out_dir, f"personal_{t}.safetensors"))) for t in tasks}
    return comm, personal, comm_round, flags
