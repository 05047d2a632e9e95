"""FedAvg of the shared adapter (reference: get_average_net, src/train/main.py:50-65).

Two forms:
  * get_average_net(server, c_models, nums, ordered_tasks, device): drop-in for the reference's function for
    clients that live on ONE device (the reference visits clients sequentially): accumulates
    net[key] * num / total in client order with the HIP kernel -- bit-exact with the reference's loop.
  * allreduce_average(engine, world): one client per GPU: pre-scale the flat adapter_1 buffer by num/total on
    device, ONE RCCL all-reduce(SUM) over xGMI (torch.distributed backend "nccl" == RCCL), write back, refresh the
    bf16 operand copies.  Replaces K x 48 tiny host-driven kernels and the CPU-side loop by one 3.58 MB collective.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch

from . import lib as L


def get_average_net(server, c_models: List[Dict[str, torch.Tensor]], nums: Sequence[float], ordered_tasks=None,
                    device=None):
    """server: object with .comm_state_dict_names and .state_dict() (e.g. feddat_amd.modeling.ViltContinualLearner)
    or a plain dict name -> device tensor.  Keys containing 'clf' are skipped (main.py:54)."""
    sd = server if isinstance(server, dict) else server.state_dict()
    names = list(sd.keys()) if isinstance(server, dict) else list(server.comm_state_dict_names)
    total = float(sum(nums))
    for key in names:
        if "clf" in key:
            continue
        dst = sd[key]
        acc = torch.empty_like(dst, dtype=torch.float32)
        for k, (net, num) in enumerate(zip(c_models, nums)):
            L.fedavg_accumulate(acc, net[key].to(dst.device, torch.float32).contiguous(), float(num), total, k == 0)
        dst.copy_(acc)
    if hasattr(server, "after_load"):
        server.after_load()
    return server


def all_reduce_sum(t: torch.Tensor):
    """In-place SUM over all ranks.  Backend "nccl" (== RCCL): on the device buffer, over xGMI.  Any other backend (gloo:
    the single-GPU / CPU test rigs) is staged through the host when the tensor lives on the device."""
    import torch.distributed as dist
    if dist.get_backend() == "nccl" or not t.is_cuda:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    else:
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        t.copy_(host)
    return t


def allreduce_flat(flat: torch.Tensor, buf: torch.Tensor, num: float, total: float, prescale=None):
    """buf = flat * num / total (device kernel, reference operation order) ; all_reduce(SUM) ; flat <- buf.
    `prescale(acc, x, num, total)` defaults to the HIP kernel; the world_size-2 gloo test on CPU injects a host
    stand-in so that rendezvous, collective and write-back are exercised without a GPU."""
    (prescale or (lambda acc, x, n, t: L.fedavg_accumulate(acc, x, n, t, True)))(buf, flat, num, total)
    all_reduce_sum(buf)
    flat.copy_(buf)
    return flat


def make_rccl_comm(world: int, rank: int):
    """RCCL communicator through the C ABI (feddat_comm_*); the unique id travels over the already-initialised
    torch.distributed group (any backend) as a host object."""
    import torch.distributed as dist
    # every rank says whether it can bind RCCL BEFORE anyone enters the collective ncclCommInitRank: a rank without a loadable
    # librccl makes all ranks fail fast here, together, instead of its peers waiting out the bootstrap watchdog
    flags = [None] * world
    try:                    # the gather below is a collective: EVERY rank must reach it, whatever went wrong locally
        mine = bool(L.rccl_available())
    except Exception:       # noqa: BLE001 -- e.g. the library itself is missing on this rank
        mine = False
    dist.all_gather_object(flags, mine)
    if not all(flags):
        raise L.FeddatHipError(f"RCCL is not loadable on rank(s) {[r for r, f in enumerate(flags) if not f]}")

    def exchange(ident: bytes) -> bytes:
        box = [ident]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    return L.RcclComm(world, rank, exchange)


def allreduce_average(engine, world: int, num: float = 1.0, total: float = None, comm=None):
    """One client per GPU: adapter_1 <- sum_k adapter_1[k] * num_k / total.  With `comm` (L.RcclComm) the whole exchange is
    the C-ABI call feddat_fedavg_allreduce; otherwise torch.distributed issues the all-reduce (backend "nccl" == RCCL)."""
    flat = engine.comm_flat()
    if not hasattr(engine, "_fedavg_buf"):
        engine._fedavg_buf = torch.empty_like(flat)
    total = float(world) if total is None else float(total)
    if comm is not None:
        comm.fedavg_allreduce(flat, engine._fedavg_buf, num, total)
    else:
        allreduce_flat(flat, engine._fedavg_buf, num, total)
    engine.repack_adapter(1)
    return flat
