"""N > 1 path on CPU: world_size-2 gloo run of the FedAvg exchange (pre-scale, all-reduce(SUM), write-back)
against the oracle's sequential get_average_net (main.py:50-65)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import feddat_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nums, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from feddat_amd.fedavg import allreduce_flat
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(894528, generator=g)          # one client's adapter_1 payload (48 tensors, ViLT)
    mine = flat.clone()
    buf = torch.empty_like(flat)

    def host_prescale(acc, x, num, total):           # stand-in for feddat_fedavg_accumulate (same op order)
        acc.copy_(x * num / total)
    allreduce_flat(flat, buf, nums[rank], float(sum(nums)), prescale=host_prescale)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    if rank == 0:
        server = {"adapter_1.w": torch.zeros_like(mine)}
        O.get_average_net(server, [{"adapter_1.w": t} for t in gathered], nums)
        out.put(float((flat - server["adapter_1.w"]).abs().max()))
    dist.destroy_process_group()


def test_allreduce_average_two_clients_matches_sequential_fedavg():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    nums = [1.0, 3.0]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nums, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # ring order differs from the reference's sequential order only in the last bit
    assert q.get(timeout=10) < 1e-6


def test_flat_payload_layout_matches_reference_state_dict_order():
    """The all-reduced flat buffer must be the concatenation of the reference's comm_state_dict_names tensors
    (main.py:160-163) in state-dict order: 12 x {down.weight [48,768], down.bias [48], up.weight [768,48], up.bias [768]}."""
    from feddat_amd import vilt_spec
    shapes = vilt_spec.param_shapes(12, ["art"])
    names = [k for k in shapes if "adapter_1" in k]
    assert len(names) == 48
    total = sum(int(torch.tensor(shapes[k]).prod()) for k in names)
    assert total == 894528
    d = O.ViltDims(layers=12)
    assert names == [k for k in O.param_shapes(d, ["art"]) if "adapter_1" in k]
