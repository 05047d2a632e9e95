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


def _albef_worker(rank, world, port, nums, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from feddat_amd import albef_spec
    from feddat_amd.fedavg import allreduce_flat
    shapes = albef_spec.param_shapes()
    names = [k for k in shapes if "adapter_1" in k]                 # main.py:160-163: the communicated keys, state-dict order
    g = torch.Generator().manual_seed(500 + rank)
    mine = {k: torch.randn(shapes[k], generator=g) * 0.02 for k in names}
    flat = torch.cat([mine[k].flatten() for k in names])           # the engine's flat adapter_1 buffer (AlbefDatEngine.comm_flat)
    buf = torch.empty_like(flat)

    def host_prescale(acc, x, num, total):           # stand-in for feddat_fedavg_accumulate (same op order)
        acc.copy_(x * num / total)
    allreduce_flat(flat, buf, nums[rank], float(sum(nums)), prescale=host_prescale)
    sent = torch.cat([mine[k].flatten() for k in names])
    gathered = [torch.empty_like(sent) for _ in range(world)]
    dist.all_gather(gathered, sent)
    if rank == 0:
        def unflat(t):
            o, d = 0, {}
            for k in names:
                n = mine[k].numel()
                d[k] = t[o:o + n].view(shapes[k]).clone()
                o += n
            return d
        clients = [unflat(t) for t in gathered]
        server = {k: torch.zeros(shapes[k]) for k in names}
        O.get_average_net(server, clients, nums)
        got = unflat(flat)
        out.put((len(names), int(flat.numel()), max(float((got[k] - server[k]).abs().max()) for k in names)))
    dist.destroy_process_group()


def test_albef_exchange_two_clients_matches_sequential_fedavg():
    """configs[3]'s exchange: the 120 adapter_1 tensors of ALBEF's 30 adapter modules (12 ViT blocks, 12 text-encoder and 6
    decoder layers: albef.py:139-147 via main.py:160-163) as ONE flat 2 236 320-float (8.95 MB) all-reduce between two ranks,
    against the reference's sequential per-key get_average_net (main.py:50-65) with unequal client weights."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    nums = [2.0, 1.0]
    procs = [ctx.Process(target=_albef_worker, args=(r, 2, port, nums, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    n_tensors, n_floats, err = q.get(timeout=10)
    assert (n_tensors, n_floats) == (120, 2236320)
    assert err < 1e-7
