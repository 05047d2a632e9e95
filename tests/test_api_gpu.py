"""The drop-in mirrors of the reference's module API (feddat_amd.modeling / feddat_amd.train) -- written the way a
test of the reference itself would read: build the model, flip adapter modes, run train(), average the clients."""
import types

import pytest
import torch

from oracle import feddat_oracle as O
from tests.golden_util import assert_update_parity, load

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def api():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from feddat_amd import lib, modeling, train
    return types.SimpleNamespace(modeling=modeling, train=train, L=lib)


def _dev(b):
    return {k: v.to(DEV) for k, v in b.items()}


def test_adapter_module_modes_and_flags(api, golden_dir):
    g = load(golden_dir, "g1_adapter.npz")
    ad = api.modeling.Adapter(["adapter_0", "adapter_1", "adapter_2"], DEV)
    for n, p in ad.named_parameters():
        p.data.copy_(torch.from_numpy(g["p." + n]))
    ad.refresh()
    x = torch.from_numpy(g["x"]).to(DEV)
    ad.deactivate_gating()
    ad.set_active_adapter("adapter_1")
    assert ad.adapter_1_down.weight.requires_grad and not ad.adapter_0_down.weight.requires_grad
    assert not ad.adapter_2_up.bias.requires_grad
    y = ad(x, x)
    assert (y.cpu() - torch.from_numpy(g["adapter_1.y"])).abs().max() < 1.2e-2
    ad.activate_gating()
    ad.set_active_adapter("adapter_0")
    assert ad.adapter_0_down.weight.requires_grad and not ad.adapter_1_down.weight.requires_grad
    y = ad(x, x)
    assert (y.cpu() - torch.from_numpy(g["gating.y"])).abs().max() < 1.2e-2
    from feddat_amd.lib import FeddatHipError
    with pytest.raises(FeddatHipError):
        ad(x, x.clone())                     # only the adapter(h, h) form exists in the reference
    with pytest.raises(FeddatHipError):
        ad(x.cpu(), x.cpu())                 # no CPU path


def test_adaptered_vilt_output(api):
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(768, 3072, generator=g) * 0.02).to(DEV)
    b = (torch.randn(768, generator=g) * 0.02).to(DEV)
    layer = types.SimpleNamespace(dense=types.SimpleNamespace(weight=w, bias=b))
    mod = api.modeling.Adaptered_ViltOutput(layer, {"names": ["adapter_0", "adapter_1", "adapter_2"], "device": DEV})
    x = torch.randn(2, 50, 3072, generator=g).to(DEV)
    res = torch.randn(2, 50, 768, generator=g).to(DEV)
    mod.adapter.deactivate_gating()
    mod.adapter.set_active_adapter("adapter_1")
    y = mod(x, res)
    h = x.to(torch.bfloat16).float() @ w.to(torch.bfloat16).float().t() + b + res
    a = mod.adapter
    ref = O.adapter_single(h, h, a.adapter_1_down.weight.data, a.adapter_1_down.bias.data,
                           a.adapter_1_up.weight.data, a.adapter_1_up.bias.data)
    assert (y - ref).abs().max() < 5e-3


def test_continual_learner_train_and_fedavg_round(api, golden_dir):
    """Two clients x 3 train_steps + get_average_net, against the reference's own round (tests/golden/g5_round.npz)."""
    r = load(golden_dir, "g5_round.npz")
    tasks = ["art", "gqa"]
    d = O.ViltDims(layers=2)
    P = O.make_params(d, tasks, bias_std=0.02)
    args = types.SimpleNamespace(local_epochs=1, num_epochs=15, lr=1e-4, optimizer_mode="dat", debug=0, hip_graph=False)
    server = api.modeling.create_vilt_continual_learner_model(P, tasks, DEV, batch_size=4, image_size=224, num_layers=2)
    assert len(server.comm_state_dict_names) == 8 and all("adapter_1" in n for n in server.comm_state_dict_names)
    server_sd = {k: v.clone() for k, v in server.state_dict().items()}
    c_models = []
    for ci, t in enumerate(tasks):
        server.load_state_dict(server_sd)                 # main.py:472 deepcopy(server) + personal params
        server.adapter_requires_grad.update({0: True, 1: True})
        batches = [_dev(O.synthetic_batch(4, 224, 777 + 10 * ci + s)) for s in range(3)]
        trainer = api.train.TaskTrainer(args, t, batches)
        score, c_model = trainer.train(server)
        assert score == 0.0 and c_model is server
        assert server.gating and server.active == "adapter_0"        # final mode of train_step, task_trainer.py:311-312
        c_models.append({n: server.state_dict()[n].clone() for n in server.comm_state_dict_names})
        # parity on the UPDATE of the personal adapter_0 tensors (they move ~1.5e-4 in 3 steps: a bare 1e-3 bound on the
        # weights could not fail), against the reference's own round
        pk = {k[len(f"personal.{t}."):]: torch.from_numpy(r[k]) for k in r if k.startswith(f"personal.{t}.")}
        assert_update_parity(list(pk), server.state_dict(), pk, P, 1e-3, 0.1, f"personal {t}")
    server.load_state_dict(server_sd)
    api.train.get_average_net(server, c_models, [1, 1], tasks, DEV)
    sk = {k[7:]: torch.from_numpy(r[k]) for k in r if k.startswith("server.")}
    assert_update_parity(list(sk), server.state_dict(), sk, P, 1e-3, 0.1, "averaged adapter_1")
    # eval leaves the model in the adapter_1 state -> the next optimizer would not hold adapter_0 (reference quirk)
    ev = api.train.TaskTrainer(args, "art", [], [_dev(O.synthetic_batch(4, 224, 5))]).eval(server)
    assert len(ev) == 3 and server.optimizer_adapters() == (1,)


def test_heterogeneous_clients_round_vs_oracle(api):
    """SURVEY 8d config 3 in small: two clients whose answers come from DIFFERENT Dirichlet(0.5) label priors and whose loaders
    have different lengths (5 and 3 batches), one FL round -- local updates from the same server state, personal parameters kept
    per client, get_average_net of adapter_1 -- against the oracle doing the same on the same batches.  The clients' adapter_1
    updates must really disagree (cosine well below 1), and the average must match."""
    from feddat_amd import vilt_spec
    tasks = ["art", "gqa"]
    d = O.ViltDims(layers=2)
    P = O.make_params(d, tasks, bias_std=0.02)
    P_ref = {k: v.clone() for k, v in P.items()}
    args = types.SimpleNamespace(local_epochs=1, num_epochs=15, lr=1e-4, optimizer_mode="dat", debug=0, hip_graph=True)
    server = api.modeling.create_vilt_continual_learner_model(P, tasks, DEV, batch_size=4, image_size=224, num_layers=2)
    server_sd = {k: v.clone() for k, v in server.state_dict().items()}
    comm = server.comm_state_dict_names
    steps = {"art": 5, "gqa": 3}
    c_models, c_ref = [], []
    for ci, t in enumerate(tasks):
        prior = vilt_spec.client_label_prior(ci)
        host = [O.synthetic_batch(4, 224, 900 + 10 * ci + s, label_prior=prior) for s in range(steps[t])]
        server.load_state_dict(server_sd)
        server.adapter_requires_grad.update({0: True, 1: True})
        api.train.TaskTrainer(args, t, [_dev(b) for b in host]).train(server)
        c_models.append({n: server.state_dict()[n].clone() for n in comm})
        Pc = {k: v.clone() for k, v in P_ref.items()}
        client = O.DatClient(Pc, d, t, lr=1e-4, steps_per_epoch=steps[t])
        for b in host:
            client.train_step(b)
        c_ref.append({n: Pc[n].clone() for n in comm})
        assert_update_parity(comm, server.state_dict(), Pc, P_ref, 1e-3, 0.1, f"client {t} adapter_1")
    big = max(comm, key=lambda n: P_ref[n].numel())
    da, db = (c_ref[0][big] - P_ref[big]).flatten(), (c_ref[1][big] - P_ref[big]).flatten()
    cos = float(torch.dot(da, db) / (da.norm() * db.norm()))
    assert cos < 0.9, cos                                  # heterogeneous clients: the updates point in different directions
    server.load_state_dict(server_sd)
    api.train.get_average_net(server, c_models, [1, 1], tasks, DEV)
    ref_server = {n: torch.zeros_like(P_ref[n]) for n in comm}
    O.get_average_net(ref_server, c_ref, [1, 1])
    assert_update_parity(comm, server.state_dict(), ref_server, P_ref, 1e-3, 0.1, "averaged adapter_1 of heterogeneous clients")


def test_kl_loss_op(api, golden_dir):
    g = load(golden_dir, "g2_loss.npz")
    kl = api.train.kl_loss(torch.from_numpy(g["logits"]).to(DEV), torch.from_numpy(g["teacher"]).to(DEV))
    assert abs(float(kl) - float(g["kl"])) < 1e-5


def test_checkpoint_roundtrip_uses_reference_key_names(api, tmp_path):
    from feddat_amd import checkpoint
    d = O.ViltDims(layers=2)
    P = O.make_params(d, ["art"], bias_std=0.02)
    m = api.modeling.create_vilt_continual_learner_model(P, ["art"], DEV, batch_size=2, image_size=224, num_layers=2)
    comm, personal = checkpoint.save_round(m, str(tmp_path), "art")
    assert comm == sorted(k for k in O.param_shapes(d, ["art"]) if "adapter_1" in k)
    assert all(("adapter_0" in k or "adapter_2" in k or k.startswith("task_layer.art.")) for k in personal)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    for v in m.state_dict().values():
        v.zero_()
    checkpoint.load_round(m, str(tmp_path), "art")
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k


def test_eval_three_way_scores_match_oracle(api):
    """task_trainer.py:230-244: [gated (adapter_0 + adapter_2), adapter_0, adapter_1] VQA scores of one loader."""
    d = O.ViltDims(layers=2)
    P = O.make_params(d, ["art"], bias_std=0.02)
    m = api.modeling.create_vilt_continual_learner_model(P, ["art"], DEV, batch_size=4, image_size=224, num_layers=2)
    args = types.SimpleNamespace(local_epochs=1, num_epochs=15, lr=1e-4, optimizer_mode="dat", debug=0, hip_graph=False)
    loader = [O.synthetic_batch(4, 224, 900 + s) for s in range(3)]
    got = api.train.TaskTrainer(args, "art", [], [_dev(b) for b in loader]).eval(m)
    want, margins = [], []
    for mode in ("gating", "adapter_0", "adapter_1"):
        sc = 0.0
        for b in loader:
            with torch.no_grad():
                _, lg = O.vilt_forward(P, d, b, mode, "art")
            top2 = lg.topk(2, dim=1).values
            margins.append(float((top2[:, 0] - top2[:, 1]).min()))
            sc += float(b["target_scores"].gather(1, lg.argmax(1, keepdim=True)).sum())
        want.append(100.0 * sc / 12)
    if min(margins) > 6e-2:          # arg-max is only comparable when no row is a near-tie at the logit tolerance
        assert got == pytest.approx(want, abs=1e-4)
    assert len(got) == 3 and all(0.0 <= s <= 100.0 for s in got)


def test_main_rounds_and_resume(api, tmp_path):
    """feddat_amd.train.main: 2 clients x 2 rounds on one GPU, then the same run split as 1 round + resume from the
    round state on disk -- identical averaged adapter and personal tensors (bit-exact: same kernels, same order)."""
    common = ["--ordered_cl_tasks", "art,gqa", "--num_layers", "2", "--image_size", "224", "--batch_size", "2",
              "--synthetic_steps", "2", "--no_hip_graph", "--save_every", "1"]
    a = api.train.main(common + ["--comm_rounds", "2", "--output_dir", str(tmp_path / "a")])
    sd_a = {k: v.clone() for k, v in a.state_dict().items()}
    api.train.main(common + ["--comm_rounds", "1", "--output_dir", str(tmp_path / "b")])
    b = api.train.main(common + ["--comm_rounds", "2", "--output_dir", str(tmp_path / "b2"),
                                 "--checkpoint", str(tmp_path / "b")])
    sd_b = b.state_dict()
    for n in a.comm_state_dict_names:
        assert torch.isfinite(sd_a[n]).all()
        assert torch.equal(sd_a[n], sd_b[n]), n
    from safetensors.torch import load_file
    for t in ("art", "gqa"):
        pa = load_file(str(tmp_path / "a" / f"personal_{t}.safetensors"))
        pb = load_file(str(tmp_path / "b2" / f"personal_{t}.safetensors"))
        assert pa.keys() == pb.keys() and all(torch.equal(pa[k], pb[k]) for k in pa), t


def test_main_default_exchange_is_the_c_abi_collective(api, tmp_path):
    """train.main with one rank: the round's FedAvg goes through feddat_fedavg_allreduce on a communicator made by
    feddat_comm_* (RCCL, one rank: the identity) -- the default exchange, not torch.distributed -- and gives the bits of
    --exchange torch (no collective at world 1)."""
    common = ["--ordered_cl_tasks", "art,gqa", "--num_layers", "2", "--image_size", "224", "--batch_size", "2",
              "--synthetic_steps", "2", "--no_hip_graph", "--comm_rounds", "2"]
    a = api.train.main(common)
    assert a.exchange_used == "feddat_fedavg_allreduce"
    sd_a = {k: v.clone() for k, v in a.state_dict().items()}
    b = api.train.main(common + ["--exchange", "torch"])
    assert b.exchange_used == "none"
    sd_b = b.state_dict()
    for n in a.comm_state_dict_names:
        assert torch.isfinite(sd_a[n]).all() and torch.equal(sd_a[n], sd_b[n]), n
    info = api.L.RcclComm(1, 0, lambda ident: ident).info()
    assert info["ranks"] == 1 and info["rank"] == 0 and info["rccl_version"] > 20000, info


def test_device_prefetcher_matches_synchronous_upload(api):
    """feddat_amd.data.DevicePrefetcher: same batches, same order, same training result as uploading synchronously;
    worker exceptions surface in the consumer."""
    from feddat_amd.data import DevicePrefetcher, pin_batch
    d = O.ViltDims(layers=2)
    host = [pin_batch(O.synthetic_batch(2, 224, 40 + s)) for s in range(5)]
    up = lambda b: {k: v.to(DEV, non_blocking=True) for k, v in b.items()}
    finals = []
    for mode in ("sync", "prefetch"):
        P = O.make_params(d, ["art"], bias_std=0.02)
        m = api.modeling.create_vilt_continual_learner_model(P, ["art"], DEV, batch_size=2, image_size=224, num_layers=2)
        m.engine.begin_local_update("art", steps_per_epoch=5)
        it = DevicePrefetcher(host, up, DEV, depth=2) if mode == "prefetch" else (up(b) for b in host)
        n = 0
        for i, b in enumerate(it):
            assert torch.equal(b["input_ids"].cpu(), host[i]["input_ids"])
            m.engine.train_step(b)
            n += 1
        assert n == 5
        torch.cuda.synchronize()
        finals.append({k: v.clone() for k, v in m.state_dict().items() if "adapter_1" in k})
    for k in finals[0]:
        assert torch.equal(finals[0][k], finals[1][k]), k

    def boom(b):
        raise ValueError("bad batch")
    with pytest.raises(ValueError):
        for _ in DevicePrefetcher(host, boom, DEV):
            pass


def test_train_with_pageable_host_batches_prefetch_and_graph(api):
    """TaskTrainer.train as main() drives it, with PAGEABLE host batches, the upload worker and hipGraph replay together
    (the graph is captured before the worker starts; capture mode is thread-local): same result as device-resident
    batches without graph."""
    d = O.ViltDims(layers=2)
    host = [O.synthetic_batch(2, 224, 70 + s) for s in range(4)]
    finals = []
    for graph, prefetch, src in ((False, False, [_dev(b) for b in host]), (True, True, host)):
        P = O.make_params(d, ["art"], bias_std=0.02)
        m = api.modeling.create_vilt_continual_learner_model(P, ["art"], DEV, batch_size=2, image_size=224, num_layers=2)
        args = types.SimpleNamespace(local_epochs=2, num_epochs=15, lr=1e-4, optimizer_mode="dat", debug=0,
                                     hip_graph=graph, prefetch=prefetch)
        api.train.TaskTrainer(args, "art", src).train(m)
        torch.cuda.synchronize()
        finals.append({k: v.clone() for k, v in m.state_dict().items() if "adapter_2" not in k})
    for k in finals[0]:
        assert torch.equal(finals[0][k], finals[1][k]), k
