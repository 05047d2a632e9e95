"""The N > 1 path on hardware, as far as a single-GPU box allows: TWO ranks (processes) share cuda:0 and exchange over gloo
(RCCL refuses two ranks on one device), everything else is the production path -- feddat_amd.train.main under
torch.distributed.run, the HIP pre-scale kernel (feddat_fedavg_accumulate), the all-reduce, write-back, repack, personal
parameter shuttle -- and `python bench.py --gpus 2` starting its own ranks.  Reference loop replaced: main.py:453-510."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    e = dict(os.environ)
    e.update(FEDDAT_FORCE_DEVICE="0", FEDDAT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


def _port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# heterogeneous clients (SURVEY 8d config 3 in small): different len(loader) AND each client's answers from its own Dirichlet(0.5)
# label prior, so the two clients' adapter_1 updates really disagree before they are averaged
COMMON = ["--num_layers", "2", "--image_size", "224", "--batch_size", "2", "--synthetic_steps", "3,2", "--comm_rounds", "2",
          "--save_every", "1", "--synthetic_label_alpha", "0.5"]


@pytest.mark.timeout(900)
def test_two_ranks_through_train_main_match_single_process(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from safetensors.torch import load_file
    from feddat_amd import train
    single = train.main(COMMON + ["--ordered_cl_tasks", "art,gqa", "--output_dir", str(tmp_path / "single")])
    sd1 = {k: v.cpu().clone() for k, v in single.state_dict().items()}
    out = tmp_path / "two"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), "-m", "feddat_amd.train"] + COMMON + [
               "--ordered_cl_tasks", "art,gqa", "--output_dir", str(out)]
    r = subprocess.run(cmd, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    srv = load_file(str(out / "server_adapter.safetensors"))
    assert len(srv) == 8
    moved = 0.0
    for k, v in srv.items():
        # two addends per element: the all-reduce and the sequential loop give the same bits
        assert torch.equal(v, sd1[k]), (k, float((v - sd1[k]).abs().max()))
        moved = max(moved, float(v.abs().max()))
    assert moved > 0
    for t in ("art", "gqa"):        # personal tensors stay on the rank that owns the client
        a = load_file(str(tmp_path / "single" / f"personal_{t}.safetensors"))
        b = load_file(str(out / f"personal_{t}.safetensors"))
        assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a), t


@pytest.mark.timeout(900)
def test_more_ranks_than_clients_idle_rank_joins_the_allreduce(tmp_path):
    """3 ranks, 2 clients: the third rank contributes zeros; the average over the two real clients is unchanged."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from safetensors.torch import load_file
    outs = []
    for n in (2, 3):
        out = tmp_path / f"n{n}"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
               "127.0.0.1", "--master-port", str(_port()), "-m", "feddat_amd.train"] + COMMON + [
                   "--comm_rounds", "1", "--ordered_cl_tasks", "art,gqa", "--output_dir", str(out)]
        r = subprocess.run(cmd, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=800)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        outs.append(load_file(str(out / "server_adapter.safetensors")))
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k


@pytest.mark.timeout(900)
def test_bench_gpus_2_starts_its_own_ranks():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4",
           "--res", "224", "--no-roofline", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and out["config"]["clients"] == 2
    assert out["value"] > 0 and out["scaling"] == "weak"
    # WORLD_SIZE != --gpus must fail loudly instead of silently running single-GPU
    e = _env()
    e.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()))
    r = subprocess.run(cmd, env=e, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stdout + r.stderr)
