"""Host-side logic that needs no GPU: client -> rank dealing, the heterogeneity bound, the exchange agreement."""
import pytest
import torch


def test_deal_clients_and_round_efficiency():
    from feddat_amd.train import deal_clients, round_efficiency
    # at least as many ranks as clients: client k <-> rank k (the reference's client order, one client per GPU)
    assert deal_clients([40, 50, 60, 70, 80], 8) == [0, 1, 2, 3, 4]
    assert deal_clients([8, 8], 2) == [0, 1]
    # equal loads: longest-processing-time dealing degenerates to round-robin (what round 3 did)
    assert deal_clients([8, 8, 8, 8, 8], 2) == [0, 1, 0, 1, 0]
    # heterogeneous: 5 clients on 2 ranks -- round-robin gives loads (180, 120): efficiency 0.83; LPT (170, 130): 0.88
    steps = [40, 50, 60, 70, 80]
    own = deal_clients(steps, 2)
    loads = [sum(s for s, o in zip(steps, own) if o == r) for r in range(2)]
    assert sorted(loads) == [130, 170] and round_efficiency(loads) == pytest.approx(300 / 340)
    assert round_efficiency([sum(steps[0::2]), sum(steps[1::2])]) == pytest.approx(300 / 360)
    # 8 clients = 8 ranks, len(loader) in {40..80}: the all-reduce barrier bounds scaling at 5.625x of 8
    hetero8 = [40 + 10 * (r % 5) for r in range(8)]
    assert deal_clients(hetero8, 8) == list(range(8))
    assert round_efficiency(hetero8) == pytest.approx(5.625 / 8)
    # 16 heterogeneous clients on 8 ranks: two per rank, balanced to within one client's difference
    c16 = hetero8 + hetero8
    own = deal_clients(c16, 8)
    loads = [sum(s for s, o in zip(c16, own) if o == r) for r in range(8)]
    assert max(loads) - min(loads) <= 20 and round_efficiency(loads) > 0.9
    assert round_efficiency([sum(c16[r::8]) for r in range(8)]) < round_efficiency(loads)      # better than round-robin


def test_agree_on_exchange_falls_back_without_raising_on_one_rank():
    import logging
    from feddat_amd import lib as L
    from feddat_amd.train import agree_on_exchange

    def broken():
        raise L.FeddatHipError("librccl not loadable")
    assert agree_on_exchange(broken, 1, logging.getLogger("t")) is None      # single rank: identity exchange, no exception
    sentinel = object()
    assert agree_on_exchange(lambda: sentinel, 1, logging.getLogger("t")) is sentinel


def test_finite_check_is_deferred_to_the_collective_under_several_ranks():
    """a rank whose local update went non-finite must not raise on its own while its peers enter the FedAvg
    all-reduce.  Standalone TaskTrainer.train raises (engine.assert_finite); driven by train.main with several ranks it only records
    the outcome, main() all-reduces it over the rendezvous group and every rank raises together."""
    import logging
    import types
    from feddat_amd import lib as L
    from feddat_amd.train import TaskTrainer

    class Eng:
        def __init__(self, bad):
            self.bad = bad

        def nonfinite_groups(self):
            return list(self.bad)

        def assert_finite(self):
            if self.bad:
                raise L.FeddatHipError("non-finite values in " + ", ".join(self.bad))

        def scaler_state(self):
            return dict(dynamic=True, skipped_substeps=3, skipped_batches=2, scale=4096.0)
    args = types.SimpleNamespace(local_epochs=1, num_epochs=15, lr=1e-4, hip_graph=True)
    tr = TaskTrainer(args, "art", [], [], logging.getLogger("t"))
    tr._finite_check(Eng([]))                                   # clean: nothing happens
    with pytest.raises(L.FeddatHipError):
        tr._finite_check(Eng(["head"]))                         # standalone: raises here
    tr.defer_finite_check = True
    tr._finite_check(Eng(["head", "adapter_0"]))                # under main() with world > 1: recorded, not raised
    assert tr.nonfinite == ["head", "adapter_0"]
    tr._finite_check(Eng([]))
    assert tr.nonfinite == []


def test_bench_round_split_reports_the_heterogeneity_bound():
    import bench
    rows = [dict(rank=r, steps=s, compute_s=0.01 * s, wait_s=0.0, allreduce_ms=0.1) for r, s in enumerate([40, 50, 60, 70, 80, 40, 50, 60])]
    rs = bench.round_split(rows, 0.8)
    assert rs["hetero_bound"]["scaling_x_bound"] == pytest.approx(5.625) and rs["hetero_bound"]["round_efficiency_bound"] == pytest.approx(0.703, abs=1e-3)
    assert rs["hetero_bound"]["efficiency_vs_bound"] == pytest.approx(1.0)


def test_bench_reads_the_committed_profile_summaries():
    """bench.py's roofline object carries two figures from the committed rocprofv3 summaries (profiles/rNN_*.csv): HBM bytes per
    launch of the dominant kernel (PMC: FETCH_SIZE x 2 + WRITE_SIZE, KiB) and the same kernels' per-step time in the kernel
    trace.  Both parsers on the files of the latest round: plausible magnitudes, and None for a pattern that matches nothing."""
    import bench
    tr = bench.profiled_traffic()
    assert tr is not None and tr["source"].endswith("_pmc_per_kernel.csv")
    assert 50e6 < tr["bytes_per_launch"] < 400e6              # the step's K1 launches move ~100-150 MB each
    kt = bench.profiled_kernel_time()
    assert kt is not None and kt["steps"] >= 5 and 2.0 < kt["ms_per_step"] < 8.0
    assert bench.profiled_traffic(pattern="r[0-9][0-9]_nothing.csv") is None
    assert bench.profiled_kernel_time(pattern="r[0-9][0-9]_nothing.csv") is None
    assert bench.profiled_kernel_time(step_marker=("no_such_kernel",)) is None


def test_bench_round_split_reports_allreduce_bandwidth_and_shares():
    import bench
    rows = [dict(rank=r, steps=20, samples_per_sec=4000.0, compute_s=0.14, wait_s=0.001, allreduce_ms=0.5) for r in range(8)]
    rs = bench.round_split(rows, 0.15, payload_bytes=894528 * 4)
    bw = rs["allreduce_bandwidth"]
    assert abs(bw["algbw_GBps"] - 894528 * 4 / 0.5e-3 / 1e9) < 1e-2
    assert abs(bw["busbw_GBps"] - bw["algbw_GBps"] * 2 * 7 / 8) < 1e-2
    assert 0.9 < rs["share_of_round"]["compute"] < 1.0 and rs["share_of_round"]["allreduce"] < 0.01
    assert bench.round_split(rows, 0.15)["allreduce_bandwidth"] is None


def test_bench_second_roofline_names_the_worst_row_kernel():
    """roofline.second: among the HBM-bound row kernels that take >= 0.1 ms of the step, the one furthest below 8 TB/s."""
    import bench
    T = 11840
    by = bench.row_kernel_bytes(T)
    assert abs(by["layernorm_bwd_dx"] - 145.5e6) < 1e6 and abs(by["adapter_fwd_ln"] - 95.5e6) < 1e6
    other = {"adapter_fwd_ln": 0.344, "layernorm_bwd_dx": 0.510, "layernorm_fwd": 0.05, "attn_fwd": 0.231}
    n = {"adapter_fwd_ln": 11, "layernorm_bwd_dx": 23, "layernorm_fwd": 13, "attn_fwd": 11}
    s = bench.second_roofline(other, n, T)
    assert s["kernel"] == "adapter_fwd_ln" and s["bound"] == "hbm" and 0.3 < s["frac"] < 0.45
    assert [r["kernel"] for r in s["all_row_kernels"]] == ["adapter_fwd_ln", "attn_fwd", "layernorm_bwd_dx"]     # >= 0.1 ms only
    assert bench.second_roofline({}, {}, T) is None


def test_client_label_priors_are_heterogeneous_and_reproducible():
    """Dirichlet(0.5) label prior per client; the product's generator draws what the checker's draws."""
    import torch
    from feddat_amd import vilt_spec
    from oracle import feddat_oracle as O
    p0, p1 = vilt_spec.client_label_prior(0), vilt_spec.client_label_prior(1)
    assert torch.equal(p0, vilt_spec.client_label_prior(0)) and not torch.equal(p0, p1)
    assert abs(float(p0.sum()) - 1) < 1e-6 and float(p0.max()) > 0.03 and int((p0 < 1e-3).sum()) > 10      # a few answers dominate
    a, b = vilt_spec.synthetic_batch(8, 32, 77, label_prior=p0), O.synthetic_batch(8, 32, 77, label_prior=p0)
    assert all(torch.equal(a[k], b[k]) for k in a)
    # the clients' answer distributions differ: label mass of 64 batches of each client
    m = [sum((vilt_spec.synthetic_batch(8, 32, 100 + s, label_prior=p)["target_scores"] > 0).float().sum(0) for s in range(64))
         for p in (p0, p1)]
    cos = float((m[0] * m[1]).sum() / (m[0].norm() * m[1].norm()))
    assert cos < 0.6


def test_bench_albef_flop_count():
    """configs[3]'s whole-step MFMA fractions rest on albef_flops(): the ViT's 12 blocks of 577 tokens dominate (~70 % of
    the FLOPs), the reference runs 3 forwards + 2 backwards, the engine 2 + 2 minus the shared prefix below the first adapter."""
    import bench
    ref, exe = bench.albef_flops(577, 25, 4)
    vit_fwd = 12 * (2.0 * 577 * 768 * 9216 + 4.0 * 577 * 577 * 768)
    assert 0.80 < 5 * vit_fwd / ref < 0.95             # 3 fwd + ~2 bwd of the ViT against everything
    assert 0.75 < exe / ref < 0.82
    assert 6.0e11 < ref < 6.8e11


def test_bench_help_renders():
    """argparse interpolates '%' in help strings: `bench.py --help` must not raise (it did in round 4)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "--operands" in r.stdout, r.stderr[-500:]
