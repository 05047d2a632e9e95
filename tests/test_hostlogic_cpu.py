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
    assert bench.profiled_kernel_time(step_marker="no_such_kernel") is None
