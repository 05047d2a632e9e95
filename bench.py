#!/usr/bin/env python
"""Headline benchmark: VQA samples/sec of the ViLT-B/32 dual-adapter (DAT + MKD) local step on MI355X.

  python bench.py --gpus N --steps K --warmup W
  N > 1: one rank = one GPU = one federated client.  Either launched by `python -m torch.distributed.run --nproc-per-node N
  ... bench.py --gpus N ...` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or started plainly, in which
  case this script re-executes itself under torch.distributed.run with N ranks on 127.0.0.1.  WORLD_SIZE != N is an error.

A "step" is one full reference train_step (task_trainer.py:280-330: P0 + P1 + P2, two AdamW/scheduler steps) on one
batch of B=32 synthetic 384x384 image / 40-token question pairs per client (BASELINE.json configs[1]; configs[2] for
N > 1).  Inputs are resident in HBM before the timed region.  With N clients the timed region also contains the
round's FedAvg exchange (one RCCL all-reduce of the 3.58 MB adapter_1 buffer, main.py:50-65).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16 = 2.5e15          # dense bf16 MFMA peak of gfx950 (MI355X_MICROARCH.md)
REF_FLOPS_PER_SAMPLE = 1.6564e11   # algorithmic FLOPs of one reference DAT step, S=185


def flops_tables(B, S, layers=12, H=768, I=3072, heads=12, r=48, npatch=144):
    """Executed FLOPs of OUR step per batch (see DESIGN.md 'step algebra') and the GEMM launch list."""
    R = B * S
    lin = lambda M, N, K: 2.0 * M * N * K  # noqa: E731
    gemms = []  # (M, N, K, epi, count)
    fwd_shapes = [(3 * H, H, 0), (H, H, 1), (I, H, 2), (H, I, 1)]     # (FFN1 of the batched layers: epi 5, set below)
    bwd_shapes = [(I, H, 6 if 2 * R >= 1024 else 3), (H, I, 0), (H, H, 0), (H, 3 * H, 0)]
    # layer 0: shared body on R rows; layers 1..L-2: both passes batched (2R rows); top layer: QKV (and its dX) on all
    # tokens, everything behind the attention only on the 2B token-0 rows (negligible, not listed)
    for N, K, epi in fwd_shapes:
        gemms.append((R, N, K, epi, 1))
        gemms.append((2 * R, N, K, 5 if epi == 2 and 2 * R >= 1024 else epi, layers - 1 if N == 3 * H else layers - 2))
    for N, K, epi in bwd_shapes:
        gemms.append((2 * R, N, K, epi, layers - 1 if K == 3 * H else layers - 2))
    gemms.append((B * npatch, H, 3 * 32 * 32, 4, 1))
    gemm_flops = sum(lin(M, N, K) * c for M, N, K, _, c in gemms)
    attn_fwd = 4.0 * S * S * 64 * heads * B           # per B samples
    # dense attention: layer 0 once (shared body), layers 1..L-2 for both passes; backward (2.5 x the forward's matmul work)
    # for layers 1..L-2.  The LAST layer computes one query per (sample, head) (feddat_attn_cls_fwd / _bwd): 1 / S of a dense
    # layer's work, on the VALU -- not counted as MFMA work.
    attn = attn_fwd * (1 + 2 * (layers - 2)) + 2.5 * attn_fwd * 2 * (layers - 2)
    ad1 = 2.0 * R * H * r * 2                          # one adapter forward over R rows
    la = layers - 1                                    # the top layer's adapter sees 2B rows only
    adapters = la * 3 * ad1 + la * 3 * 3 * ad1 + la * 2 * 2 * ad1  # fwd + (recompute, g, dx) + dW
    head = 3 * 2.0 * B * (H * 2 * H + 2 * H * 100) * 3
    return gemms, gemm_flops, gemm_flops + attn + adapters + head


def measure_gemms(L, gemms, iters=10, operands="bf16"):
    """Average launch duration of the dominant kernel (gemm_nt_kernel) per shape, HIP events on the launch stream."""
    with L.operands(operands):
        return _measure_gemms(L, gemms, iters, L.OPERAND_DTYPE[operands])


def _measure_gemms(L, gemms, iters, op_dtype):
    dev = "cuda"
    tot_t, tot_f, rows = 0.0, 0.0, []
    for M, N, K, epi, count in gemms:
        A = torch.randn(M, K, device=dev).to(op_dtype)
        Bw = (torch.randn(N, K, device=dev) * 0.02).to(op_dtype)
        bias = torch.randn(N, device=dev)
        kw = {}
        if epi in (0, 3, 5, 6):
            kw["out_bf16"] = torch.empty(M, N, dtype=op_dtype, device=dev)
        if epi == 3:
            kw["aux"] = torch.randn(M, N, device=dev).to(op_dtype)
        if epi == 5:
            kw["out2_bf16"] = torch.empty(M, N, dtype=torch.uint8, device=dev)
        if epi == 6:
            kw["aux"] = torch.randint(0, 255, (M, N), dtype=torch.uint8, device=dev)
        if epi == 2:
            kw["out_bf16"] = torch.empty(M, N, dtype=op_dtype, device=dev)
            kw["out2_bf16"] = torch.empty(M, N, dtype=op_dtype, device=dev)
        if epi == 1:
            kw["resid"] = torch.randn(M, N, device=dev)
            kw["out_f32"] = torch.empty(M, N, device=dev)
        if epi == 4:
            kw["out_f32"] = torch.empty(M, N, device=dev)
        if epi not in (3, 6):
            kw["bias"] = bias
        for _ in range(3):
            L.gemm_bf16_nt(A, Bw, epi, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            L.gemm_bf16_nt(A, Bw, epi, **kw)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / iters
        f = 2.0 * M * N * K
        rows.append(dict(M=M, N=N, K=K, epi=epi, count=count, us=round(t * 1e6, 2), tflops=round(f / t / 1e12, 1)))
        tot_t += t * count
        tot_f += f * count
    return tot_f / tot_t, tot_t, rows


def gemm_algorithmic_bytes(M, N, K, epi):
    """HBM bytes one launch has to move: A and B once (bf16), the epilogue's operands and outputs once, the bias."""
    out = {0: 2 * M * N,              # bf16 out
           1: 4 * M * N + 4 * M * N,  # fp32 residual in + fp32 out
           2: 2 * M * N + 2 * M * N,  # gelu(u) + u, both bf16
           3: 2 * M * N + 2 * M * N,  # u in (bf16) + bf16 out
           4: 4 * M * N,
           5: 2 * M * N + M * N,      # gelu(u) bf16 + 8-bit gelu'(u) codes
           6: M * N + 2 * M * N}[epi]  # codes in + bf16 out
    return 2 * M * K + 2 * N * K + out + 4 * N


def measure_gemms_in_step(L, eng, batches, steps=3):
    """K1 durations INSIDE the train_step (tools/step_breakdown.measure): the engine's own launches, an eager replay of the
    kernel sequence the graph holds, EVERY op bracketed by HIP events on the launch stream and the whole step queued
    behind a spin kernel, so each GEMM sees the caches as the preceding kernels of the step left them -- not the
    MALL-warm state of back-to-back repeats; the cost of an empty event bracket is measured in the same queue and
    subtracted.  Returns FLOP/s over all K1 launches of a step, their summed time, per-shape rows, the count-weighted mean
    algorithmic bytes per launch and the launch count."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from step_breakdown import measure
    agg, step_ms, empty_us = measure(eng, L, batches, steps, detail=True)
    g = {k[1:5]: v for k, v in agg.items() if isinstance(k, tuple) and k[0] == "gemm" and not k[5]}
    tot_t = sum(ms for _, ms in g.values()) * 1e-3
    tot_f = sum(2.0 * M * N * K * n for (M, N, K, _), (n, _) in g.items())
    launches = sum(n for n, _ in g.values())
    alg = sum(gemm_algorithmic_bytes(M, N, K, epi) * n for (M, N, K, epi), (n, _) in g.items()) / launches
    rows = [dict(M=M, N=N, K=K, epi=epi, count=n, us=round(ms / n * 1e3, 2),
                 tflops=round(2.0 * M * N * K * n / (ms * 1e-3) / 1e12, 1)) for (M, N, K, epi), (n, ms) in sorted(g.items())]
    other = {str(k): round(v[1], 4) for k, v in agg.items() if not (isinstance(k, tuple) and not k[5])}
    other_n = {str(k): v[0] for k, v in agg.items() if not isinstance(k, tuple)}
    return tot_f / tot_t, tot_t, rows, alg, launches, dict(eager_step_ms=round(step_ms, 3),
                                                           empty_bracket_us=round(empty_us, 2), other_ops_ms_per_step=other,
                                                           second=second_roofline(other, other_n, 2 * eng.R))


def row_kernel_bytes(T, H=768, I=3072, r=48, heads=12, S=185):
    """Algorithmic HBM bytes per launch of the step's HBM-bound row kernels at T rows (every operand and result once;
    DESIGN.md section 4): name of the C-ABI wrapper -> bytes."""
    nb = T // S
    return {
        "adapter_fwd_ln": T * H * (4 + 4 + 2) + T * 2 * r * 4,           # h3 in, h_in out (fp32), next LN out (16 bit), z saved
        "adapter_bwd": T * H * (4 + 4 + 2) + T * 2 * r * 4 * 2 + T * r * 4 * 2,    # dy in, dx out (fp32 + 16 bit), z saved in, z / dz out
        "adapter_wgrad_partial": T * H * 4 * 2 + T * r * 4 * 2,          # x, dy (fp32), z, dz
        "layernorm_bwd_dx": T * H * (2 + 4 + 4 + 4 + 2),                 # dy (16 bit), x, dres in, out fp32 (+ 16-bit copy)
        "layernorm_fwd": T * H * (4 + 2),
        "attn_fwd": T * 3 * H * 2 + T * H * 2 + nb * heads * S * 4,      # qkv in, ctx out, lse
        "attn_bwd": T * 3 * H * 2 * 2 + T * H * 2 * 2 + nb * heads * S * 4,   # qkv, ctx, dctx in, dqkv out
    }


PEAK_HBM = 8.0e12           # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def second_roofline(other_ms, launches, T):
    """`roofline.second`: the step's HBM-bound row kernel furthest below its roof (the one the next kernel work goes to), among
    those that take >= 0.1 ms of the step, from the same in-step event brackets as the GEMM figure."""
    by = row_kernel_bytes(T)
    rows = []
    for name, nbytes in by.items():
        if name in other_ms and launches.get(name, 0) > 0 and other_ms[name] >= 0.1:
            us = other_ms[name] / launches[name] * 1e3
            rows.append(dict(kernel=name, bound="hbm", launches_per_step=launches[name], us=round(us, 2),
                             algorithmic_bytes_per_launch=nbytes, achieved=round(nbytes / (us * 1e-6) / 1e9, 1),
                             peak=PEAK_HBM / 1e9, unit="GB/s", frac=round(nbytes / (us * 1e-6) / PEAK_HBM, 4),
                             ms_per_step=round(other_ms[name], 4)))
    if not rows:
        return None
    rows.sort(key=lambda r: r["frac"])
    worst = dict(rows[0])
    worst["all_row_kernels"] = [{k: r[k] for k in ("kernel", "us", "frac", "ms_per_step")} for r in rows]
    return worst


def profiled_traffic(kernel_substrs=("gemm_nt_v2_kernel", "gemm_nt_v3_kernel"), pattern="r[0-9][0-9]_pmc_per_kernel.csv"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/*_pmc_per_kernel.csv,
    separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
    FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> doubled."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None
    tot, n = 0.0, 0        # the kernel is templated on its epilogue: dispatch-weighted mean over all instantiations
    for r in csv.DictReader(open(files[-1])):
        if any(k in r["kernel"] for k in kernel_substrs) and r.get("FETCH_SIZE_avg") and r.get("WRITE_SIZE_avg"):
            k = int(r["dispatches"])
            tot += k * (2 * float(r["FETCH_SIZE_avg"]) + float(r["WRITE_SIZE_avg"])) * 1024
            n += k
    if not n:
        return None
    return {"bytes_per_launch": round(tot / n), "source": os.path.basename(files[-1]),
            "note": "mean over the launches of one step (all shapes and epilogues)"}


def profiled_kernel_time(kernel_substrs=("gemm_nt_v2_kernel", "gemm_nt_v3_kernel", "gemm_skinny", "gemm_nt_mid_kernel", "gemm_nt_kernel"),
                         pattern="r[0-9][0-9]_kernel_stats.csv", step_marker=("dat_step_finish", "step_tick_multi")):
    """Per-step time of the dominant kernel's launches in the committed `rocprofv3 --kernel-trace --stats` summary of this
    command (profiles/*_kernel_stats.csv: the hipGraph-replayed steps under the profiler), next to the live event-bracket
    figure: the two must agree (the brackets sit around an EAGER replay, whose launches run a few per cent longer)."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None
    ms, steps = 0.0, 0
    for r in csv.DictReader(open(files[-1])):
        if any(k in r["kernel"] for k in kernel_substrs):
            ms += float(r["total_ms"])
        if any(m in r["kernel"] for m in step_marker):      # the step's last kernel (dynamic loss scale: dat_step_finish)
            steps = max(steps, int(r["calls"]))
    if not steps or not ms:
        return None
    out = {"ms_per_step": round(ms / steps, 3), "steps": steps, "source": os.path.basename(files[-1]),
           "provenance": "HISTORICAL: the committed rocprofv3 summary of an earlier run of this command (commit / device below "
                         "when recorded), not this run -- a cross-check of the live figure, stale after a kernel change"}
    meta = files[-1].replace("_kernel_stats.csv", "_meta.json")
    if os.path.exists(meta):
        try:
            out.update({k: v for k, v in json.load(open(meta)).items() if k in ("commit", "device", "operands", "date")})
        except Exception:       # noqa: BLE001 -- provenance only
            pass
    return out


def cpu_baseline(params_cpu, batches_cpu, B, res, task, timed_steps=3):
    """The CPU restatement of the reference path (oracle/, pinned to the reference's goldens incl. a 40-step round) timed
    on this node's host cores on a BOUNDED sample of the SAME workload: same model, the same seeded B=32 batches the GPU
    steps ran on, 1 warm-up + 3 timed train_steps.  PyTorch-CPU does not scale to all cores on this model (it gets
    slower beyond a few dozen threads), so the thread count is swept once on a B=8 slice and the best one is used and
    stated in `cores`."""
    from oracle import feddat_oracle as O
    d = O.ViltDims(layers=12)
    ncpu = os.cpu_count()
    small = {k: v[:8].clone() for k, v in batches_cpu[0].items()}
    sweep = {}
    for nt in sorted({min(ncpu, n) for n in (4, 8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        c = O.DatClient({k: v.clone() for k, v in params_cpu.items()}, d, task, lr=1e-4, steps_per_epoch=50)
        c.train_step(small)
        t0 = time.time()
        c.train_step(small)
        sweep[nt] = round(8 / (time.time() - t0), 2)
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    client = O.DatClient({k: v.clone() for k, v in params_cpu.items()}, d, task, lr=1e-4, steps_per_epoch=50)
    client.train_step(batches_cpu[0])
    t1 = time.time()
    for i in range(timed_steps):
        client.train_step(batches_cpu[(1 + i) % len(batches_cpu)])
    dt = time.time() - t1
    return dict(value=round(B * timed_steps / dt, 3), unit="samples/s", cores=best, host_cores=ncpu, kind="port",
                thread_sweep_samples_per_s_B8=sweep,
                sample=f"{timed_steps} timed train_steps after 1 warm-up, B={B} (the same seeded batches as the GPU run), "
                       f"{res}x{res}, 40 tokens, 12 layers, fp32, torch {torch.__version__} CPU, {best} threads "
                       f"(best of the sweep)")


def hetero_steps(K, rank):
    """--hetero: client r's len(loader) from {40, 50, 60, 70, 80}, scaled so that the longest is K."""
    return max(1, round(K * (40 + 10 * (rank % 5)) / 80))


def make_exchange(eng, world, rank, dist):
    """The round's FedAvg exchange.  Real launch (one GPU per rank): the C-ABI collective feddat_fedavg_allreduce on a
    communicator made through feddat_comm_* (RCCL bound by dlopen, unique id shipped over the torch.distributed rendezvous) --
    not torch.distributed's all_reduce.  The torch.distributed group is only the HOST-side rendezvous (barriers, the id, the
    per-rank rows) and defaults to gloo, so each GPU holds exactly ONE RCCL communicator -- this one -- and a failure of RCCL
    cannot be confused with a failure of the rendezvous (FEDDAT_DIST_BACKEND=nccl puts the rendezvous on RCCL as well).  The
    single-GPU test rig (FEDDAT_FORCE_DEVICE: several ranks on one device, where RCCL refuses the duplicate device) keeps the
    torch.distributed path."""
    if dist is None:
        return None, None
    from feddat_amd.fedavg import allreduce_average, make_rccl_comm
    nbytes = eng.comm_flat().numel() * 4
    why = "single-GPU test rig"
    if os.environ.get("FEDDAT_FORCE_DEVICE") is None:
        comm, err = None, ""
        t_init = time.perf_counter()
        try:
            comm = make_rccl_comm(world, rank)
            info = comm.info()
        except Exception as e:      # noqa: BLE001 -- reported on the line; the ranks agree on the fallback below
            comm, err = None, f"{type(e).__name__}: {e}"
        init_ms = (time.perf_counter() - t_init) * 1e3
        ok = torch.tensor([1.0 if comm is not None else 0.0], device=eng.dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)          # every rank takes the same path
        if float(ok) > 0:
            v = info["rccl_version"]
            desc = {"library": "RCCL %d.%d.%d (C ABI: feddat_fedavg_allreduce)" % (v // 10000, v // 100 % 100, v % 100),
                    "rccl_version": v, "path": "feddat_comm_create_timeout -> ncclCommInitRank -> feddat_fedavg_allreduce",
                    "rendezvous": "torch.distributed " + dist.get_backend() + " (host side only)",
                    "rccl_communicators_per_gpu": 2 if dist.get_backend() == "nccl" else 1, "init_ms": round(init_ms, 1),
                    "ranks_requested": world, "ranks_in_communicator": info["ranks"], "payload_bytes": nbytes, "per_round": 1}
            if info["ranks"] != world:      # a communicator that does not span --gpus ranks measures something else: fail loudly
                raise SystemExit(f"bench.py: feddat_comm_info reports {info['ranks']} ranks in the communicator, --gpus {world} "
                                 f"were asked for (rank {rank}); refusing to print a throughput line")
            return (lambda: allreduce_average(eng, world, comm=comm)), desc
        if comm is not None:
            comm.close()
        why = "C-ABI communicator unavailable on some rank" + (f" ({err[:200]})" if err else "")
    desc = {"library": "torch.distributed " + dist.get_backend() + f" ({why})", "path": "torch.distributed.all_reduce",
            "ranks_requested": world, "ranks_in_communicator": dist.get_world_size(), "payload_bytes": nbytes, "per_round": 1,
            "fallback_reason": why}
    if dist.get_world_size() != world:
        raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {world} were asked for")
    return (lambda: allreduce_average(eng, world)), desc


def timed_round(step_fn, steps_r, warmup, exchange, dist, dev):
    """W warm-up steps (+ one exchange), barrier; then this rank's steps_r steps, [barrier], the exchange, barrier -- all
    inside the timed region.  Returns the round time (max over ranks is taken by the caller) and its split: this rank's
    compute time, its wait at the barrier for the slowest client, and the all-reduce (HIP events on the launch stream)."""
    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    for i in range(warmup):
        step_fn(i)
    if exchange is not None:
        exchange()
    barrier()
    t0 = time.perf_counter()
    for i in range(steps_r):
        step_fn(i)
    torch.cuda.synchronize()
    t_c = time.perf_counter() - t0
    t_w, ar_ms = 0.0, 0.0
    if exchange is not None:
        dist.barrier()                       # every client has finished its local epoch
        t_w = time.perf_counter() - t0 - t_c
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        exchange()
        e1.record()
    barrier()
    dt = time.perf_counter() - t0
    if exchange is not None:
        ar_ms = e0.elapsed_time(e1)
    return dict(dt=dt, compute_s=t_c, wait_s=t_w, allreduce_ms=ar_ms, steps=steps_r)


def gather_ranks(tr, B, world, rank, dist, dev):
    """-> (round time = max over ranks, total steps over ranks, per-rank rows) on every rank."""
    if dist is None:
        return tr["dt"], tr["steps"], None
    mine = torch.tensor([tr["dt"], tr["compute_s"], tr["wait_s"], tr["allreduce_ms"], float(tr["steps"])], dtype=torch.float64,
                        device=dev if dist.get_backend() == "nccl" else "cpu")
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    rows = [dict(rank=r, steps=int(t[4]), samples_per_sec=round(B * float(t[4]) / float(t[1]), 1),
                 compute_s=round(float(t[1]), 4), wait_s=round(float(t[2]), 4), allreduce_ms=round(float(t[3]), 3))
            for r, t in enumerate(allr)]
    return max(float(t[0]) for t in allr), int(sum(float(t[4]) for t in allr)), rows


def round_split(rows, dt, payload_bytes=None):
    """+ `hetero_bound`: with K = N heterogeneous clients every rank waits at the all-reduce for the one with the most steps,
    so N ranks can deliver at most sum(steps) / max(steps) ranks' worth of work per round (SURVEY 8d config 3's
    len(loader) in {40..80} on 8 ranks: 5.625 of 8 = 0.70; homogeneous: N) -- `scaling_x_bound` is what a reading of the
    N-GPU value / the 1-GPU value has to be compared with, `efficiency_vs_bound` how much of it the run delivered."""
    if rows is None:
        return None
    steps = [r["steps"] for r in rows]
    bound = sum(steps) / max(steps)
    busy = sum(r["compute_s"] for r in rows)
    ar = max(r["allreduce_ms"] for r in rows)
    n = len(rows)
    bus = None
    if payload_bytes and ar > 0:      # ring all-reduce moves 2 (N - 1) / N of the payload over every rank's links
        bus = {"algbw_GBps": round(payload_bytes / (ar * 1e-3) / 1e9, 3),
               "busbw_GBps": round(2.0 * (n - 1) / n * payload_bytes / (ar * 1e-3) / 1e9, 3),
               "note": "one 3.58 MB (ViLT) / 8.95 MB (ALBEF) all-reduce per round: latency-bound, far below the ~153 GB/s of an xGMI link"}
    return {"round_s": round(dt, 4), "compute_s_max": max(r["compute_s"] for r in rows),
            "compute_s_min": min(r["compute_s"] for r in rows), "wait_s_mean": round(sum(r["wait_s"] for r in rows) / len(rows), 4),
            "allreduce_ms_max": ar, "allreduce_bandwidth": bus,
            "share_of_round": {"compute": round(max(r["compute_s"] for r in rows) / dt, 4) if dt > 0 else None,
                               "allreduce": round(ar * 1e-3 / dt, 5) if dt > 0 else None},
            "hetero_bound": {"scaling_x_bound": round(bound, 3), "round_efficiency_bound": round(bound / len(rows), 3),
                             "efficiency_vs_bound": round(busy / (dt * bound), 3) if dt > 0 else None}}


def albef_roofline(L, eng, batches):
    """K1 over the GEMM launches of one ALBEF train_step and K2b (attn2) as its own entry, durations measured IN the step
    (tools/step_breakdown.measure: every C-ABI call of an eager step bracketed by HIP events on its launch stream, the
    step queued behind a spin kernel); for the measurement the two passes run on ONE stream so that a bracket times its
    own kernel and not its neighbour on the other stream."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from step_breakdown import measure
    side, eng.side = eng.side, torch.cuda.current_stream()
    try:
        agg, step_ms, empty_us = measure(eng, L, batches, steps=2, detail=True)
    finally:
        eng.side = side
    g = {k[1:5]: v for k, v in agg.items() if isinstance(k, tuple) and k[0] == "gemm"}
    tot_t = sum(ms for _, ms in g.values()) * 1e-3
    tot_f = sum(2.0 * M * N * K * n for (M, N, K, _), (n, _) in g.items())
    launches = sum(n for n, _ in g.values())
    alg = sum(gemm_algorithmic_bytes(M, N, K, epi) * n for (M, N, K, epi), (n, _) in g.items()) / launches
    big = sorted(g.items(), key=lambda kv: -kv[1][1])[:10]
    rows = [dict(M=M, N=N, K=K, epi=epi, count=n, us=round(ms / n * 1e3, 2),
                 tflops=round(2.0 * M * N * K * n / (ms * 1e-3) / 1e12, 1)) for (M, N, K, epi), (n, ms) in big]
    att = {k: v for k, v in agg.items() if isinstance(k, tuple) and k[0] in ("attn2_fwd", "attn2_bwd")}
    # matmul units: forward 2 (QK^T, PV); backward 5 executed (S, dP in both kernels counted once each, dV, dK, dQ) + 2 recomputed
    a_f = sum((2 if k[0] == "attn2_fwd" else 7) * 2.0 * k[1] * k[4] * k[2] * k[3] * 64 * n * (0.5 if k[5] else 1.0)
              for k, (n, _) in att.items())
    a_t = sum(ms for _, ms in att.values()) * 1e-3
    other = {str(k): round(v[1], 4) for k, v in agg.items() if not isinstance(k, tuple)}
    tr = profiled_traffic(("gemm_nt_v2_kernel", "gemm_nt_v3_kernel", "gemm_nt_mid_kernel", "gemm_nt_kernel"),
                          "r[0-9][0-9]_albef_pmc_per_kernel.csv")
    return {"kernel": "gemm_nt_v3_kernel / gemm_nt_v2_kernel / gemm_nt_mid_kernel (K1: every GEMM launch of one ALBEF train_step, "
                      "FLOP-weighted, durations measured in-step)",
            "bound": "mfma", "achieved": round(tot_f / tot_t / 1e12, 2), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
            "frac": round(tot_f / tot_t / PEAK_BF16, 4), "traffic": tr["bytes_per_launch"] if tr else None,
            "algorithmic_bytes_per_launch": round(alg),
            "traffic_ratio": round(tr["bytes_per_launch"] / alg, 3) if tr else None,
            "traffic_source": (tr["source"] + ": " + tr["note"]) if tr else None,
            "launches_per_step": launches, "gemm_ms_per_step": round(tot_t * 1e3, 3), "shapes_top10_by_time": rows,
            "attn2": {"kernel": "attn2_fwd_kernel / attn2_bwd_dq_kernel / attn2_bwd_dkv_kernel (K2b)", "bound": "mfma",
                      "achieved": round(a_f / a_t / 1e12, 2), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                      "frac": round(a_f / a_t / PEAK_BF16, 4), "ms_per_step": round(a_t * 1e3, 3),
                      "launches_per_step": sum(n for n, _ in att.values()),
                      "flops_counted": "executed matmul units (fwd 2, bwd 7 incl. the 2 recomputed), causal halved"},
            "in_step": {"eager_step_ms_one_stream": round(step_ms, 3), "empty_bracket_us": round(empty_us, 2),
                        "other_ops_ms_per_step": other}}


def albef_cpu_baseline(B_cpu=4, timed_steps=2):
    """The ALBEF oracle (oracle/albef_oracle.py, pinned to the reference's own modules: G10 / G11 / G12) on this node's host
    cores, full architecture, a BOUNDED sample: B_cpu-sample batches of the same synthetic recipe, 1 warm-up + 2 timed
    train_steps (dropout off, as the GPU line)."""
    from oracle import albef_oracle as A
    d = A.AlbefDims()
    ncpu = os.cpu_count()
    nt = min(ncpu, 32)
    torch.set_num_threads(nt)
    P = A.make_params(d)
    c = A.AlbefDatClient(P, d, lr=1e-4, steps_per_epoch=50)
    bs = [A.synthetic_batch(B_cpu, d, 1234 + i) for i in range(1 + timed_steps)]
    c.train_step(bs[0])
    t1 = time.time()
    for i in range(timed_steps):
        c.train_step(bs[1 + i])
    dt = time.time() - t1
    return dict(value=round(B_cpu * timed_steps / dt, 3), unit="samples/s", cores=nt, host_cores=ncpu, kind="port",
                sample=f"{timed_steps} timed train_steps after 1 warm-up at B={B_cpu} (the GPU line runs B=32/client), 384x384, "
                       f"25-token questions, one 4-token answer, ViT-B/16 + BERT-base 12 + 6 layers, fp32, torch "
                       f"{torch.__version__} CPU, {nt} threads")


def albef_flops(Ni, Lq, La, n_ans=1, H=768, I=3072, r=48, V=30522, vit=12, enc=12, fusion=6, dec=6):
    """Algorithmic FLOPs per sample of ALBEF's dual-adapter + MKD train_step, counted the way SURVEY.md 8d counts ViLT's:
    linears 2 M N K, attention 4 Sq Skv 64 per head forward, backward = dX only through the frozen linears (the same FLOPs as
    their forward), attention backward = 2 x its forward, an active adapter forward 2 x 2 H r per token and dX + dW = 2 x that;
    nothing below the first ViT adapter is back-propagated.
    -> (reference: P0 gated no-grad + P1 + P2 gated = 3 forwards + 2 backwards (task_trainer.py:280-330 around
        albef_model.py:69-145), executed: the engine runs the gated forward once (P0 == P2 here: only adapters train, dropout 0)
        = 2 forwards + 2 backwards, and the image encoder ahead of block 0's adapter once)."""
    lin = lambda M, N, K: 2.0 * M * N * K  # noqa: E731
    att = lambda Sq, Skv: 4.0 * Sq * Skv * H  # noqa: E731
    blk = lambda S: lin(S, 3 * H, H) + lin(S, H, H) + lin(S, I, H) + lin(S, H, I)  # noqa: E731
    ad = lambda S: 2.0 * S * H * r * 2  # noqa: E731
    vit_blk_lin, vit_blk_att = blk(Ni), att(Ni, Ni)
    patch = lin(Ni - 1, H, H)
    enc_self = enc * (blk(Lq) + att(Lq, Lq))
    enc_cross = (enc - fusion) * (lin(Lq, H, H) * 2 + lin(Ni, 2 * H, H) + att(Lq, Ni))
    dec_self = dec * (blk(La) + att(La, La)) * n_ans
    dec_cross = dec * (lin(La, H, H) * 2 * n_ans + lin(Lq, 2 * H, H) + att(La, Lq) * n_ans)
    head = n_ans * (La - 1) * (2.0 * H * H + 2.0 * H * V)
    lin_all = patch + vit * vit_blk_lin + enc_self - enc * att(Lq, Lq) + enc_cross - (enc - fusion) * att(Lq, Ni) + \
        dec_self - dec * att(La, La) * n_ans + dec_cross - dec * att(La, Lq) * n_ans + head
    att_all = vit * vit_blk_att + enc * att(Lq, Lq) + (enc - fusion) * att(Lq, Ni) + dec * n_ans * (att(La, La) + att(La, Lq))
    ad_tok = vit * Ni + enc * Lq + dec * La * n_ans           # adapter applications per sample (30 modules)
    fwd = lambda k: lin_all + att_all + k * ad(1) * ad_tok  # noqa: E731
    # backward: everything above block 0's adapter; block 0's own attention / MLP and the patch embedding are not differentiated
    below = patch + vit_blk_lin + vit_blk_att
    bwd = lambda k: (lin_all - patch - vit_blk_lin) + 2.0 * (att_all - vit_blk_att) + 2.0 * k * ad(1) * ad_tok  # noqa: E731
    reference = fwd(2) + fwd(1) + bwd(1) + fwd(2) + bwd(2)
    executed = fwd(2) + fwd(1) - below + bwd(1) + bwd(2)
    return reference, executed


def bench_albef(args, world, rank, dev, dist):
    """configs[3]: one ALBEF dual-adapter + MKD train_step per step (one hipGraph replay); with N clients the timed region ends with the FedAvg all-reduce of the 8.95 MB adapter_1 payload."""
    from feddat_amd import albef_engine, albef_spec, lib as L
    B = args.batch
    params = albef_spec.random_init(seed=0, image=args.res)
    eng = albef_engine.AlbefDatEngine(params, dev, batch=B, n_answers=B, image=args.res, dropout=args.albef_dropout,
                                      seed=1234 + rank, operands=args.operands)
    batches = [albef_spec.synthetic_batch(B, 1234 + 100 * rank + i, image=args.res, device=dev) for i in range(2)]
    eng.begin_local_update(steps_per_epoch=max(args.steps + args.warmup, 40))
    use_graph = not args.no_graph
    steps_r = hetero_steps(args.steps, rank) if args.hetero else args.steps
    exchange, coll = make_exchange(eng, world, rank, dist)
    tr = timed_round(lambda i: eng.train_step(batches[i % 2], use_graph=use_graph), steps_r, args.warmup, exchange, dist, dev)
    dt, total_steps, rows = gather_ranks(tr, B, world, rank, dist, dev)
    loss = float(eng.acts["gating"]["loss"][0])
    if not (loss == loss):
        raise RuntimeError("non-finite loss")
    if rank == 0:
        Ni = eng.Ni
        # executed FLOPs per sample: 2 ViT forwards + 2 backwards (dX only, weight grads of the adapters only) dominate
        vit_fwd = 12 * (2.0 * Ni * 768 * (3 * 768 + 768 + 2 * 3072) + 4.0 * Ni * Ni * 768)
        flops = 2 * vit_fwd + 2 * (vit_fwd * 11 / 12 + 12 * 4.0 * Ni * Ni * 768 * 1.5)
        sps = B * total_steps / dt
        out = {
            "metric": "VQA samples/sec, ALBEF dual-adapter local step", "value": round(sps, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "f16": "fp16"}[eng.operands], "data": "synthetic",
            "config": {"workload": "configs[3]: ALBEF (ViT-B/16 577 tokens + BERT-base 12 + 6 layers) dual-adapter + MKD, "
                                   f"batch={B}/client, {args.res}x{args.res}, 25-token questions, one 4-token answer each, "
                                   f"BERT dropout {args.albef_dropout}" + (" (0 = the parity configuration, SURVEY 8d)"
                                                                            if args.albef_dropout == 0 else "") +
                                   f", {'fp16 MFMA operands (the engine default; dynamic loss scale, GradScaler semantics on the device, initial 2^14; 3.2e-4 / 2.6e-4 on the reference 40-step full-size rounds)' if eng.operands == 'f16' else 'bf16 MFMA operands (7.6e-4 / 8.4e-4 on the reference 40-step full-size rounds; the default is fp16)'}",
                       "clients": world, "hip_graph": use_graph, "hetero_steps": bool(args.hetero), "collective": coll,
                       "last_loss_0": round(loss, 4)},
            "samples_per_sec_per_gpu": round(sps / world, 2),
            "mfma_frac_vit_flops_only": round(flops * B * total_steps / world / dt / PEAK_BF16, 4)}
        ref_f, exe_f = albef_flops(eng.Ni, eng.Lq, eng.La)
        out["flops_per_sample"] = {"reference_step": ref_f, "executed": exe_f,
                                   "note": "albef_flops(): counted as SURVEY.md 8d counts ViLT's step (3 fwd + 2 bwd of the reference; "
                                           "the engine runs 2 + 2: P0 == P2 with dropout 0)"}
        out["mfma_frac_reference_flops"] = round(sps / world * ref_f / PEAK_BF16, 4)      # speed-equivalent at the reference's count
        out["mfma_frac_executed_flops"] = round(sps / world * exe_f / PEAK_BF16, 4)
        if rows is not None:
            out["per_rank"], out["round_split"] = rows, round_split(rows, dt, coll and coll.get("payload_bytes"))
        if not args.no_roofline:
            try:
                out["roofline"] = albef_roofline(L, eng, batches)
            except Exception as e:
                out["roofline"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = albef_cpu_baseline()
            except Exception as e:
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def roofline_block(L, eng, batches, gemms):
    """achieved = FLOPs of all K1 launches of a step / their summed in-step durations (HIP events around the engine's own
    launches); the isolated figure (10 back-to-back launches per shape, operands MALL/L2-warm) is reported beside it and is
    NOT what `frac` is."""
    ach, tsum, rows, alg_bytes, launches, in_step_info = measure_gemms_in_step(L, eng, batches)
    ach_iso, tsum_iso, rows_iso = measure_gemms(L, gemms, operands=eng.operands)
    tr = profiled_traffic()
    kt = profiled_kernel_time()
    if kt:       # the same FLOPs over the kernel-trace durations of the committed profile (another run, maybe another box)
        kt["frac"] = round(ach * tsum / (kt["ms_per_step"] * 1e-3) / PEAK_BF16, 4)
    second = in_step_info.pop("second", None)
    return {"kernel_trace": kt, "second": second,
            "kernel": "gemm_nt_v3_kernel / gemm_nt_v2_kernel (K1, frozen-linear "
                      + ("fp8 (e4m3, block-scaled) + bf16" if eng.fp8 else "fp16" if eng.operands == "f16" else "bf16")
                      + " MFMA GEMM; all launches of one train_step, "
                      "FLOP-weighted, durations measured in-step)",
            "bound": "mfma", "achieved": round(ach / 1e12, 2), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_BF16, 4), "frac_in_step": round(ach / PEAK_BF16, 4),
            "frac_isolated": round(ach_iso / PEAK_BF16, 4),
            "traffic": tr["bytes_per_launch"] if tr else None, "algorithmic_bytes_per_launch": round(alg_bytes),
            "traffic_ratio": round(tr["bytes_per_launch"] / alg_bytes, 3) if tr else None,
            "traffic_source": (tr["source"] + ": " + tr["note"]) if tr else None, "launches_per_step": launches,
            "gemm_ms_per_step": round(tsum * 1e3, 3), "gemm_ms_per_step_isolated": round(tsum_iso * 1e3, 3),
            "shapes": rows, "shapes_isolated": rows_iso, "in_step": in_step_info}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # 50 warm + 200 timed
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--res", type=int, default=384)
    ap.add_argument("--workload", default="vilt", choices=["vilt", "albef"],
                    help="vilt = configs[1] / configs[2] (the headline metric); albef = configs[3]: ALBEF (ViT-B/16 + BERT-base) "
                         "dual-adapter + MKD step, 25-token questions, one 4-token answer per question")
    ap.add_argument("--fp8", action="store_true",
                    help="configs[4]: e4m3 MFMA for the QKV / FFN1 forward products of the frozen backbone (bf16 adapters); "
                         "quoted at --batch 64")
    ap.add_argument("--fp8-products", dest="fp8_products", type=int, default=7, choices=[6, 7],
                    help="--fp8: 7 (default) = also QKV^T on the block-scaled fp8 MFMA with MX-scaled e4m3 dqkv from the attention "
                         "backward; 6 = the round-3 / 4 configuration (A/B)")
    ap.add_argument("--operands", default=None, choices=["bf16", "f16"],
                    help="16-bit MFMA operand format of the frozen products, attention and adapters: IEEE half with a 2^14 loss "
                         "scale (default: the reference's own GPU arithmetic is fp16 autocast, and the format that meets the "
                         "north-star parity bar at round length) or bf16 -- same MFMA instruction rate, same bytes; both workloads "
                         "default to fp16 (ALBEF in bf16: inside the bar at the tested round lengths with a fifth to spare, -1.5 percent)")
    args_fixup = lambda a: setattr(a, "operands", a.operands or "f16")  # noqa: E731
    ap.add_argument("--hetero", action="store_true",
                    help="N > 1, SURVEY.md 8d config 3: rank r runs K * {40,50,60,70,80}[r mod 5] / 80 steps (heterogeneous "
                         "len(loader)) on answers drawn from its own Dirichlet(0.5) label prior; the imbalance is absorbed at the "
                         "round's barrier and shows up as wait_s")
    ap.add_argument("--albef-dropout", dest="albef_dropout", type=float, default=0.0,
                    help="--workload albef: BERT hidden / attention dropout inside train_step (reference recipe: 0.1; the "
                         "default 0 is the deterministic configuration SURVEY.md 8d quotes config 4 in)")
    ap.add_argument("--unfused-tail", action="store_true",
                    help="A/B switch: the round-3 serial tail (46 single-purpose launches) instead of csrc/head_tail.hip")
    ap.add_argument("--no-operand-ab", dest="no_operand_ab", action="store_true",
                    help="skip the short run of the other 16-bit operand format that the N = 1 line carries as other_operand_format")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--host-input", action="store_true",
                    help="also time the same steps with batches starting in pinned HOST memory (PCIe-inclusive rate, "
                         "uploads overlapped by feddat_amd.data.DevicePrefetcher); reported as extra fields")
    args = ap.parse_args()
    args_fixup(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the N-rank job (one process per GPU, rendezvous on 127.0.0.1)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"or without torch.distributed.run (the script then starts {args.gpus} ranks itself)")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # FEDDAT_FORCE_DEVICE / FEDDAT_DIST_BACKEND exist only so that the multi-rank control flow can be exercised on a
    # single-GPU box (two ranks on cuda:0 over gloo); the real launch uses one GPU per rank and RCCL ("nccl").
    if os.environ.get("FEDDAT_FORCE_DEVICE") is not None:
        local = int(os.environ["FEDDAT_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        # host-side rendezvous only (barriers, the communicator's unique id, the per-rank rows): gloo by default, so that the
        # data path's C-ABI communicator is the single RCCL communicator each GPU holds (make_exchange)
        backend = os.environ.get("FEDDAT_DIST_BACKEND", "gloo")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus
    from feddat_amd import engine, lib as L, vilt_spec
    dev = torch.device("cuda", local)
    if args.workload == "albef":
        return bench_albef(args, world, rank, dev, dist)
    B, res = args.batch, args.res
    tasks = [f"client{r}" for r in range(world)]
    task = tasks[rank]
    # identical frozen backbone + server adapter on every client (seed 0); heterogeneous data per client
    params = vilt_spec.random_init(12, tasks, seed=0, device="cpu")
    eng = engine.ViltDatEngine(params, [task], dev, batch=B, res=res, layers=12, fp8=args.fp8,
                               operands="bf16" if args.fp8 else args.operands, fp8_mx_dqkv=args.fp8_products == 7)
    eng.fused_tail = not args.unfused_tail
    nb = 4
    # --hetero: client r also draws its answers from its own Dirichlet(0.5) label prior
    prior = vilt_spec.client_label_prior(rank) if args.hetero else None
    batches = [vilt_spec.synthetic_batch(B, res, 1234 + 100 * rank + i, device=dev, label_prior=prior) for i in range(nb)]
    steps_per_epoch = max(args.steps + args.warmup, 40)
    eng.begin_local_update(task, steps_per_epoch=steps_per_epoch)
    use_graph = not args.no_graph
    steps_r = hetero_steps(args.steps, rank) if args.hetero else args.steps
    exchange, coll = make_exchange(eng, world, rank, dist)
    tr = timed_round(lambda i: eng.train_step(batches[i % nb], use_graph=use_graph), steps_r, args.warmup, exchange, dist, dev)
    dt, total_steps, rank_rows = gather_ranks(tr, B, world, rank, dist, dev)
    loss = float(eng.loss_buf["p2"][0])
    host_ms = None
    if args.host_input and world == 1:
        from feddat_amd.data import DevicePrefetcher, pin_batch
        host = [pin_batch({k: v.cpu() for k, v in b.items()}) for b in batches]
        up = lambda b: {k: v.to(dev, non_blocking=True) for k, v in b.items()}
        res_ms = {}
        for mode in ("sync", "prefetch"):
            src = (host[i % nb] for i in range(args.steps))
            it = DevicePrefetcher(src, up, dev) if mode == "prefetch" else (up(b) for b in src)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for b in it:
                eng.train_step(b, use_graph=use_graph)
            torch.cuda.synchronize()
            res_ms[mode] = (time.perf_counter() - t1) / args.steps * 1e3
        host_ms = res_ms
    if not (loss == loss) or abs(loss) > 1e6:
        raise RuntimeError(f"non-finite loss {loss}")
    # the OTHER 16-bit operand format on the same box, same batches, right behind the timed region (N = 1, default workload only):
    # box-to-box spread is +-4 %, so the fp16 / bf16 ratio is only meaningful measured in one run
    other_fmt = None
    if world == 1 and not args.fp8 and not args.no_operand_ab:
        alt = "bf16" if eng.operands == "f16" else "f16"
        e2 = engine.ViltDatEngine(params, [task], dev, batch=B, res=res, layers=12, operands=alt)
        e2.fused_tail = eng.fused_tail
        e2.begin_local_update(task, steps_per_epoch=steps_per_epoch)
        n2 = max(20, min(100, args.steps))
        for i in range(10):
            e2.train_step(batches[i % nb], use_graph=use_graph)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n2):
            e2.train_step(batches[i % nb], use_graph=use_graph)
        torch.cuda.synchronize()
        ms2 = (time.perf_counter() - t1) / n2 * 1e3
        other_fmt = {"operands": {"bf16": "bf16", "f16": "fp16"}[alt], "steps": n2, "ms_per_step": round(ms2, 3),
                     "samples_per_sec": round(B * 1e3 / ms2, 1),
                     "mfma_frac_reference_flops": round(B * 1e3 / ms2 * REF_FLOPS_PER_SAMPLE / PEAK_BF16, 4),
                     "parity": ("bf16 operands: max |ddW| < 1e-3 for rounds of up to 60 steps at B = 32, 1.3e-3 at 80" if alt == "bf16"
                                else "fp16 operands: max |ddW| 8.1e-4 after the longest (80-step) round at B = 32") +
                               " (tests/test_round_b32_gpu.py)"}
        del e2

    out = None
    if rank == 0:
        S = eng.S
        gemms, gemm_flops, exec_flops = flops_tables(B, S)
        sps = B * total_steps / dt
        extra_host = {} if host_ms is None else {
            "host_input_ms_per_step": {k: round(v, 3) for k, v in host_ms.items()},
            "host_input_samples_per_sec": {k: round(B * 1e3 / v, 1) for k, v in host_ms.items()}}
        if args.fp8:
            n8 = "seven" if eng.fp8_mx_dqkv else "six"
            workload = (
                f"configs[4]: ViLT-B/32 FedDAT, fp8 (e4m3) MFMA for {n8} of the eight frozen products per layer (QKV, FFN1, FFN2 forward; "
                "FFN2^T, FFN1^T, attention-output^T backward" +
                ("; QKV^T on MX block-scaled e4m3 dqkv written by the attention backward" if eng.fp8_mx_dqkv else "") +
                "), bf16 for the attention-output projection" + ("" if eng.fp8_mx_dqkv else " and QKV^T") +
                " (measured and declined: profiles/r05_fp8_remaining_products.txt); measured parity of this configuration at its own "
                "batch (tests/test_round40_gpu.py::test_b64_round_40_steps_vs_reference_golden, B=64, the reference's own 40-step "
                "round): mean |ddW| / mean |dW| 0.165, update norm within 4.1 %, max |ddW| 2.9e-3 adapters / 3.9e-3 head (the default "
                f"fp16-operand engine on the same round: 0.004, 0.2 %, 3.0e-4), batch={B}/client, ")
        else:
            workload = ("configs[1]" + (" with fp16 operands (BASELINE.json's configs[1] string says bf16: same operand width and MFMA "
                                        "rate; the bf16 step is in other_operand_format)" if eng.operands == "f16" else "")
                        + f": ViLT-B/32 FedDAT, 1 client per MI355X, {'fp16' if eng.operands == 'f16' else 'bf16'} MFMA operands"
                        + (f" (dynamic loss scale, GradScaler semantics on the device, initial 2^{int(math.log2(eng.loss_scale))})"
                           if eng.operands == "f16" and eng.scaler_state()["dynamic"] else
                           f" (static loss scale 2^{int(math.log2(eng.loss_scale))})" if eng.operands == "f16" else "")
                        + f", fp32 accumulate / masters, batch={B}/client, ")
        workload += "384x384 synthetic + 40-token questions, MKD on" + (
            f"; {world} clients + FedAvg all-reduce per round (configs[2])" if world > 1 else "")
        out = {
            "metric": "VQA samples/sec, ViLT-B/32 dual-adapter local step", "value": round(sps, 2),
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "fp8(e4m3)+bf16" if args.fp8 else {"bf16": "bf16", "f16": "fp16"}[eng.operands],
            "data": "synthetic",
            "config": {"workload": workload,
                       "batch_per_client": B, "seq_len": S, "clients": world, "hip_graph": use_graph,
                       "ranks": (dist.get_world_size() if dist is not None else 1), "hetero_steps": bool(args.hetero),
                       "hetero_label_prior": "Dirichlet(alpha=0.5) per client" if args.hetero else None,
                       "collective": coll, "last_loss_0": round(loss, 4)},
            "samples_per_sec_per_gpu": round(sps / world, 2),
            "mfma_frac_executed_flops": round(exec_flops * total_steps / world / dt / PEAK_BF16, 4),
            "mfma_frac_reference_flops": round(sps / world * REF_FLOPS_PER_SAMPLE / PEAK_BF16, 4),
        }
        out.update(extra_host)
        if other_fmt is not None:
            out["other_operand_format"] = other_fmt
        if rank_rows is not None:      # configs[2]: per-GPU rate and the round's split (compute / wait at the barrier / all-reduce)
            out["per_rank"], out["round_split"] = rank_rows, round_split(rank_rows, dt, coll and coll.get("payload_bytes"))
        if not args.no_roofline:
            try:
                out["roofline"] = roofline_block(L, eng, batches, gemms)
            except Exception as e:      # the throughput line must survive a failure of the auxiliary measurement
                out["roofline"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline({k: v.float().cpu() for k, v in params.items()},
                                                   [{k: v.cpu() for k, v in b.items()} for b in batches], B, res, task)
            except Exception as e:
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
