#!/usr/bin/env python
"""Turn the rocprofv3 outputs under gpurun_out/ into the small per-round summaries kept under profiles/.
   python tools/summarize_profiles.py r01
"""
import collections
import csv
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
os.makedirs("profiles", exist_ok=True)


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    return n.split("(")[0][:60]


# 1. kernel-trace --stats summary of one bench run
st = f"gpurun_out/prof_{tag}/trace/step_kernel_stats.csv"
if not os.path.exists(st):
    st = "gpurun_out/prof_r1/step_kernel_stats.csv"
if os.path.exists(st):
    rows = list(csv.DictReader(open(st)))
    with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"])
        for r in rows:
            if float(r["Percentage"]) < 0.05 and not any(m in r["Name"] for m in ("dat_step_finish", "step_tick_multi")):
                continue          # (the step's last kernel stays whatever its share: bench.py counts the steps of the trace by it)
            w.writerow([short(r["Name"]), r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 3),
                        round(float(r["AverageNs"]) / 1e3, 2), round(float(r["MinNs"]) / 1e3, 2),
                        round(float(r["MaxNs"]) / 1e3, 2), r["Percentage"]])
    print("wrote", f"profiles/{tag}_kernel_stats.csv")

# 2. PMC passes: per-kernel averages
out = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_sq_wave_cycles", "pmc_fetch_size", "pmc_write_size", "pmc_sq", "pmc_fetch", "pmc_write"):
    p = f"gpurun_out/prof_{tag}/{d}/p_counter_collection.csv"
    if not os.path.exists(p):
        p = f"gpurun_out/{d}/p_counter_collection.csv" if tag == "r01" else p
    if not os.path.exists(p):
        continue
    per_dispatch = collections.defaultdict(lambda: collections.defaultdict(float))
    names = {}
    for r in csv.DictReader(open(p)):
        per_dispatch[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
    for did, c in per_dispatch.items():
        for k, v in c.items():
            out[names[did]][k].append(v)
if out:
    keys = sorted({k for c in out.values() for k in c})
    with open(f"profiles/{tag}_pmc_per_kernel.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches"] + [k + "_avg" for k in keys])
        for n, c in sorted(out.items(), key=lambda x: -len(next(iter(x[1].values())))):
            nd = max(len(v) for v in c.values())
            if nd < 2:
                continue
            w.writerow([n, nd] + [round(sum(c[k]) / len(c[k]), 1) if k in c else "" for k in keys])
    print("wrote", f"profiles/{tag}_pmc_per_kernel.csv")


# 3. ALBEF (configs[3]): kernel-trace summary + PMC passes of `bench.py --workload albef`
st = f"gpurun_out/prof_{tag}/albef_trace/step_kernel_stats.csv"
if os.path.exists(st):
    rows = list(csv.DictReader(open(st)))
    with open(f"profiles/{tag}_albef_kernel_stats.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"])
        for r in rows:
            if float(r["Percentage"]) < 0.05 and not any(m in r["Name"] for m in ("dat_step_finish", "step_tick_multi")):
                continue          # (the step's last kernel stays whatever its share: bench.py counts the steps of the trace by it)
            w.writerow([short(r["Name"]), r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 3),
                        round(float(r["AverageNs"]) / 1e3, 2), round(float(r["MinNs"]) / 1e3, 2),
                        round(float(r["MaxNs"]) / 1e3, 2), r["Percentage"]])
    print("wrote", f"profiles/{tag}_albef_kernel_stats.csv")
out = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("albef_pmc_fetch_size", "albef_pmc_write_size"):
    p = f"gpurun_out/prof_{tag}/{d}/p_counter_collection.csv"
    if not os.path.exists(p):
        continue
    per_dispatch = collections.defaultdict(lambda: collections.defaultdict(float))
    names = {}
    for r in csv.DictReader(open(p)):
        per_dispatch[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
    for did, c in per_dispatch.items():
        for k, v in c.items():
            out[names[did]][k].append(v)
if out:
    keys = sorted({k for c in out.values() for k in c})
    with open(f"profiles/{tag}_albef_pmc_per_kernel.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches"] + [k + "_avg" for k in keys])
        for n, c in sorted(out.items(), key=lambda x: -len(next(iter(x[1].values())))):
            nd = max(len(v) for v in c.values())
            if nd < 2:
                continue
            w.writerow([n, nd] + [round(sum(c[k]) / len(c[k]), 1) if k in c else "" for k in keys])
    print("wrote", f"profiles/{tag}_albef_pmc_per_kernel.csv")
