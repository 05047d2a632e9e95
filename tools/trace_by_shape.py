#!/usr/bin/env python
"""Group a rocprofv3 kernel trace (…_kernel_trace.csv) by (kernel, grid size): calls, total ms, average us.
   python tools/trace_by_shape.py <trace.csv> [steps]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "").split("(")[0][:48]
    key = (name, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r.get("Grid_Size_Y", "1"))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[key][0] += 1
    agg[key][1] += d
tot = sum(v[1] for v in agg.values())
print(f"total {tot / 1e3 / steps:.3f} ms/step over {steps:g} steps")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{k[0]:50s} blocks {k[1]:6d} y {k[2]:>4s}  calls/step {v[0] / steps:7.1f}  ms/step {v[1] / 1e3 / steps:7.3f}  avg {v[1] / v[0]:7.1f} us")
