#!/usr/bin/env python
"""Per-kernel averages of a rocprofv3 --pmc counter_collection CSV: python tools/pmc_kernel_avg.py <csv> [substr]"""
import collections, csv, sys
per = collections.defaultdict(lambda: collections.defaultdict(float)); names = {}
for r in csv.DictReader(open(sys.argv[1])):
    per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d, c in per.items():
    for k, v in c.items(): agg[names[d].split("(")[0][:70]][k].append(v)
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for n, c in agg.items():
    if sub in n:
        print(n, {k: round(sum(v) / len(v)) for k, v in sorted(c.items())}, "n=", max(len(v) for v in c.values()))
