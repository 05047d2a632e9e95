"""Timeline of ONE replayed step from a rocprofv3 --kernel-trace CSV: per kernel its duration and the idle gap in front of
it, totals of busy / idle time, and the serial tail (everything between the last forward GEMM and the first backward
attention).   python tools/trace_gaps.py gpurun_out/prof_r04/trace/step_kernel_trace.csv [--all]"""
import csv
import sys
import collections

path = sys.argv[1]
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
# steps are delimited by the stage_inputs kernel (first kernel of set_batch)
starts = [i for i, r in enumerate(rows) if "stage_inputs" in r[2] or "im2col" in r[2]]
marks = [i for i, r in enumerate(rows) if "im2col" in r[2]]
if len(marks) < 3:
    print("no step markers found")
    sys.exit(1)
a, b = marks[-2], marks[-1]          # the last complete step
step = rows[a:b]
t0, t1 = step[0][0], rows[b][0]
busy = sum(e - s for s, e, _ in step)
print(f"step: {len(step)} kernels, wall {(t1 - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us")
agg = collections.OrderedDict()
prev_end = step[0][0]
lines = []
for s, e, n in step:
    gap = s - prev_end
    prev_end = max(prev_end, e)
    k = short(n)
    d = agg.setdefault(k, [0, 0.0, 0.0])
    d[0] += 1
    d[1] += (e - s) / 1e3
    d[2] += max(gap, 0) / 1e3
    lines.append(f"{(s - t0) / 1e3:9.1f}  gap {gap / 1e3:6.2f}  dur {(e - s) / 1e3:7.2f}  {k}")
print(f"{'kernel':50s} {'n':>4s} {'busy us':>9s} {'gap-before us':>14s}")
for k, (c, d, g) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print(f"{k:50s} {c:4d} {d:9.1f} {g:14.1f}")
if "--all" in sys.argv:
    print("\n".join(lines))
