# kernel-trace of the graph-replayed step in both operand formats on one box (which kernels carry the f16 - bf16 difference)
R=$(pwd)
for f in bf16 f16; do
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ab/$f -o step --output-format csv -- python $R/bench.py --operands $f --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-operand-ab > $R/gpurun_out/prof_ab/$f.log 2>&1)
  st=$(find gpurun_out/prof_ab/$f -name "step_kernel_stats.csv" | head -1)
  cp $st gpurun_out/ab_${f}_kernel_stats.csv
done
python - <<'PY'
import csv
def rd(f):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = rd("gpurun_out/ab_bf16_kernel_stats.csv"), rd("gpurun_out/ab_f16_kernel_stats.csv")
rows = []
for k in a:
    if k in b:
        rows.append((b[k][1] - a[k][1], k[:90], a[k][0], a[k][1] / a[k][0] / 1e3, b[k][1] / b[k][0] / 1e3))
rows.sort(reverse=True)
print("total bf16 %.1f ms f16 %.1f ms" % (sum(v[1] for v in a.values()) / 1e6, sum(v[1] for v in b.values()) / 1e6))
for d, k, n, ua, ub in rows[:14] + rows[-4:]:
    print(f"{d/1e6:8.3f} ms  {n:5d} calls  {ua:8.2f} -> {ub:8.2f} us  {k}")
PY
