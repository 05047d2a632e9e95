# configs[4] at B = 64: six-product fp8 (r03 / r04) vs seven (MX e4m3 dqkv + QKV^T on the block-scaled MFMA), kernel traces on one box
R=$(pwd)
mkdir -p gpurun_out/prof_fp8mx
for p in 6 7; do
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fp8mx/p$p -o step --output-format csv -- python $R/bench.py --batch 64 --fp8 --fp8-products $p --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_fp8mx/p$p.log 2>&1)
  cp $(find gpurun_out/prof_fp8mx/p$p -name "step_kernel_stats.csv" | head -1) gpurun_out/fp8mx_p${p}_kernel_stats.csv
done
python - <<'PY'
import csv
def rd(f):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = rd("gpurun_out/fp8mx_p6_kernel_stats.csv"), rd("gpurun_out/fp8mx_p7_kernel_stats.csv")
print("total six %.1f ms seven %.1f ms" % (sum(v[1] for v in a.values()) / 1e6, sum(v[1] for v in b.values()) / 1e6))
keys = sorted(set(a) | set(b), key=lambda k: -abs(b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1]))
for k in keys[:10]:
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    print(f"{(tb - ta) / 1e6:8.3f} ms  six {ca:5d} x {ta / max(ca, 1) / 1e3:7.2f} us | seven {cb:5d} x {tb / max(cb, 1) / 1e3:7.2f} us  {k[:100]}")
PY
