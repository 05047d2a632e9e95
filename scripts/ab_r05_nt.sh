# A/B on one box: the fp16 library as built vs the same with non-temporal hints on the heavy epilogues' streams (-DFD_EPI_NT:
# python -m feddat_amd.build --nt).  Step time (hipGraph), then FETCH_SIZE / WRITE_SIZE of the two code-epilogue GEMMs.
R=$(pwd)
cp feddat_amd/libfeddat_hip_f16.so /tmp/f16_base.so
run() {
  for i in 1 2; do python bench.py --no-cpu-baseline --no-roofline --no-operand-ab --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; done
  for set in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/prof_nt/$1_$set -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --no-operand-ab > /dev/null 2>&1)
    python - "$R/gpurun_out/prof_nt/$1_$set" $set $1 <<'PY'
import csv, sys, glob, collections
d, cname, tag = sys.argv[1:4]
f = glob.glob(d + "/**/p_counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(float); name = {}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == cname:
        per[r["Dispatch_Id"]] += float(r["Counter_Value"]); name[r["Dispatch_Id"]] = r["Kernel_Name"]
agg = collections.defaultdict(list)
for k, v in per.items():
    n = name[k]
    for key in ("gemm_nt_v2_kernel<5", "gemm_nt_v2_kernel<6", "gemm_nt_v3_kernel<0, 6", "gemm_nt_v3_kernel<1, 6"):
        if key in n:
            agg[key].append(v)
for k, v in sorted(agg.items()):
    print(f"{tag:5s} {cname:10s} {k:26s} {sum(v) / len(v) * (2 if cname == 'FETCH_SIZE' else 1) * 1024 / 1e6:8.1f} MB per launch ({len(v)} launches; FETCH doubled per the guide)")
PY
  done
}
run base
cp feddat_amd/libfeddat_hip_f16_nt.so feddat_amd/libfeddat_hip_f16.so
run nt
cp /tmp/f16_base.so feddat_amd/libfeddat_hip_f16.so
