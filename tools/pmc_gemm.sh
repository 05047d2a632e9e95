#!/usr/bin/env bash
# SQ counters of one K1 launch shape: bash tools/pmc_gemm.sh M N K epi flags
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM"; do
  n=$(echo $set | awk '{print tolower($1)}')
  timeout 120 rocprofv3 --pmc $set --kernel-trace -d /tmp/pg_$n -o p --output-format csv -- python $R/tools/gemm_one.py "$@" > /tmp/pg_$n.log 2>&1
  python - <<PY
import csv, collections
try:
    rows=list(csv.DictReader(open("/tmp/pg_$n/p_counter_collection.csv")))
except Exception as e:
    print("no csv", e); rows=[]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "gemm_nt" in r["Kernel_Name"]:
        agg[r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
done
