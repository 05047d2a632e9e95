cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY"; do
  n=$(echo $set | awk '{print tolower($1)}')
  REPS=1 timeout 120 rocprofv3 --pmc $set --kernel-trace -d /tmp/pm_$n -o p --output-format csv -- python /root/repo/tools/attn_bench.py > /tmp/pm_$n.log 2>&1
  echo "$n rc=$?"
  python - <<PY
import csv, collections
try:
    rows=list(csv.DictReader(open("/tmp/pm_$n/p_counter_collection.csv")))
except Exception as e:
    print("no csv", e); rows=[]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "attn_" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:40], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
done
