#!/usr/bin/env bash
# Run on the GPU box (through gpurun) from the repo root: kernel-trace summary of the default bench command plus the
# separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ set; counters in their own runs, --kernel-trace only), every pass under
# its own timeout.  Outputs under gpurun_out/prof_$TAG; tools/summarize_profiles.py $TAG turns them into profiles/$TAG_*.csv.
TAG=${1:-r02}
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o step --output-format csv -- \
  python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-operand-ab > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo "$set" | awk '{print tolower($1)}')
  timeout 400 rocprofv3 --pmc $set --kernel-trace -d "$OUT/pmc_$name" -o p --output-format csv -- \
    python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --no-operand-ab > "$OUT/pmc_$name.log" 2>&1
  echo "pmc $name rc=$?"
done
