# The ALBEF part of scripts/collect_r04.sh alone (bench line, kernel trace, PMC traffic passes).
set -x
R=$(pwd)
mkdir -p gpurun_out/r04
python bench.py --workload albef > gpurun_out/r04/bench_albef.json 2> gpurun_out/r04/bench_albef.err
rm -rf gpurun_out/prof_r04/albef_trace gpurun_out/prof_r04/albef_pmc_fetch_size gpurun_out/prof_r04/albef_pmc_write_size
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r04/albef_trace -o step --output-format csv -- python $R/bench.py --workload albef --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/r04/albef_trace.log 2>&1)
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo "$set" | awk '{print tolower($1)}')
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/prof_r04/albef_pmc_$name -o p --output-format csv -- python $R/bench.py --workload albef --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $R/gpurun_out/r04/albef_pmc_$name.log 2>&1)
done
python bench.py --workload albef --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r04/bench_albef_2.json
