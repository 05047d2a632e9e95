# Round-4 evidence, run on the GPU box through gpurun from the repo root; outputs under gpurun_out/r04 and gpurun_out/prof_r04,
# summarised into profiles/r04_* by tools/summarize_profiles.py r04 (committed copies).
set -x
R=$(pwd)
mkdir -p gpurun_out/r04
python bench.py > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err
tail -c 400 gpurun_out/r04/bench_default.json
python bench.py --unfused-tail --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/r04/bench_unfused_tail.json
python tools/step_breakdown.py --detail > gpurun_out/r04/step_breakdown.txt 2>&1
bash tools/collect_profiles.sh r04 > gpurun_out/r04/collect.log 2>&1
python tools/trace_gaps.py $(find gpurun_out/prof_r04/trace -name step_kernel_trace.csv | head -1) > gpurun_out/r04/step_timeline.txt 2>&1
python tools/head_tail_bench.py > gpurun_out/r04/head_tail_bench.txt 2>&1
python bench.py --workload albef > gpurun_out/r04/bench_albef.json 2> gpurun_out/r04/bench_albef.err
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r04/albef_trace -o step --output-format csv -- python $R/bench.py --workload albef --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/r04/albef_trace.log 2>&1)
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo "$set" | awk '{print tolower($1)}')
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/prof_r04/albef_pmc_$name -o p --output-format csv -- python $R/bench.py --workload albef --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $R/gpurun_out/r04/albef_pmc_$name.log 2>&1)
done
for f in "" "--fp8"; do python bench.py --batch 64 $f --no-cpu-baseline --no-roofline --steps 100 --warmup 10 2>/dev/null | tail -1; done > gpurun_out/r04/bench_b64.json
ls -la gpurun_out/r04 gpurun_out/prof_r04 | head -40
