# Round-6 evidence, run on the GPU box through gpurun from the repo root; outputs under gpurun_out/r06 and gpurun_out/prof_r06,
# summarised into profiles/r06_* by tools/summarize_profiles.py r06 (committed copies).
set -x
R=$(pwd)
mkdir -p gpurun_out/r06
rocminfo 2>/dev/null | grep -m1 "Marketing Name.*MI" > gpurun_out/r06/device.txt
python bench.py > gpurun_out/r06/bench_default.json 2> gpurun_out/r06/bench_default.err
tail -c 300 gpurun_out/r06/bench_default.json
# the operand formats next to each other on this box (no roofline / CPU legs): fp16 (default), bf16, fp16, bf16
for f in f16 bf16 f16 bf16; do python bench.py --operands $f --no-cpu-baseline --no-roofline --no-operand-ab 2>/dev/null | tail -1; done > gpurun_out/r06/bench_operands_ab.json
python tools/step_breakdown.py --detail > gpurun_out/r06/step_breakdown.txt 2>&1
bash tools/collect_profiles.sh r06 > gpurun_out/r06/collect.log 2>&1
python tools/trace_gaps.py $(find gpurun_out/prof_r06/trace -name step_kernel_trace.csv | head -1) > gpurun_out/r06/step_timeline.txt 2>&1
python bench.py --workload albef > gpurun_out/r06/bench_albef.json 2> gpurun_out/r06/bench_albef.err
python bench.py --workload albef --operands bf16 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/r06/bench_albef_bf16.json
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r06/albef_trace -o step --output-format csv -- python $R/bench.py --workload albef --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/r06/albef_trace.log 2>&1)
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo "$set" | awk '{print tolower($1)}')
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/prof_r06/albef_pmc_$name -o p --output-format csv -- python $R/bench.py --workload albef --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $R/gpurun_out/r06/albef_pmc_$name.log 2>&1)
done
for f in "" "--fp8"; do python bench.py --batch 64 $f --no-cpu-baseline --no-roofline --steps 100 --warmup 10 2>/dev/null | tail -1; done > gpurun_out/r06/bench_b64.json
ls -la gpurun_out/r06 gpurun_out/prof_r06 | head -40
