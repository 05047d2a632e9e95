set -x
R=$(pwd)
mkdir -p gpurun_out/r03
python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err
tail -c 600 gpurun_out/r03/bench_default.json
python tools/step_breakdown.py --detail > gpurun_out/r03/step_breakdown.txt 2>&1
bash tools/collect_profiles.sh r03 > gpurun_out/r03/collect.log 2>&1
python bench.py --workload albef > gpurun_out/r03/bench_albef.json 2> gpurun_out/r03/bench_albef.err
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r03/albef_trace -o step --output-format csv -- python $R/bench.py --workload albef --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/r03/albef_trace.log 2>&1)
for f in "" "--fp8"; do python bench.py --batch 64 $f --no-cpu-baseline --no-roofline --steps 100 --warmup 10 2>/dev/null | tail -1; done > gpurun_out/r03/bench_b64.json
python tools/gemm_fp8_vs_bf16.py > gpurun_out/r03/gemm_fp8_vs_bf16.txt 2>&1
python tools/adapter_ablate.py > gpurun_out/r03/adapter_ablate.txt 2>&1
python tools/gemm_defer_probe.py > gpurun_out/r03/gemm_defer_probe.txt 2>&1
python tools/attn_ablate.py > gpurun_out/r03/attn_ablate.txt 2>&1
python tools/gemm_epi_split.py > gpurun_out/r03/gemm_epi_split.txt 2>&1
ls -la gpurun_out/r03 gpurun_out/prof_r03 | head -40
