# Same-box A/B of the step time: this tree against the round-3 library.  scratch/r03_head is a checkout of the r03 tree with
# its built libfeddat_hip.so (git worktree add scratch/r03_head 8a07f95 && (cd scratch/r03_head && python -m feddat_amd.build));
# scratch/ is not tracked.  Run through gpurun from the repo root: bash scripts/ab_r03.sh
for i in 1 2 3; do
  echo NEW $(python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")
  echo "OLD(r03)" $(cd scratch/r03_head && python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")
done
