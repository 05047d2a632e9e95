# Same-box A/B of the step time: this tree against the round-2 library.  scratch/old_head is a checkout of the r02 tree with
# its built libfeddat_hip.so (git worktree add scratch/old_head aea45ab && (cd scratch/old_head && python __graft_entry__.py));
# scratch/ is not tracked.  Run through gpurun from the repo root: bash scripts/ab_r02.sh
for i in 1 2; do
  echo NEW $(python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")
  echo "OLD(r02)" $(cd scratch/old_head && python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")
done
