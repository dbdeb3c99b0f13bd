# usage: bash tools/gpu_inflight_sweep.sh  -- bench.py's step with 2, 3, 4 genomes queued in the pool
cd $GRAFT_REPO_ROOT
for pass in 1 2; do for i in 2 3 4; do
  timeout -k 10 200 python bench.py --steps 36 --warmup 3 --no-single --no-cpu-baseline --inflight $i 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight $i:', d['ms_per_step'], d['steps_identical'])"
done; done
