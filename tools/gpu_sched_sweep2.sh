# usage: bash tools/gpu_sched_sweep2.sh  -- bench.py's step under a few pool schedules, twice each (workers x per-base phases in flight), one process per run
cd $GRAFT_REPO_ROOT
one() { # workers streamers
  RSI_HOT_STREAMERS=$2 timeout -k 10 200 python bench.py --steps 24 --warmup 3 --no-single --no-cpu-baseline --workers $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('workers $1 streamers $2:', d['ms_per_step'], d['steps_identical'])"
}
for rep in 1 2; do one 16 3 && one 16 4 && one 16 5 && one 20 4 && one 12 4; done
