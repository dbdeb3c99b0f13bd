# usage: bash tools/gpu_sched_sweep.sh [bench args, e.g. --config 5]  -- bench.py's step under pool schedules (workers x per-base phases in flight x spins of a wait before it naps)
cd $GRAFT_REPO_ROOT
EXTRA="$@"
one() { # workers streamers spins
  RSI_HOT_SPIN=${3:-2000} RSI_HOT_STREAMERS=$2 timeout -k 10 200 python bench.py --steps 24 --warmup 3 --no-single --no-cpu-baseline --workers $1 $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('workers $1 streamers $2 spins ${3:-2000}:', d['ms_per_step'], d['steps_identical'])"
}
one 16 3 0; one 20 3 0; one 24 3 0; one 24 4 0; one 32 4 0; one 16 3 2000; one 20 4 0; one 24 3 100
