# usage: bash tools/gpu_sched_sweep.sh [bench args, e.g. --config 5]  -- bench.py's step under pool schedules (workers x per-base phases in flight)
cd $GRAFT_REPO_ROOT
EXTRA="$@"
one() { # workers streamers inflight
  RSI_HOT_STREAMERS=$2 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-single --no-cpu-baseline --workers $1 --inflight ${3:-2} $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('workers $1 streamers $2 inflight ${3:-2}:', d['ms_per_step'], d['steps_identical'])"
}
one 16 3; one 12 2; one 16 2; one 12 3; one 16 3; one 12 2; one 12 2 1; one 16 4
