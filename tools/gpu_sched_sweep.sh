# usage: bash tools/gpu_sched_sweep.sh  -- bench.py's step under pool schedules (workers x per-base phases in flight), two genomes queued
cd $GRAFT_REPO_ROOT
one() { # workers streamers
  RSI_HOT_STREAMERS=$2 timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-single --no-cpu-baseline --workers $1 --inflight ${3:-2} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('workers $1 streamers $2 inflight ${3:-2}:', d['ms_per_step'], d['steps_identical'])"
}
one 16 3; one 12 2; one 20 3; one 16 4; one 16 3; one 20 4; one 14 3; one 12 2; one 16 3; one 18 3
