# usage: bash tools/gpu_streamers.sh  -- bench.py's step and the dominant kernel's in-situ fraction under 1 .. 4 per-base phases in flight
cd $GRAFT_REPO_ROOT
one() { # workers streamers
  RSI_HOT_STREAMERS=$2 timeout -k 10 200 python bench.py --steps 24 --warmup 3 --no-single --no-cpu-baseline --workers $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('workers $1 streamers $2:', d['ms_per_step'], 'one-at-a-time', d['one_genome_at_a_time']['ms_per_step'], r['kernel'], 'in situ', r['frac'], r['avg_launch_ms'], 'isolated', r['isolated']['frac'], d['rows_match_reference'])"
}
one 16 3; one 16 2; one 16 1; one 16 4; one 12 2; one 16 3
