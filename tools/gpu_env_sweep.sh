# usage: bash tools/gpu_env_sweep.sh "ENV=V ENV2=V2|--workers 16" "..." ...  -- bench.py's step under environment / argument variants, one process per run,
# every variant twice (interleaved)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for V in "$@"; do
    E="${V%%|*}"; A=""; case "$V" in *"|"*) A="${V#*|}";; esac
    env $E timeout -k 10 200 python bench.py --steps 24 --warmup 3 --no-single --no-cpu-baseline $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$V]', d['ms_per_step'], d['steps_identical'], d['rows_match_reference'], 'K2j in situ', d['roofline']['avg_launch_ms'])" || echo "[$V] failed"
  done
done
