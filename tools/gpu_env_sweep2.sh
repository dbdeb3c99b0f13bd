# usage: bash tools/gpu_env_sweep2.sh "A=1 B=2" "A=3" ... [-- bench args]  -- bench.py's step under sets of library switches, twice over
cd $GRAFT_REPO_ROOT
SETS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do SETS+=("$1"); shift; done
[ "$1" = "--" ] && shift
for pass in 1 2; do
  for v in "${SETS[@]}"; do
    env $v timeout -k 10 200 python bench.py --steps 24 --warmup 3 --no-single --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v:', d['ms_per_step'], d['steps_identical'], d['rows_sha256'][:8])"
  done
done
