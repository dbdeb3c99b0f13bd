# usage: bash tools/gpu_env_sweep.sh VAR v1 v2 ... [-- bench args]  -- bench.py's step with an environment switch of the library at several values
cd $GRAFT_REPO_ROOT
VAR=$1; shift
VALS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done
[ "$1" = "--" ] && shift
for pass in 1 2; do
  for v in "${VALS[@]}"; do
    env $VAR=$v timeout -k 10 200 python bench.py --steps 24 --warmup 3 --no-single --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v:', d['ms_per_step'], d['steps_identical'], d['rows_sha256'][:8])"
  done
done
