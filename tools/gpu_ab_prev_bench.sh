# usage: bash tools/gpu_ab_prev_bench.sh [bench args]  -- bench.py's step with rsicnv_amd/librsi_hot_prev.so (built from an earlier commit) and
# with the current library, alternating, three times over
cd $GRAFT_REPO_ROOT
one() { timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-single --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['steps_identical'], d['rows_sha256'][:8])"; }
for pass in 1 2 3; do
  echo -n "prev: "; RSI_HOT_LIB=$GRAFT_REPO_ROOT/rsicnv_amd/librsi_hot_prev.so one "$@"
  echo -n "now:  "; one "$@"
done
