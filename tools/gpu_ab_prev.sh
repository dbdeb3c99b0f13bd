# usage: bash tools/gpu_ab_prev.sh [variant] [rounds]  -- rsicnv_amd/librsi_hot_prev.so (built from an earlier commit) and the current
# library on the same box, one process each, twice over (box-to-box noise is larger than most differences)
cd $GRAFT_REPO_ROOT
VAR=${1:-"w12:workers=12,timing=3"}; R=${2:-8}
for pass in 1 2; do
  RSI_HOT_LIB=$GRAFT_REPO_ROOT/rsicnv_amd/librsi_hot_prev.so timeout -k 10 300 python tools/ab_bench.py --rounds $R --variants "$VAR" 2>&1 | grep "mean" | sed "s/^/prev /"
  timeout -k 10 300 python tools/ab_bench.py --rounds $R --variants "$VAR" 2>&1 | grep "mean" | sed "s/^/now  /"
done
