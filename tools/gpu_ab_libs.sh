# usage: bash tools/gpu_ab_libs.sh [variant] [rounds]  -- the round-1 library and the current one on the same box, one process each
cd $GRAFT_REPO_ROOT
VAR=${1:-"w12:workers=12,timing=2"}; R=${2:-8}
for pass in 1 2; do
  RSI_HOT_LIB=$GRAFT_REPO_ROOT/rsicnv_amd/librsi_hot_r1.so timeout -k 10 300 python tools/ab_bench.py --rounds $R --variants "$VAR" 2>&1 | grep "mean" | sed "s/^/r1  /"
  timeout -k 10 300 python tools/ab_bench.py --rounds $R --variants "$VAR" 2>&1 | grep "mean" | sed "s/^/now /"
done
