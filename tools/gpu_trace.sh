# usage: bash tools/gpu_trace.sh TAG [bench args]  -> gpurun_out/trace_TAG/{analysis.txt,kernel_stats.csv}
set -e
cd $GRAFT_REPO_ROOT
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_analyze.py $F > $OUT/analysis.txt
cat $OUT/analysis.txt
S=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp $S $OUT/kernel_stats.csv
rm -f $F   # the full trace is tens of MB
