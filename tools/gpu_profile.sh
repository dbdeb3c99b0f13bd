# usage: bash tools/gpu_profile.sh TAG [bench args]   -> gpurun_out/prof_TAG/ (kernel trace + stats)
set -e
cd $GRAFT_REPO_ROOT
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
ls -R $OUT | head -30
tail -3 $OUT/bench.json
