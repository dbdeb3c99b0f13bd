# usage: bash tools/gpu_round_profiles.sh TAG   -- the round's judged profiles in one call: the default bench line, the same command under
# rocprofv3 --kernel-trace --stats (pooled), and the genome through ONE worker (every kernel alone on the chip) -> gpurun_out/round_TAG/
set -e
cd $GRAFT_REPO_ROOT
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/round_$TAG
mkdir -p $OUT
timeout -k 10 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
bash tools/gpu_trace.sh ${TAG}_pool > $OUT/trace_pool.log 2>&1
cp gpurun_out/trace_${TAG}_pool/kernel_stats.csv $OUT/kernel_stats_3Gb_pool.csv
cp gpurun_out/trace_${TAG}_pool/bench.json $OUT/bench_under_rocprof.json
cp gpurun_out/trace_${TAG}_pool/analysis.txt $OUT/trace_analysis_pool.txt
echo "pooled trace done"
bash tools/gpu_trace.sh ${TAG}_w1 --workers 1 --inflight 1 --steps 15 --warmup 2 --no-single --no-cpu-baseline > $OUT/trace_w1.log 2>&1
cp gpurun_out/trace_${TAG}_w1/kernel_stats.csv $OUT/kernel_stats_3Gb_w1.csv
echo "one-worker trace done"
timeout -k 10 300 python bench.py --config 5 --no-cpu-baseline --no-single > $OUT/bench_config5.json 2> $OUT/bench_config5.err
echo "config 5 done"
timeout -k 10 400 python tools/rank_probe.py > $OUT/rank_probe.txt 2> $OUT/rank_probe.err
echo "rank probe done"
