# usage: bash tools/gpu_prof_ab.sh TAG "<ab_bench variant>" [rounds]  -> gpurun_out/prof_TAG/ (rocprofv3 kernel stats of tools/ab_bench.py)
set -e
cd $GRAFT_REPO_ROOT
TAG=$1; VAR=$2; ROUNDS=${3:-3}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python3 $GRAFT_REPO_ROOT/tools/ab_bench.py --rounds $ROUNDS --variants "$VAR" > $OUT/ab.log 2> $OUT/ab.err || { tail -20 $OUT/ab.err; exit 1; }
tail -4 $OUT/ab.log
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:32]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} min_us {float(r['MinNs'])/1e3:8.1f} max_us {float(r['MaxNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
