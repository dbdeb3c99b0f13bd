# usage: bash tools/gpu_prof_lone.sh TAG [config]  -> rocprofv3 kernel stats of tools/lone_phases.py
set -e
cd $GRAFT_REPO_ROOT
TAG=$1; CFG=${2:-3}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python3 $GRAFT_REPO_ROOT/tools/lone_phases.py $CFG > $OUT/lone.log 2> $OUT/lone.err || { tail -20 $OUT/lone.err; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:40]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} min_us {float(r['MinNs'])/1e3:8.1f} max_us {float(r['MaxNs'])/1e3:9.1f}")
PY
