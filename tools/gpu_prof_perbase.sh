# usage: bash tools/gpu_prof_perbase.sh TAG "CFG:CHR[,CFG:CHR]" ["ENV=VAL,..." ...]  -> rocprofv3 kernel stats of tools/perbase_probe.py
# (every per-base kernel alone on the chip: one chromosome at a time through one context)
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
TAG=$1; shift
export PERBASE_CASES=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o "$TAG" -- python3 "$GRAFT_REPO_ROOT/tools/perbase_probe.py" "$@" > "$OUT/probe.log" 2> "$OUT/probe.err" || { tail -20 "$OUT/probe.err"; exit 1; }
cat "$OUT/probe.log"
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} min_us {float(r['MinNs'])/1e3:8.1f} max_us {float(r['MaxNs'])/1e3:9.1f}")
PY
find "$OUT" -name "*kernel_trace.csv" -delete
