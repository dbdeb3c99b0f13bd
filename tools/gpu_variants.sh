# usage: bash tools/gpu_variants.sh "CFG:CHR" lib1.so lib2.so ...   -> rocprofv3 kernel stats of perbase_probe per library build
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
export PERBASE_CASES=$1; shift
export TMPDIR=/tmp
for LIB in "$@"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/var_$(basename $LIB .so)
  mkdir -p "$OUT"
  (cd /tmp && RSI_HOT_LIB=$GRAFT_REPO_ROOT/$LIB timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o v -- python3 "$GRAFT_REPO_ROOT/tools/perbase_probe.py" "" > "$OUT/probe.log" 2> "$OUT/probe.err") || { tail -5 "$OUT/probe.err"; exit 1; }
  echo "== $LIB: $(grep default "$OUT/probe.log")"
  python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ("k_rescale_compact","k_bin_median","k_gc_joint","k_fasta_classify")):
        print(f"   {r['Name'].split('(')[0][-40:]:40s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:8.1f} min_us {float(r['MinNs'])/1e3:8.1f}")
PY
  find "$OUT" -name "*kernel_trace.csv" -delete
done
