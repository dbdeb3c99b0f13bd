# usage: bash tools/gpu_ab.sh REPS "ENV=V ..." "ENV=V ..."  -- bench.py's step under two environments, interleaved REPS times, one process per run; prints both lists and their medians
cd $GRAFT_REPO_ROOT
REPS=$1; A="$2"; B="$3"
run() { env $1 timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-single --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
LA=""; LB=""
for i in $(seq $REPS); do LA="$LA $(run "$A")"; LB="$LB $(run "$B")"; done
python - <<PY
import statistics as st
a=[float(x) for x in "$LA".split()]; b=[float(x) for x in "$LB".split()]
print("A [$A]:", a, "median", st.median(a)); print("B [$B]:", b, "median", st.median(b))
PY
