# usage: bash tools/gpu_pmc.sh TAG "<bench args>"   -> gpurun_out/pmc_TAG/passN/ (one rocprofv3 --pmc pass each)
# Counters are collected alone (no --kernel-trace/--stats), as the pool's rules require.
set -e
cd $GRAFT_REPO_ROOT
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 600 rocprofv3 --pmc $C --output-format csv -d $OUT/pass$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py $@ > $OUT/pass$i.json 2> $OUT/pass$i.err || { tail -5 $OUT/pass$i.err; exit 1; }
  echo "pass $i done: $C"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT --json $OUT/pmc_traffic.json ${PMC_BASES:-60000000} > $OUT/summary.txt
cat $OUT/summary.txt
