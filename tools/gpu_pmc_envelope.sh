# usage: bash tools/gpu_pmc_envelope.sh  -> gpurun_out/pmc_env/: PMC passes over tools/envelope_probe.py (the int32 K4 of the 300x case: k_cap_compact_bin)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_env
mkdir -p $OUT
cd /tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pass$i -o p -- python3 $GRAFT_REPO_ROOT/tools/envelope_probe.py > $OUT/pass$i.log 2> $OUT/pass$i.err || { tail -5 $OUT/pass$i.err; exit 1; }
  echo "pass $i done: $C"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt
grep -E "^kernel|k_cap_compact_bin|k_rescale_compact|k_gc_rescale|k_gc_joint" $OUT/summary.txt
