# usage: bash tools/gpu_pmc_probe.sh TAG "<perbase_probe variants...>"  -> gpurun_out/pmcp_TAG/summary.txt  (SQ / TA / TCP counters of the per-base kernels)
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcp_$TAG
mkdir -p "$OUT"
cd /tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_WAIT_INST_LDS" \
         "TA_TA_BUSY_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 500 rocprofv3 --pmc $C --output-format csv -d "$OUT/pass$i" -o p -- python3 "$GRAFT_REPO_ROOT/tools/perbase_probe.py" "$@" > "$OUT/pass$i.log" 2> "$OUT/pass$i.err" || { tail -5 "$OUT/pass$i.err"; exit 1; }
  echo "pass $i done"
done
python3 "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt"
python3 - "$OUT/summary.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(k in r["kernel"] for k in ("k_gc_joint", "k_rescale_compact", "k_bin_median", "k_cap_compact_bin8", "k_fasta")):
        print(r["kernel"])
        print("   " + "  ".join(f"{k}={v}" for k, v in r.items() if k != "kernel" and v))
PY
