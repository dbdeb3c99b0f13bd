# usage: bash tools/gpu_k4s_grid.sh  -- K4s's grid (workgroups): the kernel alone on the chip, then the pooled step, one process per setting
cd $GRAFT_REPO_ROOT
PERBASE_CASES="4:11,3:0" timeout -k 10 300 python tools/perbase_probe.py "" "RSI_HOT_K4S_GRID=768" "RSI_HOT_K4S_GRID=512" "RSI_HOT_K4S_GRID=1536" 2>&1 | grep -v amdgpu.ids
for g in 1024 768 512 1024 768 512; do
  RSI_HOT_K4S_GRID=$g timeout -k 10 200 python bench.py --steps 24 --warmup 3 --no-single --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid $g:', d['ms_per_step'], d['steps_identical'], d['rows_match_reference'])"
done
