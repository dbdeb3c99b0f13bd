cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -4
  python bench.py --steps 3 --warmup 1 --workers 8 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; p=d['worker_phase_ms_per_step']; print(round(d['value']/1e9,2), d['ms_per_step'], 'scan', k['rsi_scan'], 'k4', k['cap_compact_bin'], 'calls', p['a16-19.calls'], 'filt', p['scan.filterstatus'])"
  [ $i = 2 ] && sleep 20
done
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -6
