cd $GRAFT_REPO_ROOT
W=${1:-8}
for i in 1 2 3; do
  echo "--- run $i: threads alive $(ps -eLf | wc -l), procs $(ps -e | wc -l)"; grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo
  cat /proc/pressure/cpu 2>/dev/null | head -1
  python bench.py --steps 3 --warmup 1 --workers $W --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; p=d['worker_phase_ms_per_step']; print(round(d['value']/1e9,2), d['ms_per_step'], 'scan', k['rsi_scan'], 'k4', k['cap_compact_bin'], 'calls', p['a16-19.calls'], 'filt', p['scan.filterstatus'])"
done
grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo
