# usage: gpu_ab_env.sh "ENV=VAL" [reps]: the pooled bench step with and without one environment switch, alternating
sw="$1"; reps="${2:-2}"
for r in $(seq 1 $reps); do
  for mode in off on; do
    if [ $mode = on ]; then export "$sw"; else unset "${sw%%=*}"; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$mode $sw', d['ms_per_step'], d['one_genome_at_a_time']['ms_per_step'], d['rows_match_reference'], {k[:10]: round(v['ms'], 3) for k, v in d['single_chromosome_configs'].items()})"
  done
done
