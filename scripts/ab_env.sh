# A/B of one environment switch over alternating bench.py runs:  bash scripts/ab_env.sh VAR valueA valueB [rounds]
VAR=$1; A=$2; B=$3; R=${4:-3}
sleep ${SLEEP:-10}
for i in $(seq $R); do
  for v in "$A" "$B"; do
    env $VAR="$v" python bench.py --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['config']
print('$VAR=$v', d['value'], d['ms_per_step'], c.get('host_issue_ms_per_step'))"
  done
done
