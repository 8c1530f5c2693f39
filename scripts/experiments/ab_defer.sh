export RSLO_BENCH_MULTIRANK_CHILD=0 RSLO_BENCH_OTHER_CONFIGS=0
sleep 5
for i in 1 2 3; do
  for v in 0 1 8 16; do
    RSLO_DEFER_WGRAD_REDUCE=$v python bench.py --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['config']
print('defer=$v', d['value'], d['ms_per_step'], c.get('host_issue_ms_per_step'))"
  done
done
