export RSLO_BENCH_MULTIRANK_CHILD=0 RSLO_BENCH_OTHER_CONFIGS=0
sleep 5
run() { env "$@" python bench.py --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['config']
print('$*', d['value'], d['ms_per_step'], c.get('host_issue_ms_per_step'))"; }
for i in 1 2 3; do
  run A=1
  run RSLO_ROW_ORDER=0
  run RSLO_PLAN_GATE=head
  run RSLO_PLAN_GATE=none
  run RSLO_HOST_LEAD=2
done
