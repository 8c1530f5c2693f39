# where the structure plan of a coming batch may start (RSLO_PLAN_GATE), re-measured on the shortened step of the end of round 6
export RSLO_BENCH_MULTIRANK_CHILD=0 RSLO_BENCH_OTHER_CONFIGS=0
sleep 5
run() { env "$@" python bench.py --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['config']
print('$*', d['value'], d['ms_per_step'], c.get('host_issue_ms_per_step'))"; }
for i in 1 2 3; do
  run RSLO_PLAN_GATE=loss
  run RSLO_PLAN_GATE=head
  run RSLO_PLAN_GATE=fwd_end
  run RSLO_PLAN_GATE=none
  run RSLO_PLAN_GATE=loss RSLO_COV_STREAM=2
done
