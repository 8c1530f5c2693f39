# alternating bench runs on one box: the round-6 changes off (A) / on (B).  (As recorded in profiles/r06_ab_round6_changes.txt the A
# side also ran with RSLO_TUNING=spconv_skip=0: k_spconv_v6 without the dead-block skip, deleted since.)
export RSLO_BENCH_MULTIRANK_CHILD=0 RSLO_BENCH_OTHER_CONFIGS=0
sleep 5
for i in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then E="RSLO_DEFER_WGRAD_REDUCE=0 RSLO_HEAD_GRAPH_INPUT=copy"; else E="RSLO_DEFER_WGRAD_REDUCE=1 RSLO_HEAD_GRAPH_INPUT=direct"; fi
    env $E python scripts/bench_with_tuning.py --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['config']
print('$v', d['value'], d['ms_per_step'], c.get('host_issue_ms_per_step'))"
  done
done
