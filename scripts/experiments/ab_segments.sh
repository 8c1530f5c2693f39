# target segments per query wave of the pruned chamfer search (chamfer_segments; 0 = the library's choice, 4 on C3) inside the step
export RSLO_BENCH_MULTIRANK_CHILD=0 RSLO_BENCH_OTHER_CONFIGS=0
sleep 5
for i in 1 2; do
  for v in 0 1 2 3 6 8; do
    RSLO_TUNING="chamfer_segments=$v" python scripts/bench_with_tuning.py --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['config']
print('chamfer_segments=$v', d['value'], d['ms_per_step'], c.get('final_loss'))"
  done
done
