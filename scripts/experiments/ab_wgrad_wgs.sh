export RSLO_BENCH_MULTIRANK_CHILD=0 RSLO_BENCH_OTHER_CONFIGS=0
sleep 5
for i in 1 2; do
  for v in 768 512 384 1024; do
    RSLO_TUNING="conv2d_wgrad_wgs=$v" python scripts/bench_with_tuning.py --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['config']
print('wgs=$v', d['value'], d['ms_per_step'])"
  done
done
