# C2 (one scan per step, replayed encoder pass + structure plan of the coming scans): what moves the 0.75 ms
run() { env "$@" python bench.py --config c2 --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['config'].get('runner_host_split'))"; }
run A=1
run RSLO_INFER_PLAN_PRIORITY=-1
run RSLO_INFER_PLAN_PRIORITY=1
run RSLO_INFER_PLAN_PRIORITY=-1 RSLO_INFER_PLAN_STREAMS=2
run GPU_MAX_HW_QUEUES=8
run RSLO_C2_GRAPH=0
