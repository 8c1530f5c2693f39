# The measurement pass behind profiles/<round>_*: three bench runs, the bf16 line, rocprof kernel stats + one-step
# breakdowns (whole step and training stream alone), the three PMC passes (stamped with the library hash).
R=${ROUND:-r02}
# the PMC passes first: their summaries (stamped with the hash of the library sources) go to profiles/ on the box, so that the
# bench lines written below quote them (roofline.traffic, mfma_busy_pct)
ROUND=$R bash scripts/pmc_bench_traffic.sh > gpurun_out/pmc_traffic.log 2>&1; grep "v6<64, 64, 2" gpurun_out/pmc_traffic.log
ROUND=$R bash scripts/pmc_mfma_busy.sh > gpurun_out/pmc_busy.log 2>&1; grep "v6<64, 64, 2\|conv2d_fwd\|conv2d_str" gpurun_out/pmc_busy.log
ROUND=$R bash scripts/pmc_sq_waits.sh > gpurun_out/pmc_sq.log 2>&1; grep "v6<64, 64, 2" gpurun_out/pmc_sq.log | cut -c1-200
for f in pmc_mfma_busy pmc_sq_waits pmc_traffic_bench; do cp gpurun_out/${R}_$f.json profiles/${R}_$f.json; done
for i in 1 2 3; do timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_${R}_$i.json; cut -c100-230 gpurun_out/bench_${R}_$i.json; done
STEP_MARK=kb_vox_insert VOX_PER_STEP=1 TAG=$R STEPS=5 TOP=70 bash scripts/prof_stats.sh > gpurun_out/prof_$R.log 2>&1; head -9 gpurun_out/prof_${R}_last_step.txt
export RSLO_BENCH_FIXED_PLAN=1 BENCH_ARGS="--no-voxelize"
STEP_MARK=k_conv2d_wsplit_many VOX_PER_STEP=1 TOP=0 HEAD=400 TAG=${R}fix STEPS=4 bash scripts/prof_stats.sh > gpurun_out/prof_${R}fix.log 2>&1; head -3 gpurun_out/prof_${R}fix_last_step.txt
unset RSLO_BENCH_FIXED_PLAN BENCH_ARGS
# per-queue timeline of the same trace, the default invocation (with the CPU baseline), C2 / C5 encoder passes, the step
# through RCCL at N = 1, and the torch-native launches left in the step
python scripts/timeline_gaps.py $(find /tmp/prof_$R -name "*kernel_trace.csv" | head -1) > gpurun_out/${R}_timeline_gaps.txt 2>&1
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_${R}_with_cpu.json; cut -c100-230 gpurun_out/bench_${R}_with_cpu.json
for c in c2 c4 c5; do timeout 600 python bench.py --config $c 2>/dev/null | tail -1 > gpurun_out/bench_${R}_$c.json; cut -c1-200 gpurun_out/bench_${R}_$c.json; done
# the driver's own invocation (C3 headline + multirank child + host pricing + C2 / C4 / C5 under config.other_configs)
timeout 1500 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/bench_${R}_driver_invocation.json; cut -c100-230 gpurun_out/bench_${R}_driver_invocation.json
# the two-rank step on ONE GPU (functional mode): peer SyncBN exchange + overlapped gradient exchange over gloo, replicas compared
RSLO_PEER_TIMEOUT_MS=20000 RSLO_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${R}_two_rank_one_gpu.log 2>&1; tail -1 gpurun_out/${R}_two_rank_one_gpu.log | cut -c100-230
RSLO_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${R}_nccl_n1.log 2>&1; tail -1 gpurun_out/${R}_nccl_n1.log | cut -c100-230
timeout 200 python scripts/torch_launch_sites.py > gpurun_out/${R}_torch_launch_sites.txt 2>&1; grep "torch-native" gpurun_out/${R}_torch_launch_sites.txt
