# The measurement pass behind profiles/<round>_*: three bench runs, the bf16 line, rocprof kernel stats + one-step
# breakdowns (whole step and training stream alone), the three PMC passes (stamped with the library hash).
R=${ROUND:-r02}
for i in 1 2 3; do timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_${R}_$i.json; cut -c100-230 gpurun_out/bench_${R}_$i.json; done
timeout 200 python bench.py --dtype bf16 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_${R}_bf16.json; cut -c100-230 gpurun_out/bench_${R}_bf16.json
TAG=$R STEPS=5 TOP=70 bash scripts/prof_stats.sh > gpurun_out/prof_$R.log 2>&1; head -9 gpurun_out/prof_${R}_last_step.txt
ROUND=$R bash scripts/pmc_bench_traffic.sh > gpurun_out/pmc_traffic.log 2>&1; grep "v6<64, 64, 2" gpurun_out/pmc_traffic.log
ROUND=$R bash scripts/pmc_mfma_busy.sh > gpurun_out/pmc_busy.log 2>&1; grep "v6<64, 64, 2\|conv2d_fwd\|conv2d_str" gpurun_out/pmc_busy.log
ROUND=$R bash scripts/pmc_sq_waits.sh > gpurun_out/pmc_sq.log 2>&1; grep "v6<64, 64, 2" gpurun_out/pmc_sq.log | cut -c1-200
export RSLO_BENCH_FIXED_PLAN=1 BENCH_ARGS="--no-voxelize"
STEP_MARK=k_conv2d_wsplit_many VOX_PER_STEP=1 TOP=0 HEAD=400 TAG=${R}fix STEPS=4 bash scripts/prof_stats.sh > gpurun_out/prof_${R}fix.log 2>&1; head -3 gpurun_out/prof_${R}fix_last_step.txt
