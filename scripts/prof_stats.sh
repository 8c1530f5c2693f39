# rocprofv3 kernel stats of a short bench run.  Output goes to /tmp on the box; only the small
# CSV summaries are copied back into gpurun_out/ (the raw trace can be large).
set -x
STEPS=${STEPS:-3}
TAG=${TAG:-r1}
OUT=/tmp/prof_$TAG
rm -rf $OUT; mkdir -p $OUT gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 ${LIMIT:-420} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- \
   python $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench.log 2>&1
echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
tail -3 $OUT/bench.log
du -sh $OUT
find $OUT -name "*stats*.csv" -size -2M -exec cp {} gpurun_out/ \;
cp $OUT/bench.log gpurun_out/prof_${TAG}_bench.log
ls -la gpurun_out | tail
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
head -45 "$f" | cut -c1-220
t=$(find $OUT -name "*kernel_trace.csv" | head -1)
python scripts/trace_last_steps.py "$t" ${LAST:-1} ${VOX_PER_STEP:-8} gpurun_out/prof_${TAG}_last_step.txt | head -${HEAD:-90}
