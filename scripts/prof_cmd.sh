# rocprofv3 kernel stats of an arbitrary command: TAG=name bash scripts/prof_cmd.sh <command...>
TAG=${TAG:-cmd}
OUT=/tmp/prof_$TAG
rm -rf $OUT; mkdir -p $OUT $GRAFT_REPO_ROOT/gpurun_out
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 ${LIMIT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- "$@" > $OUT/cmd.log 2>&1
echo "rocprof rc=$?"
cd $ROOT
tail -${TAILN:-3} $OUT/cmd.log
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${TAG}_kernel_stats.csv
head -${TOPN:-12} "$f" | cut -c1-200
