# usage: SCRIPT=scripts/bench_spconv.py KERN=k_spconv_v6 bash scripts/pmc_tcp.sh  -- vector-L1 (TCP) request counters per launch
OUT=/tmp/pmc_tcp; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  tag=$1; shift
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$tag -o p -- python $GRAFT_REPO_ROOT/$SCRIPT > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$KERN" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for d in csv.DictReader(open(sys.argv[1])):
    k = d["Kernel_Name"][:40]
    if sys.argv[2] not in k: continue
    agg[k][d["Counter_Name"]] += float(d["Counter_Value"]); cnt[k][d["Counter_Name"]] += 1
for k, v in agg.items():
    print(k, {a: round(b / max(cnt[k][a], 1)) for a, b in v.items()})
PY
}
run a TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_DATA_STALL_CYCLES_sum SQ_INSTS_VMEM_RD
run b TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
