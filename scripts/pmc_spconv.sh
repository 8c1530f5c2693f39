# PMC counters for the sparse-conv micro-benchmark (separate passes; kernel-trace only, as the pool requires).
set -x
V=${V:-42}
OUT=/tmp/pmc_$V; rm -rf $OUT; mkdir -p $OUT gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "Name:\s+[A-Za-z0-9_]+" | awk '{print $2}' | sort -u > $OUT/counters.txt
wc -l $OUT/counters.txt
grep -E "^(SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_WAIT_ANY|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|SQ_VALU_MFMA_BUSY_CYCLES|SQ_INSTS_VALU_MFMA|SQ_INST_CYCLES_VMEM|SQ_WAIT_INST_LDS|TCC_HIT_sum|TCC_MISS_sum|TCP_TCC_READ_REQ_sum|TCP_TOTAL_CACHE_ACCESSES_sum|TCP_TCC_READ|TCC_EA0_RDREQ_sum|TCC_REQ_sum|TCP_TA_DATA_STALL|TA_BUSY|GRBM_GUI_ACTIVE|SQ_INSTS_VMEM|SQ_WAVES|FETCH_SIZE|MfmaUtil|SQ_LDS_BANK_CONFLICT|TCP_PENDING_STALL_CYCLES_sum|TA_TA_BUSY_sum|TCP_TCP_TA_DATA_STALL_CYCLES_sum)" $OUT/counters.txt | tr '\n' ' '
run() {  # $1 tag, rest counters
  tag=$1; shift
  RSLO_TUNING=spconv_v=$V ONLY="${ONLYK:-subm2 64->64 fwd}" REPS=3 timeout -k 5 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$tag -o p -- python $GRAFT_REPO_ROOT/scripts/bench_spconv.py > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d in csv.DictReader(open(sys.argv[1])):
    k = d["Kernel_Name"][:60]
    if "spconv" not in k: continue
    agg[k][d["Counter_Name"]] += float(d["Counter_Value"]); 
    if d["Counter_Name"] == list(agg[k].keys())[0]: cnt[k] += 1
for k, v in agg.items():
    print(k, "dispatches", cnt[k], {a: round(b / max(cnt[k], 1)) for a, b in v.items()})
PY
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
run b TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
run c SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU
run d TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
cp $OUT/*.log gpurun_out/ 2>/dev/null; tail -3 $OUT/a.log
