# usage: pmc_any.sh <script.py> <kernel-substring>   -- PMC counters (separate passes, kernel-trace only)
OUT=/tmp/pmc_any; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  tag=$1; shift
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$tag -o p -- python $GRAFT_REPO_ROOT/$SCRIPT > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$KERN" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for d in csv.DictReader(open(sys.argv[1])):
    k = d["Kernel_Name"][:50]
    if sys.argv[2] not in k: continue
    agg[k][d["Counter_Name"]] += float(d["Counter_Value"]); cnt[k][d["Counter_Name"]] += 1
for k, v in agg.items():
    print(k, {a: round(b / max(cnt[k][a], 1)) for a, b in v.items()})
PY
}
SCRIPT=$1; KERN=$2
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE
run b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM
