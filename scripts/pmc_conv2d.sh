# usage: [SHAPE=..] [CFG=..] pmc_conv2d.sh  -- PMC passes (kernel-trace + --pmc only) over scripts/bench_conv2d_one.py for
# the forward convolution kernel: issue / wait split, MFMA busy, LDS and vector-memory pipe activity, L1 stalls
OUT=/tmp/pmc_c2; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -n "$CFG" ] && export RSLO_TUNING="conv2d_fwd_tr=${CFG%%,*},conv2d_fwd_mtw=$(echo $CFG | cut -d, -f2)"
[ -n "$TUNING" ] && export RSLO_TUNING="$TUNING"      # explicit switch list (scripts/_tuning.py), e.g. conv2d_ablate=1
run() {
  tag=$1; shift
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$tag -o p -- python $GRAFT_REPO_ROOT/scripts/bench_conv2d_one.py > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for d in csv.DictReader(open(sys.argv[1])):
    if not any(k in d["Kernel_Name"] for k in ("k_conv2d_fwd", "k_conv2d_g2", "k_conv2d_wl")): continue
    agg[d["Counter_Name"]] += float(d["Counter_Value"]); cnt[d["Counter_Name"]] += 1
print({a: round(b / max(cnt[a], 1)) for a, b in agg.items()})
PY
}
echo "== SHAPE=$SHAPE CFG=$CFG"
if [ "$PASSES" = "ab" ]; then
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES
run b SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU
run i SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA
exit 0
fi
if [ "$PASSES" = "l1" ]; then
run f TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_TOTAL_ACCESSES_sum
run g TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run h TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
exit 0
fi
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES
run b SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU
run c SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE
run d TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
run e TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_READ_sum
