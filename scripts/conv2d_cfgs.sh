# usage: conv2d_cfgs.sh "cin,cout,H,W" cfg1 cfg2 ...   -- kernel durations (rocprofv3 kernel trace) of rslo_conv2d_fwd per tile
# configuration ("TR,MTW" -> RSLO_TUNING=conv2d_fwd_tr=TR,conv2d_fwd_mtw=MTW; "kc2" = conv2d_fwd_kc=2; "def" = library default)
cd /tmp && export TMPDIR=/tmp
S=$1; shift
for cfg in "$@"; do
  rm -rf /tmp/c2; E=""
  case $cfg in def) ;; kc2) E="RSLO_TUNING=conv2d_fwd_kc=2";; *) E="RSLO_TUNING=conv2d_fwd_tr=${cfg%%,*},conv2d_fwd_mtw=$(echo $cfg | cut -d, -f2)";; esac
  env SHAPE=$S $E $EXTRA rocprofv3 --kernel-trace --output-format csv -d /tmp/c2 -o p -- python $GRAFT_REPO_ROOT/scripts/bench_conv2d_one.py > /tmp/c2.log 2>&1
  python3 - "$(find /tmp/c2 -name '*kernel_trace.csv' | head -1)" "$S" "$cfg" <<'PY'
import csv, sys, statistics
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(sys.argv[1])) if "k_conv2d_fwd" in r["Kernel_Name"] or "k_conv2d_g2" in r["Kernel_Name"]]
print("%-16s cfg %-6s  median %.1f us  min %.1f  (n=%d)" % (sys.argv[2], sys.argv[3], statistics.median(d), min(d), len(d)))
PY
  grep -h "rel err" /tmp/c2.log
done
