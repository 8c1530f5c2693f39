# forward-accuracy A/B of the implementation switches against the float64 arbiter (scripts/parity_report.py)
run() { echo "== $1"; env $1 python scripts/parity_report.py --bs 1 --rings 16 2>&1 | grep -E "^(translation_preds|rotation_preds|C_loss|ratio|gpu-f64)"; }
run "RSLO_DUMMY=1"
run "RSLO_SPCONV_SPLIT=0"
run "RSLO_CONV2D_PASSES="
run "RSLO_FUSED_BN=0"
run "RSLO_SPCONV_SPLIT=0 RSLO_CONV2D_PASSES= RSLO_FUSED_BN=0"
