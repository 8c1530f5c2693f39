# k_spconv_v6 with / without skipping dead 16-row blocks (switch spconv_skip): the 32/64-channel layers of the encoder
for s in 0 1; do echo "== spconv_skip=$s"; RSLO_TUNING="spconv_skip=$s" ONLY="subm1 32->32 fwd,subm2 64->64 fwd,subm2 64->64 dgradT,subm3 64->64 fwd,conv1,conv2,inv1,inv2" REPS=30 python scripts/bench_spconv.py 2>&1 | grep "us "; done
