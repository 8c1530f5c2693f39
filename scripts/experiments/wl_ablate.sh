# Recorded in profiles/r06_conv2d_wl_ablations.txt with the ablation instantiations k_conv2d_wl<4, ABL> compiled in and a switch
# (conv2d_ablate) selecting them; both were taken out again once the numbers were in (round 6).  To repeat: re-add the ABL cases to
# conv2d_wl_launch and the switch, then:
#   for a in 0 1 2 3 4 7 8 15; do RSLO_TUNING="conv2d_fwd_wl=1,conv2d_ablate=$a" N=50 python scripts/bench_conv2d_one.py | tail -1; done
echo "see the header of this file"
