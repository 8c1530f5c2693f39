for a in 0 1 2 3 4 7 8 15; do echo "ablate=$a"; RSLO_TUNING="conv2d_fwd_wl=1,conv2d_ablate=$a" N=50 python scripts/bench_conv2d_one.py 2>&1 | tail -1; done
