# Recorded in profiles/r06_spconv_skip.txt while k_spconv_v6 existed with and without the dead-block skip (switch spconv_skip);
# the form without it is deleted (round 6, bits in tests/golden/kernel_bits.json).  The command was:
#   for s in 0 1; do RSLO_TUNING="spconv_skip=$s" ONLY=... python scripts/bench_spconv.py; done
# for s in 0 1; do echo "== spconv_skip=$s"; RSLO_TUNING="spconv_skip=$s" ONLY="subm1 32->32 fwd,subm2 64->64 fwd,subm2 64->64 dgradT,subm3 64->64 fwd,conv1,conv2,inv1,inv2" REPS=30 python scripts/bench_spconv.py 2>&1 | grep "us "; done
echo "see the header of this file"
