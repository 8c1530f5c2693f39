# What bounds k_spconv_v6?  Rebuilds the library on the GPU box with parts of the kernel compiled out
# (SPC_ABLATE bits: 1 no weight loads, 2 no row gathers, 4 no MFMAs, 8 no operand split) and times the 64->64 layer.
set -e
OUT=/tmp/ablate; mkdir -p $OUT
for a in ${ABLATES:-0 1 2 3 4 8 5 6 7 15}; do
  objs=""
  for f in rslo_amd/csrc/*.hip; do
    o=$OUT/$(basename $f).$a.o
    if [ "$(basename $f)" = "spconv.hip" ] || [ ! -f $OUT/$(basename $f).o ]; then
      if [ "$(basename $f)" = "spconv.hip" ]; then
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-slp-vectorize -Wno-unused-result -DSPC_ABLATE=$a -c $f -o $o
      else
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-slp-vectorize -Wno-unused-result -c $f -o $OUT/$(basename $f).o &
        o=$OUT/$(basename $f).o
      fi
    else
      o=$OUT/$(basename $f).o
    fi
    objs="$objs $o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib$a.so $objs
  echo "== SPC_ABLATE=$a"
  RSLO_HIP_LIB=$OUT/lib$a.so RSLO_SPCONV_PIPE=0 RSLO_ROW_ORDER=${ROW_ORDER:-0} ONLY=${ONLY:-subm2} python scripts/bench_spconv.py 2>&1 | grep -E "fwd " | cut -c1-110
done
