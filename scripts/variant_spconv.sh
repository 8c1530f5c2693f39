# Times kernel variants selected by compile-time defines: VARIANTS="name:-DX=1,-DY=2 ..." (library rebuilt on the box)
set -e
OUT=/tmp/variant; mkdir -p $OUT
for f in rslo_amd/csrc/*.hip; do
  [ "$(basename $f)" = "spconv.hip" ] && continue
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-slp-vectorize -Wno-unused-result -c $f -o $OUT/$(basename $f).o &
done
wait
for v in $VARIANTS; do
  name=${v%%:*}; defs=$(echo ${v#*:} | tr ',' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-slp-vectorize -Wno-unused-result $defs -c rslo_amd/csrc/spconv.hip -o $OUT/spconv.$name.o
  objs=$(ls $OUT/*.hip.o | grep -v spconv)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib$name.so $objs $OUT/spconv.$name.o
  for pipe in ${PIPES:-0}; do
    echo "== $name ($defs) PIPE=$pipe"
    RSLO_HIP_LIB=$OUT/lib$name.so RSLO_ROW_ORDER=${ROW_ORDER:-0} ONLY=${ONLY:-subm} python scripts/bench_spconv.py 2>&1 | grep -E "${GREP:-(32->32|64->64) fwd }" | cut -c1-110
  done
done
