# Times variants of ONE source file selected by compile-time defines (library rebuilt on the box):
#   SRC=chamfer_grid.hip VARIANTS="name:-DX=1,-DY=2 ..." CMD="python scripts/bench_chamfer_pair.py" bash scripts/variant_any.sh
set -e
OUT=/tmp/variant; mkdir -p $OUT
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-slp-vectorize -Wno-unused-result"
for f in rslo_amd/csrc/*.hip; do
  [ "$(basename $f)" = "$SRC" ] && continue
  /opt/rocm/bin/hipcc $FL -c $f -o $OUT/$(basename $f).o &
done
wait
base=${SRC%.hip}
for v in $VARIANTS; do
  name=${v%%:*}; defs=$(echo ${v#*:} | tr ',' ' ')
  /opt/rocm/bin/hipcc $FL $defs -c rslo_amd/csrc/$SRC -o $OUT/$base.$name.v.o
  objs=$(ls $OUT/*.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib$name.so $objs $OUT/$base.$name.v.o
  echo "== $name ($defs)"
  RSLO_HIP_LIB=$OUT/lib$name.so $CMD 2>&1 | grep -E "${GREP:-.}" | cut -c1-${CUT:-160}
done
