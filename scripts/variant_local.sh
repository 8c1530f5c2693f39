# Builds variants of ONE source file HERE (cross-compile; the variant libraries travel to the GPU box with the snapshot,
# rslo_amd/_variants/ is git-ignored through *.so) so that the box only runs them:
#   SRC=spconv.hip VARIANTS="name:-DX=1,-DY=2 ..." bash scripts/variant_local.sh
#   on the box: RSLO_HIP_LIB=rslo_amd/_variants/lib<name>.so python scripts/bench_spconv.py
set -e
OUT=rslo_amd/_variants; mkdir -p $OUT
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-slp-vectorize -Wno-unused-result"
base=${SRC%.hip}
for v in $VARIANTS; do
  name=${v%%:*}; defs=$(echo ${v#*:} | tr ',' ' ')
  ( /opt/rocm/bin/hipcc $FL $defs -c rslo_amd/csrc/$SRC -o $OUT/$base.$name.v.o
    objs=$(ls rslo_amd/_obj/*.hip.o | grep -v "/$SRC.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib$name.so $objs $OUT/$base.$name.v.o
    rm -f $OUT/$base.$name.v.o; echo "built $OUT/lib$name.so ($defs)" ) &
done
wait
