#!/bin/bash
# Variant / ablation builds of the fp32 expand + reduce chain kernel (csrc/pwchain_f32.hip) -> build/pwc_<tag>/libmindpose_hip.so:
#   bash tools/pwchain32_variants.sh "a:-DPWC_RES_AHEAD=0 b:-DPWC_BATCH_B=0 c:-DPWC_ABLATE=1"
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
C="$ROOT/mindpose_amd/csrc"
make -C "$C" -j8 > /dev/null
for spec in $1; do
  tag=${spec%%:*}; flags=$(echo ${spec#*:} | tr ',' ' ')
  d="$ROOT/build/pwc_$tag"; mkdir -p "$d"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c "$C/pwchain_f32.hip" -o "$d/pwchain_f32.o"
  objs=$(ls "$C"/*.o | grep -v pwchain_f32.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$d/libmindpose_hip.so" $objs "$d/pwchain_f32.o" -ldl
  echo "built $d ($flags)"
done
