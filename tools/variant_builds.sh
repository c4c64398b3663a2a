#!/bin/bash
# Variant / ablation builds of ONE kernel source: the library relinked with that object compiled under extra -D flags (diagnostic
# builds - results may be wrong, timings meaningful; never the product library):
#   bash tools/variant_builds.sh <source stem> "<tag:flag[,flag...]> ..."   ->  build/<stem>_<tag>/libmindpose_hip.so  (MINDPOSE_HIP_LIB=...)
# Macros: conv_small_f32  -DMP_SMALL_ABLATE=<mask>  1 no staging, 2 no MFMA loop, 4 no weight loads, 8 no fold        (tools/bench_small.py)
#         conv_gemm_f32   -DMP_GEMM_ABLATE=<mask>   1 no stores, 2 no MFMA                                            (tools/probes/gemm_expand_probe.py)
#         pwchain_f32     -DPWC_ABLATE=<mask>       1 no y stores, 2 no MFMA, 4 no residual loads                     (tools/bench_pwchain32.py)
# e.g.  bash tools/variant_builds.sh pwchain_f32 "s:-DPWC_ABLATE=1 m:-DPWC_ABLATE=2 sr:-DPWC_ABLATE=5"
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
C="$ROOT/mindpose_amd/csrc"
stem=$1
make -C "$C" -j8 > /dev/null
for spec in $2; do
  tag=${spec%%:*}; flags=$(echo ${spec#*:} | tr ',' ' ')
  d="$ROOT/build/${stem}_$tag"; mkdir -p "$d"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c "$C/$stem.hip" -o "$d/$stem.o"
  objs=$(ls "$C"/*.o | grep -v "/$stem.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$d/libmindpose_hip.so" $objs "$d/$stem.o" -ldl
  echo "built $d ($flags)"
done
