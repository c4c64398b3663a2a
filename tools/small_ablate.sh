#!/bin/bash
# Ablation builds of the small-problem fp32 conv kernel (-DMP_SMALL_ABLATE=<mask>: 1 no staging, 2 no MFMA loop, 4 no weight loads,
# 8 no fold; results wrong, timings meaningful) -> build/small_ablate_<mask>/libmindpose_hip.so (use with MINDPOSE_HIP_LIB=...):
#   bash tools/small_ablate.sh "1 2 4 8 3"      then on the GPU box:  MINDPOSE_HIP_LIB=build/small_ablate_1/libmindpose_hip.so python tools/bench_small.py 32
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
C="$ROOT/mindpose_amd/csrc"
make -C "$C" -j8 > /dev/null
for m in $1; do
  d="$ROOT/build/small_ablate_$m"; mkdir -p "$d"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMP_SMALL_ABLATE=$m -c "$C/conv_small_f32.hip" -o "$d/conv_small_f32.o"
  objs=$(ls "$C"/*.o | grep -v conv_small_f32.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$d/libmindpose_hip.so" $objs "$d/conv_small_f32.o" -ldl
  echo "built $d"
done
