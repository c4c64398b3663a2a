#!/bin/bash
# Ablation builds of the fp32 blocked-GEMM 1x1 kernel (-DMP_GEMM_ABLATE=<mask>: 1 no stores, 2 no MFMA; results
# wrong, timings meaningful) -> build/gemm_ablate_<mask>/libmindpose_hip.so (use with MINDPOSE_HIP_LIB=...):
#   bash tools/gemm_ablate.sh "1 2 3"     then on the GPU box:  MINDPOSE_HIP_LIB=build/gemm_ablate_1/libmindpose_hip.so python tools/probes/gemm_expand_probe.py
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
C="$ROOT/mindpose_amd/csrc"
make -C "$C" -j8 > /dev/null
for m in $1; do
  d="$ROOT/build/gemm_ablate_$m"; mkdir -p "$d"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMP_GEMM_ABLATE=$m -c "$C/conv_gemm_f32.hip" -o "$d/conv_gemm_f32.o"
  objs=$(ls "$C"/*.o | grep -v conv_gemm_f32.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$d/libmindpose_hip.so" $objs "$d/conv_gemm_f32.o" -ldl
  echo "built $d"
done
