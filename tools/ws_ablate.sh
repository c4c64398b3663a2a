#!/bin/bash
# Ablation builds of the weight-stationary kernel (diagnostic; results are wrong, timings are the point):
#   bash tools/ws_ablate.sh "1 2 4 8 16 32 63"   ->  build/abl<mask>/libmindpose_hip.so  (needs build/stamps from tools/build_stamps.sh)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
for m in $1; do
  (
    rm -rf "$ROOT/build/abl$m"; cp -r "$ROOT/build/stamps" "$ROOT/build/abl$m"
    cp "$ROOT"/mindpose_amd/csrc/*.hip "$ROOT"/mindpose_amd/csrc/*.h "$ROOT/build/abl$m/"
    cd "$ROOT/build/abl$m"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DMP_WS_STAMPS=1 -DMP_WS_QUICK=1 -DMP_WS_ABLATE=$m -mllvm -pragma-unroll-threshold=131072 -c conv_f16_ws.hip -o conv_f16_ws.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmindpose_hip.so *.o -ldl
  ) &
done
wait
ls -la "$ROOT"/build/abl*/libmindpose_hip.so
