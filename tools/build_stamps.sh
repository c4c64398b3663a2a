#!/bin/bash
# Diagnostic build of the HIP library with per-workgroup phase stamps (never used by the product path):
#   tools/build_stamps.sh  ->  build/stamps/libmindpose_hip.so   (use with MINDPOSE_HIP_LIB=...)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$ROOT/build/stamps"
cp "$ROOT"/mindpose_amd/csrc/*.hip "$ROOT"/mindpose_amd/csrc/*.h "$ROOT"/mindpose_amd/csrc/Makefile "$ROOT/build/stamps/"
make -C "$ROOT/build/stamps" -j8 EXTRA="-DMP_CONV_STAMPS=1 -DMP_BLOCK_STAMPS=1 -DMP_WS_STAMPS=1 $MP_STAMPS_EXTRA"
