#!/bin/bash
# Build libcft_hip.so for gfx950; exits non-zero (and removes a stale library) on any compile error.
set -e
cd "$(dirname "$0")/.."
OUT=${CFT_OUT:-multispectral-object-detection_amd/libcft_hip.so}
rm -f "$OUT"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o "$OUT" multispectral-object-detection_amd/csrc/*.hip "$@"
ls -la "$OUT"
