#!/bin/bash
# Build libcft_hip.so for gfx950; exits non-zero (and removes a stale library) on any compile error.
set -e
cd "$(dirname "$0")/.."
OUT=${CFT_OUT:-multispectral-object-detection_amd/libcft_hip.so}
rm -f "$OUT"
# product sources: csrc/*.hip; the probe build (tools/build_probes.sh sets CFT_EXTRA) adds csrc/probes/*.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o "$OUT" multispectral-object-detection_amd/csrc/*.hip $CFT_EXTRA "$@"
ls -la "$OUT"
