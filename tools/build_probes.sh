#!/bin/bash
# Probe build of libcft_hip.so: the timing-probe / A-B variants (ABLATE / ABL template arguments, variants 1xx..97xx of
# cft_set_conv_variant) are compiled only here - the product library (tools/build.sh, __graft_entry__.build) has none.
# Written next to the product library as libcft_hip_probes.so (tools/gemm_bench.py --lib ...), never in its place.
CFT_EXTRA="multispectral-object-detection_amd/csrc/probes/*.hip" CFT_OUT=${CFT_OUT:-multispectral-object-detection_amd/libcft_hip_probes.so} exec "$(dirname "$0")/build.sh" -DCFT_PROBES "$@"
