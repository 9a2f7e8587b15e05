#!/bin/bash
# Per-kernel register / occupancy report of one HIP source (hipcc remarks):  tools/kernel_resources.sh <file.hip> [grep-filter]
hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -o /dev/null "$1" -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|  VGPRs:|AGPRs:|ScratchSize|Occupancy" \
 | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - | c++filt | grep -E "${2:-.}"
