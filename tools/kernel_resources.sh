#!/bin/bash
# Registers, spills, scratch and occupancy of every kernel of librt_hip.so, as the compiler reports them.
cd "$(dirname "$0")/.." || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize \
  -fhip-fp32-correctly-rounded-divide-sqrt -Iinclude -Irt_amd/csrc "$@" -c rt_amd/csrc/rt_hip.hip -o /tmp/rt_hip_res.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep "remark:" | sed 's/.*remark: *//; s/ \[-Rpass.*//' |
  awk '/^Function Name/ { if (line) print line; cmd = "c++filt " $3; cmd | getline n; close(cmd);
                          gsub(/\(anonymous namespace\)::/, "", n); sub(/\(.*/, "", n); line = n; next }
       /^VGPRs:|Spill|ScratchSize|Occupancy/ { line = line " | " $0 }
       END { print line }'
