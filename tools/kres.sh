#!/bin/bash
# kernel resource usage of one model's translation unit: tools/kres.sh <model> [-D...]   (registers, spills, scratch, LDS, occupancy)
cd "$(dirname "$0")/.."
m=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -fPIC -Wno-unused-value -Wno-pass-failed "$@" \
  -Rpass-analysis=kernel-resource-usage -c gusto.jl_amd/csrc/model_$m.hip -o /tmp/kres_$m.o 2>&1 | \
  grep -E "Function Name|VGPRs:|AGPRs|Spill|ScratchSize|SGPRs:|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//' | paste - - - - - - - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | cut -c1-300
