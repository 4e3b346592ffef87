#!/bin/bash
# A/B builds: tools/build_variant.sh NAME MODEL [flags...] -> gusto.jl_amd/variants/NAME.so (that model only, like build_dev.sh).
# tools/ab_variants.sh then times every variant on the GPU box.
set -e
cd "$(dirname "$0")/.."
NAME=$1; M=$2; shift 2
D=gusto.jl_amd; V=$D/variants; W=/tmp/gusto_var_$NAME
mkdir -p $V $W
F="--offload-arch=gfx950 -O3 -std=c++17 -Iinclude -I$D/csrc -fPIC -Wno-unused-value -Wno-pass-failed $@"
cat > $W/stub.hip <<EOS
#include "handle.hpp"
#define STUB(i) int gusto_launch_init_m##i(gusto_handle h, bool) { h->err = "model not in this dev build"; return GUSTO_ERR_ARG; } \
                int gusto_launch_scp_m##i(gusto_handle h, int, int, int) { h->err = "model not in this dev build"; return GUSTO_ERR_ARG; }
#define STUBT(i) int gusto_launch_init_m##i(gusto_handle h, bool) { h->err = "model not in this dev build"; return GUSTO_ERR_ARG; } \
                 int gusto_launch_trajopt_m##i(gusto_handle h, int, int) { h->err = "model not in this dev build"; return GUSTO_ERR_ARG; }
EOS
for i in 0 1 2 3; do [ $i != $M ] && echo "STUB($i)" >> $W/stub.hip; done
for i in 4 5 6; do [ $i != $M ] && echo "STUBT($i)" >> $W/stub.hip; done
/opt/rocm/bin/hipcc $F -c $D/csrc/gusto_hip.hip -o $W/gusto_hip.o &
/opt/rocm/bin/hipcc $F -c $W/stub.hip -o $W/stub.o &
/opt/rocm/bin/hipcc $F -c $D/csrc/shoot.hip -o $W/shoot.o &
/opt/rocm/bin/hipcc $F -c $D/csrc/model_$M.hip -o $W/model_$M.o -Rpass-analysis=kernel-resource-usage > $W/model.log 2>&1 || { grep -B2 -A6 "error" $W/model.log | head -40; echo "FAILED"; wait; exit 1; }
grep -A9 "Function Name: .*scp_kernel" $W/model.log | grep -E "Name|VGPRs:|AGPRs|Scratch|Occupancy" | head -12 || true
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $W/gusto_hip.o $W/shoot.o $W/stub.o $W/model_$M.o -o $V/$NAME.so
echo "built $V/$NAME.so (model $M, flags: $@)"
