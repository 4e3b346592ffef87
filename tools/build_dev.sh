#!/bin/bash
# development build: one model only (default freeflyer), optional extra flags, e.g. tools/build_dev.sh 0 -DGUSTO_PROFILE
set -e
cd "$(dirname "$0")/.."
M=${1:-0}; shift || true
D=gusto.jl_amd
F="--offload-arch=gfx950 ${GUSTO_OPT:--O3} -std=c++17 -Iinclude -fPIC -Wno-unused-value -Wno-pass-failed $@"
mkdir -p $D/build
cat > $D/build/stub.hip <<EOS
#include "../csrc/handle.hpp"
#define STUB(i) int gusto_launch_init_m##i(gusto_handle h, bool) { h->err = "model not in this dev build"; return GUSTO_ERR_ARG; } \
                int gusto_launch_scp_m##i(gusto_handle h, int, int, int) { h->err = "model not in this dev build"; return GUSTO_ERR_ARG; }
#define STUBT(i) int gusto_launch_init_m##i(gusto_handle h, bool) { h->err = "model not in this dev build"; return GUSTO_ERR_ARG; } \
                 int gusto_launch_trajopt_m##i(gusto_handle h, int, int) { h->err = "model not in this dev build"; return GUSTO_ERR_ARG; }
EOS
for i in 0 1 2 3; do [ $i != $M ] && echo "STUB($i)" >> $D/build/stub.hip; done
for i in 4 5 6; do [ $i != $M ] && echo "STUBT($i)" >> $D/build/stub.hip; done
/opt/rocm/bin/hipcc $F -c $D/csrc/gusto_hip.hip -o $D/build/gusto_hip.o &
/opt/rocm/bin/hipcc $F -c $D/build/stub.hip -o $D/build/stub.o &
/opt/rocm/bin/hipcc $F -c $D/csrc/shoot.hip -o $D/build/shoot.o &
/opt/rocm/bin/hipcc $F -c $D/csrc/model_$M.hip -o $D/build/model_$M.o -Rpass-analysis=kernel-resource-usage > $D/build/model_$M.log 2>&1 || { grep -B2 -A6 "error" $D/build/model_$M.log | head -40; echo "model_$M.hip FAILED"; wait; exit 1; }
grep -A9 "scp_kernel\|trajopt_kernel" $D/build/model_$M.log | grep -E "VGPRs:|Scratch|Occupancy" || true
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $D/build/gusto_hip.o $D/build/shoot.o $D/build/stub.o $D/build/model_$M.o -o $D/libgusto_hip.so
echo built dev lib for model $M
