#!/bin/bash
# Runs on the GPU box (through gpurun): refreshes everything profiles/ holds for one round.
#   tools/profile_round.sh <tag>      outputs under gpurun_out/<tag>/
# kernel-trace/stats and every --pmc pass are separate rocprofv3 runs (never combined with other trace domains).
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --overlap 1 --no-cpu-baseline > $OUT/bench_serial.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc -o $c -- python tools/pmc_probe.py > $OUT/pmc_$c.log 2>&1
done
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sq -o "$name" -- python tools/pmc_probe.py > "$OUT/sq_$name.log" 2>&1
done
ls -R $OUT | head -80
