#!/bin/bash
# Runs on the GPU box (through gpurun): refreshes everything profiles/ holds for one round.
#   tools/profile_round.sh <tag>      outputs under gpurun_out/<tag>/
# kernel-trace/stats and every --pmc pass are separate rocprofv3 runs (never combined with other trace domains).
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python bench.py --overlap 2 --steps 60 --no-cpu-baseline > $OUT/bench_overlap2.json 2>> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python bench.py --steps 20 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
: > $OUT/configs.jsonl
for c in 2 3 4 5; do   # every BASELINE config through the contract harness itself
  timeout 900 python bench.py --config $c --steps 8 --warmup 2 --no-extras >> $OUT/configs.jsonl 2>> $OUT/bench.err
done
for c in 4 5; do   # kernel stats of BASELINE configs 4 and 5 (astrobeeSE3 B=8192, manifold B=2048)
  m=$((c - 2))
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_m$m -o stats -- python bench.py --config $c --steps 4 --warmup 1 --no-extras --no-cpu-baseline > $OUT/stats_m$m.log 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc -o $c -- python tools/pmc_probe.py > $OUT/pmc_$c.log 2>&1
done
# HBM traffic of the other BASELINE configs and of TrajOpt: two --pmc passes each, summarised by tools/pmc_config.py
for w in "3 gusto" "4 gusto" "5 gusto" "2 trajopt"; do
  set -- $w
  D=$OUT/pmc_c$1_$2; mkdir -p $D
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $D -o $c -- python tools/pmc_probe.py $1 $2 > $D/$c.log 2>&1
    f=$(find $D -name "${c}_counter_collection.csv" | head -1); [ -n "$f" ] && [ "$f" != "$D/${c}_counter_collection.csv" ] && cp "$f" $D/${c}_counter_collection.csv
  done
  name=$([ $2 = trajopt ] && echo trajopt_config$1 || echo config$1)
  python tools/pmc_config.py $D $OUT/pmc_$name.json > $D/summary.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -o stats -- python bench.py --config 3 --steps 4 --warmup 1 --no-extras --no-cpu-baseline > $OUT/stats_c3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_trajopt -o stats -- python bench.py --algo trajopt --steps 6 --warmup 1 --no-extras --no-cpu-baseline > $OUT/stats_trajopt.log 2>&1
timeout 600 python bench.py --algo trajopt > $OUT/bench_trajopt.json 2>> $OUT/bench.err
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA" \
           "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE" \
           "VmemLatency" "LdsLatency" "InstrFetchLatency" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_TC_INST_REQ" "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_IFETCH"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sq -o "$name" -- python tools/pmc_probe.py > "$OUT/sq_$name.log" 2>&1
done
for m in 2 3; do   # matrix-core counters of the 12/13-state kernels (v_mfma_f64_16x16x4_f64 in the factor sweep)
  timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/mfma_m$m -o c -- python bench.py --config $((m + 2)) --steps 1 --warmup 0 --no-extras --no-cpu-baseline > $OUT/mfma_m$m.log 2>&1
done
# the wave-per-chain kernels (csrc/segw.hpp): batch-size sweep of the decompositions, kernel statistics of the one-GPU shards of configs 4 / 5
timeout 900 python tools/chains_sweep.py $OUT/chains_sweep.jsonl > $OUT/chains_sweep.log 2>&1
for w in "2 1024" "3 256"; do
  set -- $w
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_shard_m$1 -o stats -- python tools/gpu_time.py $1 $2 50 > $OUT/stats_shard_m$1.log 2>&1
done
ls -R $OUT | head -80
