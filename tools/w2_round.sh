#!/bin/bash
# on the GPU box: what profiles/r06_wave_per_chain.txt is made of -- the batch-size sweep of one / two / four waves per problem (dev
# variants, GUSTO_DEV_W2), the phase profiles, and the rocprofv3 kernel statistics of the two shard sizes with the shipped library
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/w2_round; mkdir -p $O
bash tools/w2_ab.sh "128 256 512 1024 2048 4096" "128 256 512 1024 2048 4096 8192" "0 2 4" > $O/sweep.log 2>&1
bash tools/w2_prof.sh 3 256 "0 2 4" > $O/prof_m3_256.log 2>&1
bash tools/w2_prof.sh 2 1024 "0 2 4" > $O/prof_m2_1024.log 2>&1
for w in "2 1024" "3 256" "3 2048"; do
  set -- $w
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_m$1_$2 -o stats -- python tools/gpu_time.py $1 $2 50 > $O/stats_m$1_$2.log 2>&1
  f=$(find $O/stats_m$1_$2 -name "stats_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_m$1_$2.csv
done
ls $O
