cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06j; mkdir -p $O
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 900 python tools/chains_sweep.py $O/chains_sweep.jsonl > $O/chains_sweep.log 2>&1
for w in "2 1024" "3 256"; do set -- $w
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_shard_m$1 -o stats -- python tools/gpu_time.py $1 $2 50 > $O/stats_shard_m$1.log 2>&1
done
: > $O/configs.jsonl
for c in 2 3 4 5; do timeout 900 python bench.py --config $c --steps 8 --warmup 2 --no-extras >> $O/configs.jsonl 2>> $O/bench.err; done
ls $O
