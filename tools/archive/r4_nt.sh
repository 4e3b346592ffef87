#!/bin/bash
# waves per problem (GUSTO_DEV_WAVES) at the batch sizes one GPU of an 8-way strong-scaled config sees
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04b; : > gpurun_out/r04b/waves.jsonl
python -m pytest tests/test_gpu_env_batch.py tests/test_gpu_seam.py -x -q -m gpu > gpurun_out/r04b/tests.log 2>&1; tail -5 gpurun_out/r04b/tests.log
for cfg in "5 256" "4 1024" "2 512"; do
  set -- $cfg
  for w in 1 2 4; do
    echo "config $1 batch $2 waves $w" >> gpurun_out/r04b/waves.jsonl
    GUSTO_DEV_WAVES=$w timeout 300 python bench.py --config $1 --batch $2 --steps 6 --warmup 2 --no-extras --no-cpu-baseline >> gpurun_out/r04b/waves.jsonl 2>> gpurun_out/r04b/err.log
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r04b/waves.jsonl'):
    if l.startswith('config'): print(l.strip(), end=' -> ')
    else:
        d=json.loads(l); print(round(d['ms_per_step'],2), 'ms', d['converged'], 'conv', d['roofline']['kkt_solves_per_launch'])
PY
timeout 600 python bench.py --algo trajopt --steps 10 > gpurun_out/r04b/bench_trajopt.json 2>> gpurun_out/r04b/err.log; cut -c1-400 gpurun_out/r04b/bench_trajopt.json
tail -3 gpurun_out/r04b/err.log
