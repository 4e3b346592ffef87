#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04d; : > gpurun_out/r04d/cfg.jsonl
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "astrobee or manifold or golden" > gpurun_out/r04d/tests.log 2>&1; tail -4 gpurun_out/r04d/tests.log
python -m pytest tests/test_gpu_trajopt.py -x -q -m gpu > gpurun_out/r04d/trajopt.log 2>&1; tail -6 gpurun_out/r04d/trajopt.log
for cfg in "2 0" "4 0" "5 0" "4 1024" "5 256" "3 0"; do
  set -- $cfg
  extra=""; [ "$2" != "0" ] && extra="--batch $2"
  timeout 400 python bench.py --config $1 $extra --steps 8 --warmup 2 --no-extras --no-cpu-baseline >> gpurun_out/r04d/cfg.jsonl 2>> gpurun_out/r04d/err.log
done
python - <<'PY'
import json
for l in open('gpurun_out/r04d/cfg.jsonl'):
    d=json.loads(l); print(d['config']['baseline_config'], d['config']['batch_total'], round(d['ms_per_step'],2), 'ms', round(d['value']), d['converged'], d['roofline']['kkt_solves_per_launch'])
PY
