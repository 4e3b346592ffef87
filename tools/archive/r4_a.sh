#!/bin/bash
# round 4, first GPU pass: new tests, the whole GPU suite, config-2 bench, TrajOpt bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04a
python -m pytest tests/test_gpu_env_batch.py -x -q -m gpu > gpurun_out/r04a/env.log 2>&1; tail -5 gpurun_out/r04a/env.log
python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r04a/suite.log 2>&1; tail -15 gpurun_out/r04a/suite.log
timeout 600 python bench.py --steps 40 > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err; cat gpurun_out/r04a/bench.json | cut -c1-600
timeout 600 python bench.py --algo trajopt --steps 10 > gpurun_out/r04a/bench_trajopt.json 2>> gpurun_out/r04a/bench.err; cat gpurun_out/r04a/bench_trajopt.json | cut -c1-600
tail -5 gpurun_out/r04a/bench.err
