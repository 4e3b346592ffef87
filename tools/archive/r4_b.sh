#!/bin/bash
# round 4, after the warm-start change: whole GPU suite, then the four configs with the old fixed level and the new defaults
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04b
python -m pytest tests -q -m gpu --durations=5 > gpurun_out/r04b/suite.log 2>&1; tail -15 gpurun_out/r04b/suite.log
MUW=1e-4,auto python tools/ipm_opts_scan.py 0 4096 50
MUW=1e-4,auto python tools/ipm_opts_scan.py 1 65536 30
MUW=1e-4,auto python tools/ipm_opts_scan.py 2 8192 50
MUW=1e-4,auto python tools/ipm_opts_scan.py 3 2048 50
