#!/bin/bash
# The CPU suite with the oracle built under AddressSanitizer + UndefinedBehaviorSanitizer.
set -e
cd "$(dirname "$0")/.."
make -C oracle -s asan
export GUSTO_ORACLE_LIB=$PWD/oracle/libgusto_oracle_asan.so
export LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
python -m pytest tests/test_oracle_scp.py tests/test_oracle_subproblem.py tests/test_oracle_models_slsqp.py tests/test_shooting.py -x -q -m "not gpu" "$@"
