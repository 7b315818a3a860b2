#!/bin/bash
# development aid: the tests added this round, then the default bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --durations=8 -k "digit_plan or plans_covered or arena or high_chunk or default_path or every_source" > gpurun_out/new_tests.log 2>&1
rc=$?
tail -25 gpurun_out/new_tests.log
exit $rc
