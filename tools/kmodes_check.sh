#!/bin/bash
# development aid: the k-modes tests, then the operator at config 5's shape with and without the fast leg
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kmodes" > gpurun_out/kmodes_tests.log 2>&1
rc=$?
tail -4 gpurun_out/kmodes_tests.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" gpurun_out/kmodes_tests.log | tail -20; exit $rc; }
TM_PP_DEBUG=1 timeout -k 10 300 python tools/kmodes_probe.py 2>&1 | grep "kmodes" | tail -12
TM_KMODES_BINWISE=1 timeout -k 10 300 python tools/kmodes_probe.py 2>&1 | grep "kmodes n=" | tail -3
