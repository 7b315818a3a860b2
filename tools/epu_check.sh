#!/bin/bash
# development aid: the k-nearest / extended-palette tests, then tools/epu_probe.sh
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_encoder.py -x -q -m gpu -k "topk or epu or extended or palette_usage or arena" > gpurun_out/epu_tests.log 2>&1
rc=$?
tail -3 gpurun_out/epu_tests.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" gpurun_out/epu_tests.log | tail -20; exit $rc; }
bash tools/epu_probe.sh
