#!/bin/bash
# development aid: the multi-process / sharded tests
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1200 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_rehearsal.py tests/test_gpu_native_comm.py tests/test_gpu_rccl.py -x -q -m gpu -k "sharded or ranks or native or rccl or rehears" --durations=6 > gpurun_out/dist_tests.log 2>&1
rc=$?
tail -22 gpurun_out/dist_tests.log
exit $rc
