#!/bin/bash
# development aid: KNN tests, then the KNN micro-benchmark for both scan shapes (run on the GPU box through gpurun)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knn" > gpurun_out/knn_tests.log 2>&1
rc=$?
tail -5 gpurun_out/knn_tests.log
[ $rc -ne 0 ] && exit $rc
TM_KNN_DEBUG=1 timeout -k 10 300 python tools/knn_bench.py 60 > gpurun_out/knn_bench_v2.log 2>&1 || { tail -20 gpurun_out/knn_bench_v2.log; exit 1; }
grep -v "^\[tm_knn\] nq\|curve" gpurun_out/knn_bench_v2.log | tail -8
TM_KNN_V1=1 TM_KNN_DEBUG=1 timeout -k 10 300 python tools/knn_bench.py 60 > gpurun_out/knn_bench_v1.log 2>&1
grep -v "^\[tm_knn\] nq\|curve" gpurun_out/knn_bench_v1.log | tail -8
(rocprofv3 -L 2>/dev/null | grep -i "mfma\|SQ_BUSY_CY\|SQ_WAVE_CYCLES\|SQ_INSTS_VALU \|SQ_WAIT_ANY\|SQ_ACTIVE_INST_ANY\|LDS_BANK" | head -60) > gpurun_out/counters.txt 2>&1
wc -l gpurun_out/counters.txt
