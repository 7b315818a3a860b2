#!/bin/bash
# development aid: KNN parity tests, then the headline bench for the in-tree build against named variant builds, alternating
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "knn or KNN or match" > gpurun_out/knn_tests.log 2>&1
rc=$?
tail -3 gpurun_out/knn_tests.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" gpurun_out/knn_tests.log | tail -20; exit $rc; }
for rep in 1 2; do
for v in base "$@"; do
  if [ $v = base ]; then unset TM_LIB_VARIANT; else export TM_LIB_VARIANT=$v; fi
  TM_KNN_DEBUG=1 timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra --no-dense-extra --no-h2d-extra --no-frozen-extra --no-kmodes-extra > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { tail -5 gpurun_out/ab_$v.err; exit 1; }
  python -c "
import json
j=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1])
print('$v fps=%.0f ms=%.2f knn_ms=%.3f frac=%.4f'%(j['value'],j['ms_per_step'],j['roofline']['launch_ms'],j['roofline']['frac']), j['stage_ms'])"
  grep "kernel\|big columns" gpurun_out/ab_$v.err | tail -2
done
done
