#!/bin/bash
# development aid: KNN parity tests on the current build, then the headline bench (no extras) with the third scan shape
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_encoder.py -x -q -m gpu -k "knn or KNN or match or motion_search_q or topk or epu or extended" > gpurun_out/knn_tests.log 2>&1
rc=$?
tail -5 gpurun_out/knn_tests.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" gpurun_out/knn_tests.log | tail -20; exit $rc; }
ARGS="--steps ${AB_STEPS:-5} --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra --no-dense-extra --no-h2d-extra --no-kmodes-extra ${AB_ARGS:-}"
for rep in 1 2; do
for v in v3 "$@"; do
  unset TM_LIB_VARIANT
  if [ $v != v3 ]; then export TM_LIB_VARIANT=$v; fi
  TM_KNN_DEBUG=1 timeout -k 10 400 python bench.py $ARGS > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { tail -5 gpurun_out/ab_$v.err; exit 1; }
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
j = json.loads(open('gpurun_out/ab_%s.json' % v).read().strip().splitlines()[-1])
print('%s fps=%.0f ms=%.2f knn_ms=%.3f frac=%.4f' % (v, j['value'], j['ms_per_step'], j['roofline']['launch_ms'], j['roofline']['frac']), j['stage_ms'])
w = j.get('with_frozen_columns')
if w: print('   frozen: fps=%.0f ms=%.2f' % (w['value'], w['ms_per_step']), w['knn_kernels_ms'], w['stage_ms'])
PY
  grep "tm_knn\] seeds\|tm_knn\] scan\|stamps" gpurun_out/ab_$v.err | tail -${AB_TAIL:-4}
done
done
