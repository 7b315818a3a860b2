#!/bin/bash
# development aid: the headline bench (no extras) once per named variant build (no base run).   bash tools/ab_one.sh v1 v2 ...
set -o pipefail
mkdir -p gpurun_out
ARGS="--steps ${AB_STEPS:-3} --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra --no-dense-extra --no-h2d-extra --no-kmodes-extra --no-frozen-extra ${AB_ARGS:-}"
for v in "$@"; do
  if [ $v = base ]; then unset TM_LIB_VARIANT; else export TM_LIB_VARIANT=$v; fi
  TM_KNN_DEBUG=1 timeout -k 10 300 python bench.py $ARGS > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { tail -5 gpurun_out/ab_$v.err; exit 1; }
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
j = json.loads(open('gpurun_out/ab_%s.json' % v).read().strip().splitlines()[-1])
print('%s fps=%.0f ms=%.2f knn_ms=%.3f frac=%.4f gate=%s' % (v, j['value'], j['ms_per_step'], j['roofline']['launch_ms'], j['roofline']['frac'], j.get('parity_gate')), j['stage_ms'])
PY
  grep "stamps\]" gpurun_out/ab_$v.err | grep -v seeds | tail -9
done
