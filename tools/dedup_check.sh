#!/bin/bash
# development aid: the tests that go through the dedup kernels, then the headline bench for the in-tree build with and without TM_DEDUP_SORT
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_encoder.py tests/test_gpu_fullsize.py -x -q -m gpu -k "dedup or reduce or budget or run_all or full_size or sharded or encoder or dither" > gpurun_out/dedup_tests.log 2>&1
rc=$?
tail -4 gpurun_out/dedup_tests.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" gpurun_out/dedup_tests.log | tail -20; exit $rc; }
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra --no-dense-extra --no-h2d-extra --no-kmodes-extra"
for rep in 1 2; do
for v in table sort; do
  unset TM_DEDUP_SORT
  [ $v = sort ] && export TM_DEDUP_SORT=1
  timeout -k 10 400 python bench.py $ARGS > gpurun_out/dd_$v.json 2> gpurun_out/dd_$v.err || { tail -5 gpurun_out/dd_$v.err; exit 1; }
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
j = json.loads(open('gpurun_out/dd_%s.json' % v).read().strip().splitlines()[-1])
print('%s fps=%.0f ms=%.2f gate=%s' % (v, j['value'], j['ms_per_step'], j.get('parity_gate')), j['stage_ms'])
w = j.get('with_frozen_columns')
if w: print('   frozen: fps=%.0f ms=%.2f' % (w['value'], w['ms_per_step']), w['stage_ms'])
d = j.get('stage_rooflines', {}).get('dedup')
if d: print('   dedup operator: %.3f ms, frac %.3f' % (d['ms'], d['frac']))
PY
done
done
