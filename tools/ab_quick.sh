#!/bin/bash
# development aid: the headline bench (no extras) for the in-tree build against named variant builds, alternating; then the L2 passes
# (FETCH_SIZE; WRITE_SIZE + TCC hit/miss) of the last named variant.   bash tools/ab_quick.sh [--pmc] v1 v2 ...
set -o pipefail
mkdir -p gpurun_out
PMC=0; [ "$1" = "--pmc" ] && { PMC=1; shift; }
ARGS="--steps ${AB_STEPS:-5} --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra --no-dense-extra --no-h2d-extra --no-kmodes-extra ${AB_ARGS:---no-frozen-extra}"
for rep in 1 2; do
for v in base "$@"; do
  if [ $v = base ]; then unset TM_LIB_VARIANT; else export TM_LIB_VARIANT=$v; fi
  TM_KNN_DEBUG=1 timeout -k 10 300 python bench.py $ARGS > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { tail -5 gpurun_out/ab_$v.err; exit 1; }
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
j = json.loads(open('gpurun_out/ab_%s.json' % v).read().strip().splitlines()[-1])
print('%s fps=%.0f ms=%.2f knn_ms=%.3f frac=%.4f' % (v, j['value'], j['ms_per_step'], j['roofline']['launch_ms'], j['roofline']['frac']), j['stage_ms'])
w = j.get('with_frozen_columns')
if w: print('   frozen: fps=%.0f ms=%.2f' % (w['value'], w['ms_per_step']), w['knn_kernels_ms'], w['stage_ms'])
PY
  grep "kernel" gpurun_out/ab_$v.err | tail -1
done
done
if [ $PMC = 1 ]; then
  cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
  for v in base "$@"; do
    if [ $v = base ]; then unset TM_LIB_VARIANT; else export TM_LIB_VARIANT=$v; fi
    OUT=gpurun_out/pmcq_$v; mkdir -p $OUT
    B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra --no-dense-extra --no-h2d-extra --no-frozen-extra --no-kmodes-extra"
    timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_knn" --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.json 2> $OUT/fetch.err || { tail -5 $OUT/fetch.err; exit 1; }
    timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "k_knn" --output-format csv -d $OUT/write -- $B > $OUT/write.json 2> $OUT/write.err || { tail -5 $OUT/write.err; exit 1; }
    python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(out, k, {c: (len(v), sum(v) / len(v)) for c, v in d.items()})
PY
  done
fi
