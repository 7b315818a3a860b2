#!/bin/bash
# development aid: the headline bench for the in-tree build and for each named variant build (tools/build_variant.sh), alternating, in one GPU call
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for v in base "$@"; do
  if [ $v = base ]; then unset TM_LIB_VARIANT; else export TM_LIB_VARIANT=$v; fi
  timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra --no-dense-extra --no-h2d-extra --no-frozen-extra --no-kmodes-extra > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { tail -5 gpurun_out/ab_$v.err; exit 1; }
  python -c "
import json
j=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1])
print('$v fps=%.0f ms=%.2f'%(j['value'],j['ms_per_step']), j['stage_ms'])"
done
done
