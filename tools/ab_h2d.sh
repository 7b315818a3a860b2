#!/bin/bash
# development aid: the host-clip figures of the bench under several environment settings
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for v in base "$@"; do
  if [ "$v" = base ]; then pre=""; else pre="${v//,/ }"; fi
  env $pre timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra --no-dense-extra --no-frozen-extra --no-kmodes-extra > gpurun_out/ab_h2d.json 2> gpurun_out/ab_h2d.err || { tail -5 gpurun_out/ab_h2d.err; exit 1; }
  python -c "
import json
j=json.loads(open('gpurun_out/ab_h2d.json').read().strip().splitlines()[-1])
print('%-28s resident %.2f ms  with_h2d %.2f ms  overlapped %.2f ms'%('$v',j['ms_per_step'],j['with_h2d']['ms_per_step'],j['with_h2d_overlapped']['ms_per_step']), j['with_h2d_overlapped'].get('stage_ms'))"
done
done
