#!/bin/bash
# development aid: PreparePalettes' sub-step times under several environment settings
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for v in base "$@"; do
  if [ "$v" = base ]; then pre=""; else pre="${v//,/ }"; fi
  env $pre TM_PP_DEBUG=1 timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra --no-dense-extra --no-h2d-extra --no-frozen-extra --no-kmodes-extra > gpurun_out/ab_pp.json 2> gpurun_out/ab_pp.err || { tail -5 gpurun_out/ab_pp.err; exit 1; }
  echo "$v: $(grep 'tile -> palette' gpurun_out/ab_pp.err | tail -3 | awk '{print $(NF-1)}' | tr '\n' ' ') | colours $(grep 'palette colours' gpurun_out/ab_pp.err | tail -3 | awk '{print $(NF-1)}' | tr '\n' ' ')"
done
done
