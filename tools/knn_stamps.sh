#!/bin/bash
# development aid: the phase stamps of diagnostic builds (tools/build_variant.sh <name> -DTM_KNN2_STAMPS=1 ...)
set -o pipefail
mkdir -p gpurun_out
for v in "$@"; do
  TM_LIB_VARIANT=$v TM_KNN_DEBUG=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra --no-dense-extra --no-h2d-extra --no-frozen-extra --no-kmodes-extra > gpurun_out/st_$v.json 2> gpurun_out/st_$v.err || { tail -5 gpurun_out/st_$v.err; exit 1; }
  echo "== $v"; grep "stamps\|kernel" gpurun_out/st_$v.err | tail -13
done
