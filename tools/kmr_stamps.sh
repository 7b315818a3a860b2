#!/bin/bash
# development aid: the resident tile k-means' phase stamps (variant build "stamps": -DTM_KMR_STAMPS=1) on the bench clip
set -o pipefail
mkdir -p gpurun_out
export TM_LIB_VARIANT=stamps
TM_PP_DEBUG=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra --no-dense-extra --no-h2d-extra --no-frozen-extra --no-kmodes-extra > gpurun_out/kmr_stamps.json 2> gpurun_out/kmr_stamps.err || { tail -5 gpurun_out/kmr_stamps.err; exit 1; }
grep "tm_kmr\|tile -> palette" gpurun_out/kmr_stamps.err | tail -40
