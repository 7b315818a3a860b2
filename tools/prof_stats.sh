#!/bin/bash
# rocprofv3 --kernel-trace --stats of the headline bench (no extras): gpurun_out/prof_<tag>/
TAG=${1:-r04}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra --no-h2d-extra --no-dense-extra --no-frozen-extra --no-kmodes-extra > $OUT/bench.json 2> $OUT/bench.err
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
head -45 $OUT/kernel_stats.csv | cut -c1-150
