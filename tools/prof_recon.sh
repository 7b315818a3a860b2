#!/bin/bash
# development aid: kernel trace of the headline bench, then the kernels of one step from Dither on (tools/trace_tail.py)
TAG=${1:-x}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra --no-h2d-extra --no-dense-extra > $OUT/bench.json 2> $OUT/bench.err
python tools/trace_tail.py $OUT > $OUT/tail.txt
tail -1 $OUT/bench.json | cut -c1-200
