#!/bin/bash
# development aid: kernel stats of the extended-palette run (motion prediction off) on the bench clip:  tools/prof_epu.sh [radius]
OUT=gpurun_out/prof_epu
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/defaults_probe.py 3 ${1:-0} > $OUT/run.log 2> $OUT/run.err
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
cat $OUT/run.log
