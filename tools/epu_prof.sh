#!/bin/bash
# development aid: kernel stats of the extended-palette configuration alone (literal clip, motion prediction off), and the collection passes' debug lines
set -o pipefail
OUT=gpurun_out/prof_epu2
rm -rf $OUT; mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
export TM_PROBE_LITERAL=1
TM_KNN_DEBUG=1 timeout -k 10 300 python tools/defaults_probe.py 2 0 1 > $OUT/plain.log 2> $OUT/plain.err || { tail -5 $OUT/plain.err; exit 1; }
cat $OUT/plain.log; grep "top-64 pass\|topk" $OUT/plain.err | tail -14
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/defaults_probe.py 2 0 1 > $OUT/run.log 2> $OUT/run.err || { tail -5 $OUT/run.err; exit 1; }
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
python3 - <<'PY'
import csv
rows = list(csv.reader(open('gpurun_out/prof_epu2/kernel_stats.csv')))[1:]
for r in rows[:28]:
    print("%-60s %6s calls  total %9.2f ms  avg %9.1f us" % (r[0].split('(')[0][-60:], r[1], float(r[2]) / 1e6, float(r[3]) / 1e3))
PY
find $OUT -name "*kernel_trace.csv" -size +30M -delete
