#!/bin/bash
# development aid: kernel stats of a bench run with the motion extra (the motion-only kernels are recognisable by name)
set -o pipefail
OUT=gpurun_out/prof_motion
rm -rf $OUT; mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-defaults-extra --no-h2d-extra --no-dense-extra --no-frozen-extra --no-kmodes-extra > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
head -40 $OUT/kernel_stats.csv | cut -c1-160
python3 - <<'PY'
import json
j = json.loads(open('gpurun_out/prof_motion/bench.json').read().strip().splitlines()[-1])
k = "with_motion_prediction"
print(k, round(j[k]["value"]), j[k]["stage_ms"], j[k].get("steps"))
PY
find $OUT -name "*kernel_trace.csv" -size +30M -delete
