#!/bin/bash
# development aid: every launch of the feature kernels in a headline bench run (kernel trace), longest first
set -o pipefail
OUT=gpurun_out/prof_feat
rm -rf $OUT; mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra --no-h2d-extra --no-dense-extra --no-frozen-extra --no-kmodes-extra > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_features" in r["Kernel_Name"]]
for r in rows:
    print(r["Kernel_Name"][:40], r.get("Grid_Size_X") or r.get("Grid_Size"), "%.3f ms" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6), "start %.3f" % (int(r["Start_Timestamp"]) / 1e6 % 100000))
PY
find $OUT -name "*.csv" -size +20M -delete
