#!/bin/bash
# development aid: rocprofv3 kernel stats of the headline bench (no extras) -> gpurun_out/prof_<tag>/ + a top-N printout
set -o pipefail
TAG=${1:-x}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/prof_$TAG
rm -rf $OUT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra --no-dense-extra --no-h2d-extra --no-frozen-extra --no-kmodes-extra > $OUT.json 2> $OUT.err || { tail -5 $OUT.err; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print("%-70s calls %5s total %8.3f ms avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
