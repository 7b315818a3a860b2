#!/bin/bash
# development aid: one PMC pass (SQ issue / wait counters) over the kernels matching a regex, headline bench, one step
# usage on the GPU box: bash tools/pmc_kernel.sh <regex> <tag> [counters...]
set -o pipefail
RE=$1; TAG=$2; shift 2
CNT=${@:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE}
OUT=gpurun_out/pmck_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 200 rocprofv3 --pmc $CNT --kernel-include-regex "$RE" --output-format csv -d $OUT/p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra --no-dense-extra --no-h2d-extra --no-frozen-extra --no-kmodes-extra > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys
from collections import OrderedDict
rows = OrderedDict()
for p in glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = (r["Kernel_Name"].split("(")[0][-50:], int(r["Dispatch_Id"]))
        rows.setdefault(k, {})
        rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for (k, d), c in rows.items():
    print(k, d, " ".join("%s=%.4g" % kv for kv in c.items()))
PY
