#!/bin/bash
# development aid: PMC passes (one counter set per run) over the kernels matching a regex while a python script runs
# usage on the GPU box: bash tools/pmc_script.sh <regex> <tag> <script.py> [args]
set -o pipefail
RE=$1; TAG=$2; shift 2
OUT=gpurun_out/pmcs_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
i=0
for CNT in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $CNT --kernel-include-regex "$RE" --output-format csv -d $OUT/p$i -- python3 "$@" > $OUT/run$i.log 2> $OUT/run$i.err || { tail -5 $OUT/run$i.err; exit 1; }
done
python3 - $OUT <<'PY'
import csv, glob, sys
from collections import OrderedDict
rows = OrderedDict()
for p in sorted(glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        rows.setdefault(k, {})
        d = rows[k].setdefault(r["Counter_Name"], [0.0, 0])
        d[0] += float(r["Counter_Value"])
    seen = set()
for k, c in rows.items():
    print(k, " ".join("%s=%.4g" % (n, v[0]) for n, v in c.items()))
PY
