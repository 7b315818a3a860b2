#!/bin/bash
set -o pipefail
timeout -k 10 120 python tools/time_motion.py || exit 1
export TM_TIME_REPS=2
RE="k_mo_search_mfma"; TAG=mosearch
OUT=gpurun_out/pmcs_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
i=0
for CNT in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $CNT --kernel-include-regex "$RE" --output-format csv -d $OUT/p$i -- python3 tools/time_motion.py > $OUT/run$i.log 2> $OUT/run$i.err || { tail -5 $OUT/run$i.err; }
done
python3 - $OUT <<'PY'
import csv, glob, sys
from collections import OrderedDict
rows = OrderedDict()
for p in sorted(glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True)):
    n = {}
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        rows.setdefault(k, {})
        d = rows[k].setdefault(r["Counter_Name"], [0.0, 0])
        d[0] += float(r["Counter_Value"]); d[1] += 1
for k, c in rows.items():
    print(k)
    for n, v in c.items():
        print("   %-28s %.5g per launch (%d samples)" % (n, v[0] / max(1, v[1]), v[1]))
PY
