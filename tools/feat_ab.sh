#!/bin/bash
# development aid: the bench (motion extra on) with the features eight tiles a wave and a tile at a time
set -o pipefail
mkdir -p gpurun_out
for v in "" 1 "" 1; do
if [ -n "$v" ]; then export TM_FEATURES_BY_TILE=1; else unset TM_FEATURES_BY_TILE; fi
timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-dense-extra --no-h2d-extra --no-kmodes-extra --no-frozen-extra --no-defaults-extra > gpurun_out/feat.json 2> gpurun_out/feat.err || { tail -5 gpurun_out/feat.err; exit 1; }
python - "$v" <<'PY'
import json, sys
j = json.loads(open('gpurun_out/feat.json').read().strip().splitlines()[-1])
print("by tile" if sys.argv[1] else "8 a wave", "value %.0f  %.2f ms" % (j["value"], j["ms_per_step"]), j["stage_ms"], "features %.2f ms" % j["stage_rooflines"]["features"]["ms"], j["parity_gate"])
k = "with_motion_prediction"
print("  ", k, round(j[k]["value"]), {a: b for a, b in j[k]["stage_ms"].items() if b > 5})
PY
done
