#!/bin/bash
# runs the headline bench (no extras) for the default build and each named variant; prints value / reconstruct / knn launch ms
for v in "" "$@"; do
  out=$(TM_LIB_VARIANT=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-motion-extra --no-defaults-extra ${AB_ARGS:-} 2>&1 | tail -1)
  python - "$v" "$out" <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2])
    print("variant=%-8s fps=%.0f ms=%.2f recon=%.2f knn_ms=%.2f dense_ms=%.1f staged/WG?=%s" % (sys.argv[1] or "default", j["value"], j["ms_per_step"], j["stage_ms"]["reconstruct"], j["roofline"]["launch_ms"], j["roofline_dense"]["launch_ms"], j["roofline"].get("pairs_per_launch")))
except Exception as e:
    print("variant", sys.argv[1], "failed:", sys.argv[2][-400:])
PY
done
