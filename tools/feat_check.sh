#!/bin/bash
# development aid: the feature tests, then tools/motion_check.sh's bench part and the headline
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_encoder.py -x -q -m gpu -k "features or window or motion or every_source or run_all or first_look or default_path" > gpurun_out/feat_tests.log 2>&1
rc=$?
tail -3 gpurun_out/feat_tests.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" gpurun_out/feat_tests.log | tail -20; exit $rc; }
for rep in 1 2; do
timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-dense-extra --no-h2d-extra --no-kmodes-extra --no-frozen-extra --no-defaults-extra > gpurun_out/feat.json 2> gpurun_out/feat.err || { tail -5 gpurun_out/feat.err; exit 1; }
python - <<'PY'
import json
j = json.loads(open('gpurun_out/feat.json').read().strip().splitlines()[-1])
print("value %.0f  %.2f ms" % (j["value"], j["ms_per_step"]), j["stage_ms"], "features %.2f ms" % j["stage_rooflines"]["features"]["ms"], j["parity_gate"])
k = "with_motion_prediction"
print("  ", k, round(j[k]["value"]), {a: b for a, b in j[k]["stage_ms"].items() if b > 20})
PY
done
