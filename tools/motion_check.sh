#!/bin/bash
# development aid: the motion / window-DCT tests, then the bench's motion extras
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_encoder.py -x -q -m gpu -k "window or motion or every_source or run_all or y4m" > gpurun_out/motion_tests.log 2>&1
rc=$?
tail -3 gpurun_out/motion_tests.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" gpurun_out/motion_tests.log | tail -20; exit $rc; }
for env in "" "TM_MOTION_PACK_SEPARATE=1"; do
env $env timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dense-extra --no-h2d-extra --no-kmodes-extra --no-frozen-extra > gpurun_out/motion.json 2> gpurun_out/motion.err || { tail -5 gpurun_out/motion.err; exit 1; }
python - "$env" <<'PY'
import json, sys
j = json.loads(open('gpurun_out/motion.json').read().strip().splitlines()[-1])
print(sys.argv[1] or "default", "value %.0f" % j["value"])
for k in ("with_motion_prediction", "with_extended_palette_usage", "with_motion_and_extended_palette_usage"):
    print("  ", k, round(j[k]["value"]), {a: b for a, b in j[k]["stage_ms"].items() if b > 20})
PY
done
