#!/bin/bash
# development aid: the extended-palette (k = 64) run of the bench on the frozen and the literal clip, with the collection passes' debug lines
set -o pipefail
mkdir -p gpurun_out
for clip in "--frozen-columns" ""; do
  TM_KNN_DEBUG=1 timeout -k 10 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-motion-extra --no-dense-extra --no-h2d-extra --no-kmodes-extra --no-frozen-extra $clip > gpurun_out/epu.json 2> gpurun_out/epu.err || { tail -5 gpurun_out/epu.err; exit 1; }
  python - "$clip" <<'PY'
import json, sys
j = json.loads(open('gpurun_out/epu.json').read().strip().splitlines()[-1])
for k in ("with_extended_palette_usage", "with_motion_and_extended_palette_usage"):
    print(sys.argv[1] or "literal", k, round(j[k]["value"]), j[k]["stage_ms"]["reconstruct"])
PY
  grep "top-64 pass" gpurun_out/epu.err | tail -12
done
