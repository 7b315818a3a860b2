#!/bin/bash
# development aid: the extended-palette run (literal and frozen clip) for the in-tree build and named variants
set -o pipefail
mkdir -p gpurun_out
for v in base "$@"; do
  if [ $v = base ]; then unset TM_LIB_VARIANT; else export TM_LIB_VARIANT=$v; fi
  for clip in "" "--frozen-columns"; do
    TM_KNN_DEBUG=1 timeout -k 10 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-motion-extra --no-dense-extra --no-h2d-extra --no-kmodes-extra --no-frozen-extra $clip > gpurun_out/epu_$v.json 2> gpurun_out/epu_$v.err || { tail -5 gpurun_out/epu_$v.err; exit 1; }
    python - "$v" "$clip" <<'PY'
import json, sys
j = json.loads(open('gpurun_out/epu_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
r = j["with_extended_palette_usage"]
print(sys.argv[1], sys.argv[2] or "literal", "EPU only: %.0f fps, reconstruct %.1f ms" % (r["value"], r["stage_ms"]["reconstruct"]), "| with motion: reconstruct %.1f" % j["with_motion_and_extended_palette_usage"]["stage_ms"]["reconstruct"])
PY
    grep "top-[0-9]* pass" gpurun_out/epu_$v.err | tail -7 | sed 's/^/    /'
  done
done
