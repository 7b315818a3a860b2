#!/bin/bash
# development aid: the extended-palette run (motion prediction off) on both bench clips for the current build and the variants named
for v in cur "$@"; do
  for clip in "" "--frozen-columns"; do
    unset TM_LIB_VARIANT; if [ $v != cur ]; then export TM_LIB_VARIANT=$v; fi
    TM_KNN_DEBUG=1 timeout -k 10 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-motion-extra --no-dense-extra --no-h2d-extra --no-kmodes-extra --no-frozen-extra $clip > gpurun_out/epu_$v.json 2> gpurun_out/epu_$v.err
    python - $v "$clip" <<'PY'
import json,sys
j=json.loads(open('gpurun_out/epu_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
e=j['with_extended_palette_usage']
print(sys.argv[1], sys.argv[2] or 'literal', 'EPU fps %.0f reconstruct %.1f'%(e['value'], e['stage_ms']['reconstruct']))
PY
    grep "top-64 pass" gpurun_out/epu_$v.err | head -9 | cut -c1-100
  done
done
