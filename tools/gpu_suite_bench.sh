#!/bin/bash
# development aid: the whole GPU suite, then the default bench line (what the driver runs at round end)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/suite.log 2>&1
rc=$?
tail -25 gpurun_out/suite.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail -5 gpurun_out/bench_default.err; exit 1; }
python - <<'PY'
import json
j = json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print('value %.0f fps, %.2f ms/step, gate %s, roofline.frac %.3f' % (j['value'], j['ms_per_step'], j.get('parity_gate'), j['roofline']['frac']))
print(j['stage_ms'])
print(j.get('parity_gate_detail'))
for k in ('with_frozen_columns', 'with_motion_prediction', 'with_extended_palette_usage', 'with_motion_and_extended_palette_usage'):
    if k in j: print(k, '%.0f' % j[k]['value'], j[k].get('stage_ms'))
print('cpu', j.get('cpu_baseline', {}).get('value'))
PY
