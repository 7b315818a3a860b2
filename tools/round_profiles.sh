#!/bin/bash
# what profiles/ keeps of a round: kernel stats of the headline step (rocprofv3 --kernel-trace --stats), the profiled run's JSON line, the
# default bench line (every extra, not under the profiler).   bash tools/round_profiles.sh r05
TAG=${1:-r05}
mkdir -p gpurun_out
bash tools/prof_stats.sh $TAG > gpurun_out/prof_$TAG.log 2>&1
cp gpurun_out/prof_$TAG/kernel_stats.csv gpurun_out/${TAG}_bench_720p300_kernel_stats.csv
tail -1 gpurun_out/prof_$TAG/bench.json > gpurun_out/${TAG}_bench_720p300_profiled_run.json
timeout -k 10 900 python bench.py > gpurun_out/${TAG}_bench_720p300_unprofiled_run.json 2> gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
python - $TAG <<'PY'
import json, sys
j = json.loads(open('gpurun_out/%s_bench_720p300_unprofiled_run.json' % sys.argv[1]).read().strip().splitlines()[-1])
print('value %.0f fps, %.2f ms/step, gate %s, roofline.frac %.3f launch %.2f ms traffic %s' % (j['value'], j['ms_per_step'], j.get('parity_gate'), j['roofline']['frac'], j['roofline']['launch_ms'], j['roofline']['traffic']))
print(j['stage_ms'])
for k in ('with_frozen_columns', 'with_motion_prediction', 'with_extended_palette_usage', 'with_motion_and_extended_palette_usage'):
    if k in j: print(k, '%.0f' % j[k]['value'], j[k].get('stage_ms'))
print('h2d', j.get('with_h2d_d2h'), j.get('with_h2d_overlapped_d2h'))
print('kmeans', {k: v for k, v in j['stage_rooflines']['kmeans'].items() if k != 'note'})
print('kmodes', j['stage_rooflines'].get('kmodes', {}).get('ms_per_iteration'))
print('dense', j.get('roofline_dense', {}).get('frac'), j.get('roofline_dense', {}).get('launch_ms'))
print('cpu', j.get('cpu_baseline', {}).get('value'), j.get('cpu_baseline', {}).get('cores'))
for k in ('load', 'features', 'dedup', 'dither'):
    print(k, round(j['stage_rooflines'][k]['frac'], 3), round(j['stage_rooflines'][k]['ms'], 3))
PY
