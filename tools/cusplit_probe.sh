# development aid: the headline bench with the tile k-means and the query features on disjoint sets of compute units (TM_CU_SPLIT)
mkdir -p gpurun_out
set -o pipefail
for v in ${SPLITS:-0 64 32 128}; do
  if [ $v = 0 ]; then unset TM_CU_SPLIT; else export TM_CU_SPLIT=$v; fi
  timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra --no-dense-extra --no-h2d-extra > gpurun_out/cs_$v.json 2> gpurun_out/cs_$v.err || { tail -5 gpurun_out/cs_$v.err; exit 1; }
  python -c "
import json
j=json.loads(open('gpurun_out/cs_$v.json').read().strip().splitlines()[-1])
print('split=$v fps=%.0f ms=%.2f'%(j['value'],j['ms_per_step']), j['stage_ms'])"
done
