#!/bin/bash
# development aid: the k-means parity tests, then the headline bench twice with PreparePalettes' sub-step times
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_encoder.py -x -q -m gpu -k "kmeans or palett or quantize or run_all" > gpurun_out/km_tests.log 2>&1
rc=$?
tail -3 gpurun_out/km_tests.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" gpurun_out/km_tests.log | tail -20; exit $rc; }
for rep in 1 2 3; do
TM_PP_DEBUG=1 timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra --no-dense-extra --no-h2d-extra --no-frozen-extra --no-kmodes-extra > gpurun_out/km_bench.json 2> gpurun_out/km_bench.err || { tail -5 gpurun_out/km_bench.err; exit 1; }
python -c "
import json
j=json.loads(open('gpurun_out/km_bench.json').read().strip().splitlines()[-1])
print('fps=%.0f ms=%.2f'%(j['value'],j['ms_per_step']), j['stage_ms'], j['stage_rooflines']['kmeans']['tile_iters'], j['stage_rooflines']['kmeans']['pixel_iters'])"
grep "tm_pp" gpurun_out/km_bench.err | tail -6 | grep "192-D\|3-D\|colours"
done
