#!/bin/bash
# development aid: KNN tests, then the headline bench with the scan's debug line, then the phase stamps of a diagnostic build
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knn" > gpurun_out/knn_tests.log 2>&1
rc=$?
tail -3 gpurun_out/knn_tests.log
[ $rc -ne 0 ] && exit $rc
TM_KNN_DEBUG=1 timeout -k 10 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra > gpurun_out/bench_dbg.json 2> gpurun_out/bench_dbg.err || { tail -5 gpurun_out/bench_dbg.err; exit 1; }
python - <<'PY'
import json
j = json.loads(open("gpurun_out/bench_dbg.json").read().strip().splitlines()[-1])
print("fps=%.0f ms=%.2f stages=%s knn_ms=%.2f frac=%.4f dense_ms=%.1f dense_pipe=%.3f" % (j["value"], j["ms_per_step"], j["stage_ms"], j["roofline"]["launch_ms"], j["roofline"]["frac"], j["roofline_dense"]["launch_ms"], j["roofline_dense"]["mfma_pipe_frac"]))
PY
grep "kernel" gpurun_out/bench_dbg.err | tail -2
if [ -f tiler_amd/lib/variants/libtilemotion_stamps.so ]; then
  TM_LIB_VARIANT=stamps TM_KNN_DEBUG=1 timeout -k 10 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra > gpurun_out/bench_st.json 2> gpurun_out/bench_st.err
  grep "stamps\|kernel" gpurun_out/bench_st.err | head -11
fi
