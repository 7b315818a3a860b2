#!/bin/bash
# development aid: the whole GPU suite, then the headline bench (no extras)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -4 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" gpurun_out/gpu_tests.log | tail -20; exit $rc; }
timeout -k 10 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra ${BENCH_ARGS:-} > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err || { tail -5 gpurun_out/bench_q.err; exit 1; }
python - <<'PY'
import json
j = json.loads(open("gpurun_out/bench_q.json").read().strip().splitlines()[-1])
print("fps=%.0f ms=%.2f stages=%s knn_ms=%.2f frac=%.4f" % (j["value"], j["ms_per_step"], j["stage_ms"], j["roofline"]["launch_ms"], j["roofline"]["frac"]))
print("h2d+d2h", j.get("with_h2d_d2h"), j.get("with_h2d_overlapped_d2h"), "scan", j["roofline"].get("scan", {}).get("kernels_ms"), "dense", j.get("roofline_dense", {}).get("mfma_pipe_frac"))
PY
