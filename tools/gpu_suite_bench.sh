#!/bin/bash
# development aid: the whole GPU suite, then the default bench line (all extras) into gpurun_out/bench_full.json
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -4 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" gpurun_out/gpu_tests.log | tail -20; exit $rc; }
timeout -k 10 900 python bench.py --steps ${BENCH_STEPS:-5} --warmup 1 ${BENCH_ARGS:-} > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err || { tail -5 gpurun_out/bench_full.err; exit 1; }
python - <<'PY'
import json
j = json.loads(open("gpurun_out/bench_full.json").read().strip().splitlines()[-1])
print("fps=%.0f ms=%.2f stages=%s knn_ms=%.2f frac=%.4f" % (j["value"], j["ms_per_step"], j["stage_ms"], j["roofline"]["launch_ms"], j["roofline"]["frac"]))
print("with_h2d_d2h", j.get("with_h2d_d2h"), "overlapped", j.get("with_h2d_overlapped_d2h"), j.get("transfers", {}).get("with_h2d_d2h"))
print("scan", j["roofline"].get("scan"))
for k in ("with_frozen_columns", "with_motion_prediction", "with_extended_palette_usage", "with_motion_and_extended_palette_usage"):
    if k in j: print(k, round(j[k]["value"]), j[k].get("ms_per_step", j[k].get("ms")))
print("kmeans", j.get("stage_rooflines", {}).get("kmeans"))
print("cpu", {k: v for k, v in j.get("cpu_baseline", {}).items() if k in ("value", "cores", "legs", "with_32_threads")})
PY
