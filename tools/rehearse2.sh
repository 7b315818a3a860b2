#!/bin/bash
# development aid: bench.py --gpus 2 as two processes sharing the one GPU over gloo (everything of the multi-process path except RCCL itself)
set -o pipefail
mkdir -p gpurun_out
TM_BENCH_REHEARSE=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node ${1:-2} --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus ${1:-2} --steps 2 --warmup 1 > gpurun_out/rehearse.json 2> gpurun_out/rehearse.err || { tail -20 gpurun_out/rehearse.err; exit 1; }
python - <<'PY'
import json
j = json.loads(open("gpurun_out/rehearse.json").read().strip().splitlines()[-1])
print("n_gpus", j["n_gpus"], "fps=%.0f ms=%.2f" % (j["value"], j["ms_per_step"]), j["stage_ms"])
print("collectives_per_step", j["collectives_per_step"])
print("tiles", j["config"]["final_tiles_after_reindex"], j["config"]["global_tiles_T"])
PY
