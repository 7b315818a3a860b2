#!/bin/bash
# PMC passes over the three kernels of the third scan shape in one bench step, each counter set in its own run (no tracing beside --pmc).
# The run keeps bench.py's dense diagnostic launch (the last k_knn_consume dispatch); PMC_ARGS adds bench flags (e.g. --frozen-columns).
# Run on the GPU box:  bash tools/pmc_knn3.sh <tag> [passes...]   -> gpurun_out/pmc_<tag>/*.csv + summary + traffic.json
set -o pipefail
TAG=${1:-r04}; shift
PASSES=${*:-sq sq2 fetch write}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
BENCH="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-motion-extra --no-defaults-extra --no-h2d-extra --no-frozen-extra --no-kmodes-extra $PMC_ARGS"
run() {  # name, counters...
  local name=$1; shift
  timeout -k 10 600 rocprofv3 --pmc "$@" --kernel-include-regex "k_knn_seed|k_knn_lists|k_knn_consume" --output-format csv -d $OUT/$name -- $BENCH > $OUT/$name.json 2> $OUT/$name.err || { tail -5 $OUT/$name.err; return 1; }
}
for p in $PASSES; do
  case $p in
    sq) run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE || exit 1 ;;
    sq2) run sq2 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS || exit 1 ;;
    sq3) run sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_ADDR_CONFLICT || exit 1 ;;
    fetch) run fetch FETCH_SIZE || exit 1 ;;
    write) run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1 ;;
  esac
done
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt && cat $OUT/summary.txt &&
python3 tools/pmc_traffic.py $OUT $TAG --dense > $OUT/traffic.json && cat $OUT/traffic.json
