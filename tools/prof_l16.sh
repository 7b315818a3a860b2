#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in cur base; do
  unset TM_LIB_VARIANT; [ $v = base ] && export TM_LIB_VARIANT=base
  for c in frozen literal; do
    a=""; [ $c = frozen ] && a="--frozen-columns"
    O=gpurun_out/prof_l16_${v}_$c; mkdir -p $O
    timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 bench.py --steps 2 --warmup 1 $a --no-cpu-baseline --no-motion-extra --no-defaults-extra --no-h2d-extra --no-dense-extra --no-frozen-extra --no-kmodes-extra > $O/bench.json 2> $O/bench.err
    echo "$v $c"; grep "k_assign192_list\|k_h_bounds\|k_h_update" $O/p_kernel_stats.csv | awk -F'","' '{printf "  %-40s calls %s total_ms %.2f avg_us %.1f\n", substr($1,2,40), $2, $3/1e6, $4/1e3}'
  done
done
