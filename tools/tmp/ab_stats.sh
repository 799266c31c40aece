#!/bin/bash
# single-stream kernel stats, lazy on vs off
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for tag in on off; do
  EXTRA=""; [ $tag = off ] && EXTRA="--set ops.LAZY_ACT=False --set ops.W2_BWD_FUSED=False"
  EFGH_SIDE_STREAM=0 EFGH_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$tag -- python3 $ROOT/bench.py --no-cpu-baseline --no-forward-section --no-config-r --no-branch-section --steps 6 --warmup 2 $EXTRA --detail $ROOT/gpurun_out/ss_$tag.json > /dev/null 2> /tmp/st_$tag.err
  cp /tmp/st_$tag/*/*kernel_stats.csv $ROOT/gpurun_out/ss_${tag}_kernel_stats.csv
  tail -2 /tmp/st_$tag.err
done
