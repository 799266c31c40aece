#!/bin/bash
# rocprofv3 --stats of the training step on ONE stream (GPU box, through gpurun from the repo root) -> gpurun_out/ss_kernel_stats.csv
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
SCR=/tmp/efgh_ss_$$
mkdir -p "$SCR" "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
EFGH_SIDE_STREAM=0 EFGH_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $SCR/s1 -- python3 $ROOT/bench.py --no-cpu-baseline --no-forward-section --steps 10 --warmup 2 > $ROOT/gpurun_out/ss_bench.json 2> $SCR/s1.err
cp $SCR/s1/*/*kernel_stats.csv $ROOT/gpurun_out/ss_kernel_stats.csv
tail -n 2 $SCR/s1.err
rm -rf $SCR
