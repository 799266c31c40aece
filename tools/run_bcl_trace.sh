#!/bin/bash
# per-kernel split of the isolated pyramid build (GPU box, through gpurun): rocprofv3 kernel trace of tools/bench_bcl.py
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
SCR=/tmp/efgh_bcl_$$
mkdir -p "$SCR" "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $SCR/kt -- python3 $ROOT/tools/bench_bcl.py --iters 20 > $ROOT/gpurun_out/bcl_bench.txt 2> $SCR/kt.err
cp $SCR/kt/*/*kernel_stats.csv $ROOT/gpurun_out/bcl_kernel_stats.csv
cp $SCR/kt/*/*kernel_trace.csv $ROOT/gpurun_out/bcl_kernel_trace.csv
rm -rf $SCR
