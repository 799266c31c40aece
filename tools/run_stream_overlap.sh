#!/bin/bash
# kernel trace of the timed multi-stream training step + tools/stream_overlap.py on it (GPU box, through gpurun from the repo root)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
SCR=/tmp/efgh_overlap_$$
mkdir -p "$SCR" "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $SCR/kt -- python3 $ROOT/bench.py --no-cpu-baseline --no-forward-section --steps 6 --warmup 2 > /dev/null 2> $SCR/kt.err
python3 $ROOT/tools/stream_overlap.py $SCR/kt/*/*kernel_trace.csv > $ROOT/gpurun_out/stream_overlap.txt 2>&1
python3 $ROOT/tools/stream_timeline.py $SCR/kt/*/*kernel_trace.csv > $ROOT/gpurun_out/stream_timeline.txt 2>&1
tail -n 3 $SCR/kt.err
rm -rf $SCR
