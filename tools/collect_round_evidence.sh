#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline fields refer to, on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_round_evidence.sh r04          -> gpurun_out/evidence_r04/{*.json,*.csv}
# Separate passes, as the MI355X guide prescribes: --kernel-trace --stats alone; --pmc passes with --kernel-trace only.
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/evidence_$TAG
SCR=/tmp/efgh_evidence_$$
mkdir -p "$OUT" "$SCR"
cd /tmp && export TMPDIR=/tmp
TRAIN="$ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-forward-section --no-config-r --no-branch-section"
FWD="$ROOT/bench.py --mode fwd --steps 1 --warmup 1 --no-cpu-baseline --no-config-r --no-branch-section"
# the default bench line FIRST, on the box as it comes (two minutes of profiler passes leave it ~5 % slower: measured in round 5)
cd $ROOT && python3 bench.py --detail $OUT/bench_default.json > $OUT/bench_default_line.json 2> $OUT/bench_default.err
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $SCR/tf -- python3 $TRAIN --detail $SCR/tf.json > /dev/null 2> $SCR/tf.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $SCR/tw -- python3 $TRAIN > /dev/null 2> $SCR/tw.err
python3 $ROOT/tools/collect_traffic.py $SCR/tf $SCR/tw $OUT/hbm_traffic_train.json "bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-forward-section (training step, batch 8)" $SCR/tf.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $SCR/ff -- python3 $FWD --detail $SCR/ff.json > /dev/null 2> $SCR/ff.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $SCR/fw -- python3 $FWD > /dev/null 2> $SCR/fw.err
python3 $ROOT/tools/collect_traffic.py $SCR/ff $SCR/fw $OUT/hbm_traffic_fwd.json "bench.py --mode fwd --steps 1 --warmup 1 --no-cpu-baseline (eval forward, batch 4; exact + two fast-math passes)" $SCR/ff.json
CMD="rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-forward-section --no-config-r --no-branch-section"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $SCR/mb -- python3 $TRAIN > /dev/null 2> $SCR/mb.err
python3 $ROOT/tools/collect_mfma_busy.py $SCR/mb $OUT/mfma_busy_train.json "$CMD"
# 30 timed steps: the one-off launches of model construction (637 parameter uploads) stop weighing on the per-step launch census
rocprofv3 --kernel-trace --stats --output-format csv -d $SCR/st -- python3 $ROOT/bench.py --no-cpu-baseline --no-forward-section --no-config-r --no-branch-section --steps 30 --warmup 2 --detail $OUT/bench_train_under_rocprof.json > /dev/null 2> $SCR/st.err
cp $SCR/st/*/*kernel_stats.csv $OUT/train_kernel_stats.csv
# the same on ONE stream: the per-kernel durations the roofline objects of bench.py are computed from
EFGH_SIDE_STREAM=0 EFGH_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $SCR/s1 -- python3 $ROOT/bench.py --no-cpu-baseline --no-forward-section --no-config-r --no-branch-section --steps 10 --warmup 2 --detail $OUT/bench_train_single_stream_under_rocprof.json > /dev/null 2> $SCR/s1.err
cp $SCR/s1/*/*kernel_stats.csv $OUT/train_kernel_stats_single_stream.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $SCR/sf -- python3 $ROOT/bench.py --mode fwd --no-cpu-baseline --no-config-r --no-branch-section --detail $OUT/bench_fwd_under_rocprof.json > /dev/null 2> $SCR/sf.err
cp $SCR/sf/*/*kernel_stats.csv $OUT/fwd_kernel_stats.csv
# the reference's own configuration (batch 1, stock Adam loop): 20 training iterations, per-iteration launch census
rocprofv3 --kernel-trace --stats --output-format csv -d $SCR/cr -- python3 $ROOT/tools/config_r_loop.py 20 > $OUT/config_r_loop.txt 2> $SCR/cr.err
cp $SCR/cr/*/*kernel_stats.csv $OUT/config_r_train_kernel_stats.csv
ls -la $OUT
tail -n 2 $SCR/*.err | tail -n 30
rm -rf $SCR
