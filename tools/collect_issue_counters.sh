#!/bin/bash
# issue-side SQ counters of the MFMA kernels of one training step (GPU box, through gpurun from the repo root):
#   bash tools/collect_issue_counters.sh r06   -> gpurun_out/issue_counters_r06.{json,txt}
# --pmc passes with --kernel-trace only (no other trace domain), a handful of counters per pass.
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
SCR=/tmp/efgh_issue_$$
mkdir -p "$SCR" "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
CMD="$ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-forward-section --no-config-r --no-branch-section"
i=0
for set in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $SCR/p$i -- python3 $CMD > /dev/null 2> $SCR/p$i.err || tail -3 $SCR/p$i.err
done
python3 $ROOT/tools/collect_issue_counters.py $ROOT/gpurun_out/issue_counters_$TAG.json "rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-forward-section --no-config-r --no-branch-section (four passes)" $SCR/p1 $SCR/p2 $SCR/p3 $SCR/p4 | tee $ROOT/gpurun_out/issue_counters_$TAG.txt
tail -n 2 $SCR/*.err | tail -n 12
rm -rf $SCR
