#!/bin/bash
# same-box A/B of the committed tree (ab_old/: `git archive HEAD` + its own build) against the working tree: bench lines alternately
# usage (GPU box): bash tools/ab_old_new.sh [rounds] [bench args...]
R=${1:-2}; shift
for i in $(seq 1 $R); do
  (cd ab_old && python bench.py --no-branch-section "$@" > ../gpurun_out/ab_old_$i.json 2> ../gpurun_out/ab_old_$i.err)
  python bench.py --no-branch-section "$@" > gpurun_out/ab_new_$i.json 2> gpurun_out/ab_new_$i.err
done
python - <<'PY'
import json, glob
for tag in ('old', 'new'):
    for f in sorted(glob.glob('gpurun_out/ab_%s_*.json' % tag)):
        try:
            d = json.load(open(f))
        except Exception as e:
            print(tag, f, 'unreadable', e); continue
        cr = d.get('config_r') or {}
        print('%s ms/step %.2f fwd %.2f | config_r train loop %.2f enq %.2f gpu %.2f | eval loop %.2f enq %.2f gpu %.2f' % (
            tag, d['ms_per_step'], d.get('forward_only', {}).get('value', 0.0),
            cr.get('train', {}).get('loop_ms', 0), cr.get('train', {}).get('enqueue_ms', 0), cr.get('train', {}).get('gpu_ms', 0),
            cr.get('eval', {}).get('loop_ms', 0), cr.get('eval', {}).get('enqueue_ms', 0), cr.get('eval', {}).get('gpu_ms', 0)))
PY
