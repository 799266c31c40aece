"""issue-side SQ counters per MFMA kernel family, folded from several rocprofv3 --pmc passes of the same command (one directory per pass):
    python tools/collect_issue_counters.py <out.json> "<command>" <pmc_dir> [<pmc_dir> ...]
Every pass also carries SQ_WAVE_CYCLES (and the kernel trace): counters of different passes are compared per WAVE CYCLE of their own pass."""
import collections, csv, glob, json, sys

FAMILIES = {'k_wino43': 'k_wino43<', 'k_wino_wgrad_rows': 'k_wino_wgrad_rows(', 'k_gather_gemm_dma<1>': 'k_gather_gemm_dma<1,',
            'k_plane_gemm': 'k_plane_gemm<', 'k_plane_wgrad': 'k_plane_wgrad<', 'k_gather_wgrad<1>': 'k_gather_wgrad<1,'}
out = {'command': sys.argv[2], 'note': 'per family: counter / SQ_WAVE_CYCLES of the same pass (`per_wave_cycle`) and raw sums; launches of one '
                                       'training step (bench.py --steps 1 --warmup 1 => 2 steps + the single-stream pass)', 'families': {}}
per = {k: {} for k in FAMILIES}
raw = {k: {} for k in FAMILIES}
for d in sys.argv[3:]:
    agg = {k: collections.defaultdict(float) for k in FAMILIES}
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            for fam, pat in FAMILIES.items():
                if pat in r['Kernel_Name']:
                    agg[fam][r['Counter_Name']] += float(r['Counter_Value'])
    for fam, c in agg.items():
        wc = c.get('SQ_WAVE_CYCLES')
        for k, v in c.items():
            raw[fam][k] = v
            if wc and k != 'SQ_WAVE_CYCLES':
                per[fam][k] = v / wc
for fam in FAMILIES:
    if raw[fam]:
        out['families'][fam] = {'per_wave_cycle': {k: round(v, 5) for k, v in sorted(per[fam].items())}, 'raw': raw[fam]}
json.dump(out, open(sys.argv[1], 'w'), indent=1)
names = sorted({k for f in per.values() for k in f})
print('%-28s' % 'counter / wave cycle' + ''.join('%22s' % f[:21] for f in out['families']))
for n in names:
    print('%-28s' % n + ''.join('%22.4f' % per[f].get(n, float('nan')) for f in out['families']))
