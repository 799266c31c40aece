"""sum rocprofv3 --pmc counters per kernel name: python tools/pmc_summary.py <dir> [substr]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ''
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70]
        if sub and sub not in k:
            continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (k, r['Dispatch_Id'])
        if key not in seen:
            seen.add(key); calls[k] += 1
for k, c in agg.items():
    print(k, 'dispatches', calls[k])
    for n, v in sorted(c.items()):
        print('   %-28s %.4g' % (n, v))
    if 'SQ_WAVE_CYCLES' in c:
        w = c['SQ_WAVE_CYCLES']
        for n in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS'):
            if n in c: print('   %s / WAVE_CYCLES = %.3f' % (n, c[n] / w))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'SQ_BUSY_CYCLES' in c:
        print('   MFMA busy / SQ busy = %.3f (x? normalisation: see guide)' % (c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES']))
