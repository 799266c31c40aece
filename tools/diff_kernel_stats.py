"""diff of two rocprofv3 --stats kernel CSVs (A/B runs): python tools/diff_kernel_stats.py a.csv b.csv [steps]"""
import csv
import re
import sys


def load(f):
    d = {}
    for r in csv.DictReader(open(f)):
        n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        n = re.sub(r'\(.*', '', n)
        c, t = d.get(n, (0, 0.0))
        d[n] = (c + int(r['Calls']), t + float(r['TotalDurationNs']) / 1e6)
    return d


a, b = load(sys.argv[1]), load(sys.argv[2])
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
ta, tb = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
print('total ms per step: A %.2f  B %.2f   launches per step: A %.0f  B %.0f' % (ta / steps, tb / steps, sum(v[0] for v in a.values()) / steps,
                                                                              sum(v[0] for v in b.values()) / steps))
keys = sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, (0, 0))[1] - b.get(k, (0, 0))[1]))
for k in keys[:30]:
    x, y = a.get(k, (0, 0)), b.get(k, (0, 0))
    print('%-70s A %6.0f %8.2f   B %6.0f %8.2f   d %+8.2f' % (k[:70], x[0] / steps, x[1] / steps, y[0] / steps, y[1] / steps, (x[1] - y[1]) / steps))
