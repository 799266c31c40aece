"""per-stream picture of ONE steady-state training step in a rocprofv3 kernel trace of `bench.py --no-forward-section`: for every
stream its busy time, first start and last end relative to the step, and the kernels it runs in 20 slices of the step
    python tools/stream_timeline.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))


def nm(r):
    k = r['Kernel_Name']
    k = k.replace('(anonymous namespace)::', '').replace('void ', '')
    return k.split('(')[0][:28]


adam = [i for i, r in enumerate(rows) if nm(r).startswith('k_adam')]
# steps of the timed (multi-stream) region: bench runs warmup + steps there, then a single-stream pass; take a middle step
lo, hi = adam[len(adam) // 3], adam[len(adam) // 3 + 1]
step = rows[lo + 1:hi + 1]
t0, t1 = int(step[0]['Start_Timestamp']), int(step[-1]['End_Timestamp'])
print('step of %.1f ms, %d launches' % ((t1 - t0) / 1e6, len(step)))
streams = collections.OrderedDict()
for r in step:
    streams.setdefault(r['Stream_Id'], []).append(r)
NS = 24
for sid, rs in streams.items():
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs)
    print('\nstream %s: %d launches, busy %.1f ms, from %.1f to %.1f ms' % (
        sid, len(rs), busy / 1e6, (int(rs[0]['Start_Timestamp']) - t0) / 1e6, (max(int(r['End_Timestamp']) for r in rs) - t0) / 1e6))
    for s in range(NS):
        a, b = t0 + (t1 - t0) * s // NS, t0 + (t1 - t0) * (s + 1) // NS
        c = collections.Counter()
        for r in rs:
            o = min(int(r['End_Timestamp']), b) - max(int(r['Start_Timestamp']), a)
            if o > 0:
                c[nm(r)] += o
        tot = sum(c.values())
        top = ', '.join('%s %.1f' % (k, v / 1e6) for k, v in c.most_common(3))
        print('  %5.1f-%5.1f ms  busy %4.1f  %s' % ((a - t0) / 1e6, (b - t0) / 1e6, tot / 1e6, top))
