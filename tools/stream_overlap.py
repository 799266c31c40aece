"""how much of a training step runs with 1 / 2 / 3 / 4 streams busy, and which kernels run alone?
    python tools/stream_overlap.py <rocprofv3 kernel_trace.csv of `bench.py --no-forward-section`>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows)
rows = rows[n // 4: n * 5 // 8]                            # steady-state steps of the timed region (bench.py --steps 6 --warmup 2: 12 steps in the trace, the last 4 are the single-stream pass)
ev = []
for r in rows:
    ev.append((int(r['Start_Timestamp']), 1, r))
    ev.append((int(r['End_Timestamp']), -1, r))
ev.sort(key=lambda e: (e[0], e[1]))


def nm(r):
    k = r['Kernel_Name']
    return (k.split('(anonymous namespace)::')[-1] if 'anonymous' in k else k).split('(')[0][:34]


active, busy, alone, running = collections.Counter(), collections.Counter(), collections.Counter(), {}
last = ev[0][0]
for t, d, r in ev:
    dt = t - last
    if dt > 0:
        ns = len([q for q, c in active.items() if c > 0])
        busy[ns] += dt
        if ns == 1:
            for rr in running.values():
                alone[nm(rr)] += dt
    last = t
    active[r['Stream_Id']] += d
    if d > 0:
        running[id(r)] = r
    else:
        running.pop(id(r), None)
tot = sum(busy.values())
print('streams busy -> share of the time:', {k: round(v / tot, 3) for k, v in sorted(busy.items())}, ' window %.0f ms' % (tot / 1e6))
print('kernels running with no other stream busy (ms in this window):')
for k, v in alone.most_common(16):
    print('   %8.2f  %s' % (v / 1e6, k))
