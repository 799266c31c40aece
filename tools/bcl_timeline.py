"""timeline of ONE pyramid build from a rocprofv3 kernel trace of tools/bench_bcl.py (tools/run_bcl_trace.sh): start, duration and the
gap to the previous kernel's end for every launch of the chain - the evidence that the build is not bound by launch gaps.
    python tools/bcl_timeline.py gpurun_out/bcl_kernel_trace.csv > profiles/r04_bcl_timeline.txt"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0] for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith('k_lat_keys')]
start = None
for j in range(len(idx) - 5, -1, -1):           # the last complete pyramid: five key kernels with shrinking grids
    g = [int(rows[idx[j + k]]['Grid_Size_X']) for k in range(5)]
    if g[0] > g[1] > g[2] > g[3] > g[4]:
        start = idx[j]
        break
assert start is not None, 'no complete pyramid in the trace'
t0 = int(rows[start]['Start_Timestamp'])
prev_end, level, tot = None, -1, 0.0
print('%-28s %9s %8s %7s %10s' % ('kernel', 'start us', 'dur us', 'gap us', 'grid'))
for i in range(start, len(rows)):
    if not names[i].startswith('k_lat_'):
        break
    r = rows[i]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if names[i].startswith('k_lat_keys'):
        level += 1
        print('--- level %d' % level)
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print('%-28s %9.1f %8.1f %7.1f %10s' % (names[i][:28], (s - t0) / 1e3, (e - s) / 1e3, gap, r['Grid_Size_X']))
    prev_end = e
    tot = (e - t0) / 1e3
print('pyramid: %.1f us from the first kernel\'s start to the last one\'s end' % tot)
