"""print the kernels of the last complete pyramid build in a rocprofv3 kernel trace (csv): duration, start offset, name, grid"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def nm(r):
    k = r['Kernel_Name']
    return k.split('(anonymous namespace)::')[-1].split('(')[0][:40] if 'anonymous' in k else k[:40]
idx = [i for i, r in enumerate(rows) if nm(r).startswith('k_level_init')]
i0, i1 = idx[-10], idx[-5]
t0 = int(rows[i0]['Start_Timestamp'])
tot = {}
for i in range(i0, i1):
    r = rows[i]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot[nm(r)] = tot.get(nm(r), 0) + d
    print('%8.1f us @%8.1f  %-28s grid %s' % (d, (int(r['Start_Timestamp']) - t0) / 1e3, nm(r), r['Grid_Size_X']))
print('--- per kernel over the build:')
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print('%8.1f us  %s' % (v, k))
print('total %.1f us busy, span %.1f us' % (sum(tot.values()), (int(rows[i1 - 1]['End_Timestamp']) - t0) / 1e3))
