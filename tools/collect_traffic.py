"""HBM-side bytes per launch of the MFMA kernel families from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE):
    python tools/collect_traffic.py <fetch_dir> <write_dir> <out.json> "<workload description>"
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950."""
import collections, csv, glob, json, sys

FAMILIES = {'gemm': 'k_gather_gemm', 'wino': 'k_wino43', 'wgrad': 'k_gather_wgrad', 'wino_wgrad': 'k_wino_wgrad('}


def per_kernel(d, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            for fam, pat in FAMILIES.items():
                if pat in r['Kernel_Name']:
                    tot[fam] += float(r['Counter_Value']) * 1024.0
                    n[fam] += 1
    return tot, n


fetch, nf = per_kernel(sys.argv[1], 'FETCH_SIZE')
write, nw = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {'workload': sys.argv[4], 'note': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KB x 1024; '
       'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)', 'per_launch': {}}
for fam in FAMILIES:
    if nf[fam]:
        fb, wb = fetch[fam] / nf[fam], write[fam] / max(1, nw[fam])
        out['per_launch'][fam] = {'fetch_bytes_reported': fb, 'fetch_bytes_corrected_x2': 2 * fb, 'write_bytes': wb,
                                  'launches': nf[fam], 'traffic_bytes': 2 * fb + wb}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out['per_launch'], indent=1))
