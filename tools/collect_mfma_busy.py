"""matrix-pipe utilisation per MFMA kernel family from one rocprofv3 PMC pass:
    python tools/collect_mfma_busy.py <pmc_dir> <out.json> "<command>"
busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)   (four SIMDs per CU; BUSY_CU_CYCLES summed over the CUs),
clock = SQ_BUSY_CU_CYCLES / 256 CUs / kernel time from the same trace."""
import collections, csv, glob, json, sys

FAMILIES = {'k_wino43': 'k_wino43<', 'k_wino_wgrad_rows': 'k_wino_wgrad_rows(',
            'k_gather_gemm_dma<1> (LDS-DMA staged implicit GEMM, round 5)': 'k_gather_gemm_dma<1,',
            'k_gather_gemm<1> (register-staged instances: N <= 32 and ragged channel counts)': 'k_gather_gemm<1,',
            'k_plane_gemm (batched planes of F(4x4,3x3), round 5)': 'k_plane_gemm<',
            'k_plane_wgrad (batched weight-gradient planes of F(4x4,3x3), round 5)': 'k_plane_wgrad<',
            'k_gather_wgrad<1>': 'k_gather_wgrad<1,'}
agg = {k: collections.defaultdict(float) for k in FAMILIES}
disp = {k: set() for k in FAMILIES}
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        for fam, pat in FAMILIES.items():
            if pat in r['Kernel_Name']:
                agg[fam][r['Counter_Name']] += float(r['Counter_Value'])
                disp[fam].add(r['Dispatch_Id'])
dur = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        for fam, pat in FAMILIES.items():
            if pat in r['Kernel_Name']:
                dur[fam] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
out = {'command': sys.argv[3], 'families': {}}
for fam, c in agg.items():
    if not c.get('SQ_BUSY_CU_CYCLES'):
        continue
    busy = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * c['SQ_BUSY_CU_CYCLES'])
    out['families'][fam] = {'dispatches': len(disp[fam]), 'SQ_VALU_MFMA_BUSY_CYCLES': c['SQ_VALU_MFMA_BUSY_CYCLES'],
                            'SQ_BUSY_CU_CYCLES': c['SQ_BUSY_CU_CYCLES'], 'mfma_busy_fraction': busy,
                            'kernel_seconds': dur[fam],
                            'clock_ghz': c['SQ_BUSY_CU_CYCLES'] / 256.0 / dur[fam] / 1e9 if dur[fam] else None,
                            'executed_tflops_from_counter': c['SQ_VALU_MFMA_BUSY_CYCLES'] / 64.0 * 4096.0 / dur[fam] / 1e12 if dur[fam] else None,
                            **{k: v for k, v in c.items() if k not in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CU_CYCLES')}}
json.dump(out, open(sys.argv[2], 'w'), indent=1)
for fam, v in out['families'].items():
    print('%-72s busy %.3f  clock %.2f GHz  %.1f TFLOP/s executed  (%d dispatches)' % (
        fam, v['mfma_busy_fraction'], v['clock_ghz'] or 0, v['executed_tflops_from_counter'] or 0, v['dispatches']))
