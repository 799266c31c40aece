"""EFGHNet hot-path benchmark on MI355X (contract: task statement / DESIGN.md §5).

    python bench.py --gpus N --steps K --warmup W [--mode train|fwd] [--batch B]

One process per GPU.  The driver launches N>1 through torch.distributed.run; run bare (`python bench.py --gpus N`, no
WORLD_SIZE in the environment) the parent starts that launcher itself as a child process BEFORE anything touches the GPU,
and rank 0 of the children prints the JSON line.

* default `--mode train` = BASELINE.json's metric: frame-pairs/s of EFGHNet fwd+bwd (efghloss, gradient
  all-reduce, fused Adam) on synthetic 384x1280 RGB + 64x2048-point sweeps, batch 8 per GPU
  (configs[2]); a step = one training step on one batch, inputs resident in HBM.
* the same run also times the forward-only workload (configs[1], eval, batch 4) and reports it under
  "forward_only" (the north star's 40 frame-pairs/s/GPU target is stated on it).
Rank 0 prints ONE compact JSON line (< 8 KB, `compact_line`) as the only stdout line; the full object with every per-family
figure goes to bench_detail.json.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RAW = (768, 2560)          # raw camera size -> img 3x384x1280, range image 4x384x5120, depth 4x768x2560
NPTS = 64 * 2048
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s measured streaming)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--mode', default='train', choices=['train', 'fwd'])
    ap.add_argument('--batch', type=int, default=None, help='frame-pairs per GPU per step (8 train / 4 fwd)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-forward-section', action='store_true')
    ap.add_argument('--small', action='store_true', help='debug: 128x256 / 2048 points')
    ap.add_argument('--no-config-r', action='store_true', help='skip the batch-1 reference-loop section (config_r)')
    ap.add_argument('--no-branch-section', action='store_true', help='skip roofline_resnet_branch (profiler passes: its G-only spans would count as launches of the steps)')
    ap.add_argument('--rotate-inputs', type=int, default=4,
                    help='cycle this many resident batches of DIFFERENT synthetic frame-pairs through the steps (1 = the same batch '
                         'every step): every sweep has its own lattice sizes, so the speculative sizing of the pyramid (previous '
                         'sizes + 25 %%) and its fallback are part of what is timed, as in a real training loop (iterater.py:26-43)')
    ap.add_argument('--detail', default=None, metavar='PATH',
                    help='where the FULL result object goes (default: bench_detail.json next to bench.py, and gpurun_out/ when it '
                         'exists); stdout carries only the compact line')
    ap.add_argument('--set', action='append', default=[], metavar='MODULE.ATTR=VALUE',
                    help='builder A/B runs: set a switch of efgh_amd (e.g. ops.PLANE_DMA=0) before anything is built; the line '
                         'records it under "switches" (a default run has none)')
    return ap.parse_args()


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(raw, npts, mode, timed=3):
    """the oracle (CPU restatement of the reference) timed on the host cores on a BOUNDED sample: ONE frame-pair of the
    same workload (same sizes, same mode) per iteration, 1 warm-up + `timed` timed iterations."""
    import torch
    from efgh_amd import synthetic as syn
    from efgh_amd.nets import EFGHBackbone
    from oracle import efgh_oracle as O
    host = os.cpu_count() or 1
    cores = min(host, 32)        # torch CPU ops stop scaling (and regress badly) far below 256 threads: 256 threads took 412 s
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = EFGHBackbone(syn.default_args(raw, 'cpu'))      # parameter container only; never executed on CPU
    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    names = [k for k, _ in m.named_parameters()]
    b = syn.make_batch(raw, npts, 1)
    T = torch.from_numpy
    args = syn.default_args(raw, 'cpu')
    inp = [T(b[k]) for k in ('pc', 'img', 'calib', 'A')]

    def one():
        if mode == 'train':
            for k in names:
                P[k].requires_grad_(True)
                P[k].grad = None
            pred = O.forward(P, *inp, args, train=True)
            L, _ = O.compute_loss(inp[0], {k: T(v) for k, v in b['gt'].items()}, pred, args)
            L['total'].backward()
        else:
            with torch.no_grad():
                O.forward(P, *inp, args, train=False)
    what = 'train fwd+loss+bwd' if mode == 'train' else 'eval forward'
    t0 = time.time()
    one()                                               # warm-up (thread pool, oneDNN primitives, page faults)
    warm = time.time() - t0
    ts = []
    for _ in range(timed):
        t0 = time.time()
        one()
        ts.append(time.time() - t0)
    dt = sum(ts) / len(ts)
    # "report the lattice build time separately (C CPU vs HIP)": the five-level pyramid of the same frame
    # (nets/transforms.py:125-184, nets/generate_data.py:117-193) through oracle/lattice_oracle.c on ONE host thread (a scalar port)
    # and through efgh_amd.lattice on the GPU (launches + the one read-back of the level sizes, median of 5 after a warm-up)
    lat = None
    try:
        from efgh_amd import lattice as hip_lattice
        from oracle import lattice as c_lattice
        scales = [s_ for s_, _ in args['scale_map']]
        pc1 = b['pc'][0]
        c_lattice.generate_data(pc1, scales)
        tl = []
        for _ in range(3):
            t0 = time.time()
            c_lattice.generate_data(pc1, scales)
            tl.append(time.time() - t0)
        lat = {'cpu_c_port_ms': round(1e3 * sorted(tl)[1], 2), 'cpu_threads': 1}
        if torch.cuda.is_available():
            pcd = torch.from_numpy(b['pc']).cuda()
            th = []
            for _ in range(6):
                torch.cuda.synchronize()
                t0 = time.time()
                hip_lattice.build_pyramid_batched(pcd, scales)
                torch.cuda.synchronize()
                th.append(time.time() - t0)
            lat['hip_ms'] = round(1e3 * sorted(th[1:])[2], 3)
            lat['note'] = ('one frame (%d points), five levels; hip_ms = wall clock of build_pyramid_batched at batch 1 incl. its one host '
                           'read-back; the same arrays bit for bit (tests/test_gpu_lattice.py)' % npts)
    except Exception as e:          # noqa: BLE001
        lat = {'error': repr(e)}
    return {'value': 1.0 / dt, 'unit': 'frame-pairs/s', 'cores': cores, 'kind': 'port', 'lattice_build': lat,
            'host_cpus': host, 'cpu_model': cpu_model(), 'iterations_s': [round(t, 2) for t in ts], 'warmup_s': round(warm, 2),
            'sample': '%d timed iterations after 1 warm-up, each ONE frame-pair of the same workload (%dx%d RGB, %d points; %s, '
                      'B=1) through oracle/efgh_oracle.py + oracle/lattice_oracle.c, torch CPU fp32, %d of %d host threads, '
                      'mean %.1f s' % (timed, raw[0] // 2, raw[1] // 2, npts, what, cores, host, dt)}


def dump_shapes(prof, path):
    """per-shape launch table (debug aid: EFGH_BENCH_SHAPES=file)"""
    agg = {}
    for e0, e1, f, key in prof:
        a = agg.setdefault(key, [0, 0.0, 0.0])
        a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += f
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(path, 'a') as fh:
        fh.write('mode M N T C | calls ms TFLOP/s\n')
        for key, (n, ms, f) in rows[:40]:
            fh.write('%s | %d %.2f %.1f\n' % (' '.join(map(str, key)), n, ms, f / (ms * 1e-3) / 1e12 if ms > 0 else 0))


def committed_traffic(family, workload='train', launches_per_step=None, kernel=None):
    """HBM-side bytes per launch of one kernel family (`bcl`: per step) from the committed rocprofv3 PMC passes (FETCH_SIZE doubled
    as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE; tools/collect_traffic.py), collected offline on the same bench
    command, see profiles/README.md.  PMC collection cannot run inside the timed region, so the figure is only as good as the
    committed file is current: the file's sha256 goes into the line (`traffic_source`), and when the number of launches per step
    the file was folded from differs from what THIS run launched the figure is withheld (None) and the mismatch reported - a
    traffic number of another kernel mix must not look measured.  -> (bytes | None, error | None)"""
    try:
        raw = open(os.path.join(ROOT, 'profiles', TRAFFIC_FILE[workload]), 'rb').read()
        doc = json.loads(raw)
        ent = doc['per_launch'][family]
    except Exception as e:          # noqa: BLE001
        return None, 'no committed PMC fold for this family (%s)' % type(e).__name__
    census = doc.get('bench_launches_per_step')          # bench.py's own launch census of the PMC run (tools/collect_traffic.py)
    if census is None:
        return ent['traffic_bytes'], 'unchecked: the committed fold carries no launch census'
    if launches_per_step is not None:
        theirs = [v[1] for v in census.values() if kernel and v[0] == kernel[:24]]
        if not theirs or abs(theirs[0] - launches_per_step) > 1e-6:
            msg = ('STALE: profiles/%s was folded from a run with %s launches per step of this family, this run made %s - re-run '
                   'tools/collect_round_evidence.sh' % (TRAFFIC_FILE[workload], theirs[0] if theirs else 'no', launches_per_step))
            sys.stderr.write('bench.py: traffic[%s] withheld: %s\n' % (family, msg))
            return None, msg
    return ent['traffic_bytes'], None


def traffic_source(workload):
    import hashlib
    try:
        raw = open(os.path.join(ROOT, 'profiles', TRAFFIC_FILE[workload]), 'rb').read()
        return {'file': 'profiles/' + TRAFFIC_FILE[workload], 'sha256_16': hashlib.sha256(raw).hexdigest()[:16],
                'steps_in_trace': json.loads(raw).get('steps_in_trace')}
    except Exception:               # noqa: BLE001
        return None


# the PMC passes the `traffic` fields are read from (tools/collect_round_evidence.sh writes them, profiles/README.md)
TRAFFIC_FILE = {'train': 'r06_hbm_traffic_train.json', 'fwd': 'r06_hbm_traffic_fwd.json'}


def mfma_step_utilisation(prof, steps, ms_per_step):
    """the roofline of the TIMED schedule as a whole: MFMA FLOP executed per step by every contraction family (Winograd kernels
    counted at the products they execute, half / a quarter of the direct form) / wall time of a step / dense fp32 MFMA peak"""
    fl = 0.0
    for name, lst in prof.items():
        if name in ('bcl', 'wino2d', 'hbm_convs', 'dedicated') or not lst:   # 'wino2d' = whole layers in direct-form FLOP: its GEMM launches are 'wino2d_gemm'
            continue
        f = sum(p[2] for p in lst)
        fl += f / 2 if name in ('wino', 'wino_wgrad') else f
    per_step = fl / max(1, steps)
    tf = per_step / (ms_per_step * 1e-3) / 1e12 if ms_per_step > 0 else 0.0
    return {'executed_mfma_tflop_per_step': per_step / 1e12, 'achieved': tf, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': tf / PEAK_F32_MFMA_TFLOPS,
            'note': 'timed region (multi-stream schedule): sum of executed MFMA FLOP of all contraction kernels / ms_per_step'}


def gemm_roofline(prof, steps, kernel):
    if os.environ.get('EFGH_BENCH_SHAPES'):
        dump_shapes(prof, os.environ['EFGH_BENCH_SHAPES'])
    ms = sum(p[0].elapsed_time(p[1]) for p in prof)
    fl = sum(p[2] for p in prof)
    n = len(prof)
    ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    return {'bound': 'mfma', 'kernel': kernel, 'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': ach / PEAK_F32_MFMA_TFLOPS, 'traffic': None,
            'traffic_unit': 'HBM-side bytes per launch (committed rocprofv3 PMC passes of this bench command, profiles/r06_hbm_traffic_*.json)',
            'launches_per_step': n / max(1, steps),
            'avg_launch_ms': ms / max(1, n), 'algorithmic_gflop_per_launch': fl / max(1, n) / 1e9,
            'kernel_ms_per_step': ms / max(1, steps)}


KERNELS = {
    'wino2d_gemm': 'k_plane_gemm / k_plane_wgrad (LDS-DMA staged), batched over the 36 planes of Winograd F(4x4,3x3): the 3x3 / stride-1 '
                   'convolutions with >= 256 channels, their data gradients and (>= 128 channels) weight gradients',
    'gemm': 'k_gather_gemm / k_gather_gemm_dma (fp32 MFMA implicit GEMM: every contraction that is not a "same" 3x3 convolution)',
    'wino': 'k_wino43 (Winograd F(4,3) on fp32 MFMA: the 3x3 / stride-1 convolutions and their data gradients)',
    'wgrad': 'k_gather_wgrad (fp32 MFMA weight gradient)',
    'wino_wgrad': 'k_wino_wgrad_rows (Winograd F(3,4) weight gradient of the 3x3 / stride-1 convolutions, fp32 MFMA)',
}


def rooflines(prof, steps, workload='train'):
    """`roofline` = the MFMA kernel with the most time in the timed region; the other kernels as roofline_<name>.
    `achieved` / `frac` count the MFMA FLOPs the kernel EXECUTES (so frac <= 1): for the Winograd kernels that is half of the
    direct-form count 2*M*N*9*C, which is kept as `algorithmic_tflops` (the rate a direct kernel would need for the same time)."""
    rl = {}
    bcl = prof.get('bcl')
    w2 = prof.get('wino2d')
    hb = prof.get('hbm_convs')
    for name, lst in prof.items():
        if name in ('bcl', 'wino2d', 'hbm_convs', 'dedicated'):
            continue
        if lst:
            r = gemm_roofline(lst, steps, KERNELS[name])
            r['traffic'], err = committed_traffic(name, workload, r['launches_per_step'], r['kernel'])
            if err:
                r['traffic_error'] = err
            r['algorithmic_tflops'] = r['achieved']
            if name in ('wino', 'wino_wgrad'):
                r['achieved'] = r['achieved'] / 2
                r['frac'] = r['frac'] / 2
                r['executed_gflop_per_launch'] = r['algorithmic_gflop_per_launch'] / 2
            rl[name] = r
    if not rl:
        return {}
    top = max(rl, key=lambda k: rl[k]['kernel_ms_per_step'])
    # `roofline` = the dominant KERNEL.  The 2-D Winograd family is two kernels (k_gather_gemm<0> for forward / data gradient,
    # k_gather_wgrad<0> for the weight gradient) whose summed time is within a few per cent of k_wino43's: without this the
    # object flips between the two families from run to run
    if 'wino' in rl and rl['wino']['kernel_ms_per_step'] >= 0.7 * rl[top]['kernel_ms_per_step']:
        top = 'wino'
    out = {'roofline': rl[top]}
    if bcl:     # the HBM-bound side of the path: lattice build + BCL splat, SURVEY 8d algorithmic bytes / event time
        ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in bcl)
        by = sum(b for _, _, b, _ in bcl)
        ach = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        parts = {}
        for e0, e1, b, what in bcl:
            a = parts.setdefault(what, [0.0, 0.0])
            a[0] += e0.elapsed_time(e1); a[1] += b
        out['roofline_bcl'] = {'bound': 'hbm', 'kernel': 'BCL index + splat pipeline, all five levels and all samples of the batch: '
                               'lattice build (keys, tile-local bucket sort, per-bucket grouping in LDS, first-seen numbering, '
                               'neighbours, vertex lists) and splat gather'
                               + (' + splat adjoint' if any(k.endswith('bwd') for k in parts) else ''),
                               'achieved': ach, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': ach / PEAK_HBM_GBS,
                               'traffic': None,
                               'traffic_unit': 'HBM-side bytes per STEP, all lattice.hip + bcl.hip launches (profiles/r06_hbm_traffic_*.json)',
                               'launches_per_step': len(bcl) / max(1, steps), 'kernel_ms_per_step': ms / max(1, steps),
                               'algorithmic_mb_per_step': by / max(1, steps) / 1e6,
                               'parts': {k: {'ms_per_step': v[0] / max(1, steps), 'algorithmic_mb_per_step': v[1] / max(1, steps) / 1e6,
                                             'gbs': (v[1] / (v[0] * 1e-3) / 1e9 if v[0] > 0 else 0.0)} for k, v in parts.items()}}
    if bcl:
        rb = out['roofline_bcl']
        rb['traffic'], err = committed_traffic('bcl', workload, rb['launches_per_step'], rb['kernel'])
        if err:
            rb['traffic_error'] = err
    for k, v in rl.items():
        if k != top:
            out['roofline_' + k] = v
    out['traffic_source'] = traffic_source(workload)
    if hb and os.environ.get('EFGH_BENCH_SHAPES'):
        agg = {}
        for e0, e1, by_, key in hb:
            a_ = agg.setdefault(key, [0, 0.0, 0.0])
            a_[0] += 1; a_[1] += e0.elapsed_time(e1); a_[2] += by_
        with open(os.environ['EFGH_BENCH_SHAPES'], 'a') as fh:
            fh.write('hbm_convs: mode M N T C | calls ms GB/s\n')
            for key, (n_, ms_, by_) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
                fh.write('%s | %d %.2f %.0f\n' % (' '.join(map(str, key)), n_, ms_, by_ / (ms_ * 1e-3) / 1e9 if ms_ > 0 else 0))
    if hb:      # contractions below the ridge of the two rooflines (1x1 layers, narrow heads, thin kernels): bytes, not FLOP
        ms = sum(p[0].elapsed_time(p[1]) for p in hb)
        by = sum(p[2] for p in hb)
        ach = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        ded = prof.get('dedicated') or []
        ded_by, ded_ms = sum(p[2] for p in ded), sum(p[0].elapsed_time(p[1]) for p in ded)
        out['roofline_hbm_convs'] = {'bound': 'hbm', 'kernel': 'contractions with < %g FLOP per algorithmic byte (1x1 and 4-channel layers, narrow '
                                     'heads, the point branch\'s 32-channel layers; forward, data and weight gradients): thin / small-channel '
                                     'kernels and the generic tile' % 30.0,
                                     'achieved': ach, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': ach / PEAK_HBM_GBS, 'traffic': None,
                                     'launches_per_step': len(hb) / max(1, steps), 'kernel_ms_per_step': ms / max(1, steps),
                                     'algorithmic_mb_per_step': by / max(1, steps) / 1e6}
        # counter traffic exists for the launches the DEDICATED kernels served (thin / 4-channel / small-channel: a PMC fold goes by
        # kernel name, and the generic tile's HBM-bound launches share their name with its MFMA-bound ones)
        rh = out['roofline_hbm_convs']
        rh['dedicated'] = {'launches_per_step': len(ded) / max(1, steps), 'kernel_ms_per_step': ded_ms / max(1, steps),
                           'algorithmic_mb_per_step': ded_by / max(1, steps) / 1e6,
                           'achieved': ded_by / (ded_ms * 1e-3) / 1e9 if ded_ms > 0 else 0.0,
                           'note': 'EVERY launch served by a thin.hip / c4conv.hip / smallc.hip kernel (also the 16- / 32-channel 3x3 layers, which sit at the ridge and are listed under roofline_gemm); `traffic` covers exactly these kernels'}
        rh['traffic_unit'] = 'HBM-side bytes per STEP of the dedicated kernels (rocprofv3 PMC fold, profiles/' + TRAFFIC_FILE[workload] + ')'
        try:
            ent = json.loads(open(os.path.join(ROOT, 'profiles', TRAFFIC_FILE[workload])).read())['per_launch']['hbm_convs']
            if abs(ent.get('launches_per_step', -1) - rh['dedicated']['launches_per_step']) < 1e-6:
                rh['traffic'] = ent['traffic_bytes']
                rh['traffic_over_algorithmic'] = ent['traffic_bytes'] / max(1.0, ded_by / max(1, steps))
            else:
                rh['traffic_error'] = 'STALE: the committed fold saw %s dedicated launches per step, this run made %s' % (
                    ent.get('launches_per_step'), rh['dedicated']['launches_per_step'])
        except Exception as e:          # noqa: BLE001
            rh['traffic_error'] = 'no committed PMC fold for this family (%s)' % type(e).__name__
    if w2:      # whole F(4x4,3x3) layers (input transform + batched GEMM + output transform), direct-form FLOPs
        ms = sum(p[0].elapsed_time(p[1]) for p in w2)
        fl = sum(p[2] for p in w2)
        out['winograd2d_layers'] = {'launch_groups_per_step': len(w2) / max(1, steps), 'ms_per_step': ms / max(1, steps),
                                    'algorithmic_tflops': fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                                    'note': 'three launches per layer; the GEMM launch alone is in roofline*["wino2d_gemm"]'}
    return out


def config_r(iters=10, warm=3):
    """BASELINE configs[0], the reference's OWN configuration and loop shape (configs/train_rellis.yaml:19-29: raw 900x1600, 65 536
    points, batch 1; iterater.py:25-46,106): per iteration `.to(DEVICE).float()` of the four inputs from host memory, `model(...)`,
    `criterion.compute_loss`, `optimizer.zero_grad / backward / step` with STOCK torch.optim.Adam (main.py:181-183), `lss.update`
    = one `.item()` per loss term (helper.py:123-126), `err.update` = two `.cpu()` pose reads (helper.py:142-145), and
    `torch.cuda.empty_cache()`.  Timed three ways per iteration: host wall-clock of the loop body, host time until everything up to
    `optimizer.step()` is ENQUEUED (the forward's one lattice read-back included), and the GPU interval between two events around
    the same span.  The loop runs twice: with the real `empty_cache()` as the reference calls it, and with the no-op that
    `python -m efgh_amd.run` rebinds it to (efgh_amd/run.py:install_loop_rebinds); eval forward the same way (valid.py:21-41)."""
    import time
    import numpy as np
    import torch
    from efgh_amd import synthetic as syn
    from efgh_amd.losses import EFGHCriterion
    from efgh_amd.nets import EFGHBackbone
    raw, npts = (900, 1600), 65536
    args = syn.default_args(raw, 'cuda')
    torch.manual_seed(0)
    model = EFGHBackbone(args).cuda()
    model = torch.nn.DataParallel(model, device_ids=[0])            # main.py:127 on a one-GPU process: calls the module directly
    criterion = EFGHCriterion(args)
    optimizer = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=0)
    pairs = [syn.make_batch(raw, npts, 1, first_seed=i) for i in range(4)]          # four different frame-pairs, cycled
    host = [([torch.from_numpy(b[k]).pin_memory() for k in ('pc', 'img', 'calib', 'A')],
             {k: torch.from_numpy(v) for k, v in b['gt'].items()}) for b in pairs]
    real_empty = torch.cuda.empty_cache

    def train_iter(i, empty):
        (pcd, img, calib, A), gt = host[i % len(host)]
        t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pcd, img, calib, A = (t.to('cuda').float() for t in (pcd, img, calib, A))
        e0.record()
        pred = model(pcd, img, calib, A, False)
        losses, gt2 = criterion.compute_loss(pcd, img, calib, A, dict(gt), pred)
        optimizer.zero_grad()
        losses['total'].backward()
        optimizer.step()
        e1.record()
        t1 = time.perf_counter()
        vals = [losses[k].item() for k in list(losses.keys())]                       # Lss.update
        _ = gt2['sensor2_T_sensor1'].cpu().detach().numpy()[0], pred['sensor2_T_sensor1'].cpu().detach().numpy()[0]      # Err.update
        del pcd, img, gt2
        empty()
        t2 = time.perf_counter()
        return (t2 - t0) * 1e3, (t1 - t0) * 1e3, e0.elapsed_time(e1), len(vals)

    def eval_iter(i, empty):
        (pcd, img, calib, A), gt = host[i % len(host)]
        t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pcd, img, calib, A = (t.to('cuda').float() for t in (pcd, img, calib, A))
        e0.record()
        with torch.no_grad():
            pred = model(pcd, img, calib, A)
        e1.record()
        t1 = time.perf_counter()
        _ = pred['sensor2_T_sensor1'].cpu().numpy()[0]                               # test.py:46-53: the 3x4 pose goes to the CSV
        empty()
        t2 = time.perf_counter()
        return (t2 - t0) * 1e3, (t1 - t0) * 1e3, e0.elapsed_time(e1), 0

    def run(fn, empty):
        for i in range(warm):
            fn(i, empty)
        torch.cuda.synchronize()
        rows = [fn(warm + i, empty) for i in range(iters)]
        torch.cuda.synchronize()
        med = lambda j: float(np.median([r[j] for r in rows]))
        return {'loop_ms': med(0), 'enqueue_ms': med(1), 'gpu_ms': med(2)}
    out = {'workload': 'BASELINE.json configs[0]: the shipped RELLIS configuration (raw 900x1600, 65 536 points), batch 1, the '
                       'reference\'s loop shape (iterater.py:25-46,106) with stock torch.optim.Adam; medians of %d iterations over 4 '
                       'rotating synthetic frame-pairs' % iters}
    model.train()
    out['train_reference_loop'] = run(train_iter, real_empty)
    out['train'] = run(train_iter, lambda: None)
    losses_n = train_iter(0, lambda: None)[3]
    model.eval()
    out['eval_reference_loop'] = run(eval_iter, real_empty)
    out['eval'] = run(eval_iter, lambda: None)
    out['train_ms'], out['fwd_ms'] = out['train']['loop_ms'], out['eval']['loop_ms']
    out['host_syncs_per_train_iteration'] = losses_n + 2 + 1
    out['note'] = ('loop_ms: host wall-clock of one loop body (what the user waits for); enqueue_ms: host time until optimizer.step() '
                   '(eval: the forward) has been enqueued; gpu_ms: GPU interval between events around the same span - enqueue_ms '
                   '>= gpu_ms means the host, not the GPU, sets the pace.  *_reference_loop: torch.cuda.empty_cache() really called '
                   'every iteration (iterater.py:106); train / eval: the no-op `python -m efgh_amd.run` rebinds it to')
    del model, optimizer
    real_empty()
    return out


def _r(x, n=4):
    return round(x, n) if isinstance(x, float) else x


def _rl(o, extra=()):
    """the contract's roofline object (bound, kernel name, achieved, peak, unit, frac, traffic) + a few small figures"""
    if not isinstance(o, dict):
        return None
    keys = ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic') + tuple(extra)
    r = {k: _r(o.get(k)) for k in keys if k in o or k == 'traffic'}
    r['kernel'] = str(o.get('kernel', ''))[:64]
    return r


def compact_line(out):
    """ONE short JSON object (< 8 KB; in practice ~3 KB) for the driver: the contract keys + the roofline / cpu_baseline objects
    + the fractions of every kernel family.  The full object (20 KB) goes to bench_detail.json: round 5's line had
    grown to 20 KB and the driver's record kept only its tail (`BENCH_r05.json: parsed null`)."""
    c = {k: _r(out.get(k)) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                     'vs_baseline', 'dtype', 'data')}
    cfg = dict(out.get('config') or {})
    cfg['workload'] = str(cfg.get('workload', ''))[:200]
    c['config'] = cfg
    for k in ('rccl_ranks', 'dist_backend', 'visible_gpus', 'replicas_identical', 'compute_streams'):
        if k in out:
            c[k] = out[k]
    c['scale'] = 'unmeasured: the pool has no multi-GPU node' if (out.get('n_gpus') or 1) == 1 else 'this line'
    c['roofline'] = _rl(out.get('roofline'), ('avg_launch_ms', 'launches_per_step', 'kernel_ms_per_step'))
    fr = {}
    for name, key in (('bcl', 'roofline_bcl'), ('gemm', 'roofline_gemm'), ('wgrad', 'roofline_wgrad'), ('wino', 'roofline_wino'),
                      ('wino_wgrad', 'roofline_wino_wgrad'), ('wino2d_gemm', 'roofline_wino2d_gemm'), ('hbm_convs', 'roofline_hbm_convs'),
                      ('resnet_branch', 'roofline_resnet_branch'), ('mfma_step', 'mfma_step_utilisation')):
        o = out.get(key)
        if isinstance(o, dict) and 'frac' in o:
            fr[name] = _r(o['frac'])
    top = out.get('roofline') or {}
    for name, tag in (('wino2d_gemm', 'k_plane_gemm'), ('wino', 'k_wino43')):
        if name not in fr and str(top.get('kernel', '')).startswith(tag):
            fr[name] = _r(top.get('frac'))
    c['roofline_fracs'] = fr
    b = out.get('roofline_bcl')
    if isinstance(b, dict):
        c['roofline_bcl'] = _rl(b, ('kernel_ms_per_step', 'algorithmic_mb_per_step'))
        c['roofline_bcl']['kernel'] = 'lattice build + splat gather + splat adjoint (lattice.hip, bcl.hip)'
        parts = b.get('parts') or {}
        c['roofline_bcl']['parts_ms'] = {k: _r(v.get('ms_per_step')) for k, v in parts.items()}
    rb = out.get('roofline_resnet_branch')
    if isinstance(rb, dict):
        c['roofline_resnet_branch'] = {'frac': _r(rb.get('frac')), 'span_ms': _r(rb.get('span_ms'), 2), 'non_mfma_ms': _r(rb.get('non_mfma_ms'), 2),
                                       'mfma_kernels_alone_frac': _r((rb.get('mfma_kernels_alone') or {}).get('frac'))}
    cb = out.get('cpu_baseline')
    if isinstance(cb, dict):
        c['cpu_baseline'] = {k: _r(cb.get(k), 5) for k in ('value', 'unit', 'cores', 'host_cpus', 'kind')}
        c['cpu_baseline']['sample'] = str(cb.get('sample', ''))[:240]
    fo = out.get('forward_only')
    if isinstance(fo, dict):
        c['forward_value'] = _r(fo.get('value'), 3)
        c['forward_ms_per_step'] = _r(fo.get('ms_per_step'), 3)
        ff = {}
        for name, key in (('top', 'roofline'), ('bcl', 'roofline_bcl'), ('resnet_branch', 'roofline_resnet_branch'),
                          ('mfma_step', 'mfma_step_utilisation'), ('hbm_convs', 'roofline_hbm_convs')):
            o = fo.get(key)
            if isinstance(o, dict) and 'frac' in o:
                ff[name] = _r(o['frac'])
        c['forward_fracs'] = ff
    cr = out.get('config_r')
    if isinstance(cr, dict):
        if 'error' in cr:
            c['config_r'] = {'error': str(cr['error'])[:200]}
        else:
            c['config_r'] = {'train_ms': _r(cr.get('train_ms'), 2), 'fwd_ms': _r(cr.get('fwd_ms'), 2),
                             'enqueue_ms': _r((cr.get('train') or {}).get('enqueue_ms'), 2), 'gpu_ms': _r((cr.get('train') or {}).get('gpu_ms'), 2),
                             'host_syncs': cr.get('host_syncs_per_train_iteration')}
    for k in ('peak_hbm_gb_per_gpu', 'switches'):
        if k in out:
            c[k] = _r(out[k], 2)
    c['detail'] = 'bench_detail.json'
    return c


def emit(out, detail=None):
    """full object -> bench_detail.json (+ gpurun_out/ when it exists); the compact line is the ONLY stdout line (the full object is
    not printed anywhere: a 20-KB line on either stream is what the driver's record could not hold)"""
    full = json.dumps(out)
    paths = [detail] if detail else [os.path.join(d, 'bench_detail.json') for d in (ROOT, os.path.join(ROOT, 'gpurun_out')) if os.path.isdir(d)]
    for path in paths:
        try:
            with open(path, 'w') as fh:
                fh.write(full + '\n')
        except OSError as e:
            sys.stderr.write('bench.py: could not write %s (%s)\n' % (path, e))
    sys.stderr.write('bench.py: full object (%d bytes) written to bench_detail.json\n' % len(full))
    sys.stderr.flush()
    line = json.dumps(compact_line(out), separators=(',', ':'))
    assert len(line) < 8192, len(line)
    sys.stdout.flush()
    print(line, flush=True)


def spawn_ranks(a):
    """`python bench.py --gpus N` outside a launcher: start `torch.distributed.run` with N ranks as a CHILD process and hand its
    exit code on.  Nothing in this process has touched the GPU (no torch import yet), so no initialised process is replaced."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % a.gpus,
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault('OMP_NUM_THREADS', '8')
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(a))
    import torch
    import torch.distributed as dist
    from efgh_amd import lattice, ops, synthetic as syn
    from efgh_amd.losses import EFGHCriterion
    from efgh_amd.nets import EFGHBackbone
    from efgh_amd.train import Trainer

    switches = {}
    for item in a.set:
        import ast
        import importlib
        name, val = item.split('=', 1)
        mod, attr = name.rsplit('.', 1)
        m_ = importlib.import_module('efgh_amd.' + mod)
        if not hasattr(m_, attr):
            sys.exit('bench.py --set: efgh_amd.%s has no attribute %s' % (mod, attr))
        setattr(m_, attr, ast.literal_eval(val))
        switches[name] = ast.literal_eval(val)
    if switches:
        ops.apply_switches()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != a.gpus:
        sys.exit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks (use --nproc-per-node %d, or run '
                 '`python bench.py --gpus %d` bare and let it start the ranks itself)' % (a.gpus, world, a.gpus, a.gpus))
    ndev = torch.cuda.device_count()          # (does not initialise the GPU)
    if ndev < 1:
        sys.exit('bench.py: no GPU visible')
    shared = world > ndev                     # more ranks than devices: ranks share GPUs (plumbing test on a 1-GPU box)
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    backend, rccl_ranks = None, None
    if world > 1:
        # RCCL over xGMI ('nccl' is RCCL on ROCm).  RCCL cannot put two ranks on one device, so ranks that SHARE a GPU
        # (or EFGH_DIST_BACKEND=gloo) rendezvous over gloo: same code path above the backend, no scaling claim.
        backend = os.environ.get('EFGH_DIST_BACKEND', 'gloo' if shared else 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
            rccl_ranks = dist.get_world_size()
        else:
            dist.init_process_group(backend)
    raw, npts = ((128, 256), 2048) if a.small else (RAW, NPTS)
    args = syn.default_args(raw, 'cuda')

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return dt

    nrot = max(1, a.rotate_inputs)

    def load(B):
        """`nrot` resident batches of different frame-pairs (seed = global sample index, disjoint over ranks and rotations)"""
        sets = []
        for r in range(nrot):
            batch = syn.make_batch(raw, npts, B, first_seed=(r * world + rank) * B)
            inp = [torch.from_numpy(batch[k]).to(dev) for k in ('pc', 'img', 'calib', 'A')]
            gt = {k: torch.from_numpy(v).to(dev) for k, v in batch['gt'].items()}
            sets.append((inp, gt))
        return sets

    torch.manual_seed(0)                                  # identical weights on every rank
    model = EFGHBackbone(args).to(dev)

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        ops.PROFILE, ops.PROFILE_WGRAD, ops.PROFILE_WINO, ops.PROFILE_WINO_WGRAD, ops.PROFILE_BCL = [], [], [], [], []
        ops.PROFILE_WINO2D, ops.PROFILE_WINO2D_GEMM, ops.PROFILE_THIN, ops.PROFILE_DED = [], [], [], []
        lattice.PROFILE = ops.PROFILE_BCL
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        prof = {'gemm': ops.PROFILE, 'wgrad': ops.PROFILE_WGRAD, 'wino': ops.PROFILE_WINO,
                'wino_wgrad': ops.PROFILE_WINO_WGRAD, 'bcl': ops.PROFILE_BCL, 'wino2d': ops.PROFILE_WINO2D,
                'wino2d_gemm': ops.PROFILE_WINO2D_GEMM, 'hbm_convs': ops.PROFILE_THIN, 'dedicated': ops.PROFILE_DED}
        ops.PROFILE_THIN = ops.PROFILE_DED = None
        ops.PROFILE = ops.PROFILE_WGRAD = ops.PROFILE_WINO = ops.PROFILE_WINO_WGRAD = ops.PROFILE_BCL = lattice.PROFILE = None
        ops.PROFILE_WINO2D = ops.PROFILE_WINO2D_GEMM = None
        return dt, prof

    def attach_serialized(dst, step_fn, workload):
        """Independent branches of the network run on up to four streams (nets/efghbackbone.py, nets/fnet.py; weight gradients on
        their own stream, nets/fn.py).  `value` / `ms_per_step` come from that schedule.  Inside it the HIP-event interval of a
        launch is not the kernel's own duration any more - it shares the GPU with launches of other branches (k_wino43: 3.2 ms
        per launch instead of 1.46) - so a roofline fraction computed from it would measure the schedule, not the kernel.  The
        `roofline*` objects therefore hold the figures of a second pass inside this same process, right after the timed region:
        min(steps, 3) steps with everything on ONE stream (same kernels, same shapes, HIP events on the launch stream), and the
        timed-region figures are kept beside them under `concurrent`.  The rocprofv3 statistics committed under profiles/ exist
        for both schedules (`..._single_stream.csv` is the one the roofline durations agree with)."""
        from efgh_amd.nets import efghbackbone as bb
        if not bb.SIDE_STREAM:
            return
        nser = max(1, min(a.steps, 3))
        bb.SIDE_STREAM, wg, ops.WGRAD_SIDE = False, ops.WGRAD_SIDE, False
        try:
            _, prof2 = timed(step_fn, nser, 1)
        finally:
            bb.SIDE_STREAM, ops.WGRAD_SIDE = True, wg
        if dst is None:
            return
        mine = {v['kernel']: (k, v) for k, v in dst.items() if k.startswith('roofline') and isinstance(v, dict)}
        keep = ('achieved', 'frac', 'avg_launch_ms', 'kernel_ms_per_step', 'algorithmic_tflops', 'parts', 'dedicated')
        for k, v in rooflines(prof2, nser, workload).items():
            if k.startswith('roofline') and isinstance(v, dict) and v.get('kernel') in mine:
                key, cur = mine[v['kernel']]
                conc = {f: cur[f] for f in keep if f in cur and f not in ('parts', 'dedicated')}
                conc['note'] = 'the same launches inside the timed region, sharing the GPU with the other streams'
                for f in keep:
                    if f in v:
                        cur[f] = v[f]
                cur['concurrent'] = conc
                cur['measured'] = 'single-stream pass of %d steps right after the timed region (see bench.py attach_serialized)' % nser
        ser = rooflines(prof2, nser, workload)
        if 'winograd2d_layers' in ser:
            dst['winograd2d_layers'] = ser['winograd2d_layers']

    def resnet_branch(inp, gt, train):
        """north star: ">= 60 % MFMA for the ResNet branch".  G = two ResNet-18 trunks + the decoder + the depth / mask heads
        (nets/gnet.py:31-36,82-87,103-124; 60 % of the forward FLOPs of the path) run ALONE on one stream on the upstream outputs
        of a full forward (detached): forward - and in training the three G loss terms and the backward through G - between two
        HIP events.  `achieved` = MFMA FLOP the contraction kernels of that span EXECUTE (Winograd kernels at the products they
        run) / the whole span's GPU time - BatchNorm passes, Winograd transforms, pooling, the rasteriser, the loss sweep and every
        small launch included - / 157.3 TFLOP/s: the branch-level figure next to the per-kernel fractions."""
        from efgh_amd.nets import efghbackbone as bb
        keep = (bb.SIDE_STREAM, ops.WGRAD_SIDE)
        bb.SIDE_STREAM, ops.WGRAD_SIDE = False, False
        try:
            model.train(train)
            with torch.set_grad_enabled(train):
                full = model(*inp)
            ret = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in full.items()
                   if not k.startswith('g_') and k not in ('efgh_cam_T_velo', 'cam_T_velo')}
            ret['network'] = 'EHF'
            ret['sensor2_T_sensor1'] = torch.bmm(ret['f_l'], ret['e_l'])
            del full
            crit = EFGHCriterion(args)
            crit.dp_exact = False            # (rank 0 alone runs this span: no collective inside it)

            def once():
                with torch.set_grad_enabled(train):
                    pg = model.G(inp[0], inp[1], dict(ret))
                    if train:
                        Lg, _ = crit.compute_loss(inp[0], inp[1], inp[2], inp[3], gt, pg)
                        (Lg['g_trs'] + Lg['g_depth'] + Lg['g_mask']).backward()
                        for p_ in model.parameters():
                            p_.grad = None
            once()
            torch.cuda.synchronize()
            ops.PROFILE, ops.PROFILE_WGRAD, ops.PROFILE_WINO, ops.PROFILE_WINO_WGRAD = [], [], [], []
            ops.PROFILE_WINO2D, ops.PROFILE_WINO2D_GEMM, ops.PROFILE_THIN = [], [], []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 2
            e0.record()
            for _ in range(n):
                once()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            fl = (sum(p[2] for p in ops.PROFILE) + sum(p[2] for p in ops.PROFILE_WGRAD) + sum(p[2] for p in ops.PROFILE_WINO2D_GEMM)
                  + 0.5 * (sum(p[2] for p in ops.PROFILE_WINO) + sum(p[2] for p in ops.PROFILE_WINO_WGRAD))) / n
            mfma_ms = sum(p[0].elapsed_time(p[1]) for lst in (ops.PROFILE, ops.PROFILE_WGRAD, ops.PROFILE_WINO, ops.PROFILE_WINO_WGRAD,
                                                             ops.PROFILE_WINO2D_GEMM) for p in lst) / n
            tf = fl / (ms * 1e-3) / 1e12
            # the same span in DIRECT-form FLOP (what a kernel without Winograd would have to execute): the 2-D layers in full
            alg = (sum(p[2] for p in ops.PROFILE) + sum(p[2] for p in ops.PROFILE_WGRAD) + sum(p[2] for p in ops.PROFILE_WINO)
                   + sum(p[2] for p in ops.PROFILE_WINO_WGRAD) + sum(p[2] for p in ops.PROFILE_WINO2D)) / n
            return {'bound': 'mfma', 'kernel': 'everything G launches (%s), one stream, upstream outputs detached: contraction kernels, '
                                               'BatchNorm / activation passes, Winograd transforms, depth rasteriser, loss sweep'
                                               % ('forward + G loss terms + backward' if train else 'eval forward'),
                    'achieved': tf, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / PEAK_F32_MFMA_TFLOPS, 'traffic': None,
                    'span_ms': ms, 'executed_mfma_tflop': fl / 1e12, 'mfma_kernel_ms': mfma_ms,
                    'algorithmic_tflops': alg / (ms * 1e-3) / 1e12,
                    'note': 'frac counts the MFMA FLOP the kernels EXECUTE: moving a layer from the 1-D to the 2-D Winograd form halves '
                            'them and lowers frac while the span gets shorter (round 5: the 128-channel layers); algorithmic_tflops is '
                            'the direct-form rate of the same span',
                    'mfma_kernels_alone': {'achieved': fl / (mfma_ms * 1e-3) / 1e12 if mfma_ms > 0 else 0.0,
                                           'frac': fl / (mfma_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS if mfma_ms > 0 else 0.0},
                    'non_mfma_ms': ms - mfma_ms, 'batch': int(inp[0].shape[0])}
        finally:
            ops.PROFILE_THIN = None
            ops.PROFILE = ops.PROFILE_WGRAD = ops.PROFILE_WINO = ops.PROFILE_WINO_WGRAD = None
            ops.PROFILE_WINO2D = ops.PROFILE_WINO2D_GEMM = None
            bb.SIDE_STREAM, ops.WGRAD_SIDE = keep

    out = None
    fwd = None
    if a.mode == 'fwd' or not a.no_forward_section:
        Bf = a.batch if (a.mode == 'fwd' and a.batch) else 4
        fsets, fcount = load(Bf), [0]
        model.eval()

        def fstep():
            inp = fsets[fcount[0] % nrot][0]
            fcount[0] += 1
            with torch.no_grad():
                return model(*inp)
        dt, prof = timed(fstep, a.steps, a.warmup)
        fwd = {'metric': 'frame-pairs/sec EFGHNet forward (384x1280 RGB + 64x2048 range), whole job',
               'value': world * Bf * a.steps / dt, 'unit': 'frame-pairs/s', 'ms_per_step': dt / a.steps * 1e3,
               'workload': 'BASELINE.json configs[1]: EFGHNet forward only (eval), batch=%d per GPU' % Bf,
               }
        fwd.update(rooflines(prof, a.steps, 'fwd'))
        fwd['mfma_step_utilisation'] = mfma_step_utilisation(prof, a.steps, fwd['ms_per_step'])
        attach_serialized(fwd, fstep, 'fwd')
        if not a.no_branch_section:
            fwd['roofline_resnet_branch'] = resnet_branch(fsets[0][0], None, False)
        del fsets
    if a.mode == 'train':
        Bt = a.batch or 8
        tsets, tcount = load(Bt), [0]
        trainer = Trainer(model, EFGHCriterion(args), lr=1e-4)

        def tstep():
            inp, gt = tsets[tcount[0] % nrot]
            tcount[0] += 1
            return trainer.step(*inp, gt)
        lat0 = dict(lattice.STATS)
        dt, prof = timed(tstep, a.steps, a.warmup)
        lat1 = dict(lattice.STATS)
        if rank == 0:
            out = {
                'metric': 'frame-pairs/sec EFGHNet fwd+bwd (384x1280 RGB + 64x2048 range), whole job',
                'value': world * Bt * a.steps / dt, 'unit': 'frame-pairs/s',
                'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3,
                'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
                'data': 'synthetic',
                'config': {'workload': 'BASELINE.json configs[2]: EFGHNet fwd+bwd with efghloss + gradient all-reduce + '
                                       'fused Adam, synthetic %dx%d RGB + %d-point sweep, batch=%d per GPU, random-init '
                                       'weights' % (raw[0] // 2, raw[1] // 2, npts, Bt),
                           'global_batch': world * Bt, 'points': npts, 'parallelism': 'dp%d' % world},
                'rccl_ranks': rccl_ranks, 'dist_backend': backend, 'visible_gpus': ndev,
                'forward_only': fwd,
                'inputs': {'rotating_batches': nrot,
                           'lattice_pyramids': {k: lat1[k] - lat0[k] for k in lat1},
                           'note': 'warm-up + timed steps; speculative = all five levels enqueued from the previous sizes with ONE '
                                   'read-back, level_by_level = the five-sync fallback, reenqueued = escalated rebuilds'},
            }
            out.update(rooflines(prof, a.steps))
            out['mfma_step_utilisation'] = mfma_step_utilisation(prof, a.steps, out['ms_per_step'])
            out['peak_hbm_gb_per_gpu'] = torch.cuda.max_memory_allocated() / 1e9      # of 288 GB
        if world > 1:
            # data parallelism keeps the replicas identical: same start (broadcast), same all-reduced gradients, same Adam.  Two
            # position-weighted checksums of the flat weight buffer, max == min over the ranks
            w = trainer.flat.w.double()
            ck = torch.stack([w.sum(), (w * torch.arange(1, w.numel() + 1, device=dev, dtype=torch.float64).remainder(977.0)).sum()])
            hi, lo = ck.clone(), ck.clone()
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            if rank == 0:
                out['replicas_identical'] = bool(torch.equal(hi, lo))
                out['compute_streams'] = 1 + len(ops.side_streams())
        attach_serialized(out, tstep, 'train')
        if rank == 0 and not a.no_branch_section:
            out['roofline_resnet_branch'] = resnet_branch(tsets[0][0], tsets[0][1], True)
    elif rank == 0:
        out = {'metric': fwd['metric'], 'value': fwd['value'], 'unit': 'frame-pairs/s', 'n_gpus': world,
               'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': fwd['ms_per_step'], 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': fwd['workload'] + ', synthetic %dx%d RGB + %d-point sweep, random-init weights'
                                      % (raw[0] // 2, raw[1] // 2, npts),
                          'global_batch': world * (a.batch or 4), 'points': npts, 'parallelism': 'dp%d' % world},
               'rccl_ranks': rccl_ranks, 'dist_backend': backend, 'visible_gpus': ndev,
               }
        out.update({k: v for k, v in fwd.items() if k.startswith('roofline') or k == 'mfma_step_utilisation'})
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(raw, npts, a.mode)
        if world == 1 and not a.small and not a.no_config_r:
            # the reference's own configuration and loop (batch 1): everything of the main measurement is released first
            fsets = tsets = trainer = model = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            try:
                out['config_r'] = config_r()
            except Exception as e:          # noqa: BLE001  (a side measurement must not take the headline line down)
                out['config_r'] = {'error': repr(e)}
        if switches:
            out['switches'] = switches            # (an A/B run of the builder: not the default configuration)
        if fwd is not None and a.mode == 'train':
            # small copies at the END of the line (a log tail keeps them): the forward-only workload, BASELINE configs[1]
            out['forward_value'] = fwd['value']
            out['forward_ms_per_step'] = fwd['ms_per_step']
            out['forward_mfma_frac'] = fwd['mfma_step_utilisation']['frac']
            out['forward_roofline_frac'] = fwd.get('roofline', {}).get('frac')
            out['forward_roofline_bcl_frac'] = fwd.get('roofline_bcl', {}).get('frac')
        emit(out, a.detail)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
