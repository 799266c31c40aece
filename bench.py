"""EFGHNet hot-path benchmark on MI355X (contract: see the task statement / DESIGN.md §Measurement).

    python bench.py --gpus N --steps K --warmup W [--mode fwd|train] [--batch B]

One process per GPU (the driver launches N>1 through torch.distributed.run).  A step = one pass of
the hot path over one batch of synthetic frame-pairs (BASELINE.json configs[1]: EFGHNet forward,
384x1280 RGB + 64x2048-point sweep, batch 4 per GPU, inputs resident in HBM before the timed region).
Rank 0 prints ONE JSON line.
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
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--mode', default='fwd', choices=['fwd'])
    ap.add_argument('--batch', type=int, default=4, help='frame-pairs per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--small', action='store_true', help='debug: 128x256 / 2048 points')
    return ap.parse_args()


def cpu_baseline(raw, npts):
    """the oracle (CPU restatement of the reference) timed on the host cores: ONE frame-pair of the
    same workload, eval forward (bounded sample, ~10-30 s)."""
    import torch
    from efgh_amd import synthetic as syn
    from efgh_amd.nets import EFGHBackbone
    from oracle import efgh_oracle as O
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = EFGHBackbone(syn.default_args(raw, 'cpu'))      # parameter container only; never executed on CPU
    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    b = syn.make_batch(raw, npts, 1)
    T = torch.from_numpy
    args = syn.default_args(raw, 'cpu')
    t0 = time.time()
    with torch.no_grad():
        O.forward(P, T(b['pc']), T(b['img']), T(b['calib']), T(b['A']), args, train=False)
    dt = time.time() - t0
    return {'value': 1.0 / dt, 'unit': 'frame-pairs/s', 'cores': cores, 'kind': 'port',
            'sample': '1 frame-pair of the same workload (eval forward, B=1), oracle/efgh_oracle.py + '
                      'oracle/lattice_oracle.c, torch CPU fp32, %.1f s' % dt}


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from efgh_amd import ops, synthetic as syn
    from efgh_amd.nets import EFGHBackbone

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    assert world == a.gpus, (world, a.gpus)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)
    raw, npts = ((128, 256), 2048) if a.small else (RAW, NPTS)

    torch.manual_seed(0)                                  # identical weights on every rank
    model = EFGHBackbone(syn.default_args(raw, 'cuda')).to(dev)
    model.eval()
    B = a.batch
    batch = syn.make_batch(raw, npts, B, first_seed=rank * B)       # seed = global sample index
    inp = [torch.from_numpy(batch[k]).to(dev) for k in ('pc', 'img', 'calib', 'A')]

    def step():
        with torch.no_grad():
            return model(*inp)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ops.PROFILE = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        gemm_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in prof)
        gemm_fl = sum(f for _, _, f in prof)
        n_launch = len(prof)
        achieved = gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        out = {
            'metric': 'frame-pairs/sec EFGHNet forward (384x1280 RGB + 64x2048 range), whole job',
            'value': world * B * a.steps / dt,
            'unit': 'frame-pairs/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': dt / a.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE.json configs[1]: EFGHNet forward only (eval), synthetic %dx%d RGB + '
                                   '%d-point sweep, batch=%d per GPU, random-init weights' %
                                   (raw[0] // 2, raw[1] // 2, npts, B),
                       'global_batch': world * B, 'points': npts, 'parallelism': 'dp%d' % world},
            'roofline': {
                'bound': 'mfma', 'kernel': 'k_gather_gemm (fp32 MFMA implicit GEMM, all conv/convT/linear/BCL-blur launches)',
                'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved / PEAK_F32_MFMA_TFLOPS, 'traffic': None,
                'launches_per_step': n_launch / max(1, a.steps),
                'avg_launch_ms': gemm_ms / max(1, n_launch),
                'algorithmic_gflop_per_launch': gemm_fl / max(1, n_launch) / 1e9,
                'gemm_ms_per_step': gemm_ms / max(1, a.steps),
            },
        }
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(raw, npts)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
