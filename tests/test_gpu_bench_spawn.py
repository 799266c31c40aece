"""`python bench.py --gpus 2` with no launcher around it starts its own ranks (before touching the GPU) and rank 0 prints the one
JSON line; on a 1-GPU box the two ranks share the device and rendezvous over gloo (bench.py reports the backend it used)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_spawns_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--small', '--steps', '3', '--warmup', '1',
                        '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['parallelism'] == 'dp2' and out['value'] > 0
    import torch
    if torch.cuda.device_count() >= 2:
        assert out['dist_backend'] == 'nccl' and out['rccl_ranks'] == 2
    else:
        assert out['dist_backend'] == 'gloo' and out['rccl_ranks'] is None
    # after the timed steps every rank holds bit-identical weights (same start, same all-reduced gradients, same Adam), and the
    # communication stream has its hardware queue: three compute streams under world > 1
    assert out['replicas_identical'] is True and out['compute_streams'] == 3
    for k in ('roofline', 'forward_value', 'roofline_fracs'):
        assert k in out
    assert out['roofline']['frac'] <= 1.0 and len(lines[0]) < 8192


def test_bench_rejects_a_mismatched_launcher():
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--small'], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr


REFERENCE_LIKE_SCRIPT = '''
# the shape of the reference's entry script and loop (main.py:14-15,126-129,181-183; iterater.py:26-43), nothing package-specific
import json, os, sys
import numpy as np
import torch, torch.utils.data
import nets
import losses
from efgh_amd import synthetic as syn                 # (stands in for data_loader: synthetic frame-pairs)

args = syn.default_args((128, 256), 'cuda')
args['arch'] = 'EFGH'


class Pairs(torch.utils.data.Dataset):
    def __len__(self):
        return 8

    def __getitem__(self, i):
        s = syn.make_sample((128, 256), 2048, i)
        return s['pc'], s['img'], s['calib'], s['A'], {k: np.asarray(v) for k, v in s['gt'].items()}, 'pair%d' % i


device = torch.device('cuda')
torch.manual_seed(int(os.environ.get('RANK', 0)))     # ranks would start apart without the wrapper's broadcast
model = nets.__dict__[args['arch'] + 'Backbone'](args).to(device)
model = torch.nn.DataParallel(model)
criterion = losses.__dict__[args['arch'] + 'Criterion'](args)
loader = torch.utils.data.DataLoader(Pairs(), batch_size=4, shuffle=True, num_workers=0)
optimizer = torch.optim.Adam([p for _, p in model.named_parameters() if p.requires_grad], lr=1e-4, weight_decay=0)
model.train()
it, seen, last = 0, [], None
for pcd, img, calib, A, gt, fname in loader:
    pcd, img, calib, A = (t.to(device).float() for t in (pcd, img, calib, A))
    pred = model(pcd, img, calib, A, it == 0)
    L, gt = criterion.compute_loss(pcd, img, calib, A, gt, pred)
    optimizer.zero_grad()
    L['total'].backward()
    optimizer.step()
    seen += list(fname)
    last = float(L['total'].detach())
    it += 1
torch.cuda.synchronize()
w = torch.cat([p.detach().double().reshape(-1) for p in model.parameters()])
ck = float((w * torch.arange(1, w.numel() + 1, device=w.device, dtype=torch.float64).remainder(977.0)).sum())
torch.save({'state_dict': model.state_dict()}, sys.argv[1] + '.ckpt%s' % os.environ.get('RANK', '0'))
json.dump({'cls': type(model).__name__, 'ck': ck, 'seen': seen, 'loss': last, 'iters': it,
           'key0': next(iter(model.state_dict()))}, open(sys.argv[1] + '.r%s' % os.environ.get('RANK', '0'), 'w'))
'''


def test_launcher_runs_a_reference_shaped_script_on_two_ranks(tmp_path):
    """`python -m efgh_amd.run --gpus 2 <script>`: a script shaped like the reference's main.py + iterater.py (model and criterion
    looked up by name in `nets` / `losses`, `torch.nn.DataParallel(model)`, a shuffling DataLoader, stock `torch.optim.Adam`,
    zero_grad / backward / step) runs UNCHANGED, one process per rank (sharing the one GPU of this box over gloo): every rank
    trains on its own half of every batch, the gradients are averaged before the optimizer step, and the replicas end on
    identical weights; rank 0 alone writes the checkpoint, with `module.`-prefixed keys."""
    script = tmp_path / 'main.py'
    script.write_text(REFERENCE_LIKE_SCRIPT)
    out = str(tmp_path / 'out')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['PYTHONPATH'] = ROOT
    r = subprocess.run([sys.executable, '-m', 'efgh_amd.run', '--gpus', '2', str(script), out], env=env, capture_output=True,
                       text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    r0, r1 = json.load(open(out + '.r0')), json.load(open(out + '.r1'))
    assert r0['cls'] == 'ProcessDataParallel' and r0['key0'].startswith('module.')
    assert r0['iters'] == r1['iters'] == 2 and len(r0['seen']) == len(r1['seen']) == 4          # 8 pairs, global batch 4 = 2 per rank
    assert sorted(r0['seen'] + r1['seen']) == ['pair%d' % i for i in range(8)]
    assert r0['ck'] == r1['ck'] and np_isfinite(r0['loss']) and np_isfinite(r1['loss'])
    assert os.path.exists(out + '.ckpt0') and not os.path.exists(out + '.ckpt1')
    # the single-process form of the same launch: DataParallel proper over the one pinned device, the whole batch on it
    r = subprocess.run([sys.executable, '-m', 'efgh_amd.run', str(script), out + '_single'], env=env, capture_output=True, text=True,
                       timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    s0 = json.load(open(out + '_single.r0'))
    assert s0['cls'] == 'DataParallel' and s0['iters'] == 2 and len(s0['seen']) == 8


def np_isfinite(v):
    import math
    return math.isfinite(v)


def _two_devices():
    import torch
    return torch.cuda.device_count() >= 2            # (counting devices does not initialise the GPU)


@pytest.mark.skipif(not _two_devices(), reason='needs two visible MI355X (this pool hands out 1-GPU boxes): the first multi-GPU lease runs it')
def test_two_ranks_on_rccl_over_xgmi():
    """the RCCL path itself, on the first box that has two devices: `bench.py --gpus 2 --small` must rendezvous on `nccl` (RCCL), one
    rank per device, with the bucketed gradient all-reduce over xGMI inside the timed region - world size from RCCL's own
    communicator (`rccl_ranks == 2`), bit-identical replicas after the steps, three compute streams (the fourth hardware queue is
    the collective's), and `HSA_ENABLE_IPC_MODE_LEGACY=0` carried into the ranks (dmabuf IPC only on this pool).  The zero-edit
    launcher takes the same route with the reference-shaped script and stock Adam."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                                                            'EFGH_DIST_BACKEND')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--small', '--steps', '3', '--warmup', '1',
                        '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert out['dist_backend'] == 'nccl' and out['rccl_ranks'] == 2 and out['visible_gpus'] >= 2
    assert out['replicas_identical'] is True and out['compute_streams'] == 3 and out['n_gpus'] == 2
    assert out['scaling'] == 'weak' and out['value'] > 0


@pytest.mark.skipif(not _two_devices(), reason='needs two visible MI355X')
def test_launcher_two_ranks_on_rccl(tmp_path):
    script = tmp_path / 'main.py'
    script.write_text(REFERENCE_LIKE_SCRIPT)
    out = str(tmp_path / 'out')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                                                            'EFGH_DIST_BACKEND')}
    env['PYTHONPATH'] = ROOT
    r = subprocess.run([sys.executable, '-m', 'efgh_amd.run', '--gpus', '2', str(script), out], env=env, capture_output=True,
                       text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    r0, r1 = json.load(open(out + '.r0')), json.load(open(out + '.r1'))
    assert r0['ck'] == r1['ck'] and r0['iters'] == r1['iters'] == 2
