"""config_r (900x1600 / 65 536 points / batch 1) through Trainer.step (flat parameters, direct gradient writes, fused Adam)
instead of the reference's loop with stock Adam: the upper bound of what re-homing the optimizer buys that loop (GPU box)."""
import sys
import time
sys.path.insert(0, '/root/repo')
import numpy as np
import torch
from efgh_amd import synthetic as syn
from efgh_amd.losses import EFGHCriterion
from efgh_amd.nets import EFGHBackbone
from efgh_amd.train import Trainer

raw, npts = (900, 1600), 65536
args = syn.default_args(raw, 'cuda')
torch.manual_seed(0)
model = EFGHBackbone(args).cuda()
tr = Trainer(model, EFGHCriterion(args), lr=1e-4)
pairs = [syn.make_batch(raw, npts, 1, first_seed=i) for i in range(4)]
host = [([torch.from_numpy(b[k]).pin_memory() for k in ('pc', 'img', 'calib', 'A')],
         {k: torch.from_numpy(v) for k, v in b['gt'].items()}) for b in pairs]
rows = []
for i in range(12):
    (pcd, img, calib, A), gt = host[i % 4]
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pcd, img, calib, A = (t.to('cuda').float() for t in (pcd, img, calib, A))
    e0.record()
    losses, pred = tr.step(pcd, img, calib, A, dict(gt))
    e1.record()
    t1 = time.perf_counter()
    vals = [losses[k].item() for k in list(losses.keys())]
    _ = pred['sensor2_T_sensor1'].cpu().detach().numpy()[0]
    t2 = time.perf_counter()
    rows.append(((t2 - t0) * 1e3, (t1 - t0) * 1e3, e0.elapsed_time(e1)))
r = np.median(np.array(rows[4:]), axis=0)
print('Trainer.step at config_r: loop %.2f ms, enqueue %.2f ms, gpu %.2f ms' % tuple(r))
