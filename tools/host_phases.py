"""host time of the phases of one batch-1 training iteration in the reference's loop shape (config R): where does the enqueue
time go?  python tools/host_phases.py [iters]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from efgh_amd import synthetic as syn
from efgh_amd.losses import EFGHCriterion
from efgh_amd.nets import EFGHBackbone

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 8
raw, npts = (900, 1600), 65536
args = syn.default_args(raw, 'cuda')
torch.manual_seed(0)
model = torch.nn.DataParallel(EFGHBackbone(args).cuda(), device_ids=[0])
criterion = EFGHCriterion(args)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=0)
b = syn.make_batch(raw, npts, 1)
host = [torch.from_numpy(b[k]).pin_memory() for k in ('pc', 'img', 'calib', 'A')]
gt = {k: torch.from_numpy(v) for k, v in b['gt'].items()}
rows = []
model.train()
for it in range(iters + 3):
    t = [time.perf_counter()]
    pcd, img, calib, A = (x.to('cuda').float() for x in host); t.append(time.perf_counter())
    pred = model(pcd, img, calib, A, False); t.append(time.perf_counter())
    losses, gt2 = criterion.compute_loss(pcd, img, calib, A, dict(gt), pred); t.append(time.perf_counter())
    opt.zero_grad(); t.append(time.perf_counter())
    losses['total'].backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    vals = [losses[k].item() for k in losses]; t.append(time.perf_counter())
    if it >= 3:
        rows.append(np.diff(t) * 1e3)
names = ['h2d', 'forward', 'loss', 'zero_grad', 'backward', 'adam step', 'item() x%d (waits for the GPU)' % len(vals)]
med = np.median(np.array(rows), 0)
for n, v in zip(names, med):
    print('%-36s %7.2f ms' % (n, v))
print('%-36s %7.2f ms' % ('sum', med.sum()))
