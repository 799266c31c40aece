"""N training iterations of the reference's own configuration (900x1600 / 65 536 points / batch 1) in the reference's loop shape
(stock Adam, per-term .item(), two pose read-backs) and nothing else: the program the per-iteration kernel census under
rocprofv3 --kernel-trace --stats is taken from (tools/collect_round_evidence.sh).  python tools/config_r_loop.py [iters]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efgh_amd import synthetic as syn
from efgh_amd.losses import EFGHCriterion
from efgh_amd.nets import EFGHBackbone

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
raw, npts = (900, 1600), 65536
args = syn.default_args(raw, 'cuda')
torch.manual_seed(0)
model = torch.nn.DataParallel(EFGHBackbone(args).cuda(), device_ids=[0])
crit = EFGHCriterion(args)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=0)
pairs = [syn.make_batch(raw, npts, 1, first_seed=i) for i in range(4)]
host = [([torch.from_numpy(b[k]).pin_memory() for k in ('pc', 'img', 'calib', 'A')],
         {k: torch.from_numpy(v) for k, v in b['gt'].items()}) for b in pairs]
for i in range(iters):
    (pcd, img, calib, A), gt = host[i % 4]
    pcd, img, calib, A = (t.to('cuda').float() for t in (pcd, img, calib, A))
    pred = model(pcd, img, calib, A, False)
    losses, gt2 = crit.compute_loss(pcd, img, calib, A, dict(gt), pred)
    opt.zero_grad()
    losses['total'].backward()
    opt.step()
    vals = [losses[k].item() for k in list(losses.keys())]
    _ = gt2['sensor2_T_sensor1'].cpu().detach().numpy()[0], pred['sensor2_T_sensor1'].cpu().detach().numpy()[0]
torch.cuda.synchronize()
print('config_r_loop: %d training iterations' % iters)
