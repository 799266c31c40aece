"""N eval forwards of the reference's own configuration (900x1600 / 65 536 points / batch 1), for a kernel census under rocprofv3
(python tools/config_r_eval_loop.py [iters])"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efgh_amd import synthetic as syn
from efgh_amd.nets import EFGHBackbone

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
raw, npts = (900, 1600), 65536
args = syn.default_args(raw, 'cuda')
torch.manual_seed(0)
model = EFGHBackbone(args).cuda().eval()
pairs = [syn.make_batch(raw, npts, 1, first_seed=i) for i in range(4)]
inps = [[torch.from_numpy(b[k]).cuda() for k in ('pc', 'img', 'calib', 'A')] for b in pairs]
with torch.no_grad():
    for i in range(iters):
        pred = model(*inps[i % 4])
        _ = pred['sensor2_T_sensor1'].cpu().numpy()[0]
torch.cuda.synchronize()
print('config_r_eval_loop: %d forwards' % iters)
