"""where the HOST time of a batch-1 step goes (cProfile over a few iterations of the reference-shaped loop at config R):
python tools/host_profile.py [train|eval] [iters]"""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efgh_amd import synthetic as syn
from efgh_amd.losses import EFGHCriterion
from efgh_amd.nets import EFGHBackbone

mode = sys.argv[1] if len(sys.argv) > 1 else 'train'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
raw, npts = (900, 1600), 65536
args = syn.default_args(raw, 'cuda')
torch.manual_seed(0)
model = EFGHBackbone(args).cuda()
criterion = EFGHCriterion(args)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
b = syn.make_batch(raw, npts, 1)
inp = [torch.from_numpy(b[k]).cuda() for k in ('pc', 'img', 'calib', 'A')]
gt = {k: torch.from_numpy(v) for k, v in b['gt'].items()}


def step():
    if mode == 'train':
        pred = model(*inp)
        losses, _ = criterion.compute_loss(*inp, dict(gt), pred)
        opt.zero_grad()
        losses['total'].backward()
        opt.step()
    else:
        with torch.no_grad():
            model(*inp)


model.train(mode == 'train')
for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(iters):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
