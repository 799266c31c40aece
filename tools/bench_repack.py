"""time of the batched in-place repack (all packed layouts + their Winograd images) of the full model, tiled vs flat kernel (GPU box)"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from efgh_amd import ops, synthetic as syn
from efgh_amd.losses import EFGHCriterion
from efgh_amd.nets import EFGHBackbone
from efgh_amd.train import Trainer

raw, npts = (128, 256), 2048
args = syn.default_args(raw, 'cuda')
torch.manual_seed(0)
model = EFGHBackbone(args).cuda()
tr = Trainer(model, EFGHCriterion(args), lr=1e-4)
b = syn.make_batch(raw, npts, 2)
inp = [torch.from_numpy(b[k]).cuda() for k in ('pc', 'img', 'calib', 'A')]
gt = {k: torch.from_numpy(v).cuda() for k, v in b['gt'].items()}
for _ in range(2):
    tr.step(*inp, gt)                       # registers every layout (forward and data-gradient) and their Winograd images
dev = torch.device('cuda', torch.cuda.current_device())
for tiled in (True, False, True, False):
    ops.PACK_TILED = tiled
    ts = []
    for wino in (True, False):
        ops.BATCH_WINO = wino
        for i in range(6):
            ops.bump_epoch(tr.flat.epoch)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.repack_stale(dev, tr.flat.epoch)
            e1.record()
            torch.cuda.synchronize()
            if i >= 2:
                ts.append((wino, e0.elapsed_time(e1)))
    w = [t for k, t in ts if k]; nw = [t for k, t in ts if not k]
    print('tiled=%s: repack + Winograd images %.3f ms, repack alone %.3f ms' % (tiled, sorted(w)[len(w) // 2], sorted(nw)[len(nw) // 2]))
ops.BATCH_WINO = True; ops.PACK_TILED = True
