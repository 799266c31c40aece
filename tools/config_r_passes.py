"""Per-launch time of the BatchNorm-backward passes inside ONE training iteration of the reference's own configuration
(900x1600 / 65 536 points / batch 1): which shapes the reduce / apply kernels see there and at what rate (GPU box)."""
import sys
sys.path.insert(0, '/root/repo')
import torch
from efgh_amd import ops, synthetic as syn
from efgh_amd.losses import EFGHCriterion
from efgh_amd.nets import EFGHBackbone

raw, npts = (900, 1600), 65536
args = syn.default_args(raw, 'cuda')
torch.manual_seed(0)
model = EFGHBackbone(args).cuda()
crit = EFGHCriterion(args)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
b = syn.make_batch(raw, npts, 1, first_seed=0)
pcd, img, calib, A = (torch.from_numpy(b[k]).cuda().float() for k in ('pc', 'img', 'calib', 'A'))
gt = {k: torch.from_numpy(v) for k, v in b['gt'].items()}


def step():
    pred = model(pcd, img, calib, A, False)
    losses, _ = crit.compute_loss(pcd, img, calib, A, dict(gt), pred)
    opt.zero_grad()
    losses['total'].backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
rows = {}


def wrap(name, shape_of):
    fn = getattr(ops, name)

    def timed(*a, **k):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        torch.cuda.synchronize()
        r = rows.setdefault((name,) + shape_of(*a, **k), [0, 0.0])
        r[0] += 1; r[1] += e0.elapsed_time(e1)
        return out
    setattr(ops, name, timed)


wrap('act_bn_bwd_reduce', lambda *a, **k: (int(a[8]), int(a[9])))
wrap('act_bn_bwd_apply', lambda *a, **k: (int(a[11]), int(a[12])))
wrap('pool_bn_bwd', lambda *a, **k: (int(a[1].numel() // a[1].shape[-1]), int(a[1].shape[-1])))
wrap('scale_shift_act', lambda *a, **k: (int(a[6]), int(a[7])))
for _ in range(3):
    step()
tot = {}
print('%-20s %9s %5s | calls/iter  us/call  GB/s (2 or 3 streams)' % ('pass', 'M', 'C'))
for (name, M, C), (n, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    streams = {'act_bn_bwd_reduce': 2, 'act_bn_bwd_apply': 3, 'pool_bn_bwd': 4.25, 'scale_shift_act': 2}[name]
    print('%-20s %9d %5d | %5.1f %8.1f %8.0f' % (name, M, C, n / 3, ms / n * 1e3, streams * M * C * 4 / (ms / n * 1e-3) / 1e9))
    tot[name] = tot.get(name, 0.0) + ms / 3
print({k: round(v, 3) for k, v in tot.items()}, 'ms per iteration')
