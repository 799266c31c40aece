"""run-to-run reproducibility: the eval forward twice (all outputs bit-equal), and the gradients of one training-mode
forward + loss + backward twice from identical weights (per-parameter: bit-equal or the largest relative difference)
    python tools/check_determinism.py [--small]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from efgh_amd import synthetic as syn                       # noqa: E402
from efgh_amd.losses import EFGHCriterion                   # noqa: E402
from efgh_amd.nets import EFGHBackbone                      # noqa: E402

small = '--small' in sys.argv
RAW, NPTS = ((128, 256), 2048) if small else ((768, 2560), 131072)
manifest = json.load(open(os.path.join(ROOT, 'tests/golden/state_dict_manifest.json')))
args = syn.default_args(RAW, 'cuda')
m = EFGHBackbone(args)
m.load_state_dict(syn.synthetic_state_dict(manifest['state_dict'], 1))
m = m.cuda().eval()
b = syn.make_batch(RAW, NPTS, 2)
inp = [torch.from_numpy(b[k]).cuda() for k in ('pc', 'img', 'calib', 'A')]
gt = {k: torch.from_numpy(v).cuda() for k, v in b['gt'].items()}
with torch.no_grad():
    o1 = m(*inp)
    o2 = m(*inp)
print('eval forward:', {k: bool(torch.equal(o1[k], o2[k])) for k in o1 if torch.is_tensor(o1[k])})

crit = EFGHCriterion(args)
sd = {k: v.clone() for k, v in m.state_dict().items()}
grads = []
for run in range(2):
    m.load_state_dict(sd)
    m.train()
    m.zero_grad(set_to_none=True)
    pred = m(*inp)
    L, _ = crit.compute_loss(*inp, dict(gt), pred)
    L['total'].backward()
    grads.append({n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    print('run', run, 'total', float(L['total']))
bad = []
for n in grads[0]:
    a, c = grads[0][n], grads[1][n]
    if not torch.equal(a, c):
        bad.append((float((a - c).abs().max() / (a.abs().max() + 1e-30)), n))
bad.sort(reverse=True)
print('parameters with bit-identical gradients: %d of %d' % (len(grads[0]) - len(bad), len(grads[0])))
for r, n in bad[:25]:
    print('   %.3e  %s' % (r, n))
