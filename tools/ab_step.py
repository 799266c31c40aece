"""A/B aid: one training step at the small test size with the Winograd kernels on and off; prints every loss term and a checksum of
every sub-net's gradient.  Run once per build (EFGH_LIB=<path to libefgh_hip.so>) and diff the outputs."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efgh_amd import ops, synthetic as syn          # noqa: E402
from efgh_amd.losses import EFGHCriterion            # noqa: E402
from efgh_amd.nets import EFGHBackbone               # noqa: E402
from efgh_amd.train import Trainer                   # noqa: E402

RAW, NPTS = (128, 256), 2048
man = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'state_dict_manifest.json')))
args = syn.default_args(RAW, 'cuda')
b = syn.make_batch(RAW, NPTS, 2)
inp = [torch.from_numpy(b[k]).cuda() for k in ('pc', 'img', 'calib', 'A')]
gt = {k: torch.from_numpy(v) for k, v in b['gt'].items()}
for wino in (True, False):
    ops.USE_WINO = ops.USE_WINO_WGRAD = ops.USE_WINO2D = wino
    m = EFGHBackbone(args)
    m.load_state_dict(syn.synthetic_state_dict(man['state_dict'], 1))
    tr = Trainer(m.cuda(), EFGHCriterion(args), lr=1e-3)
    losses, pred = tr.step(*inp, gt)
    names = [n for n, _ in m.named_parameters()]
    print('wino', wino, {k: float(v) for k, v in losses.items()})
    for net in 'EHFG':
        s = sum(float(p.grad.double().abs().sum()) for n, p in zip(names, tr.flat.params) if n.startswith(net + '.'))
        print('  grad', net, repr(s))
    for k in ('e_gn_abs', 'h_hrzn_abs', 'f_score', 'g_trs'):
        print('  out', k, repr(float(pred[k].double().abs().sum())))
