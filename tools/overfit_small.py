import sys; sys.path.insert(0, '/root/repo')
import json, torch
from efgh_amd import synthetic as syn
from efgh_amd.losses import EFGHCriterion
from efgh_amd.nets import EFGHBackbone
from efgh_amd.train import Trainer
RAW, NPTS = (128, 256), 2048
manifest = json.load(open('/root/repo/tests/golden/state_dict_manifest.json'))
args = syn.default_args(RAW, 'cuda')
m = EFGHBackbone(args); m.load_state_dict(syn.synthetic_state_dict(manifest['state_dict'], 1))
tr = Trainer(m.cuda(), EFGHCriterion(args), lr=float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3)
b = syn.make_batch(RAW, NPTS, 2)
inp = [torch.from_numpy(b[k]).cuda() for k in ('pc', 'img', 'calib', 'A')]
gt = {k: torch.from_numpy(v) for k, v in b['gt'].items()}
for it in range(40):
    L, _ = tr.step(*inp, gt)
    if it % 5 == 0 or it == 39:
        print(it, ' '.join('%s %.3f' % (k, float(v.detach())) for k, v in L.items() if k in ('total', 'e_gn', 'h_hrzn', 'fov', 'g_trs', 'g_depth', 'g_mask')))
