"""Diagnose the full-size F-net logit difference between the HIP path and the oracle."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efgh_amd import synthetic as syn
from efgh_amd.nets import EFGHBackbone
from oracle import efgh_oracle as O
import torch.nn.functional as F

RAW, NPTS = (768, 2560), 131072
manifest = json.load(open('tests/golden/state_dict_manifest.json'))
P = syn.synthetic_state_dict(manifest['state_dict'], 1)
m = EFGHBackbone(syn.default_args(RAW, 'cuda')); m.load_state_dict(P, strict=True); m = m.cuda().eval()
b = syn.make_batch(RAW, NPTS, 1)
inp = [torch.from_numpy(b[k]).cuda() for k in ('pc', 'img', 'calib', 'A')]
cpu = [t.cpu() for t in inp]
torch.set_num_threads(32)
args = syn.default_args(RAW, 'cpu')
with torch.no_grad():
    r = dict(O.enet(P, cpu[0], False)); r.update(O.hnet(P, cpu[1], False)); r['network'] = 'EH'
    ko = {}
    rf = O.fnet(P, cpu[0], r, args, False, keep=ko)
    kf = {}
    dev = lambda d: {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()}
    f = m.F(inp[0], dev(r), keep=kf)
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
er = kf['e_range'].permute(0, 3, 1, 2).cpu()
print('e_range pixels differing', int(((er - ko['e_range']).abs().amax(1) > 1e-4).sum()), 'of', er.shape[2] * er.shape[3])
cam = kf['cam3'].permute(0, 3, 1, 2).cpu(); rng = kf['rng3'].permute(0, 3, 1, 2).cpu()
camn = cam / (cam.max() - cam.min()); rngn = rng / (rng.max() - rng.min())
print('cam_feat rel', rel(camn, ko['cam_feat'][0]), 'rng_feat rel', rel(rngn, ko['rng_feat'][0]))
lo = ko['f_logit'][0]
lg = kf['f_logit'].cpu()
def corr64(c, r_):
    rp = O.circular_assign(r_.double(), int(r_.size(-1) / 8))
    return (F.conv2d(rp, c.double()) / (c.size(0) * c.size(1))).view(1, -1)
l64_o = corr64(ko['cam_feat'][0], ko['rng_feat'][0])
l64_g = corr64(camn, rngn)
print('hip logit vs oracle', rel(lg, lo))
print('oracle logit vs f64(oracle feats)', rel(lo.double(), l64_o))
print('hip logit vs f64(hip feats)', rel(lg.double(), l64_g))
print('f64(hip feats) vs f64(oracle feats)', rel(l64_g, l64_o))
print('f_score rel', rel(f['f_score'].cpu(), rf['f_score']))
