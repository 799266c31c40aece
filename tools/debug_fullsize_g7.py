"""is the run-to-run variation of the G gradients caused by upstream (E/F) last-bit differences?  run G twice on frozen inputs"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efgh_amd import ops, synthetic as syn
from efgh_amd.nets import EFGHBackbone
RAW, NPTS = (768, 2560), 131072
manifest = json.load(open('tests/golden/state_dict_manifest.json'))
b = syn.make_batch(RAW, NPTS, 1)
T = torch.from_numpy
gpu = [T(b[k]).cuda() for k in ('pc', 'img', 'calib', 'A')]
m = EFGHBackbone(syn.default_args(RAW, 'cuda'))
m.load_state_dict(syn.synthetic_state_dict(manifest['state_dict'], 1))
m = m.cuda().train()
rets = []
orig_G = m.G.forward
def capG(pc, img, ret, *a, **k):
    rets.append({kk: (v.detach().clone() if torch.is_tensor(v) else v) for kk, v in ret.items()})
    return orig_G(pc, img, ret, *a, **k)
m.G.forward = capG
with torch.no_grad():
    m(*gpu); m(*gpu)
d = float((rets[0]['efh_cam_T_velo'] - rets[1]['efh_cam_T_velo']).abs().max())
print('efh_cam_T_velo run-to-run max abs diff %.3e (values ~%.1f)' % (d, float(rets[0]['efh_cam_T_velo'].abs().max())))
gp = [p for n, p in m.named_parameters() if n.startswith('G.')]
w = torch.tensor([1.0, -0.7, 0.3], device='cuda').view(1, 3, 1)
img_nhwc = ops.nchw_to_nhwc(gpu[1], 4)
def grads(ret):
    out = orig_G(gpu[0], gpu[1], dict(ret), False, img_nhwc=img_nhwc)
    return [g.double() for g in torch.autograd.grad((out['g_trs'] * w).sum(), gp, allow_unused=True) if g is not None]
rel = lambda a, c: (sum(float((x - y).pow(2).sum()) for x, y in zip(a, c)) / sum(float(y.pow(2).sum()) for y in c)) ** 0.5
g00, g01, g1 = grads(rets[0]), grads(rets[0]), grads(rets[1])
print('same frozen inputs, two evaluations: rel %.3e' % rel(g00, g01))
print('inputs of run 1 vs inputs of run 2:   rel %.3e' % rel(g1, g00))
fd0, _ = ops.depth_image(gpu[0], rets[0]['efh_cam_T_velo'], 768, 2560); fd1, _ = ops.depth_image(gpu[0], rets[1]['efh_cam_T_velo'], 768, 2560)
print('depth-image pixels that differ between the two runs:', int(((fd0 - fd1).abs().amax(-1) > 0).sum()), 'of', fd0.shape[1] * fd0.shape[2])
