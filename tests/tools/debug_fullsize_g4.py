"""is d(sum g_trs)/d(G weights) well defined in fp32?  oracle gnet in float32 vs float64 on the same inputs (CPU only)"""
import json, os, re, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from efgh_amd import synthetic as syn
from oracle import efgh_oracle as O
RAW, NPTS = (768, 2560), 131072
manifest = json.load(open('tests/golden/state_dict_manifest.json'))
b = syn.make_batch(RAW, NPTS, 1)
T = torch.from_numpy
cpu = [T(b[k]) for k in ('pc', 'img', 'calib', 'A')]
torch.set_num_threads(32)
args = syn.default_args(RAW, 'cpu')
P32 = syn.synthetic_state_dict(manifest['state_dict'], 1)
with torch.no_grad():
    r = dict(O.enet(P32, cpu[0], True)); r.update(O.hnet(P32, cpu[1], True)); r['network'] = 'EH'
    r['eh_cam_T_velo'] = O.compute_cam_T_velo(r['intrinsic_sensor2'], r['sensor2_T_sensor1'], cpu[2], cpu[3])
    rf = O.fnet(P32, cpu[0], r, args, True)
    rf['efh_cam_T_velo'] = O.compute_cam_T_velo(rf['intrinsic_sensor2'], rf['sensor2_T_sensor1'], cpu[2], cpu[3])
skip = re.compile(r'(features\.\d+|conv_gn_\d|conv_hrzn_\d|E\.bcn5\.blur_conv\.2)\.bias$')
gnames = [k for k in manifest['parameters'] if k.startswith('G.') and not skip.search(k)]
grads = {}
_orig_depth = O.depth_image
_otm = O.translation_matrix
O.translation_matrix = lambda v: _otm(v.float()).to(v.dtype)
for dt in (torch.float32, torch.float64):
    t0 = time.time()
    P = {k: (v.detach().to(dt) if v.is_floating_point() else v.detach().clone()) for k, v in P32.items()}
    for k in gnames: P[k].requires_grad_(True)
    rr = {k: (v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in rf.items()}
    O.depth_image = lambda *a, _dt=dt, **k: _orig_depth(*[x.float() if torch.is_tensor(x) and x.is_floating_point() else x for x in a], **k).to(_dt)
    out = O.gnet(P, cpu[0].to(dt), rr['h_img'], rr, args, True)
    w = torch.tensor([1.0, -0.7, 0.3], dtype=dt).view(1, 3, 1)
    (out['g_trs'] * w).sum().backward()
    grads[dt] = [(torch.zeros_like(P[k]).double() if P[k].grad is None else P[k].grad.double().clone()) for k in gnames]
    print(dt, 'g_trs', out['g_trs'].flatten().tolist(), '%.1f s' % (time.time() - t0))
num = sum(float((a - c).pow(2).sum()) for a, c in zip(grads[torch.float32], grads[torch.float64]))
den = sum(float(c.pow(2).sum()) for c in grads[torch.float64])
print('oracle fp32 vs oracle fp64, d(w.g_trs)/d(G params): rel err %.3e' % ((num / den) ** 0.5))
for k, a, c in list(zip(gnames, grads[torch.float32], grads[torch.float64]))[:6]:
    print('  %-40s rel %.3e' % (k, float((a - c).norm() / max(float(c.norm()), 1e-300))))
