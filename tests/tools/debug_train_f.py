import sys, json, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from efgh_amd import synthetic as syn
from efgh_amd.nets import EFGHBackbone
from oracle import efgh_oracle as O
RAW, NPTS = (128, 256), 2048
man = json.load(open('/root/repo/tests/golden/state_dict_manifest.json'))
train = True
m = EFGHBackbone(syn.default_args(RAW, 'cuda'))
m.load_state_dict(syn.synthetic_state_dict(man['state_dict'], 1)); m = m.cuda(); m.train(train)
b = syn.make_batch(RAW, NPTS, 1)
inp = [torch.from_numpy(b[k]).cuda() for k in ('pc', 'img', 'calib', 'A')]
keep = {}
with torch.no_grad():
    out = m(*inp, keep=keep)
P = syn.synthetic_state_dict(man['state_dict'], 1)
okeep = {}
with torch.no_grad():
    ref = O.forward(P, *[t.cpu() for t in inp], syn.default_args(RAW, 'cpu'), train=train, keep=okeep)
def rel(a, b): return float((a - b).abs().max() / (b.abs().max() + 1e-12))
er = keep['e_range'].permute(0, 3, 1, 2).cpu()
print('e_range mismatching px', int(((er - okeep['e_range']).abs().amax(1) > 1e-5).sum()), 'of', er.shape[2] * er.shape[3])
print('cam3', rel(keep['cam3'].permute(0, 3, 1, 2).cpu(), okeep['cam_feat'][0] * (okeep['cam_feat'][0].max() * 0 + 1)) if False else '')
# oracle keeps normalised feats; compare normalised
cam = keep['cam3'].permute(0, 3, 1, 2).cpu(); cam = cam / (cam.max() - cam.min())
rng = keep['rng3'].permute(0, 3, 1, 2).cpu(); rng = rng / (rng.max() - rng.min())
print('cam_feat', rel(cam, okeep['cam_feat'][0]), 'rng_feat', rel(rng, okeep['rng_feat'][0]))
print('f_logit', rel(keep['f_logit'].cpu(), okeep['f_logit'][0]), 'f_score', rel(out['f_score'].cpu(), ref['f_score']))
for k in ('e_gn_sgn', 'h_hrzn_sgn', 'g_trs', 'e_l', 'h_c'):
    print(k, rel(out[k].cpu(), ref[k]))
# teacher-forced: feed the oracle's e_range to our range trunk
from efgh_amd.nets import layers as L
ctx = L.Ctx(train)
m.load_state_dict(syn.synthetic_state_dict(man['state_dict'], 1)); m.train(train)
er_o = okeep['e_range'].permute(0, 2, 3, 1).contiguous().cuda()
with torch.no_grad():
    r0 = L.run_conv_bn_relu(ctx, m.F.conv_range, er_o)
    rng_tf = m.F._trunk(ctx, r0, 'range')
rng_tf = rng_tf.permute(0, 3, 1, 2).cpu(); rng_tf = rng_tf / (rng_tf.max() - rng_tf.min())
print('teacher-forced rng_feat', rel(rng_tf, okeep['rng_feat'][0]))
