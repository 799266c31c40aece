import sys; sys.path.insert(0, '/root/repo')
import json, torch
from efgh_amd import synthetic as syn
from efgh_amd.nets import EFGHBackbone
RAW, NPTS = (768, 2560), 131072
manifest = json.load(open('/root/repo/tests/golden/state_dict_manifest.json'))
m = EFGHBackbone(syn.default_args(RAW, 'cuda')); m.load_state_dict(syn.synthetic_state_dict(manifest['state_dict'], 1)); m = m.cuda().eval()
b = syn.make_batch(RAW, NPTS, 2)
inp = [torch.from_numpy(b[k]).cuda() for k in ('pc', 'img', 'calib', 'A')]
with torch.no_grad():
    o1 = m(*inp); o2 = m(*inp)
print({k: bool(torch.equal(o1[k], o2[k])) for k in o1 if torch.is_tensor(o1[k])})
