"""timing of the overlay images at the shipped RELLIS size (900x1600 raw, 65 536 points) on the GPU box"""
import math
import sys
import time
sys.path.insert(0, '/root/repo')
import numpy as np
import torch
from efgh_amd import synthetic as syn
from efgh_amd.common import summary as S

raw, n = (900, 1600), 65536
b = syn.make_batch(raw, n, 1)
dev = 'cuda'
pc, img, calib, A = [torch.from_numpy(b[k]).to(dev) for k in ('pc', 'img', 'calib', 'A')]
rs = np.random.RandomState(0)
eye4 = torch.eye(4, device=dev)[None]
def hc(a):
    return torch.tensor([[[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1.]]], device=dev)
gt = {'img_raw': torch.from_numpy(rs.randint(0, 256, (1, 3, 900, 1600)).astype(np.float32)).to(dev),
      'img_rot': torch.from_numpy(rs.randint(0, 256, (1, 3, 900, 1600)).astype(np.float32)).to(dev),
      'e_l': eye4, 'f_l': eye4, 'g_l': eye4, 'h_c': hc(0.05), 'f_score': torch.rand(1, 3193, device=dev),
      'g_depth': torch.rand(1, 1, 900, 1600, device=dev) * (torch.rand(1, 1, 900, 1600, device=dev) > 0.9),
      'g_mask': (torch.rand(1, 1, 900, 1600, device=dev) > 0.7).float()}
T = torch.from_numpy(b['gt']['cam_T_velo'].astype(np.float32)).to(dev)
pred = {'network': 'EHFG', 'e_l': eye4, 'f_l': eye4, 'g_l': eye4, 'h_c': hc(0.04), 'eh_cam_T_velo': T, 'efh_cam_T_velo': T,
        'efgh_cam_T_velo': T, 'f_score': torch.rand(1, 3193, device=dev), 'g_depth': torch.randn(1, 1, 900, 1600, device=dev) + 10,
        'g_mask': torch.rand(1, 2, 900, 1600, device=dev)}
for name, fn in (('image_draw', lambda: S.image_draw(pc, img, calib, A, gt, pred, raw, [0.125, -0.125])),
                 ('eval_image_draw', lambda: S.eval_image_draw(pc, img, calib, A, gt, pred, raw, [0.125, -0.125], 2))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
    print('%s: %.1f ms' % (name, (time.perf_counter() - t0) * 1e3), {k: tuple(v.shape) for k, v in out.items()})
