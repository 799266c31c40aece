"""Generates tests/golden/summary_cases.npz by running the UNMODIFIED reference (`common/numpy_utils.py:image_draw`,
`eval_image_draw`) in this container (needs /root/reference, matplotlib, Pillow; never runs on the GPU box).
Inputs are small synthetic scenes; every output image of the two functions is stored next to its inputs."""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness                             # noqa: E402
ref_harness._install_stubs()                   # open3d & co. are imported by the reference module but not used on this path
sys.path.insert(0, '/root/reference')
import matplotlib
matplotlib.use('Agg')
import matplotlib.pyplot as plt
from common import numpy_utils as NU          # noqa: E402  (the reference)


def rot_z(a):
    return np.array([[math.cos(a), -math.sin(a), 0, 0], [math.sin(a), math.cos(a), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.]])


def rot_y(a):
    return np.array([[math.cos(a), 0, math.sin(a), 0], [0, 1, 0, 0], [-math.sin(a), 0, math.cos(a), 0], [0, 0, 0, 1.]])


def trans(x, y, z):
    m = np.eye(4); m[:3, 3] = [x, y, z]
    return m


def make_case(raw_hw, n, seed):
    rs = np.random.RandomState(seed)
    H, W = raw_hw
    nb = 16
    na = n // nb
    pitch = np.linspace(-0.25, 0.25, nb)[:, None]
    yaw = np.linspace(-math.pi, math.pi, na, endpoint=False)[None, :]
    r = 4 + 30 * rs.rand(nb, na)
    pc = np.float32(np.stack([r * np.cos(pitch) * np.cos(yaw), r * np.cos(pitch) * np.sin(yaw), r * np.sin(pitch)]).reshape(3, -1))
    f = W / 2.0
    K = np.array([[f, 0, W / 2, 0], [0, f, H / 2, 0], [0, 0, 1, 0.]])
    T = np.array([[0., -1, 0, 0], [0, 0, -1, 0], [1, 0, 0, 0], [0, 0, 0, 1]])
    calib = np.float32((K @ T)[:3])
    A = np.float32([[1., 0, -W / 2], [0, 1., -H / 2], [0, 0, 1.]])
    h, w = H // 2, W // 2
    img = rs.randint(0, 256, size=(3, h, w)).astype(np.float32)

    def h_c(a):
        return np.float32([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1.]])
    gt = {'img_raw': rs.randint(0, 256, size=(3, H, W)).astype(np.float32),
          'img_rot': rs.randint(0, 256, size=(3, H - 4, W + 6)).astype(np.float32),       # resized to raw by the reference
          'e_l': np.float32(rot_y(0.05) @ rot_z(0.3)), 'f_l': np.float32(rot_z(-0.2)), 'g_l': np.float32(trans(0.3, -0.2, 0.1)),
          'h_c': h_c(0.12), 'f_score': np.float32(rs.rand(W * 2 - 7)),
          'g_depth': np.float32(rs.rand(1, H, W) * (rs.rand(1, H, W) > 0.8) * 40), 'g_mask': np.float32(rs.rand(1, H, W) > 0.7)}
    pe, pf, pg = np.float32(rot_y(0.04) @ rot_z(0.27)), np.float32(rot_z(-0.17)), np.float32(trans(0.25, -0.1, 0.0))
    ph = h_c(0.1)

    def cam_T(s2s1):
        return np.float32(np.linalg.inv(A) @ ph @ A @ calib @ s2s1)
    pred = {'network': 'EHFG', 'e_l': pe, 'f_l': pf, 'g_l': pg, 'h_c': ph, 'eh_cam_T_velo': cam_T(pe),
            'efh_cam_T_velo': cam_T(pf @ pe), 'efgh_cam_T_velo': cam_T(pg @ pf @ pe), 'f_score': np.float32(rs.rand(W * 2 - 7)),
            'g_depth': np.float32(rs.randn(1, H, W) * 3 + 10), 'g_mask': np.float32(rs.rand(2, H, W))}
    return {'pc': pc, 'img': img, 'calib': calib, 'A': A}, gt, pred


def batched(d):
    return {k: (torch.from_numpy(np.asarray(v))[None] if not isinstance(v, str) else v) for k, v in d.items()}


out = {}
cases = [('a', (48, 64), 1600, 1, 1), ('b', (40, 96), 960, 2, 3)]
fov = [0.125, -0.125]
for name, raw, n, seed, px in cases:
    inp, gt, pred = make_case(raw, n, seed)
    ti, tg, tp = batched(inp), batched(gt), batched(pred)
    draw = NU.image_draw(ti['pc'], ti['img'], ti['calib'], ti['A'], tg, tp, raw, fov, cmap=plt.cm.plasma)
    ev = NU.eval_image_draw(ti['pc'], ti['img'], ti['calib'], ti['A'], tg, tp, raw, fov, px, cmap=plt.cm.jet)
    out[name + '.meta'] = np.array([raw[0], raw[1], n, px])
    for k, v in inp.items():
        out['%s.in.%s' % (name, k)] = v
    for k, v in gt.items():
        out['%s.gt.%s' % (name, k)] = v
    for k, v in pred.items():
        if k != 'network':
            out['%s.pred.%s' % (name, k)] = v
    for k, v in draw.items():
        out['%s.draw.%s' % (name, k)] = np.asarray(v)
    for k, v in ev.items():
        out['%s.eval.%s' % (name, k)] = np.asarray(v)
    # the primitives on their own (float64 rasters before colouring)
    out[name + '.prim.depth'] = NU.depth_img_from_cartesian_pc_numpy(inp['pc'], pred['eh_cam_T_velo'], raw)
    out[name + '.prim.range'] = NU.range_img_from_cartesian_pc_numpy(inp['pc'], pred['e_l'], (raw[0] // 2, raw[1] * 2), fov)
    print(name, {k: np.asarray(v).shape for k, v in draw.items()}, {k: np.asarray(v).shape for k, v in ev.items()})
out['fov'] = np.array(fov)
dst = os.path.join(os.path.dirname(__file__), 'summary_cases.npz')
np.savez_compressed(dst, **out)
print(dst, os.path.getsize(dst))
