"""TensorBoard / evaluation overlay images on the GPU: `image_draw` (common/numpy_utils.py:8-179),
`eval_image_draw` (:181-297) and `update_summary` (common/helper.py:11-26) with the reference's signatures.  The rasters, the
raster-order colouring and the colour look-up run in csrc/summary.hip, the Pillow operations in csrc/prep.hip (data/prepare.py);
only the 3x3 / 4x4 pose matrices come to the host (their float32 products are formed with numpy exactly as the reference does).
The returned images are uint8 [H][W][3] tensors on the device; `update_summary` hands them to the writer as numpy arrays.
The colour maps are matplotlib's 256-entry tables shipped as data (colormaps.npz): 'plasma' and 'jet'."""
import ctypes
import math
import os

import numpy as np
import torch

from .. import _C
from ..data import prepare as P

_LUT = {}


def _L():
    return _C.lib()


def _st():
    return _C.stream_ptr()


def _lut(name, device):
    key = (name, str(device))
    if key not in _LUT:
        data = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'colormaps.npz'))
        if name not in data.files:
            raise _C.EfghError('unknown colour map %r (shipped: %s)' % (name, ', '.join(data.files)))
        _LUT[key] = torch.from_numpy(data[name]).to(device).contiguous()
    return _LUT[key]


def _np0(t):
    """sample 0 of a batched tensor as numpy, as `x.cpu().detach().numpy()[0]` (small pose matrices only)"""
    return t.detach().cpu().numpy()[0]


class _PaintJob(ctypes.Structure):
    _fields_ = [('inp', ctypes.c_void_p), ('out', ctypes.c_void_p), ('H', ctypes.c_int32), ('W', ctypes.c_int32),
                ('px', ctypes.c_int32), ('_pad', ctypes.c_int32)]


class _Painter:
    """collects minmax_color_img_from_img_numpy calls so that all of them run as ONE launch (one workgroup per image)"""

    def __init__(self, lut):
        self.lut, self.jobs = lut, []

    def add(self, img, px=2):
        """img: uint8 / float32 / float64 [H][W] on the device -> handle; normalisation in the image's own arithmetic"""
        if img.dtype == torch.uint8:
            mn, mx = img.amin(), img.amax()
            n = (img - mn).double() / (mx - mn).double()
        else:
            n = ((img - img.amin()) / (img.amax() - img.amin())).double()
        n = n.contiguous()
        out = torch.zeros_like(n)
        self.jobs.append((n, out, int(px)))
        return len(self.jobs) - 1

    def run(self):
        if not self.jobs:
            return []
        arr = (_PaintJob * len(self.jobs))()
        for k, (n, out, px) in enumerate(self.jobs):
            arr[k].inp, arr[k].out, arr[k].H, arr[k].W, arr[k].px = n.data_ptr(), out.data_ptr(), n.shape[0], n.shape[1], px
        dev = self.jobs[0][0].device
        jobs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        _C.check(_L().efgh_sum_paint(_C.ptr(jobs), ctypes.c_int32(len(self.jobs)), _st()))
        res = []
        for n, out, _ in self.jobs:
            H, W = out.shape
            rgb = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
            mask = torch.empty((H, W), dtype=torch.uint8, device=dev)
            _C.check(_L().efgh_sum_colorize(_C.ptr(out), ctypes.c_int64(H * W), _C.ptr(self.lut), _C.ptr(rgb), _C.ptr(mask), _st()))
            res.append((rgb, mask))
        return res


def _T34(T, dev):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(T, dtype=np.float64)[:3].reshape(-1))).to(dev)


def depth_image_last(pc0, T34, raw_hw):
    """numpy_utils.py:338-358 for one sample: pc0 (3,N) float32 on the device, T34 numpy 3x4 -> uint8 [H][W]"""
    H, W = int(raw_hw[0]), int(raw_hw[1])
    dev = pc0.device
    ws = torch.empty(H * W, dtype=torch.int32, device=dev)
    out = torch.empty((H, W), dtype=torch.uint8, device=dev)
    _C.check(_L().efgh_sum_depth_last(_C.ptr(pc0), ctypes.c_int64(pc0.stride(0)), ctypes.c_int32(pc0.shape[1]), _C.ptr(_T34(T34, dev)),
                                      ctypes.c_int32(H), ctypes.c_int32(W), _C.ptr(ws), _C.ptr(out), _st()))
    return out


def range_image_last(pc0, T44, rng_hw, fov):
    """numpy_utils.py:299-336 for one sample -> float64 [H][W]"""
    H, W = int(rng_hw[0]), int(rng_hw[1])
    dev = pc0.device
    ws = torch.empty(H * W, dtype=torch.int32, device=dev)
    out = torch.empty((H, W), dtype=torch.float64, device=dev)
    _C.check(_L().efgh_sum_range_last(_C.ptr(pc0), ctypes.c_int64(pc0.stride(0)), ctypes.c_int32(pc0.shape[1]), _C.ptr(_T34(T44, dev)),
                                      ctypes.c_int32(H), ctypes.c_int32(W), ctypes.c_double(fov[0] * math.pi),
                                      ctypes.c_double(fov[1] * math.pi), _C.ptr(ws), _C.ptr(out), _st()))
    return out


def _u8_hwc(t0, dev):
    """`x.numpy()[0].astype('uint8')`, channels last"""
    t = t0.to(dev)
    if t.dtype != torch.uint8:
        t = t.to(torch.int64).to(torch.uint8)               # float -> uint8 truncation as astype does for in-range values
    if t.shape[2] != 3:
        t = t.permute(1, 2, 0)
    return t.contiguous()


def _rot_deg(mat):
    return math.degrees(np.arctan2(mat[1, 0], mat[0, 0]))     # numpy_utils.py:436


def _prep(pcd, img, calib, A):
    _C.require_cuda(pcd, img)
    _C.require_f32(pcd)
    pc0 = pcd[0]
    if pc0.stride(1) != 1:
        pc0 = pc0.contiguous()
    return pc0, _np0(calib), _np0(A)


def image_draw(pcd, img, calib, A, gt, pred, raw_cam_img_size, lidar_fov_rad, cmap='plasma'):
    """numpy_utils.py:8-179 (sample 0 of the batch) -> {'cam', 'score', 'dimage', 'mask', 'range', 'depth'} (for a full EHFG pass)"""
    pc0, calib0, A0 = _prep(pcd, img, calib, A)
    dev = pc0.device
    raw = (int(raw_cam_img_size[0]), int(raw_cam_img_size[1]))
    net_hw, rng_hw = (int(raw[0] / 2), int(raw[1] / 2)), (int(raw[0] / 2), int(raw[1] * 2))
    paint = _Painter(_lut(cmap, dev))
    in_img = P.crop_image(_u8_hwc(img[0], dev), net_hw)
    cam_rot = P.resize_image(_u8_hwc(gt['img_rot'][0], dev), raw)
    ge, gf, gg, gh = (_np0(gt[k]) for k in ('e_l', 'f_l', 'g_l', 'h_c'))
    gt_s2s1 = gg @ gf @ ge
    gt_T = np.linalg.inv(A0) @ gh @ A0 @ calib0 @ gt_s2s1
    net = pred['network']
    depth_jobs, range_jobs, plain_jobs = {}, {}, {}

    def depth(name, T, px=2):
        depth_jobs[name] = paint.add(depth_image_last(pc0, T, raw), px)

    def rng(name, T):
        range_jobs[name] = paint.add(range_image_last(pc0, T, rng_hw, lidar_fov_rad))
    depth('in', calib0); depth('gt', gt_T)
    rng('in', np.eye(4)); rng('gt', gt_s2s1)
    if 'E' in net:
        pe = _np0(pred['e_l'])
        rng('E', pe)
    if 'E' in net and 'H' in net:
        depth('EH', _np0(pred['eh_cam_T_velo']))
    if 'F' in net:
        pf = _np0(pred['f_l'])
        rng('EF', pf @ pe)
        depth('EFH', _np0(pred['efh_cam_T_velo']))
        for who, d in (('gt', gt), ('pred', pred)):
            plain_jobs['score_' + who] = paint.add(d['f_score'][0].to(dev)[None, :].repeat(8, 1))
    if 'G' in net:
        rng('EFG', _np0(pred['g_l']) @ pf @ pe)
        depth('EFGH', _np0(pred['efgh_cam_T_velo']))
        for who, d in (('pred', pred), ('gt', gt)):
            plain_jobs['dimage_' + who] = paint.add(d['g_depth'][0][0].to(dev))
            plain_jobs['mask_' + who] = paint.add(d['g_mask'][0][0].to(dev))
    res = paint.run()
    over = {k: torch.where(res[j][1][:, :, None] != 0, res[j][0], cam_rot) for k, j in depth_jobs.items()}
    rr = {k: res[j][0] for k, j in range_jobs.items()}
    out = {}
    if 'H' in net:
        gt_img = P.crop_image(P.rotate_expand(in_img, _rot_deg(gh)), net_hw)
        img_h = P.crop_image(P.rotate_expand(in_img, _rot_deg(_np0(pred['h_c']))), net_hw)
        out['cam'] = torch.cat([in_img, img_h, gt_img], 0)
    if 'F' in net:
        out['score'] = torch.cat([P.resize_image(res[plain_jobs['score_gt']][0], rng_hw),
                                  P.resize_image(res[plain_jobs['score_pred']][0], rng_hw)], 0)
    if 'G' in net:
        out['dimage'] = torch.cat([res[plain_jobs['dimage_pred']][0], res[plain_jobs['dimage_gt']][0]], 0)
        out['mask'] = torch.cat([res[plain_jobs['mask_pred']][0], res[plain_jobs['mask_gt']][0]], 0)
    if 'E' in net and 'F' in net:
        out['range'] = torch.cat([rr['in'], rr['E'], rr['EF']] + ([rr['EFG']] if 'G' in net else []) + [rr['gt']], 0)
        out['depth'] = torch.cat([over['in'], over['EH'], over['EFH']] + ([over['EFGH']] if 'G' in net else []) + [over['gt']], 0)
    else:                                                    # the reference leaves the single images in the dict then
        if 'E' in net:
            out['pred_range_E'] = rr['E']
        if 'E' in net and 'H' in net:
            out['pred_depth_EH'] = over['EH']
        if 'F' in net:
            out['pred_range_EF'], out['pred_depth_EFH'] = rr['EF'], over['EFH']
        if 'G' in net:
            out['pred_range_EFG'], out['pred_depth_EFGH'] = rr['EFG'], over['EFGH']
    return out


def eval_image_draw(pcd, img, calib, A, gt, pred, raw_cam_img_size, lidar_fov_rad, px, cmap='jet'):
    """numpy_utils.py:181-297: the three predicted depth overlays, rotated by the predicted h_c and centre-cropped"""
    pc0, _, _ = _prep(pcd, img, calib, A)
    dev = pc0.device
    raw = (int(raw_cam_img_size[0]), int(raw_cam_img_size[1]))
    net_hw = (int(raw[0] / 2), int(raw[1] / 2))
    paint = _Painter(_lut(cmap, dev))
    cam_rot = P.resize_image(_u8_hwc(gt['img_rot'][0], dev), raw)
    names = (('pred_depth_EH', 'eh_cam_T_velo', px), ('pred_depth_EFH', 'efh_cam_T_velo', px), ('pred_depth_EFGH', 'efgh_cam_T_velo', 2))
    jobs = [paint.add(depth_image_last(pc0, _np0(pred[key]), raw), p) for _, key, p in names]
    res = paint.run()
    deg = _rot_deg(_np0(pred['h_c']))
    out = {}
    for (name, _, _), j in zip(names, jobs):
        over = torch.where(res[j][1][:, :, None] != 0, res[j][0], cam_rot)
        out[name] = P.crop_image(P.rotate_expand(over.contiguous(), deg), net_hw)
    return out


def update_summary(summary, mode, it, losses, errors, pcd, img, calib, A, gt, pred, raw_cam_img_size, lidar_fov_rad):
    """common/helper.py:11-26: scalars + images into a tensorboardX-style writer (`add_scalar`, `add_image` with CHW arrays)"""
    for k in list(losses.keys()):
        summary.add_scalar(mode + '_loss/' + k, losses[k].avg, it)
    for k in list(errors.keys()):
        summary.add_scalar(mode + '_error/' + k, errors[k], it)
    imgs = image_draw(pcd, img, calib, A, gt, pred, raw_cam_img_size, lidar_fov_rad)
    for k, v in imgs.items():
        a = v.cpu().numpy()
        if a.shape[2] == 3:
            a = np.transpose(a, (2, 0, 1))
        summary.add_image(mode + '_image/' + k, a, it)
