"""GPU-side sample preparation: drop-in for the reference's `ProcessKITTIODOM` /
`ProcessRELLIS` callables (data_loader/kitti_odom_loader.py:237-273, rellis3d_loader.py:292-339) and the
`preproc_*` helpers they call (data_loader/loader_utils.py:63-202).

Same constructor (`args` dict), same `__call__(pcd, img, calibs, posej_T_posei, fname, rand_init=None)` and the same
return tuple `(pc[:3], img, calib, A, gts, fname)` — but the pixel and point work runs in libefgh_hip.so
(csrc/prep.hip) and the results are CUDA tensors (no CPU fallback).  Host code keeps what is O(1) per sample: the random
mis-calibration draw, 4x4 matrix algebra in float64, and the geometry of the two Pillow operations
(`Image.rotate(expand=True)` -> 16.16 affine coefficients, `Image.resize` BICUBIC -> integer coefficient tables),
restated from Pillow's published algorithm so that the kernels reproduce its pixels exactly.

One deliberate difference: the reference sub-samples with `np.random.choice(..., replace=False)` (numpy's global
Mersenne stream).  Here the subset is `sampled_indices` when given (parity tests pass the reference's draw) and a
`torch.randperm` on the device otherwise.
"""
import ctypes
import math
import random
from math import cos, pi, sin

import numpy as np
import torch

from .. import _C
from .._C import c_float, c_int32, ptr


def _L():
    return _C.lib()


def _st():
    return _C.stream_ptr()


# ------------------------------------------------------------------------------------------------
# host-side O(1) pieces
# ------------------------------------------------------------------------------------------------
def rand_init_params(rand_init, rpy_range, xyz_range, t_range):
    """loader_utils.py:63-79 (python `random`, 7 draws in this order)"""
    if rand_init is not None:
        return tuple(rand_init)
    if rpy_range is None or xyz_range is None or t_range is None:
        raise ValueError('rand_init_params: neither rand_init nor the ranges are given')
    rr = (random.random() * 2. - 1.) * pi * rpy_range
    rp = (random.random() * 2. - 1.) * pi * rpy_range
    ry = (random.random() * 2. - 1.) * pi * rpy_range
    tx = (random.random() * 2. - 1.) * xyz_range
    ty = (random.random() * 2. - 1.) * xyz_range
    tz = (random.random() * 2. - 1.) * xyz_range
    rt = (random.random() * 2. - 1.) * pi * t_range
    return rr, rp, ry, tx, ty, tz, rt


def _rpy(roll, pitch, yaw):            # numpy_utils.py:519-548
    ym = np.array([[cos(yaw), -sin(yaw), 0], [sin(yaw), cos(yaw), 0], [0, 0, 1]])
    pm = np.array([[cos(pitch), 0, sin(pitch)], [0, 1, 0], [-sin(pitch), 0, cos(pitch)]])
    rm = np.array([[1, 0, 0], [0, cos(roll), -sin(roll)], [0, sin(roll), cos(roll)]])
    R4 = np.eye(4)
    R4[:3, :3] = ym @ pm @ rm
    return R4


def preproc_gt(rr, rp, ry, tx, ty, tz, rt, posej_T_posei=np.eye(4)):
    """loader_utils.py:81-102 (float64 on the host)"""
    ltrs = np.eye(4)
    ltrs[:3, 3] = (tx, ty, tz)
    rand_init_l = np.array(_rpy(rr, rp, ry) @ ltrs)
    rand_init_c = np.array([[cos(rt), -sin(rt), 0], [sin(rt), cos(rt), 0], [0, 0, 1]])
    return {'rand_init_l': rand_init_l, 'rand_init_c': rand_init_c,
            'sensor2_T_sensor1': posej_T_posei @ np.linalg.inv(rand_init_l),
            'intrinsic_sensor2': np.array(np.linalg.inv(rand_init_c))}


def _fix16(v):
    return int(math.floor(v * 65536.0 + 0.5))


def rotate_plan(w, h, angle_deg):
    """Pillow `Image.rotate(angle, expand=True)` NEAREST: ((nw, nh), six 16.16 coefficients) of the destination->source map
    (Image.py:rotate, Geometry.c:affine_fixed); the 0/90/180/270 fast paths are the same map with exact coefficients."""
    angle = angle_deg % 360.0
    one, half = 65536, 32768
    if angle == 0:
        return (w, h), (one, 0, half, 0, one, half)
    if angle == 180:
        return (w, h), (-one, 0, (w - 1) * one + half, 0, -one, (h - 1) * one + half)
    if angle == 90:                      # counter-clockwise: out[y][x] = in[x][w-1-y]
        return (h, w), (0, -one, (w - 1) * one + half, one, 0, half)
    if angle == 270:                     # out[y][x] = in[h-1-x][y]
        return (h, w), (0, one, half, -one, 0, (h - 1) * one + half)
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]

    def tf(x, y):
        return m[0] * x + m[1] * y + m[2], m[3] * x + m[4] * y + m[5]

    cx, cy = w / 2, h / 2
    m[2], m[5] = tf(-cx, -cy)
    m[2] += cx
    m[5] += cy
    xx, yy = zip(*[tf(x, y) for x, y in ((0, 0), (w, 0), (w, h), (0, h))])
    nw = math.ceil(max(xx)) - math.floor(min(xx))
    nh = math.ceil(max(yy)) - math.floor(min(yy))
    m[2], m[5] = tf(-(nw - w) / 2.0, -(nh - h) / 2.0)
    return (nw, nh), (_fix16(m[0]), _fix16(m[1]), _fix16(m[2] + m[0] * 0.5 + m[1] * 0.5),
                      _fix16(m[3]), _fix16(m[4]), _fix16(m[5] + m[3] * 0.5 + m[4] * 0.5))


_COEFF_CACHE = {}


def resample_tables(in_size, out_size, device):
    """Pillow Resample.c:precompute_coeffs + normalize_coeffs_8bpc (bicubic, a = -0.5, 22 fractional bits)"""
    key = (in_size, out_size, str(device))
    hit = _COEFF_CACHE.get(key)
    if hit is not None:
        return hit
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)

    def bicubic(x):
        x = abs(x)
        if x < 1.0:
            return ((-0.5 + 2.0) * x - (-0.5 + 3.0)) * x * x + 1
        if x < 2.0:
            return (((x - 5) * x + 8) * x - 4) * -0.5
        return 0.0

    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in k:
            ww += v
        if ww != 0.0:
            k = [v / ww for v in k]
        for x, v in enumerate(k):
            kk[xx, x] = int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22))
        bounds[xx] = (xmin, xmax)
    val = (torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device), ksize)
    _COEFF_CACHE[key] = val
    return val


# ------------------------------------------------------------------------------------------------
# device ops (uint8 (H,W,3) images)
# ------------------------------------------------------------------------------------------------
def _as_dev_u8(img, device):
    t = torch.as_tensor(np.ascontiguousarray(img)) if not torch.is_tensor(img) else img
    if t.dim() == 3 and t.shape[2] != 3:                 # (3,H,W) -> (H,W,3), as the reference helpers accept
        t = t.permute(1, 2, 0)
    t = t.to(device=device, dtype=torch.uint8).contiguous()
    _C.require_cuda(t)
    return t


def rotate_expand(img, angle_deg):
    h, w = img.shape[:2]
    (nw, nh), coef = rotate_plan(w, h, float(angle_deg))
    out = torch.empty((nh, nw, 3), dtype=torch.uint8, device=img.device)
    c6 = (ctypes.c_int64 * 6)(*coef)
    _C.check(_L().efgh_prep_affine_nearest_u8(ptr(img), c_int32(h), c_int32(w), c6, ptr(out), c_int32(nh), c_int32(nw), _st()))
    return out


def crop_image(img, target_hw, init=False):
    """numpy_utils.py:447-472 (zero pad up to the target, then crop: centred, or from the origin when `init`)"""
    h, w = img.shape[:2]
    th, tw = int(target_hw[0]), int(target_hw[1])
    ph, pw = max(h, th), max(w, tw)
    i0, j0 = int(math.floor((ph - h) / 2.)), int(math.floor((pw - w) / 2.))
    i, j = (0, 0) if init else (int(math.floor((ph - th) / 2.)), int(math.floor((pw - tw) / 2.)))
    out = torch.empty((th, tw, 3), dtype=torch.uint8, device=img.device)
    _C.check(_L().efgh_prep_crop_pad_u8(ptr(img), c_int32(h), c_int32(w), c_int32(i - i0), c_int32(j - j0), ptr(out),
                                        c_int32(th), c_int32(tw), _st()))
    return out


def resize_image(img, target_hw):
    """numpy_utils.py:474-486: `Image.resize((W, H))`, default BICUBIC with antialiasing, two 8-bit passes"""
    th, tw = int(target_hw[0]), int(target_hw[1])
    cur = img
    for axis, size in ((1, tw), (0, th)):                # horizontal pass first, as Pillow does
        h, w = cur.shape[:2]
        if size == (w if axis else h):
            continue
        bounds, kk, ksize = resample_tables(w if axis else h, size, img.device)
        out = torch.empty((h, size, 3) if axis else (size, w, 3), dtype=torch.uint8, device=img.device)
        _C.check(_L().efgh_prep_resample_u8(ptr(cur), c_int32(h), c_int32(w), c_int32(axis), c_int32(size), ptr(bounds),
                                            ptr(kk), c_int32(ksize), ptr(out), _st()))
        cur = out
    return cur.clone() if cur is img else cur


def _chw_and_mask(img, want_chw=True, want_mask=False):
    h, w = img.shape[:2]
    chw = torch.empty((3, h, w), dtype=torch.uint8, device=img.device) if want_chw else None
    mask = torch.empty((1, h, w), dtype=torch.uint8, device=img.device) if want_mask else None
    _C.check(_L().efgh_prep_hwc_to_chw_u8(ptr(img), c_int32(h), c_int32(w), ptr(chw), ptr(mask), _st()))
    return chw, mask


def preproc_img(img, gts, raw_cam_img_size, rellis=False, device='cuda'):
    """loader_utils.py:104-130 (KITTI) / :132-158 (`rellis=True`: the raw view is a resize, not a crop)"""
    raw_hw = (int(raw_cam_img_size[0]), int(raw_cam_img_size[1]))
    img = _as_dev_u8(img, device)
    img_raw = resize_image(img, raw_hw) if rellis else crop_image(img, raw_hw, init=True)
    rc = gts['rand_init_c']
    rot_deg = math.degrees(np.arctan2(rc[1, 0], rc[0, 0]))                       # numpy_utils.py:436
    img_rot = crop_image(rotate_expand(img, rot_deg), raw_hw)
    half = (int(img_rot.shape[0] / 2), int(img_rot.shape[1] / 2))
    small = resize_image(img_rot, half)
    th, tw = int(raw_hw[0] / 2), int(raw_hw[1] / 2)
    img_in = torch.empty((3, th, tw), dtype=torch.float32, device=img.device)
    oy, ox = int(math.floor((th - half[0]) / 2.)), int(math.floor((tw - half[1]) / 2.))
    _C.check(_L().efgh_prep_u8_to_f32_chw_pad(ptr(small), c_int32(half[0]), c_int32(half[1]), c_int32(oy), c_int32(ox),
                                              ptr(img_in), c_int32(th), c_int32(tw), _st()))
    raw_chw, _ = _chw_and_mask(img_raw)
    rot_chw, mask = _chw_and_mask(img_rot, want_mask=True)
    return {'in': img_in, 'raw': raw_chw, 'rot': rot_chw, 'img_mask': mask}


def lidar_line_indices(n_points, reduce_to):
    """the index list of `reduce_lidar_line` (loader_utils.py:162-175) incl. its python negative indexing"""
    rate = 64 / reduce_to
    line_num = int(n_points / 64)
    lo, hi = int(-line_num / 2), int(line_num / 2)
    rows = np.array([i for i in range(64) if i % rate == 0], np.int64)
    idx = (rows[:, None] * line_num + np.arange(lo, hi, dtype=np.int64)[None, :]).reshape(-1)
    return np.where(idx < 0, idx + n_points, idx).astype(np.int32)


def preproc_pcd(pcd, gts, num_points, lidar_line=None, radius=50., sampled_indices=None, flip_xy=False, device='cuda',
                want_float64=False):
    """loader_utils.py:160-202: (lidar-line reduction,) radius filter, sub-sample without replacement (or zero pad),
    homogeneous transform by `rand_init_l` in float64.  Returns (3, num_points) float32 on the device
    (the reference's float64 rows are cast by the training loop, iterater.py:29)."""
    p = torch.as_tensor(np.ascontiguousarray(pcd, dtype=np.float32)) if not torch.is_tensor(pcd) else pcd
    p = p.to(device=device, dtype=torch.float32).contiguous()
    _C.require_cuda(p)
    assert p.dim() == 2 and p.shape[1] == 4
    n_src = p.shape[0]
    pre = None
    n = n_src
    if lidar_line is not None:
        pre = torch.from_numpy(lidar_line_indices(n_src, lidar_line)).to(p.device)
        n = int(pre.numel())
    keep = torch.empty(n, dtype=torch.int32, device=p.device)
    count = torch.zeros(1, dtype=torch.int32, device=p.device)
    if radius is not None:
        scratch = torch.empty(_L().efgh_prep_filter_blocks(c_int32(n)), dtype=torch.int32, device=p.device)
        _C.check(_L().efgh_prep_radius_filter(ptr(p), ptr(pre), c_int32(n), c_int32(1 if flip_xy else 0), c_float(radius),
                                              ptr(scratch), ptr(keep), ptr(count), _st()))
        n_keep = int(count.item())                       # one host read per sample: the branch below depends on it
    else:
        keep = pre if pre is not None else torch.arange(n, dtype=torch.int32, device=p.device)
        n_keep = n
    sel = None
    if num_points < n_keep:
        if sampled_indices is None:
            sel = torch.randperm(n_keep, device=p.device)[:num_points].to(torch.int32)
        else:
            sel = torch.as_tensor(np.asarray(sampled_indices)).to(device=p.device, dtype=torch.int32)
            assert sel.numel() == num_points
        n_sel = num_points
    else:
        n_sel = n_keep
    T = torch.from_numpy(np.ascontiguousarray(np.asarray(gts['rand_init_l'], np.float64)[:3, :])).to(p.device)
    out32 = torch.empty((3, num_points), dtype=torch.float32, device=p.device)
    out64 = torch.empty((3, num_points), dtype=torch.float64, device=p.device) if want_float64 else None
    _C.check(_L().efgh_prep_gather_transform(ptr(p), ptr(keep), ptr(sel), c_int32(n_sel), c_int32(1 if flip_xy else 0),
                                             ptr(T), c_int32(num_points), ptr(out32), ptr(out64), _st()))
    return (out32, out64) if want_float64 else out32


# ------------------------------------------------------------------------------------------------
# the reference's callables
# ------------------------------------------------------------------------------------------------
class _Process(object):
    rellis = False

    def __init__(self, args):
        self.raw_cam_img_size = args['raw_cam_img_size']
        self.lidar_line = args['lidar_line']
        self.num_points = args['num_points']
        self.device = args.get('DEVICE', 'cuda')
        if args['test'] is False:
            self.l_rot_range = args['dclb']['l_rot_range']
            self.l_trs_range = args['dclb']['l_trs_range']
            self.c_rot_range = args['dclb']['c_rot_range']
        else:
            self.l_rot_range, self.l_trs_range, self.c_rot_range = None, None, None

    def _calib(self, calibs):
        raise NotImplementedError

    def __call__(self, pcd, img, calibs, posej_T_posei, fname, rand_init=None, sampled_indices=None):
        rr, rp, ry, tx, ty, tz, rt = rand_init_params(rand_init, self.l_rot_range, self.l_trs_range, self.c_rot_range)
        gts = preproc_gt(rr, rp, ry, tx, ty, tz, rt, posej_T_posei)
        imgs = preproc_img(img, gts, self.raw_cam_img_size, self.rellis, self.device)
        pc = preproc_pcd(pcd, gts, self.num_points, self.lidar_line, sampled_indices=sampled_indices,
                         flip_xy=self.rellis, device=self.device)
        gts['img_raw'], gts['img_rot'], gts['img_mask'] = imgs['raw'], imgs['rot'], imgs['img_mask']
        A = np.array([[1, 0, -self.raw_cam_img_size[1] / 2], [0, 1, -self.raw_cam_img_size[0] / 2], [0, 0, 1]])
        calib = self._calib(calibs)[:3, :]
        gts['cam_T_velo'] = np.linalg.inv(A) @ gts['intrinsic_sensor2'] @ A @ calib @ gts['sensor2_T_sensor1']
        return pc, imgs['in'], calib, A, gts, fname


class ProcessKITTIODOM(_Process):
    """data_loader/kitti_odom_loader.py:237-273"""

    def _calib(self, calibs):
        return calibs['P2'] @ calibs['Tr']


class ProcessRELLIS(_Process):
    """data_loader/rellis3d_loader.py:292-339: the sweep is first rotated by diag(-1,-1,1) (an exact sign flip, folded
    into the point kernels) and the calibration carries its inverse"""
    rellis = True

    def _calib(self, calibs):
        R = np.array([[-1, 0, 0, 0], [0, -1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
        return calibs['P'] @ calibs['Tr'] @ np.linalg.inv(R)
