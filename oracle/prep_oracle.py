"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy, integers + float64) of the reference's per-sample
preparation: `data_loader/loader_utils.py:63-202` and the `Process*.__call__` bodies in
`data_loader/kitti_odom_loader.py:237-273` / `rellis3d_loader.py:292-339`, including the two Pillow operations they
rely on (`Image.rotate(expand=True)` NEAREST and `Image.resize` BICUBIC), restated from Pillow's published algorithm
(Pillow 12.2: `Image.rotate`, `Geometry.c:affine_fixed`, `Resample.c:precompute_coeffs / normalize_coeffs_8bpc /
ImagingResampleHorizontal_8bpc`).  Pinned by tests/test_oracle_prep.py against fixtures produced by the unmodified
reference (tests/golden/make_golden_prep.py) and against Pillow itself.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import math

import numpy as np


# ------------------------------------------------------------------------------------------------
# Pillow pieces
# ------------------------------------------------------------------------------------------------
def _fix16(v):
    return int(math.floor(v * 65536.0 + 0.5))


def rotate_geometry(w, h, angle_deg, expand=True):
    """Image.rotate's affine matrix (destination -> source) and output size.  Returns (kind, matrix, (nw, nh)):
    kind in {'copy', 'rot180', 'rot90', 'rot270', 'affine'}."""
    angle = angle_deg % 360.0
    if angle == 0:
        return 'copy', None, (w, h)
    if angle == 180:
        return 'rot180', None, (w, h)
    if angle in (90, 270) and (expand or w == h):
        return ('rot90' if angle == 90 else 'rot270'), None, (h, w)
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]

    def tf(x, y):
        return m[0] * x + m[1] * y + m[2], m[3] * x + m[4] * y + m[5]

    cx, cy = w / 2, h / 2
    m[2], m[5] = tf(-cx, -cy)
    m[2] += cx
    m[5] += cy
    nw, nh = w, h
    if expand:
        xx, yy = [], []
        for x, y in ((0, 0), (w, 0), (w, h), (0, h)):
            tx, ty = tf(x, y)
            xx.append(tx)
            yy.append(ty)
        nw = math.ceil(max(xx)) - math.floor(min(xx))
        nh = math.ceil(max(yy)) - math.floor(min(yy))
        m[2], m[5] = tf(-(nw - w) / 2.0, -(nh - h) / 2.0)
    return 'affine', m, (nw, nh)


def affine_fixed_coeffs(m):
    """Geometry.c:affine_fixed — 16.16 fixed point; the half-pixel centre is folded into the offsets"""
    return (_fix16(m[0]), _fix16(m[1]), _fix16(m[2] + m[0] * 0.5 + m[1] * 0.5),
            _fix16(m[3]), _fix16(m[4]), _fix16(m[5] + m[3] * 0.5 + m[4] * 0.5))


def pil_rotate_nearest_u8(img, angle_deg, expand=True):
    """numpy_utils.py:426-445 — `Image.fromarray(img).rotate(rot_deg, expand=True)`; img (H,W,3) uint8"""
    h, w = img.shape[:2]
    kind, m, (nw, nh) = rotate_geometry(w, h, float(angle_deg), expand)
    if kind == 'copy':
        return img.copy()
    if kind == 'rot180':
        return img[::-1, ::-1].copy()
    if kind == 'rot90':                         # Transpose.ROTATE_90: counter-clockwise
        return np.ascontiguousarray(np.transpose(img, (1, 0, 2))[::-1])
    if kind == 'rot270':
        return np.ascontiguousarray(np.transpose(img, (1, 0, 2))[:, ::-1])
    a0, a1, a2, a3, a4, a5 = affine_fixed_coeffs(m)
    xs = np.arange(nw, dtype=np.int64)[None, :]
    ys = np.arange(nh, dtype=np.int64)[:, None]
    xin = (a2 + a0 * xs + a1 * ys) >> 16
    yin = (a5 + a3 * xs + a4 * ys) >> 16
    ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    out = np.zeros((nh, nw, img.shape[2]), np.uint8)
    out[ok] = img[yin[ok], xin[ok]]
    return out


PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size, out_size):
    """Resample.c:precompute_coeffs + normalize_coeffs_8bpc for the bicubic filter over the whole axis.
    Returns (bounds int32 [out][2] = (first input index, count), coeffs int32 [out][ksize], ksize)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in k:
            ww += v
        if ww != 0.0:
            k = [v / ww for v in k]
        for x, v in enumerate(k):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _resample_axis(img, out_size, axis):
    in_size = img.shape[axis]
    bounds, kk, ksize = resample_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        x0, n = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += src[x0 + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_resize_bicubic_u8(img, target_hw):
    """numpy_utils.py:474-486 — `Image.fromarray(img).resize((W, H))` (default BICUBIC, antialiased): horizontal pass,
    8-bit intermediate, then vertical pass; passes whose size does not change are skipped (Resample.c:ImagingResample)"""
    th, tw = int(target_hw[0]), int(target_hw[1])
    out = img
    if tw != img.shape[1]:
        out = _resample_axis(out, tw, 1)
    if th != img.shape[0]:
        out = _resample_axis(out, th, 0)
    return np.ascontiguousarray(out) if out is not img else img.copy()


# ------------------------------------------------------------------------------------------------
# numpy_utils.py:447-517
# ------------------------------------------------------------------------------------------------
def zero_pad_image(img, target_hw):
    h, w = img.shape[:2]
    i = int(math.floor((target_hw[0] - h) / 2.))
    j = int(math.floor((target_hw[1] - w) / 2.))
    out = np.zeros((target_hw[0], target_hw[1], 3))
    out[i:i + h, j:j + w, :] = img
    return out.astype('uint8')


def crop_image(img, target_hw, init=False):
    ph, pw = max(img.shape[0], target_hw[0]), max(img.shape[1], target_hw[1])
    img = zero_pad_image(img, (ph, pw))
    i = int(math.floor((ph - target_hw[0]) / 2.))
    j = int(math.floor((pw - target_hw[1]) / 2.))
    if init:
        i, j = 0, 0
    return img[i:i + target_hw[0], j:j + target_hw[1], :].astype('uint8')


def image_valid_mask(img, target_hw):
    m = np.ones((target_hw[0], target_hw[1], 1))
    m[(img[:, :, 0] == 0) & (img[:, :, 1] == 0) & (img[:, :, 2] == 0)] = 0
    return m.astype('uint8')


def rpy_to_matrix(roll, pitch, yaw):
    """numpy_utils.py:519-550 (R = yaw @ pitch @ roll, embedded in 4x4)"""
    ym = np.array([[math.cos(yaw), -math.sin(yaw), 0], [math.sin(yaw), math.cos(yaw), 0], [0, 0, 1]])
    pm = np.array([[math.cos(pitch), 0, math.sin(pitch)], [0, 1, 0], [-math.sin(pitch), 0, math.cos(pitch)]])
    rm = np.array([[1, 0, 0], [0, math.cos(roll), -math.sin(roll)], [0, math.sin(roll), math.cos(roll)]])
    R4 = np.eye(4)
    R4[:3, :3] = ym @ pm @ rm
    return R4


def xyz_to_matrix(x, y, z):
    T = np.eye(4)
    T[:3, 3] = (x, y, z)
    return T


# ------------------------------------------------------------------------------------------------
# loader_utils.py:63-202
# ------------------------------------------------------------------------------------------------
def preproc_gt(rr, rp, ry, tx, ty, tz, rt, posej_T_posei=np.eye(4)):
    rand_init_l = np.array(rpy_to_matrix(rr, rp, ry) @ xyz_to_matrix(tx, ty, tz))
    rand_init_c = np.array([[math.cos(rt), -math.sin(rt), 0], [math.sin(rt), math.cos(rt), 0], [0, 0, 1]])
    return {'rand_init_l': rand_init_l, 'rand_init_c': rand_init_c,
            'sensor2_T_sensor1': posej_T_posei @ np.linalg.inv(rand_init_l),
            'intrinsic_sensor2': np.array(np.linalg.inv(rand_init_c))}


def rot_deg_of(mat):
    return math.degrees(np.arctan2(mat[1, 0], mat[0, 0]))


def preproc_img(img, gts, raw_hw, rellis=False):
    """loader_utils.py:104-130 (`rellis=True`: :132-158, the raw view is a resize instead of a crop)"""
    img_raw = pil_resize_bicubic_u8(img, raw_hw) if rellis else crop_image(img, raw_hw, init=True)
    img_rot = crop_image(pil_rotate_nearest_u8(img, rot_deg_of(gts['rand_init_c'])), raw_hw)
    half = (int(img_rot.shape[0] / 2), int(img_rot.shape[1] / 2))
    img_in = zero_pad_image(pil_resize_bicubic_u8(img_rot, half), (int(raw_hw[0] / 2), int(raw_hw[1] / 2)))
    return {'in': np.ascontiguousarray(np.transpose(img_in, (2, 0, 1)), dtype=np.float32),
            'raw': np.transpose(img_raw, (2, 0, 1)), 'rot': np.transpose(img_rot, (2, 0, 1)),
            'img_mask': np.ascontiguousarray(np.transpose(image_valid_mask(img_rot, raw_hw), (2, 0, 1)))}


def lidar_line_indices(n_points, reduce_to):
    """index list of `reduce_lidar_line` (loader_utils.py:162-175), python negative indexing included"""
    lines = 64
    rate = lines / reduce_to
    line_num = int(n_points / lines)
    idx = []
    for i in range(64):
        if i % rate == 0:
            for j in range(int(-line_num / 2), int(line_num / 2)):
                k = i * line_num + j
                idx.append(k if k >= 0 else k + n_points)
    return np.asarray(idx, np.int64)


def preproc_pcd(pcd, gts, num_points, lidar_line=None, radius=50., sampled_indices=None):
    """loader_utils.py:160-202.  `sampled_indices` stands in for the `np.random.choice(..., replace=False)` draw"""
    if lidar_line is not None:
        pcd = pcd[lidar_line_indices(pcd.shape[0], lidar_line)]
    if radius is not None:
        keep = (pcd[:, 0] >= -radius) & (pcd[:, 0] < radius) & (pcd[:, 1] >= -radius) & (pcd[:, 1] < radius)
        pcd = pcd[np.where(keep)[0]]
    if num_points < pcd.shape[0]:
        pcd_ = pcd[sampled_indices].T
    else:
        pcd_ = np.zeros((3, num_points))
        pcd_[:3, :pcd.shape[0]] = pcd[:, :3].T
    pc = np.ones((4, pcd_.shape[1]))
    pc[:3, :] = pcd_[:3, :]
    return np.array(gts['rand_init_l'] @ pc)


def process_sample(pcd, img, calib34, posej_T_posei, rand_init, raw_hw, num_points, lidar_line=None, rellis=False,
                   sampled_indices=None):
    """`ProcessKITTIODOM.__call__` (kitti_odom_loader.py:251-273) / `ProcessRELLIS.__call__` (rellis3d_loader.py:306-339);
    `calib34` = `(P2 @ Tr)[:3]` resp. `(P @ Tr @ R_inv)[:3]` (the caller's 3x4 product)."""
    gts = preproc_gt(*rand_init, posej_T_posei)
    if rellis:
        R = np.diag([-1., -1., 1., 1.])
        pc = np.ones((4, pcd.shape[0]))
        pc[:3, :] = pcd.T[:3, :]
        pcd = (R @ pc)[:3, :].T
    imgs = preproc_img(img, gts, raw_hw, rellis)
    pc = preproc_pcd(pcd, gts, num_points, lidar_line, sampled_indices=sampled_indices)
    A = np.array([[1, 0, -raw_hw[1] / 2], [0, 1, -raw_hw[0] / 2], [0, 0, 1]])
    gts['img_raw'], gts['img_rot'], gts['img_mask'] = imgs['raw'], imgs['rot'], imgs['img_mask']
    gts['cam_T_velo'] = np.linalg.inv(A) @ gts['intrinsic_sensor2'] @ A @ calib34 @ gts['sensor2_T_sensor1']
    return pc[:3, :], imgs['in'], calib34, A, gts
