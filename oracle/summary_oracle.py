"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy, float64 / uint8) of the reference's TensorBoard / evaluation overlay
images (second half): `common/numpy_utils.py:8-179` (image_draw), `:181-297` (eval_image_draw),
`:299-413` (the float64 rasterisers, the raster-order "paint the window if it is brighter" colouring, the overlay and the score
strip).  The Pillow operations come from oracle/prep_oracle.py; the colour maps are matplotlib's 256-entry `plasma` and `jet`
look-up tables, taken as data (tests/golden/make_colormaps.py -> efgh_amd/common/colormaps.npz).
Pinned by tests/test_oracle_summary.py against fixtures produced by the unmodified reference (tests/golden/make_golden_summary.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import math

import numpy as np

from . import prep_oracle as PO


def depth_image_last(pc, cam_T_velo, hw):
    """numpy_utils.py:338-358: pinhole raster in float64, the LAST point of the sweep that lands on a pixel wins, uint8 wrap"""
    H, W = hw
    p = np.concatenate([np.asarray(pc)[:3], np.ones((1, pc.shape[1]))], 0)          # float64 from here on
    q = np.asarray(cam_T_velo) @ p
    x, y, w = q[0], q[1], q[2]
    ok = (w > 0) & (0 <= x) & (x < w * W) & (0 <= y) & (y < w * H)
    idx = np.nonzero(ok)[0]
    with np.errstate(all='ignore'):
        r, c = (y[idx] / w[idx]).astype(np.int64), (x[idx] / w[idx]).astype(np.int64)
    img = np.zeros((H, W))
    img[r, c] = w[idx]                       # numpy fancy assignment: the last occurrence wins, as the sequential loop
    return img.astype(np.int64).astype(np.uint8)


def range_image_last(pc, T, hw, fov):
    """numpy_utils.py:299-336: spherical raster in float64 (r includes the homogeneous 1: sqrt(x^2+y^2+z^2+1)), last point wins"""
    H, W = hw
    up, down = fov[0] * math.pi, fov[1] * math.pi
    p = np.asarray(T) @ np.concatenate([np.asarray(pc), np.ones((1, pc.shape[1]))], 0)
    p = np.concatenate([p[:3], np.ones((1, pc.shape[1]))], 0)
    r = np.sqrt(np.sum(np.power(p, 2), 0))
    with np.errstate(all='ignore'):
        pitch, yaw = np.arcsin(p[2] / r), np.arctan2(p[1], p[0])
    m = (pitch < up) & (pitch > down)
    u = ((up - pitch[m]) / (up - down)) * (H - 1)
    v = ((-yaw[m] + math.pi) / (2 * math.pi)) * (W - 1)
    img = np.zeros((H, W))
    img[u.astype(np.int64), v.astype(np.int64)] = r[m]
    return img


def minmax_paint(img, px=2):
    """numpy_utils.py:384-396: normalise to [0, 1] in the input's own arithmetic (uint8 -> float64, float32 stays float32), then in
    raster order every pixel with a positive value paints its (2px+1)^2 window - rows [y-px, min(H-1, y+px+1)), columns likewise,
    so the last row and column are never written - provided nothing in the window is already >= its value."""
    img = np.asarray(img)
    with np.errstate(all='ignore'):
        n = (img - np.min(img)) / (np.max(img) - np.min(img))
    H, W = n.shape
    out = np.zeros((H, W))
    ys, xs = np.nonzero(n > 0)
    for y, x in zip(ys.tolist(), xs.tolist()):           # np.nonzero returns raster order
        y0, y1, x0, x1 = max(0, y - px), min(H - 1, y + px + 1), max(0, x - px), min(W - 1, x + px + 1)
        win = out[y0:y1, x0:x1]
        if win.size == 0 or win.max() < n[y, x]:         # np.max of an empty window raises in the reference; callers never
            win[...] = n[y, x]                           # produce one (H, W >= 2)
    return out


def colorize(minmax, lut8):
    """cmap(x)[:, :, :3] * 255 -> uint8 with matplotlib's index rule: int(x * 256), x == 1 -> 255"""
    idx = (minmax * 256.0).astype(np.int64)
    idx[minmax * 256.0 == 256.0] = 255
    return lut8[np.clip(idx, 0, 255)], minmax != 0


def minmax_color(img, lut8, px=2):
    return colorize(minmax_paint(img, px), lut8)


def overlay(color, mask, cam_hwc, raw_hw):
    """numpy_utils.py:360-375: camera pixels wherever the depth raster is empty"""
    cam = PO.pil_resize_bicubic_u8(cam_hwc, raw_hw)
    return np.where(mask[:, :, None], color, cam).astype(np.uint8)


def score_image(vec, range_hw, lut8):
    """numpy_utils.py:402-413: the 1-D score as an 8-row strip, coloured, resized (Pillow bicubic) to the range image size"""
    strip, _ = minmax_color(np.tile(np.asarray(vec)[None, :], (8, 1)), lut8)
    return PO.pil_resize_bicubic_u8(strip, range_hw)


def rotate_by_matrix(img_hwc, mat):
    return PO.pil_rotate_nearest_u8(img_hwc, math.degrees(np.arctan2(mat[1, 0], mat[0, 0])), expand=True)


def _hwc(a):
    a = np.asarray(a)
    return a if a.shape[2] == 3 else np.transpose(a, (1, 2, 0))


def image_draw(pcd, img, calib, A, gt, pred, raw_hw, fov, lut8):
    """numpy_utils.py:8-179 for sample 0 of numpy inputs (pcd (3,N), img (3,h,w) float, calib (3,4), A (3,3); gt / pred: dicts of
    numpy arrays of sample 0, pred['network'] the string)"""
    net_hw, rng_hw = (int(raw_hw[0] / 2), int(raw_hw[1] / 2)), (int(raw_hw[0] / 2), int(raw_hw[1] * 2))
    in_img = PO.crop_image(_hwc(img.astype(np.uint8)), net_hw)
    cam_rot = PO.pil_resize_bicubic_u8(_hwc(gt['img_rot'].astype(np.uint8)), raw_hw)

    def depth_overlay(T, px=2):
        c, m = minmax_color(depth_image_last(pcd, T, raw_hw), lut8, px)
        return overlay(c, m, cam_rot, raw_hw)

    def rng(T):
        return minmax_color(range_image_last(pcd, T, rng_hw, fov), lut8)[0]
    gt_s2s1 = gt['g_l'] @ gt['f_l'] @ gt['e_l']
    gt_T = np.linalg.inv(A) @ gt['h_c'] @ A @ calib @ gt_s2s1
    in_depth, gt_depth = depth_overlay(calib), depth_overlay(gt_T)
    gt_img = PO.crop_image(rotate_by_matrix(in_img, gt['h_c']), net_hw)
    in_range, gt_range = rng(np.eye(4)), rng(gt_s2s1)
    out, net = {}, pred['network']
    if 'E' in net:
        out['pred_range_E'] = rng(pred['e_l'])
    if 'E' in net and 'H' in net:
        out['pred_depth_EH'] = depth_overlay(pred['eh_cam_T_velo'])
    if 'H' in net:
        out['cam'] = np.concatenate([in_img, PO.crop_image(rotate_by_matrix(in_img, pred['h_c']), net_hw), gt_img], 0)
    if 'F' in net:
        out['pred_range_EF'] = rng(pred['f_l'] @ pred['e_l'])
        out['pred_depth_EFH'] = depth_overlay(pred['efh_cam_T_velo'])
        out['score'] = np.concatenate([score_image(gt['f_score'], rng_hw, lut8), score_image(pred['f_score'], rng_hw, lut8)], 0)
    if 'G' in net:
        out['pred_range_EFG'] = rng(pred['g_l'] @ pred['f_l'] @ pred['e_l'])
        out['pred_depth_EFGH'] = depth_overlay(pred['efgh_cam_T_velo'])
        out['dimage'] = np.concatenate([minmax_color(pred['g_depth'][0], lut8)[0], minmax_color(gt['g_depth'][0], lut8)[0]], 0)
        out['mask'] = np.concatenate([minmax_color(pred['g_mask'][0], lut8)[0], minmax_color(gt['g_mask'][0], lut8)[0]], 0)
    if 'E' in net and 'F' in net:
        names = ['E', 'EF'] + (['EFG'] if 'G' in net else [])
        out['range'] = np.concatenate([in_range] + [out.pop('pred_range_' + n) for n in names] + [gt_range], 0)
        dn = ['EH', 'EFH'] + (['EFGH'] if 'G' in net else [])
        out['depth'] = np.concatenate([in_depth] + [out.pop('pred_depth_' + n) for n in dn] + [gt_depth], 0)
    return out


def eval_image_draw(pcd, img, calib, A, gt, pred, raw_hw, fov, px, lut8):
    """numpy_utils.py:181-297: the three predicted depth overlays (EH and EFH with `px`, EFGH with px = 2), each rotated by the
    predicted h_c and centre-cropped to the network input size"""
    net_hw = (int(raw_hw[0] / 2), int(raw_hw[1] / 2))
    cam_rot = PO.pil_resize_bicubic_u8(_hwc(gt['img_rot'].astype(np.uint8)), raw_hw)
    out = {}
    for name, key, p in (('pred_depth_EH', 'eh_cam_T_velo', px), ('pred_depth_EFH', 'efh_cam_T_velo', px),
                         ('pred_depth_EFGH', 'efgh_cam_T_velo', 2)):
        c, m = minmax_color(depth_image_last(pcd, pred[key], raw_hw), lut8, p)
        out[name] = PO.crop_image(rotate_by_matrix(overlay(c, m, cam_rot, raw_hw), pred['h_c']), net_hw)
    return out
