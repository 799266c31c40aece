"""Deterministic synthetic EFGH inputs.

Produces exactly what the reference's loader hands to the model
(`ProcessRELLIS.__call__`, data_loader/rellis3d_loader.py:306-339): ``pc (3,N) f32``,
``img (3,H/2,W/2) f32 in [0,255]``, ``calib (3,4)``, ``A (3,3)`` and the ground-truth dict
produced by ``preproc_gt`` (data_loader/loader_utils.py:79-103).  All numpy, no device work.
"""
import math
import re

import numpy as np


def default_args(raw_hw=(768, 2560), device='cuda'):
    """Hot-path keys of configs/train_rellis.yaml with a configurable camera size."""
    return {
        'dim': 3,
        'scale_map': [[1., 1], [0.75, 1], [0.5, 1], [0.25, 1], [0.125, 1]],
        'DEVICE': device,
        'use_leaky': True, 'bcn_use_bias': True, 'bcn_use_norm': True, 'last_relu': False,
        'raw_cam_img_size': [int(raw_hw[0]), int(raw_hw[1])],
        'lidar_fov_rad': [0.125, -0.125],
        'dataset': 'RELLIS_3D',
        'lambda': {'e_gn': 100., 'h_hrzn': 100., 'fov': 100., 'g_trs': 1000., 'g_depth': 0.1,
                   'g_mask': 1000.},
        'fov_pos_num': 30, 'fov_neg_ratio': 5,
    }


def lidar_sweep(n_points, seed=0, beams=64, pitch_range=(-0.12 * np.pi, 0.12 * np.pi)):
    """64-beam organised sweep with random ranges in [5,25) m -> (3,N) float32.  pitch_range: lowest / highest beam elevation
    in radians (default: the symmetric fan of SURVEY 8d; an HDL-64E is (-24.8 deg, +2 deg))."""
    rs = np.random.RandomState(seed)
    nb = beams
    na = n_points // nb
    pitch = np.linspace(pitch_range[0], pitch_range[1], nb)[:, None]
    yaw = np.linspace(-np.pi, np.pi, na, endpoint=False)[None, :]
    r = 5 + 20 * rs.rand(nb, na)
    x = r * np.cos(pitch) * np.cos(yaw)
    y = r * np.cos(pitch) * np.sin(yaw)
    z = r * np.sin(pitch)
    return np.float32(np.stack([x, y, z]).reshape(3, -1))


def coherent_sweep(n_points, seed=0, beams=64, pitch_range=(-24.8 / 180 * np.pi, 2.0 / 180 * np.pi)):
    """64-beam organised sweep over a ground plane and a dozen walls -> (3,N) float32.  Unlike `lidar_sweep` (independent random
    ranges: no two neighbouring returns share a lattice cell) this one is spatially coherent like a real scan - lattice cells near
    the sensor collect hundreds to thousands of points - which is the hard case for the lattice build (long vertex lists) and the
    easy one for the splat's cache locality."""
    rs = np.random.RandomState(seed)
    na = n_points // beams
    el = np.linspace(pitch_range[0], pitch_range[1], beams)[:, None]
    az = np.linspace(-np.pi, np.pi, na, endpoint=False)[None, :]
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el) * np.ones_like(az)])     # (3, beams, na)
    with np.errstate(divide='ignore', invalid='ignore'):
        r = np.minimum(80.0, np.where(d[2] < -1e-3, -1.73 / d[2], np.inf))      # ground plane z = -1.73, 80 m maximum range
        for _ in range(12):                                                     # vertical walls, 16 m wide, 3 m above the sensor
            th = rs.uniform(-np.pi, np.pi)
            nrm = np.array([np.cos(th), np.sin(th), 0.0])
            dist = rs.uniform(6, 40)
            den = d[0] * nrm[0] + d[1] * nrm[1]
            rw = np.where(den > 1e-3, dist / den, np.inf)
            lat = (d[0] * -nrm[1] + d[1] * nrm[0]) * rw
            rw = np.where((np.abs(lat) < 8) & (d[2] * rw < 3.0), rw, np.inf)
            r = np.minimum(r, rw)
    r = r * (1 + 0.002 * rs.randn(beams, na))
    return np.float32((d * r).reshape(3, -1))


def camera_image(raw_hw, seed=0):
    """uint8-valued RGB at half the raw camera size -> (3,H/2,W/2) float32."""
    h, w = raw_hw[0] // 2, raw_hw[1] // 2
    rs = np.random.RandomState(1000 + seed)
    return rs.randint(0, 256, size=(3, h, w)).astype(np.float32)


def calib_and_A(raw_hw):
    H, W = raw_hw
    K = np.array([[600., 0, W / 2, 0], [0, 600., H / 2, 0], [0, 0, 1, 0]])
    T = np.array([[0., -1, 0, 0], [0, 0, -1, 0], [1, 0, 0, 0], [0, 0, 0, 1]])
    calib = (K @ T)[:3]
    A = np.array([[1., 0, -W / 2], [0, 1., -H / 2], [0, 0, 1.]])
    return calib, A


def _rpy(roll, pitch, yaw):
    cy, sy, cp, sp, cr, sr = (math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch),
                              math.cos(roll), math.sin(roll))
    Y = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.]])
    P = np.array([[cp, 0, sp], [0, 1., 0], [-sp, 0, cp]])
    R = np.array([[1., 0, 0], [0, cr, -sr], [0, sr, cr]])
    M = np.eye(4)
    M[:3, :3] = Y @ P @ R
    return M


def ground_truth(raw_hw, seed=0, l_rot=1 / 6., l_trs=2., c_rot=1 / 6.):
    """Random mis-calibration + GT dict (loader_utils.py:63-103, rellis3d_loader.py:337)."""
    rs = np.random.RandomState(2000 + seed)
    u = rs.rand(7) * 2. - 1.
    rr, rp, ry = u[0] * np.pi * l_rot, u[1] * np.pi * l_rot, u[2] * np.pi * l_rot
    tx, ty, tz = u[3] * l_trs, u[4] * l_trs, u[5] * l_trs
    rt = u[6] * np.pi * c_rot
    return ground_truth_from_params(raw_hw, rr, rp, ry, tx, ty, tz, rt)


def ground_truth_from_params(raw_hw, rr, rp, ry, tx, ty, tz, rt):
    """GT dict of one mis-calibration given as a row of the reference's rand-init CSV (name, roll, pitch, yaw, tx, ty, tz,
    cam_roll; rellis3d_loader.py:44-48 -> rand_init_params / preproc_gt, loader_utils.py:63-103)"""
    ltrs = np.eye(4)
    ltrs[:3, 3] = [tx, ty, tz]
    rand_init_l = _rpy(rr, rp, ry) @ ltrs
    rand_init_c = np.array([[math.cos(rt), -math.sin(rt), 0], [math.sin(rt), math.cos(rt), 0],
                            [0, 0, 1.]])
    calib, A = calib_and_A(raw_hw)
    s2Ts1 = np.linalg.inv(rand_init_l)
    intr = np.linalg.inv(rand_init_c)
    return {
        'rand_init_l': rand_init_l, 'rand_init_c': rand_init_c,
        'sensor2_T_sensor1': s2Ts1, 'intrinsic_sensor2': intr,
        'cam_T_velo': np.linalg.inv(A) @ intr @ A @ calib @ s2Ts1,
        'img_mask': np.ones((1, raw_hw[0], raw_hw[1]), dtype=np.uint8),
    }


def make_sample(raw_hw, n_points, seed=0):
    calib, A = calib_and_A(raw_hw)
    return {
        'pc': lidar_sweep(n_points, seed),
        'img': camera_image(raw_hw, seed),
        'calib': calib.astype(np.float32),
        'A': A.astype(np.float32),
        'gt': ground_truth(raw_hw, seed),
    }


def make_batch(raw_hw, n_points, batch, first_seed=0):
    """Stack `batch` samples (seed = global sample index) into loader-shaped arrays."""
    ss = [make_sample(raw_hw, n_points, first_seed + i) for i in range(batch)]
    out = {k: np.stack([s[k] for s in ss]) for k in ('pc', 'img', 'calib', 'A')}
    out['gt'] = {k: np.stack([s['gt'][k] for s in ss]) for k in ss[0]['gt']}
    return out


def synthetic_state_dict(manifest, seed=0):
    """Deterministic, well-conditioned weights for parity tests.

    `manifest` is an ordered list of (key, shape, dtype-name) as stored in
    tests/golden/state_dict_manifest.json.  Every tensor is drawn from its own CPU generator
    seeded by (seed, position) so the values do not depend on construction order, module code or
    device.  Conv/linear weights are fan-in scaled (activations neither vanish nor explode through
    ~20 layers in eval mode); BatchNorm affine/running statistics are non-trivial.
    """
    import torch
    out = {}
    for i, (key, shape, dtype) in enumerate(manifest):
        g = torch.Generator(device='cpu')
        g.manual_seed(seed * 100003 + i)
        shape = tuple(shape)
        leaf = key.rsplit('.', 1)[-1]
        if leaf == 'num_batches_tracked':
            t = torch.zeros(shape, dtype=torch.int64)
        elif leaf == 'feat_indices':
            t = torch.arange(shape[0], dtype=torch.int64)
        elif leaf == 'running_mean':
            t = torch.randn(shape, generator=g) * 0.1
        elif leaf == 'running_var':
            t = torch.rand(shape, generator=g) + 0.5
        elif leaf == 'weight' and len(shape) == 1:          # BatchNorm gamma
            t = torch.rand(shape, generator=g) + 0.5
        elif leaf == 'bias':
            t = torch.randn(shape, generator=g) * 0.1
        elif leaf == 'weight':
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            gain = 2.0
            if re.search(r'(vgg_5_\d_(camera|range)|convt_\w+)\.0\.weight$', key):
                # ConvTranspose2d weight is (in, out, kh, kw): stride 2 -> ~9/4 taps per output
                fan_in = shape[0] * shape[2] * shape[3] / 4.0
            if re.search(r'(H\.vgg|F\.vgg_camera)\.features\.0\.weight$|G\.conv_i0\.0\.weight$', key):
                gain = 2.0 / (128.0 * 128.0)          # these convs see the raw [0,255] image
            if re.search(r'\.conv2\.weight$|downsample\.0\.weight$', key):
                gain = 0.5                             # keep the residual sums from growing
            if re.search(r'lin_\w+_(abs|sgn)\.weight$|conv_trs_4\.weight$', key):
                gain = 0.25                            # un-saturated softmax / O(1) translation
            if re.search(r'E\.conv_in\.0\.0\.weight$', key):
                gain = 2.0 / 225.0                     # sees raw metric coordinates (5..25 m)
            if re.search(r'E\.lin_gn_(abs|sgn)\.weight$|H\.lin_hrzn_abs\.weight$', key):
                gain = 0.02
            t = torch.randn(shape, generator=g) * math.sqrt(gain / fan_in)
        else:
            raise KeyError(key)
        out[key] = t
    return out
