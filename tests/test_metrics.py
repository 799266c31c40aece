"""Pose-error metrics: oracle vs the reference's `Err` outputs (CPU), HIP kernel + `efgh_amd.common.metrics.Err` vs both (GPU)."""
import os

import numpy as np
import pytest
import torch


def _G(golden_dir):
    return np.load(os.path.join(str(golden_dir), 'metrics_cases.npz'))


def _tol(rot_ref):
    # arccos near 1: a float32 rounding of the trace (6e-8) moves the angle by sqrt(2*6e-8) rad = 0.02 deg at 0 deg and by
    # 6e-8/sin(theta) elsewhere; the reference's own value carries that noise (its sum order is BLAS')
    return np.maximum(0.05 * (rot_ref < 1.0), 1e-4 * np.maximum(rot_ref, 1.0))


def test_oracle_odom_equals_reference(golden_dir):
    from oracle import metrics_oracle as MO
    G = _G(golden_dir)
    for g, p, r, t in zip(G['gt'], G['pred'], G['rot'], G['trs']):
        ro, to = MO.calc_error_odom(g, p)
        assert abs(ro - r) <= _tol(np.array(r)) and abs(to - t) <= 1e-6 * max(1.0, t)


def test_oracle_raw_equals_reference(golden_dir):
    """raw mode (KITTI_RAW): fixtures from the reference's own calc_error_raw_np / quaternion_distance (helper.py:166-196; run over
    a quaternion-algebra stand-in for the absent pyquaternion, tests/golden/ref_harness.py)"""
    from oracle import metrics_oracle as MO
    G = _G(golden_dir)
    assert len(G['raw_rot']) == 64
    for g, p, r, t in zip(G['gt'], G['pred'], G['raw_rot'], G['raw_trs']):
        ro, to = MO.calc_error_raw(g, p)
        assert abs(ro - r) <= 1e-9 * max(1.0, r) and abs(to - t) <= 1e-7 * max(1.0, t)


@pytest.mark.gpu
def test_hip_pose_errors_vs_reference_and_oracle(golden_dir):
    from efgh_amd.common.metrics import Err
    from oracle import metrics_oracle as MO
    G = _G(golden_dir)
    err = Err('RELLIS_3D', capacity=16)              # also exercises the history growth
    for g, p in zip(G['gt'], G['pred']):
        err.update({'sensor2_T_sensor1': torch.from_numpy(g[None]).cuda()}, {'sensor2_T_sensor1': torch.from_numpy(p[None]).cuda()})
    h = err.error_dict
    rot, trs = np.array(h['rot'], np.float64), np.array(h['trs'], np.float64)
    assert np.all(np.abs(rot - G['rot']) <= _tol(G['rot'])) and np.allclose(trs, G['trs'], rtol=1e-5, atol=1e-6)
    d = err.dict
    big = G['rot'] >= 1.0                               # the well-conditioned part decides the mean to 1e-4
    assert abs(d['trs_mean'] - G['final'][2]) < 1e-5 and abs(d['trs_std'] - G['final'][3]) < 1e-5
    assert abs(d['rot_mean'] - G['final'][0]) < 0.05 and abs(d['rot_std'] - G['final'][1]) < 0.05 and big.sum() > 10
    raw = Err('KITTI_RAW')
    for g, p in zip(G['gt'], G['pred']):
        raw.update({'sensor2_T_sensor1': torch.from_numpy(g[None]).cuda()}, {'sensor2_T_sensor1': torch.from_numpy(p[None]).cuda()})
    hr = raw.error_dict
    for i, (g, p) in enumerate(zip(G['gt'], G['pred'])):
        ro, to = MO.calc_error_raw(g, p)
        assert abs(hr['rot'][i] - ro) < 2e-3 + 1e-5 * ro and abs(hr['trs'][i] - to) < 1e-6 * max(1.0, to)
        # ... and the reference's own raw-mode outputs
        assert abs(hr['rot'][i] - G['raw_rot'][i]) < 2e-3 + 1e-5 * G['raw_rot'][i]
        assert abs(hr['trs'][i] - G['raw_trs'][i]) < 1e-6 * max(1.0, G['raw_trs'][i])
    dr = raw.dict
    assert abs(dr['rot_mean'] - G['raw_final'][0]) < 2e-3 and abs(dr['rot_std'] - G['raw_final'][1]) < 2e-3
    assert abs(dr['trs_mean'] - G['raw_final'][2]) < 1e-5 and abs(dr['trs_std'] - G['raw_final'][3]) < 1e-5
