"""tests/golden/metrics_cases.npz: `Err.update` / `calc_error_odom_np` / `calc_error_raw_np` of the unmodified reference (common/helper.py:128-207)
on random pose pairs (float32 tensors, as the training loop hands them over).  Run in the build container only."""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402


def rand_pose(rng, ang_scale, t_scale):
    from scipy.spatial.transform import Rotation
    T = np.eye(4)
    T[:3, :3] = Rotation.from_rotvec(rng.normal(0, ang_scale, 3)).as_matrix()
    T[:3, 3] = rng.normal(0, t_scale, 3)
    return T


def main():
    ref_harness._install_stubs()
    if ref_harness.REF_ROOT not in sys.path:
        sys.path.insert(0, ref_harness.REF_ROOT)
    helper = importlib.import_module('common.helper')
    assert helper.__file__.startswith(ref_harness.REF_ROOT)
    rng = np.random.default_rng(7)
    err = helper.Err('RELLIS_3D')
    gts, preds, rots, trss = [], [], [], []
    for i in range(64):
        g = rand_pose(rng, 0.5, 2.0)
        d = rand_pose(rng, [1e-4, 1e-2, 0.1, 1.0][i % 4], [1e-3, 0.1, 1.0, 0.0][i % 4])
        p = d @ g
        gt = {'sensor2_T_sensor1': torch.from_numpy(g[None]).float()}
        pr = {'sensor2_T_sensor1': torch.from_numpy(p[None]).float()}
        err.update(gt, pr)
        gts.append(gt['sensor2_T_sensor1'].numpy()[0]); preds.append(pr['sensor2_T_sensor1'].numpy()[0])
        rots.append(err.error_dict['rot'][-1]); trss.append(err.error_dict['trs'][-1])
    # raw mode (camera-LiDAR extrinsic calibration, helper.py:147-148,166-196): the reference's own calc_error_raw_np /
    # quaternion_distance on the same pairs, over the quaternion stand-in of ref_harness (pyquaternion is absent here)
    raw = helper.Err('KITTI_RAW')
    for g, p in zip(gts, preds):
        raw.update({'sensor2_T_sensor1': torch.from_numpy(g[None])}, {'sensor2_T_sensor1': torch.from_numpy(p[None])})
    path = os.path.join(HERE, 'metrics_cases.npz')
    np.savez_compressed(path, gt=np.stack(gts), pred=np.stack(preds), rot=np.array(rots, np.float64),
                        trs=np.array(trss, np.float64),
                        final=np.array([err.dict['rot_mean'], err.dict['rot_std'], err.dict['trs_mean'], err.dict['trs_std']]),
                        raw_rot=np.array(raw.error_dict['rot'], np.float64), raw_trs=np.array(raw.error_dict['trs'], np.float64),
                        raw_final=np.array([raw.dict['rot_mean'], raw.dict['rot_std'], raw.dict['trs_mean'], raw.dict['trs_std']]))
    print('wrote', path)


if __name__ == '__main__':
    main()
