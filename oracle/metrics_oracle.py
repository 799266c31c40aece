"""TEST INFRASTRUCTURE ONLY — numpy restatement of the reference's pose-error metrics (common/helper.py:163-207).
Both modes are pinned against tests/golden/metrics_cases.npz, outputs of the unmodified `Err`: `odom` directly; `raw` (which needs
pyquaternion, absent from the build container) through the reference's own calc_error_raw_np / quaternion_distance run over a
quaternion-algebra stand-in in the fixture harness (tests/golden/ref_harness.py) - this restatement agrees with it to 1e-14."""
import numpy as np


def calc_error_odom(gt, pred):
    """helper.py:198-207 on float32 inputs (numpy keeps float32 throughout)"""
    gt_R, gt_t, pred_R, pred_t = gt[:3, :3], gt[:3, 3], pred[:3, :3], pred[:3, 3]
    tmp = (np.trace(pred_R.transpose().dot(gt_R)) - 1) / 2
    tmp = np.clip(tmp, -1.0, 1.0)
    return 180 * np.arccos(tmp) / np.pi, np.linalg.norm(pred_t - gt_t)


def calc_error_raw(gt, pred):
    """helper.py:165-196: 2*atan2(|v|, |w|) of q_gt * q_pred^-1 in degrees; mean absolute translation difference"""
    from scipy.spatial.transform import Rotation
    q = (Rotation.from_matrix(gt[:3, :3].astype(np.float64)) * Rotation.from_matrix(pred[:3, :3].astype(np.float64)).inv()).as_quat()
    return 2 * np.arctan2(np.linalg.norm(q[:3]), abs(q[3])) * 180 / np.pi, np.mean(np.fabs(gt[:3, 3] - pred[:3, 3]))
