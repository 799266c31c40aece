"""Generate tests/golden/prep_cases.npz from the UNMODIFIED reference (run in the build container only):
`ProcessKITTIODOM.__call__` / `ProcessRELLIS.__call__` (data_loader/kitti_odom_loader.py:237-273,
rellis3d_loader.py:292-339) on small synthetic images / sweeps.  Fixtures are data only (inputs + the reference's
outputs).  `np.random.seed(s)` right before each call pins the `np.random.choice` draw of preproc_pcd; the drawn index
list is recorded next to the outputs so that implementations with another RNG can be checked on the same subset.

    python tests/golden/make_golden_prep.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402


def synth_image(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 100 * np.sin(xx / 7.0 + seed) * np.cos(yy / 5.0), 255.0 * xx / w, 255.0 * yy / h], -1)
    img = np.clip(base + rng.normal(0, 25, (h, w, 3)), 0, 255).astype(np.uint8)
    img[rng.random((h, w)) < 0.02] = 0            # some exactly-black pixels (valid-mask edge case)
    return img


def synth_sweep(n, seed, spread=70.0):
    rng = np.random.default_rng(seed)
    p = np.empty((n, 4), np.float32)
    p[:, 0] = rng.uniform(-spread, spread, n)
    p[:, 1] = rng.uniform(-spread, spread, n)
    p[:, 2] = rng.uniform(-3, 5, n)
    p[:, 3] = rng.uniform(0, 1, n)
    p[:7, 0] = [50.0, -50.0, 49.999996, -50.000004, 0, 0, 0]       # the half-open radius test's boundary values
    p[:7, 1] = [0, 0, 0, 0, 50.0, -50.0, 12.5]
    return p


def main():
    ref_harness._install_stubs()
    if ref_harness.REF_ROOT not in sys.path:
        sys.path.insert(0, ref_harness.REF_ROOT)
    kitti = importlib.import_module('data_loader.kitti_odom_loader')
    rellis = importlib.import_module('data_loader.rellis3d_loader')
    assert kitti.__file__.startswith(ref_harness.REF_ROOT)
    P = np.array([[700., 0, 80, 40], [0, 700, 24, 2], [0, 0, 1, 0.003], [0, 0, 0, 1]])
    Tr = np.array([[0.01, -0.999, -0.02, 0.05], [0.02, 0.02, -0.999, -0.07], [0.999, 0.01, 0.02, -0.3], [0, 0, 0, 1.]])
    pose = np.eye(4)
    pose[:3, 3] = (0.4, -0.1, 0.02)
    cases = [
        # name, dataset, img hw, raw hw, n points, num_points, lidar_line, rand_init
        ('kitti_a', 'kitti', (60, 200), (48, 160), 6000, 2048, None, (0.05, -0.03, 0.2, 0.3, -0.2, 0.1, 0.07)),
        ('kitti_b', 'kitti', (40, 150), (48, 160), 1500, 2048, None, (-0.1, 0.02, -0.4, 0.0, 0.5, -0.3, -0.31)),
        ('kitti_c', 'kitti', (52, 168), (48, 160), 6400, 1024, 32, (0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)),
        ('rellis_a', 'rellis', (75, 120), (60, 100), 5000, 2048, None, (0.3, 0.1, -0.25, -0.4, 0.2, 0.0, -0.2)),
        ('rellis_b', 'rellis', (90, 144), (60, 100), 3000, 1024, None, (-0.02, 0.04, 0.5, 0.1, 0.1, 0.1, 0.45)),
    ]
    out = {}
    for i, (name, ds, ihw, raw, n, npts, ll, ri) in enumerate(cases):
        args = {'raw_cam_img_size': list(raw), 'lidar_line': ll, 'num_points': npts, 'test': True}
        img = synth_image(ihw[0], ihw[1], 10 + i)
        pcd = synth_sweep(n, 20 + i)
        seed = 1000 + i
        np.random.seed(seed)
        if ds == 'kitti':
            proc = kitti.ProcessKITTIODOM(args)
            pc, im, calib, A, gts, _ = proc(pcd, img, {'P2': P, 'Tr': Tr}, pose, name, rand_init=ri)
        else:
            proc = rellis.ProcessRELLIS(args)
            pc, im, calib, A, gts, _ = proc(pcd, img, {'P': P, 'Tr': Tr}, pose, name, rand_init=ri)
        out[name + '.img'] = img
        out[name + '.pcd'] = pcd
        out[name + '.meta'] = np.array([raw[0], raw[1], npts, -1 if ll is None else ll, seed, 1 if ds == 'rellis' else 0],
                                       np.int64)
        out[name + '.rand_init'] = np.array(ri, np.float64)
        out[name + '.pose'] = pose
        out[name + '.out.pc'] = np.asarray(pc)
        out[name + '.out.img'] = np.asarray(im)
        out[name + '.out.calib'] = np.asarray(calib)
        out[name + '.out.A'] = np.asarray(A)
        for k, v in gts.items():
            out[name + '.gt.' + k] = np.asarray(v)
    out['P'] = P
    out['Tr'] = Tr
    path = os.path.join(HERE, 'prep_cases.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes;', len(out), 'arrays')


if __name__ == '__main__':
    main()
