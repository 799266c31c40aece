"""Fixtures for the SURVEY 8(f) rows 2-3 (on-disk formats, checkpoint interchange), produced by the UNMODIFIED reference:

* data_loader/loader_utils.py : pose_read, calib_read, pcd_read, get_lidar2cam_mtx, get_cam_mtx        (:12-61, 206-229)
* data_loader/rellis3d_loader.py:44-48 : the rand-init CSV reader (csv.reader + float) on rows of the shipped
  params/rellis3d_rand_init_30_30.csv
* test.py:13-53 : test_odom run with a stand-in loader / model, i.e. the reference's own prediction-CSV writer
* common/helper.py:40-61 : save_checkpoint on a DataParallel-wrapped toy model + torch.optim.Adam state (main.py:127,181-183)

Run in the build container only (`python tests/golden/make_golden_io.py`); writes tests/golden/io/*.  The input files are
small hand-made samples in the datasets' formats; the expected values are whatever the reference returns for them."""
import csv
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh          # noqa: E402

OUT = os.path.join(HERE, 'io')


def main():
    rh._install_stubs()
    sys.path.insert(0, rh.REF_ROOT)
    os.makedirs(OUT, exist_ok=True)
    import importlib
    lu = importlib.import_module('data_loader.loader_utils')
    assert lu.__file__.startswith(rh.REF_ROOT)
    rs = np.random.RandomState(7)
    exp = {}
    # ---- KITTI-style inputs ------------------------------------------------------------------------------------
    pts = (rs.randn(257, 4) * np.array([20, 20, 2, 0.3])).astype(np.float32)
    pts.tofile(os.path.join(OUT, 'sweep.bin'))
    exp['pcd'] = lu.pcd_read(os.path.join(OUT, 'sweep.bin'))
    poses = ['1.000000e+00 9.043680e-12 2.326809e-11 5.551115e-17 9.043683e-12 1.000000e+00 2.392370e-10 3.330669e-16 '
             '2.326810e-11 2.392370e-10 9.999999e-01 -4.440892e-16',
             '9.999978e-01 5.272628e-04 -2.066935e-03 -4.690294e-02 -5.296506e-04 9.999992e-01 -1.154865e-03 -2.839928e-02 '
             '2.066324e-03 1.155958e-03 9.999971e-01 8.586941e-01']
    open(os.path.join(OUT, 'poses.txt'), 'w').write('\n'.join(poses) + '\n')
    exp['poses'] = np.stack([lu.pose_read(l) for l in poses])
    calib = ('P0: 7.188560000000e+02 0.0 6.071928000000e+02 0.0 0.0 7.188560000000e+02 1.852157000000e+02 0.0 0.0 0.0 1.0 0.0\n'
             'P1: 7.188560000000e+02 0.0 6.071928000000e+02 -3.861448000000e+02 0.0 7.188560000000e+02 1.852157000000e+02 0.0 0.0 0.0 1.0 0.0\n'
             'P2: 7.188560000000e+02 0.0 6.071928000000e+02 4.538225000000e+01 0.0 7.188560000000e+02 1.852157000000e+02 -1.130887000000e-01 0.0 0.0 1.0 3.779761000000e-03\n'
             'P3: 7.188560000000e+02 0.0 6.071928000000e+02 -3.372877000000e+02 0.0 7.188560000000e+02 1.852157000000e+02 2.369057000000e+00 0.0 0.0 1.0 4.915215000000e-03\n'
             'Tr: 4.276802385584e-04 -9.999672484946e-01 -8.084491683471e-03 -1.198459927713e-02 -7.210626507497e-03 8.081198471645e-03 '
             '-9.999413164504e-01 -5.403984729748e-02 9.999738645903e-01 4.859485810390e-04 -7.206933692422e-03 -2.921968648686e-01\n'
             'calib_time: 09-Jan-2012 13:57:47\n')
    open(os.path.join(OUT, 'calib.txt'), 'w').write(calib)
    c = lu.calib_read(os.path.join(OUT, 'calib.txt'))
    for k, v in c.items():
        exp['calib_' + k] = v
    # ---- RELLIS-3D inputs ---------------------------------------------------------------------------------------
    open(os.path.join(OUT, 'transforms.yaml'), 'w').write(
        'os1_cloud_node-pylon_camera_node:\n  q:\n    w: -0.50507811\n    x: 0.51206185\n    y: 0.49024953\n    z: -0.49228464\n'
        '  t:\n    x: -0.13165462\n    y: 0.03870398\n    z: -0.17253834\n')
    open(os.path.join(OUT, 'camera_info.txt'), 'w').write('2813.643275 2808.326079 969.285772 624.049972\n')
    exp['lidar2cam'] = lu.get_lidar2cam_mtx(os.path.join(OUT, 'transforms.yaml'))
    exp['cam_mtx'] = lu.get_cam_mtx(os.path.join(OUT, 'camera_info.txt'))
    # ---- rand-init CSV: first rows of the reference's own parameter file, read as rellis3d_loader.py:44-48 does --------------
    src = os.path.join(rh.REF_ROOT, 'params', 'rellis3d_rand_init_30_30.csv')
    rows = open(src).read().splitlines()[:6]
    open(os.path.join(OUT, 'rand_init_head.csv'), 'w').write('\n'.join(rows) + '\n')
    ri = {}
    f = open(os.path.join(OUT, 'rand_init_head.csv'), 'r')
    for k, line in enumerate(csv.reader(f)):                # (the loader's loop, verbatim semantics)
        ri[line[0]] = [float(i) for i in line[1:]]
    f.close()
    exp['rand_init_names'] = np.array(list(ri))
    exp['rand_init_vals'] = np.array([ri[k] for k in ri], dtype=np.float64)
    # the whole file: distribution facts used by the configs[4] test (ranges of the seven columns)
    allrows = [[float(v) for v in r.split(',')[1:]] for r in open(src).read().splitlines() if r]
    a = np.array(allrows)
    exp['rand_init_all_min'], exp['rand_init_all_max'], exp['rand_init_all_count'] = a.min(0), a.max(0), np.array(len(a))
    # ---- prediction CSV through the reference's own test_odom ------------------------------------------------------------
    test_mod = importlib.import_module('test')
    helper = importlib.import_module('common.helper')
    T = np.eye(4, dtype=np.float32)[None].repeat(2, 0)
    T[0, :3, :] = np.array([[0.99862951, -0.05233596, 0.0, 1.2345678], [0.05233596, 0.99862951, 0.0, -0.5],
                            [0.0, 0.0, 1.0, 1e-05]], dtype=np.float32)
    T[1, :3, 3] = [100.125, -3.0000001e-4, 7.0]
    gt = {'sensor2_T_sensor1': torch.eye(4)[None]}
    loader = [(torch.zeros(1, 3, 4), torch.zeros(1, 3, 2, 2), torch.zeros(1, 3, 4), torch.zeros(1, 3, 3), gt, ('000000_000001',)),
              (torch.zeros(1, 3, 4), torch.zeros(1, 3, 2, 2), torch.zeros(1, 3, 4), torch.zeros(1, 3, 3), gt, ('000000_000002',))]

    class Model:
        k = 0

        def eval(self):
            return self

        def __call__(self, *a):
            out = {'sensor2_T_sensor1': torch.from_numpy(T[self.k:self.k + 1])}
            self.k += 1
            return out
    cwd = os.getcwd()
    work = os.path.join(OUT, '_work', 'a', 'b')
    os.makedirs(work, exist_ok=True)
    os.chdir(work)                    # test_odom writes to ../../test/preds/<ckpt dir>/
    try:
        test_mod.test_odom(loader, Model(), {'dataset': 'RELLIS_3D', 'ckpt_path': 'x/run1/model_best.pth.tar',
                                             'rand_init': 'params/rand_init_toy.csv', 'DEVICE': 'cpu', 'save_image': False})
    finally:
        os.chdir(cwd)
    pred_file = os.path.join(OUT, '_work', 'test', 'preds', 'run1', 'pred_toy.csv')
    text = open(pred_file).read()
    open(os.path.join(OUT, 'pred_toy.csv'), 'w').write(text)
    exp['pred_T'] = T
    import shutil
    shutil.rmtree(os.path.join(OUT, '_work'))
    # ---- checkpoint written by the reference's save_checkpoint ---------------------------------------------------------------
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, bias=False), torch.nn.BatchNorm2d(4), torch.nn.Flatten(),
                              torch.nn.Linear(16, 2))
    model = torch.nn.DataParallel(net)                                     # main.py:127
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=1e-4, weight_decay=0)   # main.py:178-183
    for _ in range(3):
        opt.zero_grad()
        model(torch.randn(5, 3, 4, 4)).pow(2).mean().backward()
        opt.step()
    ck_dir = os.path.join(OUT, 'ckpt')
    os.makedirs(ck_dir, exist_ok=True)
    helper.save_checkpoint({'iter': 2000, 'state_dict': model.state_dict(), 'min_loss': 0.75,
                            'optimizer': opt.state_dict()}, True, ck_dir, iter_iterval=1000)
    for extra in ('checkpoint_2000.pth.tar', 'model_best.pth.tar'):          # byte-identical copies: keep one file
        assert os.path.exists(os.path.join(ck_dir, extra))
        os.remove(os.path.join(ck_dir, extra))
    np.savez_compressed(os.path.join(OUT, 'io_expected.npz'), **exp)
    print('wrote', sorted(os.listdir(OUT)))


if __name__ == '__main__':
    main()
