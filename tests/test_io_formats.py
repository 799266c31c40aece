"""SURVEY §8(f) "next" rows: on-disk formats and checkpoint interchange (host side, CPU)."""
import os

import numpy as np
import torch

from efgh_amd.io import checkpoint as ck, formats as fm


def test_velodyne_pose_calib_roundtrip(tmp_path):
    pts = np.random.RandomState(0).randn(100, 4).astype(np.float32)
    p = tmp_path / 'a.bin'
    fm.write_velodyne_bin(p, pts)
    assert np.array_equal(fm.read_velodyne_bin(p), pts)
    T = fm.parse_pose_line('1 0 0 1.5 0 1 0 -2 0 0 1 3e-1')
    assert T.shape == (4, 4) and T[0, 3] == 1.5 and T[2, 3] == 0.3 and T[3, 3] == 1
    c = tmp_path / 'calib.txt'
    c.write_text('P0: ' + ' '.join(['1'] * 12) + '\nP2: 7 0 6 0 0 7 2 0 0 0 1 0\nTr: 0 -1 0 0.1 0 0 -1 0.2 1 0 0 0.3\ntime: 2011\n')
    cal = fm.read_kitti_calib(c)
    assert cal['P2'][0, 0] == 7 and np.allclose(cal['Tr'] @ cal['Tr_inv'], np.eye(4))


def test_rand_init_csv_matches_reference_file_format(tmp_path):
    p = tmp_path / 'r.csv'
    p.write_text('00000_001706_001788,0.5030576594709399,-0.5171817316539501,-0.3074713016063032,-0.0,-0.0,-0.0,0.13442123642005216\n')
    d = fm.read_rand_init_csv(p)
    assert list(d) == ['00000_001706_001788'] and len(d['00000_001706_001788']) == 7
    assert d['00000_001706_001788'][6] == 0.13442123642005216


def test_prediction_csv(tmp_path):
    p = tmp_path / 'pred.csv'
    T = np.arange(16, dtype=np.float32).reshape(4, 4)
    fm.append_prediction_csv(p, 'f0', T)
    line = p.read_text().strip()
    assert line.startswith('f0,0.0,1.0,') and line.endswith(',') and line.count(',') == 13      # test.py:46-53
    assert np.array_equal(fm.read_prediction_csv(p)['f0'], T[:3])


def test_quaternion():
    R = fm.quat_xyzw_to_matrix(np.array([0, 0, np.sin(np.pi / 4), np.cos(np.pi / 4)]))
    assert np.allclose(R, [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-12)


def test_checkpoint_interchange_with_torch_adam(tmp_path):
    from efgh_amd.train import FlatParams, FusedAdam
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.Linear(4, 2))
    flat = FlatParams(m)
    opt = FusedAdam(flat, lr=3e-4)
    opt.t = 7
    opt.m.normal_(); opt.v.uniform_()
    path = ck.save_checkpoint(str(tmp_path), m, opt, it=2000, min_loss=1.25, is_best=True, iter_interval=1000)
    d = torch.load(path, map_location='cpu', weights_only=False)
    assert set(d) == {'iter', 'state_dict', 'min_loss', 'optimizer'} and all(k.startswith('module.') for k in d['state_dict'])
    assert os.path.exists(tmp_path / 'checkpoint_2000.pth.tar') and os.path.exists(tmp_path / 'model_best.pth.tar')
    # the reference side: DataParallel-style wrapper + torch.optim.Adam resume (main.py:136,190-198)
    ref = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.Linear(4, 2))
    wrapper = torch.nn.Module(); wrapper.module = ref
    wrapper.load_state_dict(d['state_dict'], strict=True)
    ropt = torch.optim.Adam(ref.parameters(), lr=1.0)
    ropt.load_state_dict(d['optimizer'])
    assert ropt.param_groups[0]['lr'] == 3e-4
    st = ropt.state[list(ref.parameters())[0]]
    assert int(st['step']) == 7 and torch.equal(st['exp_avg'].reshape(-1), opt.m[:24])
    # and back: the reference's checkpoint into this path
    m2 = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.Linear(4, 2))
    ck.load_model_state(m2, {'state_dict': {'module.' + k: v for k, v in ref.state_dict().items()}})
    flat2 = FlatParams(m2); opt2 = FusedAdam(flat2)
    ck.load_adam_state(opt2, ropt.state_dict())
    assert opt2.t == 7 and torch.equal(opt2.m, opt.m) and torch.equal(opt2.v, opt.v) and opt2.lr == 3e-4
    for a, b in zip(m2.parameters(), m.parameters()):
        assert torch.equal(a, b)


def test_partial_load_rename_and_freeze(tmp_path):
    """main.py:162-176,212-235: `convert_dict` renaming, keys the model lacks dropped, `grad_false_keys` freezing; FlatParams /
    FusedAdam then cover exactly the parameters that still require a gradient (main.py:178-183)"""
    from efgh_amd.train import FlatParams
    torch.manual_seed(0)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.E = torch.nn.Linear(3, 4)
            self.H = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4))
            self.G = torch.nn.Linear(4, 2)
    src = Net()
    sd = {('module.' + k).replace('module.G.', 'module.Gold.'): v.clone() + 1.0 for k, v in src.state_dict().items()}
    sd['module.unknown.weight'] = torch.zeros(1)
    torch.save({'iter': 5, 'state_dict': sd, 'min_loss': 0.1, 'optimizer': {}}, tmp_path / 'pre.pth.tar')
    dst = Net()
    res = ck.load_pretrained(dst, str(tmp_path / 'pre.pth.tar'), convert_dict={'Gold.': 'G.'}, grad_false_keys=['E.', 'H.0'])
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in dst.state_dict().items():
        assert torch.equal(v, src.state_dict()[k] + 1.0), k
    frozen = {k for k, p in dst.named_parameters() if not p.requires_grad}
    assert frozen == {'E.weight', 'E.bias', 'H.0.weight', 'H.0.bias'}
    flat = FlatParams(dst)
    assert flat.n == sum(p.numel() for k, p in dst.named_parameters() if k not in frozen)
    # the reference's own functions, restated, on a bare (unwrapped) state_dict and without renaming
    dst2 = Net()
    ck.load_pretrained(dst2, {k: v for k, v in src.state_dict().items() if k.startswith('E.')})
    assert torch.equal(dst2.E.weight, src.E.weight) and not torch.equal(dst2.G.weight, src.G.weight)


# ---- pinned to the reference: fixtures written by tests/golden/make_golden_io.py (the unmodified reference's own readers,
# ---- prediction writer and save_checkpoint, run in the build container) -----------------------------------------------------------
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'io')


def test_readers_equal_the_reference_loader_utils():
    """loader_utils.py:12-61,206-229: pcd_read, pose_read, calib_read, get_lidar2cam_mtx, get_cam_mtx"""
    e = np.load(os.path.join(GOLD, 'io_expected.npz'))
    assert np.array_equal(fm.read_velodyne_bin(os.path.join(GOLD, 'sweep.bin')), e['pcd'])
    lines = open(os.path.join(GOLD, 'poses.txt')).read().splitlines()
    assert np.array_equal(np.stack([fm.parse_pose_line(l) for l in lines]), e['poses'])
    cal = fm.read_kitti_calib(os.path.join(GOLD, 'calib.txt'))
    for k in ('Tr', 'Tr_inv', 'P2', 'P2_inv'):
        assert np.array_equal(cal[k], e['calib_' + k]), k
    assert np.array_equal(fm.read_rellis_camera_info(os.path.join(GOLD, 'camera_info.txt')), e['cam_mtx'])
    # scipy's Rotation.from_quat vs the closed form: same matrix to the last bits of the inverse
    assert np.allclose(fm.read_rellis_lidar2cam(os.path.join(GOLD, 'transforms.yaml')), e['lidar2cam'], rtol=0, atol=2e-16)


def test_rand_init_csv_equals_the_reference_reader_on_the_shipped_rows():
    """rellis3d_loader.py:44-48 on the first rows of params/rellis3d_rand_init_30_30.csv"""
    e = np.load(os.path.join(GOLD, 'io_expected.npz'))
    d = fm.read_rand_init_csv(os.path.join(GOLD, 'rand_init_head.csv'))
    assert list(d) == [str(n) for n in e['rand_init_names']]
    assert np.array_equal(np.array([d[k] for k in d]), e['rand_init_vals'])
    # facts about the whole file that the configs[4] test draws from: 2413 rows, |angles| <= 30 deg, translations zero
    assert int(e['rand_init_all_count']) == 2413
    assert np.all(np.abs(e['rand_init_all_min'][[0, 1, 2, 6]]) <= np.pi / 6 + 1e-9) and np.all(e['rand_init_all_max'][[0, 1, 2, 6]] <= np.pi / 6 + 1e-9)
    assert np.all(e['rand_init_all_min'][3:6] == 0) and np.all(e['rand_init_all_max'][3:6] == 0)


def test_prediction_csv_is_byte_identical_to_the_reference_writer(tmp_path):
    """test.py:46-53 (run through the reference's own test_odom with a stand-in model): same bytes, and it reads back"""
    e = np.load(os.path.join(GOLD, 'io_expected.npz'))
    p = tmp_path / 'pred.csv'
    for name, T in zip(('000000_000001', '000000_000002'), e['pred_T']):
        fm.append_prediction_csv(p, name, T)
    assert p.read_bytes() == open(os.path.join(GOLD, 'pred_toy.csv'), 'rb').read()
    back = fm.read_prediction_csv(os.path.join(GOLD, 'pred_toy.csv'))
    assert np.array_equal(back['000000_000001'], e['pred_T'][0][:3]) and np.array_equal(back['000000_000002'], e['pred_T'][1][:3])


def test_checkpoint_written_by_the_reference_loads_here_and_back():
    """common/helper.py:40-61 save_checkpoint of a DataParallel-wrapped model + torch.optim.Adam (main.py:127,181-183): our loader
    restores weights, buffers, Adam moments and the step; what we save from that state is loadable by the reference's resume code
    (main.py:149-160,190-198: strict load into the wrapped model, optimizer.load_state_dict) and equal tensor for tensor"""
    from efgh_amd.train import FlatParams, FusedAdam
    path = os.path.join(GOLD, 'ckpt', 'checkpoint.pth.tar')
    ref = torch.load(path, map_location='cpu', weights_only=False)
    assert set(ref) == {'iter', 'state_dict', 'min_loss', 'optimizer'} and ref['iter'] == 2000

    def toy():
        return torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, bias=False), torch.nn.BatchNorm2d(4), torch.nn.Flatten(),
                                   torch.nn.Linear(16, 2))
    m = toy()
    ck.load_model_state(m, path)
    for k, v in m.state_dict().items():
        assert torch.equal(v, ref['state_dict']['module.' + k]), k
    flat = FlatParams(m)
    opt = FusedAdam(flat, lr=1.0)
    ck.load_adam_state(opt, ref['optimizer'])
    assert opt.t == 3 and opt.lr == 1e-4
    for i, (p, (off, n)) in enumerate(zip(flat.params, flat.offsets)):
        assert torch.equal(opt.m[off:off + n].view(p.shape), ref['optimizer']['state'][i]['exp_avg'])
        assert torch.equal(opt.v[off:off + n].view(p.shape), ref['optimizer']['state'][i]['exp_avg_sq'])
    # and back: the reference's resume path
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        out = ck.save_checkpoint(d, m, opt, it=2000, min_loss=0.75)
        mine = torch.load(out, map_location='cpu', weights_only=False)
    wrapped = torch.nn.DataParallel(toy())
    wrapped.load_state_dict(mine['state_dict'], strict=True)
    ropt = torch.optim.Adam(filter(lambda p: p.requires_grad, wrapped.parameters()), lr=1e-4, weight_decay=0)
    ropt.load_state_dict(mine['optimizer'])
    for k, v in ref['state_dict'].items():
        assert torch.equal(mine['state_dict'][k], v), k
    for i in ref['optimizer']['state']:
        for key in ('exp_avg', 'exp_avg_sq'):
            assert torch.equal(ropt.state_dict()['state'][i][key], ref['optimizer']['state'][i][key])
        assert float(ropt.state_dict()['state'][i]['step']) == float(ref['optimizer']['state'][i]['step'])
