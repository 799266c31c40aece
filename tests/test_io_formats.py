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
