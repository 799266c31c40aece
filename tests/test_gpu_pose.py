"""Pose heads and pose loss terms as HIP kernels (csrc/pose.hip, forward + hand-written backward) vs the same formulas as
device tensor expressions differentiated by autograd (EFGH_POSE_KERNELS=0 path, itself checked against the oracle and the
reference's golden outputs in test_gpu_forward / test_gpu_backward)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _expr(fn):
    """run fn with the tensor-expression path"""
    from efgh_amd.common import pose
    old, pose.USE_KERNELS = pose.USE_KERNELS, False
    try:
        return fn()
    finally:
        pose.USE_KERNELS = old


@pytest.mark.parametrize('nd,dest', [(3, (0., 0., 1.)), (2, (0., 1., 0.))])
def test_head_normal_forward_and_backward(nd, dest):
    from efgh_amd.common import pose
    torch.manual_seed(nd)
    B = 9
    raw = torch.randn(B, 32, device='cuda')
    sraw = torch.randn(B, 32, device='cuda')
    ga, gn, gR = torch.randn(B, nd, 1, device='cuda'), torch.randn(B, nd, 1, device='cuda'), torch.randn(B, 4, 4, device='cuda')
    res = []
    for kernels in (True, False):
        x = raw.clone().requires_grad_(True)
        run = lambda: pose.head_normal(x[:, :nd], sraw[:, :1 << nd], dest)
        a, n, R = run() if kernels else _expr(run)
        ((a * ga).sum() + (n * gn).sum() + (R * gR).sum()).backward()
        res.append((a.detach(), n.detach(), R.detach(), x.grad.clone()))
    for u, v, name in zip(res[0], res[1], ('abs', 'normal', 'R', 'grad')):
        tol = 2e-6 if name != 'grad' else 2e-5
        assert torch.allclose(u, v, rtol=1e-5, atol=tol), (name, float((u - v).abs().max()))
    assert float(res[0][3][:, nd:].abs().max()) == 0.0 and float(res[0][3][:, :nd].abs().max()) > 0


def test_head_normal_degenerate_directions():
    """normal already on (or opposite to) the destination axis: the constant rotations, zero gradient through R"""
    from efgh_amd.common import pose
    big = 200.0                                               # softmax saturates exactly: abs = (0, 0, 1)
    x = torch.tensor([[-big, -big, big], [-big, -big, big]], device='cuda').requires_grad_(True)
    s = torch.zeros(2, 8, device='cuda')
    s[0, 7] = 1.0                                             # + + +  -> normal = e3 ("same")
    s[1, 0] = 1.0                                             # - - -  -> normal = -e3 ("opposite")
    a, n, R = pose.head_normal(x, s, (0., 0., 1.))
    a2, n2, R2 = _expr(lambda: pose.head_normal(x.detach(), s, (0., 0., 1.)))
    assert torch.equal(R.detach(), R2) and torch.equal(n.detach(), n2)
    R.sum().backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().max()) == 0.0


def test_cam_T_velo_backward():
    from efgh_amd.common import pose
    torch.manual_seed(3)
    B = 5
    A = torch.eye(3, device='cuda').repeat(B, 1, 1)
    A[:, 0, 2], A[:, 1, 2] = -640.0, -192.0
    calib = torch.randn(B, 3, 4, device='cuda') * torch.tensor([700., 700., 1.], device='cuda')[None, :, None]
    g = torch.randn(B, 3, 4, device='cuda')
    res = []
    for kernels in (True, False):
        c = (torch.eye(3, device='cuda')[None] + 0.1 * torch.randn(B, 3, 3, device='cuda', generator=torch.Generator('cuda').manual_seed(1))).requires_grad_(True)
        l = torch.randn(B, 4, 4, device='cuda', generator=torch.Generator('cuda').manual_seed(2)).requires_grad_(True)
        run = lambda: pose.compute_cam_T_velo(c, l, calib, A)
        out = run() if kernels else _expr(run)
        (out * g).sum().backward()
        res.append((out.detach(), c.grad.clone(), l.grad.clone()))
    for u, v in zip(res[0], res[1]):
        assert torch.allclose(u, v, rtol=2e-5, atol=2e-5 * float(v.abs().max()))


def _loss_case(B, W, seed, raw=(128, 256)):
    from efgh_amd import synthetic as syn
    rs = np.random.RandomState(seed)
    args = syn.default_args(raw, 'cuda')
    dev = 'cuda'
    t = lambda a: torch.from_numpy(np.asarray(a, np.float32)).to(dev)

    def rot(rs, scale):
        from scipy.spatial.transform import Rotation
        M = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
        M[:, :3, :3] = Rotation.from_euler('xyz', rs.uniform(-scale, scale, (B, 3))).as_matrix()
        return M
    T4 = rot(rs, 0.2)
    T4[:, :3, 3] = rs.uniform(-1, 1, (B, 3))
    rc = rot(rs, 0.3)
    gt = {'rand_init_l': t(rot(rs, 0.5)), 'rand_init_c': t(rc[:, :3, :3].copy() if seed % 2 == 0 else rc),      # loaders hand over 3x3
          'sensor2_T_sensor1': t(T4)}
    e_l = rot(rs, 0.4)
    f_l = rot(rs, 3.0)
    f_l[:, :3, :3] = [[[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]] for a in rs.uniform(-3, 3, B)]
    soft = lambda x: np.exp(x) / np.exp(x).sum(1, keepdims=True)
    ea, ha = soft(rs.randn(B, 3)), soft(rs.randn(B, 2))
    pred = {'e_gn_abs': t(ea / np.linalg.norm(ea, axis=1, keepdims=True))[:, :, None], 'e_gn_sgn': t(rs.randn(B, 8)),
            'h_hrzn_abs': t(ha / np.linalg.norm(ha, axis=1, keepdims=True))[:, :, None], 'h_hrzn_sgn': t(rs.randn(B, 4)),
            'f_score': t(1.0 / (1.0 + np.exp(-2.0 * rs.randn(B, W)))), 'g_trs': t(rs.uniform(-1.5, 1.5, (B, 3, 1))),
            'e_l': t(e_l), 'f_l': t(f_l)}
    return args, gt, pred


@pytest.mark.parametrize('B,W,seed', [(1, 64, 0), (4, 160, 1), (8, 129, 2)])
def test_pose_loss_kernel_vs_expressions(B, W, seed):
    """every pose entry of the loss dictionary, every ground-truth tensor and the gradient w.r.t. every prediction"""
    from efgh_amd.losses import efghloss
    from efgh_amd.losses.efghloss import PoseLossFn
    args, gt, pred = _loss_case(B, W, seed)
    crit = efghloss.EFGHCriterion(args)
    cfg = (crit.lam, crit.positive_num, crit.neg_ratio)
    names = ('e_gn_abs', 'e_gn_sgn', 'h_hrzn_abs', 'h_hrzn_sgn', 'f_score', 'g_trs', 'e_l')
    weights = torch.rand(11, device='cuda') + 0.5                 # an arbitrary combination of the 11 entries
    l_dep0, l_msk0 = torch.tensor(0.37, device='cuda'), torch.tensor(0.61, device='cuda')

    # kernel
    p = {k: v.clone().requires_grad_(k in names) for k, v in pred.items()}
    l_dep, l_msk = l_dep0.clone().requires_grad_(True), l_msk0.clone().requires_grad_(True)
    Lv, gtbuf, gtcls, gtfs = PoseLossFn.apply(*[p[k] for k in names], p['f_l'], l_dep, l_msk, gt['rand_init_l'], gt['rand_init_c'],
                                              gt['sensor2_T_sensor1'], cfg)
    (Lv * weights).sum().backward()
    kg = {k: p[k].grad.clone() for k in names}
    kg['l_dep'], kg['l_msk'] = l_dep.grad.clone(), l_msk.grad.clone()

    # expressions: the pose part of _compute_loss_expressions, restated on the same inputs
    p2 = {k: v.clone().requires_grad_(k in names) for k, v in pred.items()}
    l_dep2, l_msk2 = l_dep0.clone().requires_grad_(True), l_msk0.clone().requires_grad_(True)
    L2, gt2 = crit._pose_terms_expressions(dict(gt), p2)
    L2['g_depth'] = l_dep2 * crit.lam['g_depth']
    L2['g_mask'] = (l_msk2 * crit.lam['g_mask']) * crit.lam['g_depth']
    total = 0
    for k in L2:
        total = total + L2[k]
    L2['total'] = total
    ref = torch.stack([L2[k] for k in crit.loss_name])
    (ref * weights).sum().backward()

    assert torch.allclose(Lv, ref.detach(), rtol=2e-5, atol=1e-5), (Lv, ref)
    for k in names:
        g, r = kg[k], p2[k].grad
        if r is None:
            assert float(g.abs().max()) == 0.0, k
            continue
        assert torch.allclose(g, r, rtol=1e-4, atol=1e-5 * max(1.0, float(r.abs().max()))), (k, float((g - r).abs().max()))
    assert torch.allclose(kg['l_dep'], l_dep2.grad) and torch.allclose(kg['l_msk'], l_msk2.grad)
    # ground truth
    from efgh_amd.losses.efghloss import _GT
    for k, (a, b) in _GT.items():
        want = gt2[k].reshape(B, -1)
        assert torch.allclose(gtbuf[:, a:b], want.detach(), rtol=1e-5, atol=2e-6), (k, float((gtbuf[:, a:b] - want).abs().max()))
    assert torch.equal(gtcls[:, 0], gt2['e_gn_sgn']) and torch.equal(gtcls[:, 1], gt2['h_hrzn_sgn'])
    assert torch.equal(gtfs, gt2['f_score'])


def test_pose_compose_forward_and_backward():
    from efgh_amd.common import pose
    torch.manual_seed(5)
    a0, b0, g = (torch.randn(6, 4, 4, device='cuda') for _ in range(3))
    a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    out = pose.compose(a, b)
    (out * g).sum().backward()
    a2, b2 = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    ref = torch.bmm(a2, b2)
    (ref * g).sum().backward()
    for u, v in ((out, ref), (a.grad, a2.grad), (b.grad, b2.grad)):
        assert torch.allclose(u.detach(), v.detach(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('mode', [0, 1])
def test_raster_pose_gradient_vs_expressions(mode):
    """d/d pose of the rasterised values, fused gather + contraction (float64 partial sums) vs the batched tensor expressions
    round 1 used (gather kernel + cat / bmm / sqrt / bmm)"""
    from efgh_amd import ops, synthetic as syn
    B, N, H, W = 2, 4096, 32, 128
    pc = torch.from_numpy(np.stack([syn.lidar_sweep(N, s) for s in range(B)])).cuda()
    torch.manual_seed(mode)
    if mode == 0:
        E = torch.eye(4, device='cuda').repeat(B, 1, 1)
        E[:, :3, :3] += 0.05 * torch.randn(B, 3, 3, device='cuda')
        img, pix = ops.range_image(pc, E, H, W, 0.125, -0.125)
    else:
        K = torch.tensor([[60., 0., W / 2, 0.], [0., 60., H / 2, 0.], [0., 0., 1., 0.]], device='cuda')
        T = torch.tensor([[0., -1., 0., 0.], [0., 0., -1., 0.], [1., 0., 0., 0.], [0., 0., 0., 1.]], device='cuda')
        P = (K @ T)[None].repeat(B, 1, 1).contiguous()
        img, pix = ops.depth_image(pc, P, H, W)
    g = torch.randn_like(img)
    got = ops.raster_pose_bwd(pix, g.contiguous(), pc, E if mode == 0 else None, B, N, H * W, mode)
    gv = ops.raster_bwd(pix, g.contiguous(), B, N, H * W).double()                    # (B,N,4)
    p1 = torch.cat([pc, torch.ones((B, 1, N), device='cuda')], 1).double()             # (B,4,N)
    if mode == 0:
        q = torch.bmm(E.double(), p1)
        r = torch.sqrt(torch.sum(q * q, 1, keepdim=True))
        gq = torch.zeros_like(q)
        gq[:, :3] = gv[..., :3].transpose(1, 2)
        gq = gq + gv[..., 3:4].transpose(1, 2) * (q / r)
        want = torch.bmm(gq, p1.transpose(1, 2)).reshape(B, 16)
    else:
        want = torch.zeros((B, 3, 4), dtype=torch.float64, device='cuda')
        want[:, 2] = torch.bmm(p1, gv[..., 3][:, :, None])[:, :, 0]
        want = want.reshape(B, 12)
    assert int((pix >= 0).sum()) > N // 4
    assert torch.allclose(got.double(), want, rtol=2e-5, atol=2e-5 * float(want.abs().max()))
