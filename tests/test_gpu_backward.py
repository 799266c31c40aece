"""Backward kernels (through the C-ABI + autograd shims) vs torch autograd on CPU, and the whole
training step (forward + efghloss + backward) vs the oracle and the reference's golden gradients."""
import os
import re

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from efgh_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _relerr(a, b):
    return float((a - b).norm() / (b.norm() + 1e-20))


def _mk_bn(c):
    bn = nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
        bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
    return bn


@pytest.mark.parametrize('cin,cout,k,s,p,hw,res', [
    (64, 64, 3, 1, 1, (12, 20), True), (64, 128, 3, 2, 1, (12, 20), False), (64, 128, 1, 2, 0, (12, 20), False),
    (3, 64, 3, 1, 1, (10, 14), False), (4, 3, (1, 2), 1, 0, (6, 17), False), (128, 128, 3, 1, 1, (7, 9), True),
    (1, 1, 3, 1, 1, (10, 12), False), (512, 512, 1, 1, 0, (4, 6), False),
    (64, 192, 3, 1, 1, (5, 13), False), (256, 64, 3, 1, 1, (3, 3), True), (128, 64, 3, 1, 1, (33, 70), False),
    (64, 64, 3, 1, 1, (70, 150), False), (64, 128, 3, 1, 1, (9, 200), True),     # several strips / row chunks / n-blocks of k_wino_wgrad_rows
    (64, 128, 3, 2, 1, (20, 300), False), (64, 64, 1, 2, 0, (11, 520), False),   # k_gather_wgrad mode 1 on wide grids: incremental row pointers with wraps
])
@pytest.mark.parametrize('train', [True, False])
def test_conv_bn_act_backward(cin, cout, k, s, p, hw, res, train):
    from efgh_amd import ops
    from efgh_amd.nets import layers as L
    torch.manual_seed(0)
    conv = nn.Conv2d(cin, cout, k, s, p, bias=True)
    bn = _mk_bn(cout)
    bn.train(train)
    x = torch.randn(2, cin, *hw, requires_grad=True)
    y0 = bn(conv(x))
    r = torch.randn_like(y0) if res else None
    if res:
        r.requires_grad_(True)
    y = F.relu(y0 + r) if res else F.leaky_relu(y0, 0.2)
    gy = torch.randn_like(y)
    # the activation's kink: outputs within 1e-4 of it could take the other branch under fp32 rounding of the
    # convolution (one flipped element of 3e5 is a 1e-3 relative change of the input gradient): no gradient there
    pre = (y0 + r) if res else y0
    gy = gy * (pre.detach().abs() > 1e-4)
    y.backward(gy)
    import copy
    conv_g, bn_g = copy.deepcopy(conv).cuda(), copy.deepcopy(bn).cuda()
    for m in (conv_g, bn_g):
        for p_ in m.parameters():
            p_.grad = None
    bn_g.train(train)
    cp = (cin + 3) // 4 * 4
    xg = ops.nchw_to_nhwc(x.detach().cuda(), cp).requires_grad_(True)
    rg = _nhwc(r.detach()).cuda().requires_grad_(True) if res else None
    yg = L.conv2d(L.Ctx(train), xg, conv_g, bn_g, L.ACT_RELU if res else L.ACT_LEAKY, 0.2, residual=rg)
    gyg = torch.zeros_like(yg)
    gyg[..., :cout] = _nhwc(gy).cuda()
    yg.backward(gyg)
    assert _relerr(yg[..., :cout].permute(0, 3, 1, 2).detach().cpu(), y.detach()) < 1e-5
    assert _relerr(xg.grad[..., :cin].permute(0, 3, 1, 2).cpu(), x.grad) < 2e-4
    assert _relerr(conv_g.weight.grad.cpu(), conv.weight.grad) < 2e-4
    assert _relerr(bn_g.weight.grad.cpu(), bn.weight.grad) < 2e-4
    assert _relerr(bn_g.bias.grad.cpu(), bn.bias.grad) < 2e-4
    if not train:
        assert _relerr(conv_g.bias.grad.cpu(), conv.bias.grad) < 2e-4
    if res:
        assert _relerr(rg.grad.permute(0, 3, 1, 2).cpu(), r.grad) < 2e-4


@pytest.mark.parametrize('cin,cout,pad,opad,hw', [(64, 32, 1, 0, (5, 9)), (32, 16, 0, 0, (7, 11)),
                                                  (128, 64, 1, 1, (6, 8)), (128, 2, 1, 1, (6, 8))])
def test_conv_transpose_backward(cin, cout, pad, opad, hw):
    import copy
    from efgh_amd.nets import layers as L
    torch.manual_seed(1)
    ct = nn.ConvTranspose2d(cin, cout, 3, 2, pad, opad, bias=False)
    bn = _mk_bn(cout)
    x = torch.randn(2, cin, *hw, requires_grad=True)
    y = F.leaky_relu(bn(ct(x)), 0.2)
    gy = torch.randn_like(y)
    y.backward(gy)
    ct_g, bn_g = copy.deepcopy(ct).cuda(), copy.deepcopy(bn).cuda()
    ct_g.weight.grad = None
    xg = _nhwc(x.detach()).cuda().requires_grad_(True)
    yg = L.conv_transpose2d(L.Ctx(True), xg, ct_g, bn_g, L.ACT_LEAKY, 0.2)
    gyg = torch.zeros_like(yg)
    gyg[..., :cout] = _nhwc(gy).cuda()
    yg.backward(gyg)
    assert _relerr(xg.grad.permute(0, 3, 1, 2).cpu(), x.grad) < 2e-4
    assert _relerr(ct_g.weight.grad.cpu(), ct.weight.grad) < 2e-4
    assert _relerr(bn_g.weight.grad.cpu(), bn.weight.grad) < 2e-4


def test_linear_pool_colmax_backward():
    import copy
    from efgh_amd import ops
    from efgh_amd.nets import fn as FN, layers as L
    torch.manual_seed(2)
    lin, bn = nn.Linear(128, 64), nn.BatchNorm1d(64)
    a = torch.randn(50, 128, requires_grad=True)
    y = F.relu(bn(lin(a)))
    seg = [0, 20, 50]
    z = torch.stack([y[:20].max(0)[0], y[20:].max(0)[0]])
    gz = torch.randn_like(z)
    z.backward(gz)
    lin_g, bn_g = copy.deepcopy(lin).cuda(), copy.deepcopy(bn).cuda()
    lin_g.weight.grad = lin_g.bias.grad = None
    ag = a.detach().cuda().requires_grad_(True)
    yg = L.linear_rows(L.Ctx(True), ag, 50, 128, lin_g.weight, lin_g.bias, bn=bn_g, act=L.ACT_RELU)
    zg = FN.SegmentColMaxFn.apply(yg, torch.tensor(seg, dtype=torch.int32).cuda(), 2, 64)
    zg.backward(gz.cuda())
    assert _relerr(zg.detach().cpu(), z.detach()) < 1e-5
    assert _relerr(ag.grad.cpu(), a.grad) < 2e-4 and _relerr(lin_g.weight.grad.cpu(), lin.weight.grad) < 2e-4
    x = torch.randn(2, 8, 10, 14, requires_grad=True)
    p = F.max_pool2d(x, 2, 2)
    gp = torch.randn_like(p)
    p.backward(gp)
    xg = _nhwc(x.detach()).cuda().requires_grad_(True)
    pg = FN.MaxPool2Fn.apply(xg)
    pg.backward(_nhwc(gp).cuda())
    assert torch.equal(xg.grad.permute(0, 3, 1, 2).cpu(), x.grad)


def test_bcl_level_backward():
    from efgh_amd import lattice
    from efgh_amd.nets import fn as FN, layers as L
    from efgh_amd.nets.builders import BilateralConvFlex
    from oracle import efgh_oracle as O, lattice as olat
    torch.manual_seed(4)
    pc = syn.lidar_sweep(2048, 2)
    ref_lv = olat.generate_data(pc)[0]
    lv = lattice.build_pyramid(torch.from_numpy(pc).cuda(), (1.0,))[0]
    C = 36
    feat = torch.randn(2048, C)
    m = BilateralConvFlex(C, [32, 48])
    with torch.no_grad():
        for p in m.parameters():
            p.normal_(0, 0.2)
    P = {'b.' + k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in m.state_dict().items()}
    fr = feat.t().contiguous().requires_grad_(True)
    ref = O.bcl(P, 'b', fr, torch.from_numpy(ref_lv['bary']), torch.from_numpy(ref_lv['off']),
                torch.from_numpy(ref_lv['nbr']))
    g = torch.randn_like(ref)
    ref.backward(g)
    m = m.cuda()
    fg = feat.cuda().requires_grad_(True)
    splat = FN.SplatFn.apply(fg, lv, C, False)            # all 36 channels from the feature rows (no el_minus_gr part)
    out = L.blur_conv(L.Ctx(True), splat, lv.H, C, lv, m.blur_conv[0], m.blur_conv[2])
    out.backward(g.t().contiguous().cuda())
    assert _relerr(fg.grad.t().cpu(), fr.grad) < 2e-4
    assert _relerr(m.blur_conv[0].weight.grad.cpu(), P['b.blur_conv.0.weight'].grad) < 2e-4
    assert _relerr(m.blur_conv[0].bias.grad.cpu(), P['b.blur_conv.0.bias'].grad) < 2e-4
    assert _relerr(m.blur_conv[2].weight.grad.cpu(), P['b.blur_conv.2.weight'].grad) < 2e-4


@pytest.mark.parametrize('mfma', [True, False])
@pytest.mark.parametrize('h,wc,wr', [(9, 21, 85), (8, 37, 150), (3, 5, 24)])
def test_corr_head_backward(mfma, h, wc, wr):
    """forward + both gradients of the F correlation head: VALU kernels and the Toeplitz-GEMM (MFMA) formulation"""
    from efgh_amd import ops
    from efgh_amd.nets import fn as FN
    from oracle import efgh_oracle as O
    torch.manual_seed(5)
    cam = torch.randn(2, 16, h, wc, requires_grad=True)
    rng = torch.randn(2, 16, h, wr, requires_grad=True)
    refs = []
    for b in range(2):                                 # B independent evaluations
        c = cam[b:b + 1] / (cam[b].max() - cam[b].min())
        r = rng[b:b + 1] / (rng[b].max() - rng[b].min())
        refs.append(torch.sigmoid(F.conv2d(O.circular_assign(r, int(wr / 8)), c).view(1, -1) / 16))
    ref = torch.cat(refs, 0)
    g = torch.randn_like(ref)
    ref.backward(g)
    cg, rg = _nhwc(cam.detach()).cuda().requires_grad_(True), _nhwc(rng.detach()).cuda().requires_grad_(True)
    old = ops.USE_MFMA_CORR
    ops.USE_MFMA_CORR = mfma
    try:
        s = FN.CorrHeadFn.apply(cg, rg)
        s.backward(g.cuda())
    finally:
        ops.USE_MFMA_CORR = old
    assert _relerr(s.detach().cpu(), ref.detach()) < 1e-5
    assert _relerr(cg.grad.permute(0, 3, 1, 2).cpu(), cam.grad) < 2e-4
    assert _relerr(rg.grad.permute(0, 3, 1, 2).cpu(), rng.grad) < 2e-4


def test_training_step_vs_oracle_and_golden(golden_dir, manifest, monkeypatch):
    """forward + efghloss + backward at the golden size; the uint8 rotate is teacher-forced to the
    oracle's h_img (a 1-ulp angle difference would move pixels, see test_gpu_forward)."""
    from efgh_amd import ops
    from efgh_amd.losses import EFGHCriterion
    from efgh_amd.nets import EFGHBackbone
    from oracle import efgh_oracle as O
    RAW, NPTS = (128, 256), 2048
    G = np.load(os.path.join(golden_dir, 'e2e_small.npz'))
    args_c, args_g = syn.default_args(RAW, 'cpu'), syn.default_args(RAW, 'cuda')
    b = syn.make_batch(RAW, NPTS, 1)
    T = torch.from_numpy
    cpu = [T(b[k]) for k in ('pc', 'img', 'calib', 'A')]
    # ---- oracle
    P = syn.synthetic_state_dict(manifest['state_dict'], 1)
    for k in manifest['parameters']:
        P[k].requires_grad_(True)
    pred_o = O.forward(P, *cpu, args_c, train=True)
    L_o, _ = O.compute_loss(cpu[0], {k: T(v) for k, v in b['gt'].items()}, pred_o, args_c)
    L_o['total'].backward()
    # ---- ours (rotate teacher-forced)
    h_img_o = pred_o['h_img'].detach().cuda()
    monkeypatch.setattr(ops, 'rotate_nearest_u8',
                        lambda img, rot, **kw: (h_img_o, ops.nchw_to_nhwc(h_img_o, 4)))
    m = EFGHBackbone(args_g)
    m.load_state_dict(syn.synthetic_state_dict(manifest['state_dict'], 1))
    m = m.cuda().train()
    crit = EFGHCriterion(args_g)
    gpu = [t.cuda() for t in cpu]
    pred = m(*gpu)
    L, gt2 = crit.compute_loss(gpu[0], gpu[1], gpu[2], gpu[3], {k: T(v) for k, v in b['gt'].items()}, pred)
    for k in L_o:
        assert abs(L[k].item() - L_o[k].item()) <= 2e-4 * abs(L_o[k].item()) + 1e-6, (k, L[k].item(), L_o[k].item())
        assert abs(L[k].item() - float(G['train.loss.' + k])) <= 2e-4 * abs(float(G['train.loss.' + k])) + 1e-6, k
    L['total'].backward()
    names = manifest['parameters']
    params = dict(m.named_parameters())
    ref_norm = G['train.grad_norm']
    errs = {'E': [], 'H': [], 'F': [], 'G': []}
    for i, k in enumerate(names):
        g_o = P[k].grad
        g = params[k].grad
        assert g is not None, k
        g = g.cpu()
        if re.search(r'(features\.\d+|conv_gn_\d|conv_hrzn_\d|E\.bcn5\.blur_conv\.2)\.bias$', k):
            continue            # bias in front of a train-mode BatchNorm: analytically zero, rounding noise
        errs[k[0]].append((_relerr(g, g_o), k))
        if k[0] != 'F':
            assert abs(float(g.double().norm()) - ref_norm[i]) <= 3e-2 * ref_norm[i] + 1e-9, (k, float(g.norm()), ref_norm[i])
    # Tolerances are set from the ORACLE'S OWN reproducibility: the same torch-CPU oracle run with 1 thread /
    # native convs vs 8 threads / oneDNN differs by <=1.8e-5 (E, H), <=1.4e-2 (G: BatchNorm backward over 128
    # positions behind a mean-pool head cancels catastrophically in fp32) and up to 0.39 (F: max/min
    # normalisation + hard-negative mining are discontinuous) on this very case (measured).
    # Every backward kernel is separately checked at 2e-4 against torch autograd in the tests above.
    worst = {n: max(v) for n, v in errs.items()}
    assert worst['E'][0] < 2e-3, worst['E']
    assert worst['H'][0] < 2e-3, worst['H']
    assert worst['G'][0] < 5e-2, worst['G']
    med_f = sorted(e for e, _ in errs['F'])[len(errs['F']) // 2]
    assert med_f < 5e-2, (med_f, worst['F'])
    # the WORST F parameter as well (round 5; measured 5.9e-3 - the bound is the oracle's own 1- vs 8-thread spread of the terms above, not
    # the 0.39 of its mined / normalised ones: those flip as a whole or not at all, and on this case they do not)
    assert worst['F'][0] < 5e-2, worst['F']
    print('grad rel err: ' + ', '.join('%s max %.1e' % (n, worst[n][0]) for n in 'EHFG') + ', F median %.1e' % med_f)


def test_training_step_batch2_vs_batched_oracle(manifest, monkeypatch):
    """train-mode B = 2: the only place the samples couple is BatchNorm over the per-GPU batch.  The oracle
    restates that definition on the CPU (per-sample lattices / rasters / correlation, BatchNorm statistics over both samples,
    E head over the vertices of both) - an external check of what round 1 only tested against itself: every loss term, the
    BatchNorm running statistics after the step, and the per-net gradients."""
    from efgh_amd import ops
    from efgh_amd.losses import EFGHCriterion
    from efgh_amd.nets import EFGHBackbone
    from oracle import efgh_oracle as O
    RAW, NPTS = (128, 256), 2048
    args_c, args_g = syn.default_args(RAW, 'cpu'), syn.default_args(RAW, 'cuda')
    b = syn.make_batch(RAW, NPTS, 2)
    T = torch.from_numpy
    cpu = [T(b[k]) for k in ('pc', 'img', 'calib', 'A')]
    P = syn.synthetic_state_dict(manifest['state_dict'], 1)
    for k in manifest['parameters']:
        P[k].requires_grad_(True)
    pred_o = O.forward(P, *cpu, args_c, train=True)
    L_o, _ = O.compute_loss(cpu[0], {k: T(v) for k, v in b['gt'].items()}, pred_o, args_c)
    L_o['total'].backward()
    h_img_o = pred_o['h_img'].detach().cuda()
    monkeypatch.setattr(ops, 'rotate_nearest_u8', lambda img, rot, **kw: (h_img_o, ops.nchw_to_nhwc(h_img_o, 4)))
    m = EFGHBackbone(args_g)
    m.load_state_dict(syn.synthetic_state_dict(manifest['state_dict'], 1))
    m = m.cuda().train()
    gpu = [t.cuda() for t in cpu]
    pred = m(*gpu)
    L, _ = EFGHCriterion(args_g).compute_loss(gpu[0], gpu[1], gpu[2], gpu[3], {k: T(v) for k, v in b['gt'].items()}, pred)
    for k in ('e_gn_sgn', 'e_gn_abs', 'h_hrzn_sgn', 'h_hrzn_abs', 'g_trs'):
        assert _relerr(pred[k].detach().cpu(), pred_o[k].detach()) < 2e-4, k
    for k in L_o:
        assert abs(L[k].item() - L_o[k].item()) <= 5e-4 * abs(L_o[k].item()) + 1e-6, (k, L[k].item(), L_o[k].item())
    # running statistics after one train-mode forward (momentum 0.1, unbiased variance over BOTH samples)
    sd = m.state_dict()
    for k in ('H.vgg.features.1.running_mean', 'H.vgg.features.1.running_var', 'E.bn_gn_1.running_mean', 'E.bn_gn_3.running_var',
              'G.conv_img2.0.bn1.running_var', 'F.vgg_range.features.5.running_mean'):
        assert _relerr(sd[k].cpu(), P[k].detach()) < 1e-4, k
    L['total'].backward()
    params = dict(m.named_parameters())
    num, den = {n: 0.0 for n in 'EHG'}, {n: 0.0 for n in 'EHG'}
    for k in manifest['parameters']:
        if k[0] == 'F' or re.search(r'(features\.\d+|conv_gn_\d|conv_hrzn_\d|E\.bcn5\.blur_conv\.2)\.bias$', k):
            continue
        g, ref = params[k].grad.cpu().double(), P[k].grad.double()
        num[k[0]] += float((g - ref).pow(2).sum())
        den[k[0]] += float(ref.pow(2).sum())
    rel = {n: (num[n] / max(den[n], 1e-300)) ** 0.5 for n in num}
    print('B=2 gradient rel err per net:', {n: '%.2e' % v for n, v in rel.items()})
    assert rel['E'] < 2e-3 and rel['H'] < 2e-3 and rel['G'] < 5e-2, rel


def test_g_image_losses_vs_torch_autograd():
    """GImageLossFn (masked L2 on the depth image + BCE on the mask image) against the torch expressions of loss_utils.py:186-199"""
    from efgh_amd.nets import fn as FN
    torch.manual_seed(3)
    B, H, W = 2, 37, 53
    pd = torch.randn(B, 1, H, W, requires_grad=True)
    pm = torch.softmax(torch.randn(B, 2, H, W) * 3, 1).requires_grad_(True)
    gd = torch.relu(torch.randn(B, 1, H, W)) * 5
    gd[0, 0, :5] = 0
    im = (torch.rand(B, 1, H, W) > 0.3).to(torch.uint8)
    valid = (gd > 0) & (im > 0)
    l_dep = (((gd - pd) * valid) ** 2).sum() / valid.sum()
    l_msk = F.binary_cross_entropy(pm[:, 0].reshape(B, -1), (gd > 0).float().view(B, -1))
    (3.0 * l_dep + 0.5 * l_msk).backward()
    gdep4 = torch.zeros(B, H, W, 4)
    gdep4[..., 3] = gd[:, 0]
    pdg, pmg = pd.detach().cuda().requires_grad_(True), pm.detach().cuda().requires_grad_(True)
    a, b, gtd, gtm, nval = FN.GImageLossFn.apply(pdg, pmg, gdep4.cuda(), im.cuda())
    (3.0 * a + 0.5 * b).backward()
    assert abs(float(a) - float(l_dep)) < 1e-5 * float(l_dep) and abs(float(b) - float(l_msk)) < 1e-5 * float(l_msk)
    assert torch.equal(gtd.cpu(), gd) and torch.equal(gtm.cpu(), (gd > 0).float()) and float(nval) == float(valid.sum())
    assert _relerr(pdg.grad.cpu(), pd.grad) < 1e-5 and _relerr(pmg.grad.cpu(), pm.grad) < 1e-5


@pytest.mark.parametrize('hw', [(12, 20), (9, 13)])
def test_vgg_block_with_fused_bn_relu_pool_backward(hw):
    """conv3x3 + train-mode BatchNorm + ReLU + MaxPool2d(2,2) on the training path (BatchNorm/ReLU/pool fused over the raw conv
    output, the full-resolution activation is never stored) against torch autograd; odd sizes leave the last row/column unpooled"""
    from efgh_amd import ops
    from efgh_amd.nets import layers as L
    import copy
    torch.manual_seed(2)
    feats = nn.Sequential(nn.Conv2d(64, 64, 3, padding=1), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(2, 2),
                          nn.Conv2d(64, 128, 3, padding=1), nn.BatchNorm2d(128), nn.ReLU())
    for mod in feats:
        if isinstance(mod, nn.BatchNorm2d):
            with torch.no_grad():
                mod.weight.uniform_(0.5, 1.5); mod.bias.normal_(0, 0.2)
    feats.train()
    x = torch.randn(2, 64, *hw, requires_grad=True)
    y = feats(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    fg = copy.deepcopy(feats).cuda()
    for p_ in fg.parameters():
        p_.grad = None
    fg.train()
    xg = _nhwc(x.detach()).cuda().requires_grad_(True)
    yg = L.run_vgg(L.Ctx(True), fg, xg)
    yg.backward(_nhwc(gy).cuda())
    assert yg.shape[1:3] == (hw[0] // 2, hw[1] // 2)
    assert _relerr(yg.permute(0, 3, 1, 2).detach().cpu(), y.detach()) < 1e-5
    assert _relerr(xg.grad.permute(0, 3, 1, 2).cpu(), x.grad) < 3e-4
    for (n, p_), (_, q_) in zip(fg.named_parameters(), feats.named_parameters()):
        if n.endswith('0.bias') or n.endswith('4.bias'):
            continue                                      # conv bias in front of a train-mode BatchNorm: rounding noise
        assert _relerr(p_.grad.cpu(), q_.grad) < 3e-4, n


def test_fold_unpack_in_one_launch_equals_fold_then_unpack():
    """a split weight gradient's final fold writes the reference (out, in, kh, kw) layout itself (the explicit `efgh_wgrad_out_desc *out` argument of the weight-gradient entry points, ABI 3; k_fold_splits<true>)
    instead of a packed plane that k_unpack_weight re-reads: bit-identical to the two-launch form for the generic kernel, the
    small-channel and 4-channel kernels and a padded-channel layer"""
    from efgh_amd import ops
    from efgh_amd.nets import layers as L
    import torch.nn as nn
    torch.manual_seed(4)
    hits0 = ops.FOLD_UNPACK_HITS[0]
    cases = [(nn.Conv2d(64, 128, 3, 2, 1), (2, 64, 40, 56)), (nn.Conv2d(128, 64, 1, 1, 0), (2, 128, 30, 44)),
             (nn.Conv2d(4, 64, 3, 1, 1), (2, 4, 64, 96)), (nn.Conv2d(32, 32, 3, 1, 1), (2, 32, 64, 96)),
             (nn.Conv2d(3, 64, 3, 1, 1), (1, 3, 48, 80)),
             (nn.Conv2d(128, 256, 3, 1, 1), (2, 128, 24, 40)),      # 2-D Winograd weight gradient: k_w2_wfinish writes the layout
             (nn.Conv2d(64, 64, 3, 1, 1), (2, 64, 48, 80))]         # 1-D Winograd weight gradient: k_wino_wgrad_finish3 does
    for conv, shp in cases:
        conv = conv.cuda()
        x = torch.randn(*shp, device='cuda')
        cin = shp[1]
        xg = ops.nchw_to_nhwc(x, -(-cin // 4) * 4).requires_grad_(True)
        grads = {}
        for fused in (True, False):
            ops.FOLD_UNPACK = fused
            try:
                conv.zero_grad()
                y = L.conv2d(L.Ctx(True), xg, conv, None, L.ACT_NONE)
                (y * torch.linspace(-1, 1, y.numel(), device='cuda').view_as(y)).sum().backward()
                grads[fused] = conv.weight.grad.clone()
            finally:
                ops.FOLD_UNPACK = True
        assert torch.equal(grads[True], grads[False]), (type(conv), shp)
        assert float(grads[True].abs().max()) > 0
    assert ops.FOLD_UNPACK_HITS[0] >= hits0 + 5, (ops.FOLD_UNPACK_HITS[0], hits0)      # (most of these split their rows; the Winograd finishes always take it)
