"""Unit parity of the HIP ops (through the C-ABI) against torch fp32 CPU references."""
import math

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.fixture(scope='module')
def L():
    from efgh_amd.nets import layers
    return layers


@pytest.mark.parametrize('cin,cout,k,s,p,hw', [
    (64, 64, 3, 1, 1, (24, 40)), (64, 128, 3, 2, 1, (24, 40)), (64, 128, 1, 2, 0, (24, 40)),
    (3, 64, 3, 1, 1, (16, 24)), (128, 256, 3, 1, 1, (9, 13)), (4, 32, 3, 2, 1, (20, 28)),
    (512, 512, 1, 1, 0, (6, 10)), (16, 16, 3, 1, 1, (29, 61)), (1, 1, 3, 1, 1, (16, 24)),
    (260, 72, 3, 1, 1, (7, 9)),
])
@pytest.mark.parametrize('train', [False, True])
def test_conv_bn_act(L, cin, cout, k, s, p, hw, train):
    torch.manual_seed(0)
    conv = nn.Conv2d(cin, cout, k, s, p, bias=True)
    bn = nn.BatchNorm2d(cout)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
        bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
    x = torch.randn(2, cin, *hw)
    bn.train(train)
    ref = F.leaky_relu(bn(conv(x)), 0.2)
    rm_ref, rv_ref = bn.running_mean.clone(), bn.running_var.clone()
    # reset running stats, run ours
    conv_g, bn_g = nn.Conv2d(cin, cout, k, s, p, bias=True).cuda(), nn.BatchNorm2d(cout).cuda()
    conv_g.load_state_dict(conv.state_dict())
    torch.manual_seed(0)
    bn2 = nn.BatchNorm2d(cout)
    with torch.no_grad():
        nn.Conv2d(cin, cout, k, s, p, bias=True)       # consume the same RNG stream
        bn2.weight.uniform_(0.5, 1.5); bn2.bias.normal_(0, 0.2)
        bn2.running_mean.normal_(0, 0.2); bn2.running_var.uniform_(0.5, 1.5)
    bn_g.load_state_dict(bn2.state_dict())
    bn_g.train(train)
    from efgh_amd import ops
    xg = ops.nchw_to_nhwc(x.cuda(), (cin + 3) // 4 * 4)
    y = L.conv2d(L.Ctx(train), xg, conv_g, bn_g, L.ACT_LEAKY, 0.2)
    got = y[..., :cout].permute(0, 3, 1, 2).cpu()
    assert _rel(got, ref) < 2e-5
    if train:
        assert _rel(bn_g.running_mean.cpu(), rm_ref) < 1e-5 and _rel(bn_g.running_var.cpu(), rv_ref) < 1e-5


@pytest.mark.parametrize('cin,cout,pad,opad,hw', [(512, 128, 1, 0, (4, 8)), (128, 32, 0, 0, (7, 15)),
                                                  (32, 16, 1, 0, (15, 31)), (512, 256, 1, 1, (6, 10)),
                                                  (128, 1, 1, 1, (12, 20)), (128, 2, 1, 1, (12, 20)), (128, 3, 1, 1, (12, 20)),
                                                  (64, 3, 0, 0, (7, 9)), (128, 2, 1, 1, (5, 300))])
@pytest.mark.parametrize('train', [False, True])
def test_conv_transpose(L, cin, cout, pad, opad, hw, train):
    torch.manual_seed(1)
    ct = nn.ConvTranspose2d(cin, cout, 3, 2, pad, opad, bias=False)
    bn = nn.BatchNorm2d(cout)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
        bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
    sd_bn = {k: v.clone() for k, v in bn.state_dict().items()}
    x = torch.randn(2, cin, *hw)
    bn.train(train)
    ref = F.leaky_relu(bn(ct(x)), 0.2)
    ct_g, bn_g = nn.ConvTranspose2d(cin, cout, 3, 2, pad, opad, bias=False).cuda(), nn.BatchNorm2d(cout).cuda()
    ct_g.load_state_dict(ct.state_dict()); bn_g.load_state_dict(sd_bn); bn_g.train(train)
    xg = _nhwc(x).cuda().requires_grad_(train and cout <= 3)
    y = L.conv_transpose2d(L.Ctx(train), xg, ct_g, bn_g, L.ACT_LEAKY, 0.2)
    got = y[..., :cout].permute(0, 3, 1, 2).cpu()
    assert got.shape == ref.shape
    assert _rel(got.detach(), ref.detach()) < 2e-5
    if train and cout <= 3:
        # the GEMM + col2im form of the narrow heads: data and weight gradient (im2col of the gradient + plain GEMMs) vs autograd
        gy = torch.randn_like(ref)
        xr = x.clone().requires_grad_(True)
        ct.weight.grad = None
        F.leaky_relu(bn(ct(xr)), 0.2).backward(gy)
        gyp = torch.zeros(y.shape)
        gyp[..., :cout] = gy.permute(0, 2, 3, 1)
        y.backward(gyp.cuda())
        assert _rel(ct_g.weight.grad.cpu(), ct.weight.grad) < 2e-4, _rel(ct_g.weight.grad.cpu(), ct.weight.grad)
        assert _rel(xg.grad[..., :cin].permute(0, 3, 1, 2).cpu(), xr.grad) < 2e-4


def test_conv_1x2_and_residual_and_concat(L):
    torch.manual_seed(2)
    from efgh_amd import ops
    torch.set_grad_enabled(False)          # inference path: fused epilogues, channel-slice outputs
    conv = nn.Conv2d(4, 3, (1, 2), 1, 0, bias=False)
    x = torch.randn(1, 4, 8, 33)
    ref = conv(x)
    y = L.conv2d(L.Ctx(False), _nhwc(x).cuda(), conv.cuda(), None)
    assert _rel(y[..., :3].permute(0, 3, 1, 2).cpu(), ref) < 1e-5
    assert float(y[..., 3].abs().max()) == 0.0
    # residual + relu into a channel slice of a wider buffer, reading a channel slice
    c2 = nn.Conv2d(64, 64, 3, 1, 1, bias=False)
    xin, res = torch.randn(2, 128, 10, 12), torch.randn(2, 64, 10, 12)
    ref = F.relu(c2(xin[:, 64:]) + res)
    buf = torch.full((2, 10, 12, 192), 7.0).cuda()
    L.conv2d(L.Ctx(False), _nhwc(xin).cuda(), c2.cuda(), None, L.ACT_RELU, residual=_nhwc(res).cuda(),
             out=(buf, 64), in_ch=(64, 64))
    assert _rel(buf[..., 64:128].permute(0, 3, 1, 2).cpu(), ref) < 1e-5
    assert float((buf[..., :64] - 7).abs().max()) == 0 and float((buf[..., 128:] - 7).abs().max()) == 0
    torch.set_grad_enabled(True)


def test_maxpool_linear_colmax(L):
    from efgh_amd import ops
    torch.manual_seed(3)
    x = torch.randn(2, 8, 10, 14)
    assert torch.equal(ops.maxpool2(_nhwc(x).cuda()).permute(0, 3, 1, 2).cpu(), F.max_pool2d(x, 2, 2))
    lin = nn.Linear(128, 32)
    a = torch.randn(5, 128)
    y = L.linear_rows(L.Ctx(False), a.cuda(), 5, 128, lin.cuda().weight, lin.bias, act=L.ACT_RELU)
    assert _rel(y.cpu(), F.relu(F.linear(a, lin.weight.cpu(), lin.bias.cpu()))) < 1e-5
    m = torch.randn(37, 128)
    seg = torch.tensor([0, 10, 37], dtype=torch.int32).cuda()
    mx, _ = ops.segment_colmax(m.cuda(), 128, 128, seg, 2)
    assert torch.equal(mx.cpu(), torch.stack([m[:10].max(0)[0], m[10:].max(0)[0]]))


def test_splat_and_blur(golden_dir):
    """BCL level: splat + normalise + neighbour gather + blur conv vs the oracle's bcl()."""
    from efgh_amd import lattice, ops, synthetic as syn
    from efgh_amd.nets import layers as L
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
    P = {'b.' + k: v for k, v in m.state_dict().items()}
    ref = O.bcl(P, 'b', feat.t().contiguous(), torch.from_numpy(ref_lv['bary']), torch.from_numpy(ref_lv['off']),
                torch.from_numpy(ref_lv['nbr']))
    m = m.cuda()
    splat, _ = ops.splat_fwd(lv, feat.cuda(), C, use_emg=False)
    out = L.blur_conv(L.Ctx(False), splat, lv.H, C, lv, m.blur_conv[0], m.blur_conv[2])
    assert _rel(out[:, :48].t().cpu(), ref) < 2e-5
    # the production form: el_minus_gr read from the lattice's own array + 32 feature channels == the concatenated row
    feat2 = torch.cat([lv.emg.t().cpu(), feat[:, 4:]], 1)
    s_cat, w_cat = ops.splat_fwd(lv, feat2.cuda(), C, use_emg=False)
    s_two, w_two = ops.splat_fwd(lv, feat[:, 4:].contiguous().cuda(), C - 4, use_emg=True)
    assert torch.equal(s_cat, s_two) and torch.equal(w_cat, w_two)


def test_rotate_golden(golden_dir):
    import os
    from efgh_amd import ops
    R = np.load(os.path.join(golden_dir, 'rotate_cases.npz'))
    for i in range(int(R['count'])):
        img = torch.from_numpy(R[f'img{i}'].astype(np.float32)).cuda()
        mat = torch.from_numpy(R[f'mat{i}'])
        rot_deg = torch.rad2deg(torch.atan2(mat[:, 1, 0], mat[:, 0, 0]))        # CPU fp32, as the reference
        o1, o2 = ops.rotate_nearest_u8(img, rot_deg.cuda())
        assert np.array_equal(o1.cpu().numpy().astype(np.uint8), R[f'out{i}']), i
        assert np.array_equal(o2[..., :3].permute(0, 3, 1, 2).cpu().numpy().astype(np.uint8), R[f'out{i}']), i


def _last_wins(h, w, pix, vals):
    """numpy restatement of the scatter: the point with the largest index wins its pixel"""
    img = np.zeros((h * w, 4), np.float32)
    win = np.full(h * w, -1, np.int64)
    ok = pix >= 0
    np.maximum.at(win, pix[ok], np.nonzero(ok)[0])
    hit = win >= 0
    img[hit] = vals[win[hit]]
    return img.reshape(h, w, 4)


def test_rasterisers_golden(golden_dir):
    """range / depth image against the reference's outputs.  Two exact statements instead of a mismatch budget:
    (1) given the kernel's own pixel index per point, the image IS the last-point-wins scatter of (x, y, z, r) / (px, py, pz, w);
    (2) the pixel index of every point equals the float64 evaluation of the reference's formula wherever that is unambiguous -
        i.e. except for points whose continuous coordinate lies within 2e-4 of a pixel or field-of-view boundary (the reference
        evaluates asin / atan2 / the division in fp32 on the CPU, so exactly those points may legitimately fall either way) -
        and outside the pixels such points can touch the image equals the reference's golden image exactly."""
    import os
    from efgh_amd import ops
    R = np.load(os.path.join(golden_dir, 'raster_cases.npz'))
    # ---- range image (torch_utils.py:11-59)
    epc = R['range.pc'].astype(np.float32)                                     # already e_l-rotated, (1,4,N)
    H, W = 32, 256
    up, down = 0.125 * math.pi, -0.125 * math.pi
    img, pix = ops.range_image(torch.from_numpy(epc[:, :3]).cuda(), torch.eye(4)[None].cuda(), H, W, up, down)
    got, pix = img[0].cpu().numpy(), pix[0].cpu().numpy().astype(np.int64)
    x, y, z = (epc[0, i].astype(np.float64) for i in range(3))
    r = np.sqrt(x * x + y * y + z * z + 1.0)
    pitch, yaw = np.arcsin(z / r), np.arctan2(y, x)
    u, v = (up - pitch) / (up - down) * (H - 1), (-yaw + math.pi) / (2 * math.pi) * (W - 1)
    inside = (pitch < up) & (pitch > down)
    amb = (np.abs(u - np.round(u)) < 2e-4) | (np.abs(v - np.round(v)) < 2e-4) | (np.abs(pitch - up) < 1e-6) | (np.abs(pitch - down) < 1e-6)
    want = np.where(inside, np.floor(u).astype(np.int64) * W + np.floor(v).astype(np.int64), -1)
    mism = pix != want
    assert not (mism & ~amb).any() and mism.mean() < 2e-3          # only boundary points may differ, and only a handful do
    r32 = np.sqrt((epc[0, 0] * epc[0, 0] + epc[0, 1] * epc[0, 1] + epc[0, 2] * epc[0, 2] + np.float32(1.0)).astype(np.float32))
    vals = np.stack([epc[0, 0], epc[0, 1], epc[0, 2], r32], 1)
    mine = _last_wins(H, W, pix, vals)
    assert np.array_equal(got[..., :3], mine[..., :3])                          # (1): copies are exact
    assert np.abs(got[..., 3] - mine[..., 3]).max() <= 1e-6 * np.abs(mine[..., 3]).max()
    touched = np.zeros(H * W, bool)                                            # pixels a differently-assigned point touches
    for q in (pix[mism], want[mism]):
        touched[q[q >= 0]] = True
    gold = R['range.out'][0].transpose(1, 2, 0).reshape(H * W, 4)
    keep = ~touched
    assert np.array_equal(got.reshape(H * W, 4)[keep][:, :3], gold[keep][:, :3])   # (2)
    assert np.abs(got.reshape(H * W, 4)[keep][:, 3] - gold[keep][:, 3]).max() <= 2e-6 * np.abs(gold[:, 3]).max()
    # ---- depth image (torch_utils.py:61-103)
    pc, P = R['depth.pc'].astype(np.float32), R['depth.calib'].astype(np.float32)
    H, W = 64, 128
    dep, dpix = ops.depth_image(torch.from_numpy(pc).cuda(), torch.from_numpy(P).cuda(), H, W)
    got, dpix = dep[0].cpu().numpy(), dpix[0].cpu().numpy().astype(np.int64)
    p4 = np.concatenate([pc[0, :3].astype(np.float64), np.ones((1, pc.shape[2]))], 0)
    xyw = P[0].astype(np.float64) @ p4
    w_, xx, yy = xyw[2], xyw[0] / xyw[2], xyw[1] / xyw[2]
    inside = (xx < W) & (xx > 0) & (yy < H) & (yy > 0) & (w_ > 0)
    amb = (np.abs(xx - np.round(xx)) < 2e-4) | (np.abs(yy - np.round(yy)) < 2e-4) | (np.abs(w_) < 1e-6)
    want = np.where(inside, np.floor(np.where(inside, yy, 0)).astype(np.int64) * W + np.floor(np.where(inside, xx, 0)).astype(np.int64), -1)
    mism = dpix != want
    assert not (mism & ~amb).any() and mism.mean() < 2e-3
    gold = R['depth.out'][0].transpose(1, 2, 0).reshape(H * W, 4)
    touched = np.zeros(H * W, bool)
    for q in (dpix[mism], want[mism]):
        touched[q[q >= 0]] = True
    keep = ~touched
    assert np.array_equal(got.reshape(H * W, 4)[keep][:, :3], gold[keep][:, :3])
    assert np.abs(got.reshape(H * W, 4)[keep][:, 3] - gold[keep][:, 3]).max() <= 2e-6 * max(1.0, np.abs(gold[:, 3]).max())


def test_corr_head():
    from efgh_amd import ops
    from oracle import efgh_oracle as O
    torch.manual_seed(5)
    cam, rng = torch.randn(2, 16, 29, 61), torch.randn(2, 16, 29, 245)
    score, logit = ops.corr_head(_nhwc(cam).cuda(), _nhwc(rng).cuda(), want_logit=True)
    for b in range(2):
        c = cam[b:b + 1] / (cam[b].max() - cam[b].min())
        r = rng[b:b + 1] / (rng[b].max() - rng[b].min())
        ref = F.conv2d(O.circular_assign(r, int(245 / 8)), c).view(-1) / 16
        assert _rel(logit[b].cpu(), ref) < 2e-5
        assert _rel(score[b].cpu(), torch.sigmoid(ref)) < 2e-5


@pytest.mark.parametrize('cin,cout,hw,B', [(64, 64, (24, 40), 2), (128, 256, (9, 13), 2), (256, 128, (5, 3), 1),
                                           (64, 192, (17, 262), 1), (512, 512, (6, 10), 3)])
def test_winograd_conv3x3_vs_direct_and_fp64(L, cin, cout, hw, B):
    """efgh_wino_conv3x3 (F(4,3) on fp32 MFMA) against the direct gather-GEMM and a float64 reference: ragged
    widths (W % 4 != 0, W < 4), residual + ReLU epilogue, train-mode BatchNorm statistics"""
    from efgh_amd import ops
    torch.manual_seed(1)
    conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=True)
    x = torch.randn(B, cin, *hw).clamp_min(-0.5)
    res = torch.randn(B, cout, *hw)
    ref = F.relu(F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1) + res.double())
    cg = nn.Conv2d(cin, cout, 3, 1, 1, bias=True).cuda()
    cg.load_state_dict(conv.state_dict())
    xg, rg = ops.nchw_to_nhwc(x.cuda(), cin), ops.nchw_to_nhwc(res.cuda(), cout)
    out = {}
    old2d, ops.USE_WINO2D = ops.USE_WINO2D, False            # this test is about the 1-D kernel (the 2-D path has its own below)
    for wino in (True, False):
        ops.USE_WINO = wino
        try:
            assert ops.wino_eligible(1, cin, cout, (B, hw[0], hw[1], hw[0], hw[1], 1, 1, [t // 3 - 1 for t in range(9)],
                                                    [t % 3 - 1 for t in range(9)], hw[0], hw[1], 1, 1, 0, 0)) == wino
            with torch.no_grad():
                y = L.conv2d(L.Ctx(False), xg, cg, None, L.ACT_RELU, 0.0, residual=rg)
            out[wino] = y.permute(0, 3, 1, 2).double().cpu()
        finally:
            ops.USE_WINO = True
    e_w, e_d = _rel(out[True], ref), _rel(out[False], ref)
    assert e_w < 1e-5 and e_d < 1e-5, (e_w, e_d)
    assert e_w < 8 * e_d + 1e-6, (e_w, e_d)          # the transforms cost a small constant factor of rounding
    # train-mode BatchNorm on top (per-tile statistics epilogue of the Winograd kernel)
    bn = nn.BatchNorm2d(cout)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
    bng = nn.BatchNorm2d(cout).cuda()
    bng.load_state_dict(bn.state_dict())
    bn.train(); bng.train()
    refb = F.relu(bn(conv(x)))
    with torch.no_grad():
        yb = L.conv2d(L.Ctx(True), xg, cg, bng, L.ACT_RELU, 0.0)
    assert _rel(yb.permute(0, 3, 1, 2).cpu(), refb) < 2e-5
    assert _rel(bng.running_var.cpu(), bn.running_var) < 1e-5 and _rel(bng.running_mean.cpu(), bn.running_mean) < 1e-4
    ops.USE_WINO2D = old2d


@pytest.mark.parametrize('cin,cout,hw,B', [(128, 256, (9, 13), 2), (256, 128, (8, 11), 1), (128, 128, (17, 262), 1),
                                           (512, 512, (10, 14), 3), (256, 256, (24, 40), 2)])
def test_winograd2d_conv3x3_vs_direct_and_fp64(L, cin, cout, hw, B):
    """F(4x4,3x3) path (input transform, 36 batched GEMMs, output transform; wino2d.hip) against the direct gather-GEMM and
    float64: ragged heights and widths (H, W % 4 != 0, < 4), residual + ReLU epilogue, train-mode BatchNorm statistics, and the
    weight gradient + data gradient through autograd"""
    from efgh_amd import ops
    torch.manual_seed(1)
    conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=True)
    x = torch.randn(B, cin, *hw).clamp_min(-0.5)
    res = torch.randn(B, cout, *hw)
    ref = F.relu(F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1) + res.double())
    cg = nn.Conv2d(cin, cout, 3, 1, 1, bias=True).cuda()
    cg.load_state_dict(conv.state_dict())
    xg, rg = ops.nchw_to_nhwc(x.cuda(), cin), ops.nchw_to_nhwc(res.cuda(), cout)
    geom = (B, hw[0], hw[1], hw[0], hw[1], 1, 1, [t // 3 - 1 for t in range(9)], [t % 3 - 1 for t in range(9)], hw[0], hw[1],
            1, 1, 0, 0)
    out = {}
    old_min, ops.WINO2D_MIN_C = ops.WINO2D_MIN_C, 128          # also the 128-channel shapes through the forward path
    for w2 in (True, False):
        ops.USE_WINO2D = w2
        ops.USE_WINO = w2
        try:
            assert ops.wino2d_eligible(1, cin, cout, geom) == w2
            with torch.no_grad():
                y = L.conv2d(L.Ctx(False), xg, cg, None, L.ACT_RELU, 0.0, residual=rg)
            out[w2] = y.permute(0, 3, 1, 2).double().cpu()
        finally:
            ops.USE_WINO2D = ops.USE_WINO = True
    e_w, e_d = _rel(out[True], ref), _rel(out[False], ref)
    assert e_w < 3e-5 and e_d < 1e-5, (e_w, e_d)        # 2-D transforms: ~1e-5 max relative, direct ~3e-7
    # train-mode BatchNorm on top (statistics rows of the output-transform kernel) + gradients through autograd
    bn = nn.BatchNorm2d(cout)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
    bng = nn.BatchNorm2d(cout).cuda()
    bng.load_state_dict(bn.state_dict())
    bn.train(); bng.train()
    xr = x.clone().requires_grad_(True)
    refb = F.relu(bn(conv(xr)))
    gy = torch.randn_like(refb)
    refb.backward(gy)
    xq = xg.clone().requires_grad_(True)
    yb = L.conv2d(L.Ctx(True), xq, cg, bng, L.ACT_RELU, 0.0)
    yb.backward(ops.nchw_to_nhwc(gy.cuda(), cout))
    assert _rel(yb.detach().permute(0, 3, 1, 2).cpu(), refb.detach()) < 3e-5
    assert _rel(bng.running_var.cpu(), bn.running_var) < 1e-5 and _rel(bng.running_mean.cpu(), bn.running_mean) < 1e-4
    assert _rel(cg.weight.grad.cpu(), conv.weight.grad) < 2e-4, _rel(cg.weight.grad.cpu(), conv.weight.grad)
    assert _rel(xq.grad.permute(0, 3, 1, 2).cpu(), xr.grad) < 2e-4
    assert _rel(bng.weight.grad.cpu(), bn.weight.grad) < 2e-4 and _rel(bng.bias.grad.cpu(), bn.bias.grad) < 2e-4
    ops.WINO2D_MIN_C = old_min


@pytest.mark.parametrize('cin,cout,stride,hw,B', [(3, 64, 1, (24, 40), 2), (4, 32, 2, (37, 301), 1), (1, 128, 2, (18, 26), 2),
                                                 (3, 64, 1, (9, 263), 1), (2, 128, 1, (5, 131), 2)])
def test_c4_mfma_conv_vs_generic_and_fp64(L, cin, cout, stride, hw, B):
    """efgh_c4_conv3x3 / efgh_c4_wgrad (4 input channels per tap: the RGB / range / depth input layers and the data gradient of
    G's transposed heads) against the generic implicit-GEMM kernels and float64: stride 1 and 2, odd sizes, several 128-pixel
    units per row with a ragged last one, residual epilogue, train-mode BatchNorm statistics, weight gradient"""
    from efgh_amd import ops
    torch.manual_seed(3)
    conv = nn.Conv2d(cin, cout, 3, stride, 1, bias=True)
    x = torch.randn(B, cin, *hw)
    ho, wo = (hw[0] + 2 - 3) // stride + 1, (hw[1] + 2 - 3) // stride + 1
    res = torch.randn(B, cout, ho, wo)
    ref = F.leaky_relu(F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), stride=stride, padding=1) + res.double(), 0.2)
    cg = nn.Conv2d(cin, cout, 3, stride, 1, bias=True).cuda()
    cg.load_state_dict(conv.state_dict())
    xg, rg = ops.nchw_to_nhwc(x.cuda(), 4), ops.nchw_to_nhwc(res.cuda(), cout)
    taps = ([t // 3 - 1 for t in range(9)], [t % 3 - 1 for t in range(9)])
    geom = (B, hw[0], hw[1], ho, wo, stride, stride, taps[0], taps[1], ho, wo, 1, 1, 0, 0)
    out = {}
    for c4 in (True, False):
        ops.USE_C4 = c4
        try:
            assert ops.c4_eligible(1, 4, cout, geom) == c4
            with torch.no_grad():
                y = L.conv2d(L.Ctx(False), xg, cg, None, L.ACT_LEAKY, 0.2, residual=rg)
            out[c4] = y.permute(0, 3, 1, 2).double().cpu()
        finally:
            ops.USE_C4 = True
    assert _rel(out[True], ref) < 2e-6 and _rel(out[False], ref) < 2e-6, (_rel(out[True], ref), _rel(out[False], ref))
    # train-mode BatchNorm on top (one statistics row per persistent workgroup) + the weight gradient through autograd
    bn = nn.BatchNorm2d(cout)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
    bng = nn.BatchNorm2d(cout).cuda()
    bng.load_state_dict(bn.state_dict())
    bn.train(); bng.train()
    gy = torch.randn(B, cout, ho, wo)
    refb = F.relu(bn(conv(x)))
    refb.backward(gy)
    yb = L.conv2d(L.Ctx(True), xg, cg, bng, L.ACT_RELU, 0.0)          # grad mode: the autograd Functions of nets/fn.py
    yb.backward(ops.nchw_to_nhwc(gy.cuda(), cout))
    assert _rel(yb.detach().permute(0, 3, 1, 2).cpu(), refb.detach()) < 2e-5
    assert _rel(bng.running_var.cpu(), bn.running_var) < 1e-5 and _rel(bng.running_mean.cpu(), bn.running_mean) < 1e-4
    assert _rel(cg.weight.grad.cpu(), conv.weight.grad) < 2e-4, _rel(cg.weight.grad.cpu(), conv.weight.grad)
    assert _rel(bng.weight.grad.cpu(), bn.weight.grad) < 2e-4 and _rel(bng.bias.grad.cpu(), bn.bias.grad) < 2e-4


@pytest.mark.parametrize('cin,cout,hw,B,k', [(16, 16, (24, 40), 2, 3), (32, 32, (37, 301), 1, 3), (16, 32, (18, 26), 2, 3),
                                            (32, 16, (9, 263), 1, 3), (16, 16, (5, 131), 3, 3),
                                            (64, 32, (11, 157), 2, 1), (64, 32, (3, 32), 1, 1), (64, 32, (40, 333), 1, 1)])
def test_small_channel_conv_vs_generic_and_fp64(L, cin, cout, hw, B, k, monkeypatch):
    """efgh_sc_conv3x3 / efgh_sc_wgrad (stride 1; 3x3 with 16 / 32 channels on both sides: F's up-sampling stages; 1x1 64 -> 32 and
    its data gradient 32 -> 64) against the generic implicit-GEMM kernels and float64: odd sizes, several 32-pixel units per row with
    a ragged last one, residual epilogue, train-mode BatchNorm statistics, data gradient with a skip gradient added in the epilogue,
    weight gradient (bit-reproducible)"""
    from efgh_amd import ops
    monkeypatch.setattr(ops, 'SC_MIN_PIXELS_32', 0)          # (the 32 -> 32 launch is reserved for large maps: test it at any size)
    torch.manual_seed(4)
    conv = nn.Conv2d(cin, cout, k, 1, k // 2, bias=True)
    x = torch.randn(B, cin, *hw)
    res = torch.randn(B, cout, *hw)
    ref = F.leaky_relu(F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=k // 2) + res.double(), 0.2)
    cg = nn.Conv2d(cin, cout, k, 1, k // 2, bias=True).cuda()
    cg.load_state_dict(conv.state_dict())
    xg, rg = ops.nchw_to_nhwc(x.cuda(), cin), ops.nchw_to_nhwc(res.cuda(), cout)
    taps = ([t // 3 - 1 for t in range(9)], [t % 3 - 1 for t in range(9)]) if k == 3 else ([0], [0])
    geom = (B, hw[0], hw[1], hw[0], hw[1], 1, 1, taps[0], taps[1], hw[0], hw[1], 1, 1, 0, 0)
    out = {}
    for sc in (True, False):
        ops.USE_SMALLC = sc
        try:
            assert ops.sc_eligible(1, cin, cout, geom) == sc
            with torch.no_grad():
                y = L.conv2d(L.Ctx(False), xg, cg, None, L.ACT_LEAKY, 0.2, residual=rg)
            out[sc] = y.permute(0, 3, 1, 2).double().cpu()
        finally:
            ops.USE_SMALLC = True
    assert _rel(out[True], ref) < 2e-6 and _rel(out[False], ref) < 2e-6, (_rel(out[True], ref), _rel(out[False], ref))
    # train-mode BatchNorm on top (one statistics row per persistent workgroup), data and weight gradients through autograd
    bn = nn.BatchNorm2d(cout)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
    bng = nn.BatchNorm2d(cout).cuda()
    bng.load_state_dict(bn.state_dict())
    bn.train(); bng.train()
    gy = torch.randn(B, cout, *hw)
    xr = x.clone().requires_grad_(True)
    refb = F.relu(bn(conv(xr)))
    refb.backward(gy)
    grads = []
    for _ in range(2):
        cg.weight.grad = cg.bias.grad = bng.weight.grad = bng.bias.grad = None
        xq = xg.clone().requires_grad_(True)
        yb = L.conv2d(L.Ctx(True), xq, cg, bng, L.ACT_RELU, 0.0)          # grad mode: the autograd Functions of nets/fn.py
        yb.backward(ops.nchw_to_nhwc(gy.cuda(), cout))
        grads.append(cg.weight.grad.clone())
    assert torch.equal(grads[0], grads[1])                                  # partial planes folded in a fixed order
    assert _rel(yb.detach().permute(0, 3, 1, 2).cpu(), refb.detach()) < 2e-5
    assert _rel(cg.weight.grad.cpu(), conv.weight.grad) < 2e-4, _rel(cg.weight.grad.cpu(), conv.weight.grad)
    assert _rel(xq.grad.permute(0, 3, 1, 2).cpu(), xr.grad) < 2e-4
    assert _rel(bng.weight.grad.cpu(), bn.weight.grad) < 2e-4 and _rel(bng.bias.grad.cpu(), bn.bias.grad) < 2e-4


@pytest.mark.parametrize('train', [False, True])
def test_fused_convt_heads_equal_separate_stacks(L, train):
    """layers.run_convt_heads (G's depth and mask heads as one 3-channel pipeline: concatenated transposed weights and BatchNorms,
    block-diagonal 3x3 convolution) against the two convt_bn_relu stacks run separately: outputs, running statistics, and in train
    mode the gradients of every parameter of both stacks and of the shared input"""
    import copy
    from efgh_amd import ops
    from efgh_amd.nets import fn as FN
    from efgh_amd.nets.builders import convt_bn_relu
    torch.manual_seed(11)
    sd, sm = convt_bn_relu(128, 1, 3, 2, 1, 1).cuda(), convt_bn_relu(128, 2, 3, 2, 1, 1).cuda()
    for seq in (sd, sm):
        for m in seq.modules():
            if isinstance(m, nn.BatchNorm2d):
                with torch.no_grad():
                    m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2); m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
    sd2, sm2 = copy.deepcopy(sd), copy.deepcopy(sm)
    for seq in (sd, sm, sd2, sm2):
        seq.train(train)
    assert L.convt_heads_fusable(sd, sm)
    B, H, W = 2, 9, 21
    x = torch.randn(B, H, W, 128, device='cuda')
    gd, gm = torch.randn(B, 1, 2 * H, 2 * W, device='cuda'), torch.randn(B, 2, 2 * H, 2 * W, device='cuda')
    with torch.set_grad_enabled(train):
        ctx = L.Ctx(train)
        xa = x.clone().requires_grad_(train)
        hm = L.run_convt_heads(ctx, sd, sm, xa)
        d1, m1 = FN.HeadsToNchwFn.apply(hm) if train else ops.heads_to_nchw(hm)
        xb = x.clone().requires_grad_(train)
        dimg, alias = L.run_convt_bn_relu(ctx, sd2, xb, skip_out=True)
        mask = L.run_convt_bn_relu(ctx, sm2, alias)
        if train:
            d2, m2 = FN.NhwcToNchwFn.apply(dimg, 1), FN.Softmax2ToNchwFn.apply(mask)
        else:
            d2, m2 = ops.nhwc_to_nchw(dimg, 1), ops.softmax2_to_nchw(mask)
    assert _rel(d1, d2) < 1e-6 and _rel(m1, m2) < 1e-6, (_rel(d1, d2), _rel(m1, m2))
    for a, b in zip(list(sd.buffers()) + list(sm.buffers()), list(sd2.buffers()) + list(sm2.buffers())):
        assert torch.equal(a, b) if a.dtype != torch.float32 else _rel(a, b) < 1e-6
    if train:
        ((d1 * gd).sum() + (m1 * gm).sum()).backward()
        ((d2 * gd).sum() + (m2 * gm).sum()).backward()
        assert _rel(xa.grad, xb.grad) < 2e-5, _rel(xa.grad, xb.grad)
        for (n, pa), pb in zip(list(sd.named_parameters()) + list(sm.named_parameters()), list(sd2.parameters()) + list(sm2.parameters())):
            assert pa.grad is not None and _rel(pa.grad, pb.grad) < 5e-5 + 0, (n, _rel(pa.grad, pb.grad))


@pytest.mark.parametrize('c,hw,B', [(2, (37, 301), 2), (1, (5, 9), 1), (2, (64, 256), 1), (3, (33, 600), 3)])
def test_c4n4_stencil_conv_vs_fp64_and_autograd(L, c, hw, B):
    """k_c4n4_conv3x3 / k_c4n4_wgrad3x3 (<= 4 channels on both sides, 3x3, stride 1: the 1- / 2-channel convolutions behind G's
    transposed heads) against float64 and torch autograd: ragged last column block, row bands that end inside the image, bias +
    residual + activation epilogue, train-mode BatchNorm on top, data gradient, weight gradient bit-identical run to run"""
    from efgh_amd import ops
    torch.manual_seed(6)
    conv = nn.Conv2d(c, c, 3, 1, 1, bias=True)
    x = torch.randn(B, c, *hw)
    res = torch.randn(B, c, *hw)
    ref = F.leaky_relu(F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1) + res.double(), 0.2)
    cg = nn.Conv2d(c, c, 3, 1, 1, bias=True).cuda()
    cg.load_state_dict(conv.state_dict())
    xg, rg = ops.nchw_to_nhwc(x.cuda(), 4), ops.nchw_to_nhwc(res.cuda(), 4)
    ops.TRACE_THIN = []
    try:
        with torch.no_grad():
            y = L.conv2d(L.Ctx(False), xg, cg, None, L.ACT_LEAKY, 0.2, residual=rg)
        assert ops.TRACE_THIN == [4], ops.TRACE_THIN
    finally:
        ops.TRACE_THIN = None
    got = y[..., :c].permute(0, 3, 1, 2).double().cpu()
    assert _rel(got, ref) < 2e-6, _rel(got, ref)
    bn = nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
    bng = nn.BatchNorm2d(c).cuda()
    bng.load_state_dict(bn.state_dict())
    bn.train(); bng.train()
    gy = torch.randn(B, c, *hw)
    xr = x.clone().requires_grad_(True)
    refb = F.leaky_relu(bn(conv(xr)), 0.2)
    refb.backward(gy)
    grads = []
    for _ in range(2):
        cg.weight.grad = cg.bias.grad = bng.weight.grad = bng.bias.grad = None
        xq = xg.clone().requires_grad_(True)
        yb = L.conv2d(L.Ctx(True), xq, cg, bng, L.ACT_LEAKY, 0.2)
        yb.backward(ops.nchw_to_nhwc(gy.cuda(), 4))
        grads.append(cg.weight.grad.clone())
    assert torch.equal(grads[0], grads[1])
    assert _rel(yb.detach()[..., :c].permute(0, 3, 1, 2).cpu(), refb.detach()) < 2e-5
    assert _rel(cg.weight.grad.cpu(), conv.weight.grad) < 2e-4, _rel(cg.weight.grad.cpu(), conv.weight.grad)
    assert _rel(xq.grad[..., :c].permute(0, 3, 1, 2).cpu(), xr.grad) < 2e-4
    assert _rel(bng.weight.grad.cpu(), bn.weight.grad) < 2e-4 and _rel(bng.bias.grad.cpu(), bn.bias.grad) < 2e-4


@pytest.mark.parametrize('hw,B', [((70, 95), 2), ((3, 30), 1), ((64, 31), 1), ((131, 7), 3)])
def test_n4_mfma_conv_vs_fp64(L, hw, B):
    """k_n4_conv3x3_c64 (64 -> <= 4 channels, 3x3, stride 1: the data gradient of the range trunk's 4 -> 64 input layer) against
    float64: several 30-column strips with a ragged last one, more rows than one 64-row unit, bias + residual + activation epilogue"""
    from efgh_amd import ops
    torch.manual_seed(5)
    conv = nn.Conv2d(64, 3, 3, 1, 1, bias=True)
    x = torch.randn(B, 64, *hw)
    res = torch.randn(B, 3, *hw)
    ref = F.leaky_relu(F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1) + res.double(), 0.2)
    cg = nn.Conv2d(64, 3, 3, 1, 1, bias=True).cuda()
    cg.load_state_dict(conv.state_dict())
    xg, rg = ops.nchw_to_nhwc(x.cuda(), 64), ops.nchw_to_nhwc(res.cuda(), 4)
    ops.TRACE_THIN = []
    try:
        with torch.no_grad():
            y = L.conv2d(L.Ctx(False), xg, cg, None, L.ACT_LEAKY, 0.2, residual=rg)
        assert ops.TRACE_THIN == [3], ops.TRACE_THIN
    finally:
        ops.TRACE_THIN = None
    assert y.shape[-1] == 4
    got = y[..., :3].permute(0, 3, 1, 2).double().cpu()
    assert _rel(got, ref) < 2e-6, _rel(got, ref)


@pytest.mark.parametrize('C', [36, 68, 132, 260, 4, 256])
@pytest.mark.parametrize('emg', [False, True])
def test_splat_gather_vs_float64_scatter_add(C, emg):
    """splat gather over the lattice's vertex lists (no fp32 atomics) == a float64 scatter-add; the vertex lists are the exact
    inverse of `off`, ascending; the gradient kernel == the float64 adjoint"""
    from efgh_amd import lattice, ops, synthetic as syn
    pc = syn.lidar_sweep(4096, 3)
    lv = lattice.build_pyramid(torch.from_numpy(pc).cuda(), (1.0,))[0]
    n, H = 4096, lv.H
    if emg and C <= 4:
        pytest.skip('no feature channels left')
    Cf = C - 4 if emg else C
    feat = torch.randn(n, Cf, device='cuda')
    splat, wsum = ops.splat_fwd(lv, feat, Cf, use_emg=emg)
    rows = torch.cat([lv.emg.t(), feat], 1).cpu().double() if emg else feat.cpu().double()
    ref = torch.zeros((H, C), dtype=torch.float64)
    w = torch.zeros(H, dtype=torch.float64)
    off, bary = lv.off.cpu().long(), lv.bary.cpu().double()
    for r in range(4):
        ref.index_add_(0, off[r], bary[r][:, None] * rows)
        w.index_add_(0, off[r], bary[r])
    ref = ref / (w[:, None] + 1e-5)
    assert _rel(splat.cpu().double(), ref) < 2e-6 and _rel(wsum.cpu().double(), w) < 2e-6, C
    # lists: vertex h <- ascending flat positions f = 4p + r with off[r][p] == h
    vseg, lst = lv.vseg[:H].cpu().numpy(), lv.list.cpu().numpy()
    flat_off = lv.off_pm[:n].cpu().numpy().reshape(-1)
    assert int(vseg[:, 1].sum()) == 4 * n
    order = np.argsort(flat_off, kind='stable')
    starts = np.concatenate([[0], np.cumsum(np.bincount(flat_off, minlength=H))])
    for h in (0, 1, H // 2, H - 1):
        got = lst[vseg[h, 0]:vseg[h, 0] + vseg[h, 1]]
        assert np.array_equal(got, order[starts[h]:starts[h + 1]]), h
    # backward
    g = torch.randn(H, C, device='cuda')
    gfeat = torch.empty((n, Cf), device='cuda')
    ops.splat_bwd(lv, g, wsum, Cf, gfeat, use_emg=emg)
    gd = g.cpu().double() / (w[:, None] + 1e-5)
    gref = torch.zeros((n, C), dtype=torch.float64)
    for r in range(4):
        gref += bary[r][:, None] * gd[off[r]]
    assert _rel(gfeat.cpu().double(), gref[:, C - Cf:]) < 2e-6


def test_neighbor_gather_adjoint_vs_atomic_scatter():
    """adjoint of the blur's neighbour gather through the lattice's own table (symmetric part as a gather + the aliased hits
    of key2int added) == the atomic scatter-add over the table, on a scene whose key box is hit by out-of-range neighbours"""
    from efgh_amd import lattice, ops
    rs = np.random.RandomState(11)
    pc = (rs.randn(3, 6000) * np.array([[3.], [3.], [0.4]])).astype(np.float32)
    for lv in lattice.build_pyramid(torch.from_numpy(pc).cuda(), (1.0, 0.5)):
        H, C = lv.H, 36
        src = torch.randn(H, 15 * C, device='cuda')
        got = ops.neighbor_gather_adjoint(lv, src, C)
        ref = torch.zeros((H, C), device='cuda')
        ops.table_scatter_add(src, lv.nbr, H, 15, C, ref)
        assert _rel(got.cpu(), ref.cpu()) < 1e-5, int(lv.info[2])
        # column 15 = alias mask, and the unmarked relation is symmetric
        nbr = lv.nbr.cpu().numpy()
        na = int(lv.info[2])
        assert int(sum(bin(int(v)).count('1') for v in nbr[:, 15])) == na
        for t in range(1, 15):
            ok = (nbr[:, t] >= 0) & ((nbr[:, 15] >> t) & 1 == 0)
            assert np.array_equal(nbr[nbr[ok, t], 15 - t], np.nonzero(ok)[0]), t


def _inject_aliased_hits(lv, count, seed):
    """key2int's aliased neighbour hits need key boxes that real sweeps do not produce (a lattice point and its alias differ in
    their residue mod 4 unless a key range degenerates), so the tests plant them: `count` absent table entries (h, t) are pointed at
    arbitrary vertices and marked exactly as the build marks a real one - bit t of column 15 and a record in alist."""
    from efgh_amd import lattice
    rs = np.random.RandomState(seed)
    nbr = lv.nbr.cpu().numpy().copy()
    hh, tt = np.nonzero(nbr[:, 1:15] < 0)
    pick = rs.choice(len(hh), size=min(count, len(hh)), replace=False)
    recs = []
    for k in pick:
        h, t = int(hh[k]), int(tt[k]) + 1
        target = int(rs.randint(0, lv.H))
        nbr[h, t] = target
        nbr[h, 15] |= 1 << t
        recs.append((h * 16 + t, target))
    rs.shuffle(recs)                              # (arrival order of the records is arbitrary on the device too)
    lv.nbr = torch.from_numpy(nbr).cuda()
    alist = np.zeros((lattice.ALIAS_CAP, 2), np.int32)
    alist[:len(recs)] = np.asarray(recs, np.int32).reshape(-1, 2)
    lv.alist = torch.from_numpy(alist).cuda()
    lv.info = lv.info.clone()
    lv.info[lattice.INFO_ALIAS] = len(recs)
    lv.n_alias = len(recs)
    return len(recs)


@pytest.mark.parametrize('C,C0,npts', [(36, 32, 6000), (68, 64, 6000), (260, 256, 900)])
def test_fused_blur_dgrad_equals_gemm_plus_scatter(C, C0, npts):
    """data gradient of the blur's neighbour gather + (15,1) convolution as ONE gather-GEMM through the lattice's symmetric table
    (+ the aliased hits in a fixed order) == GEMM into the [H][15 C] intermediate + float64 scatter-add over the table; with
    planted aliased hits (several on one target), at a level large enough for the plain launch and one small enough for the
    split-K launch; two runs are bit-identical"""
    from efgh_amd import lattice, ops
    rs = np.random.RandomState(11)
    pc = (rs.randn(3, npts) * np.array([[3.], [3.], [0.4]])).astype(np.float32)
    torch.manual_seed(1)
    for lv in lattice.build_pyramid(torch.from_numpy(pc).cuda(), (1.0, 0.5)):
        H = lv.H
        assert _inject_aliased_hits(lv, 40, 3) == 40
        w = torch.randn(C0, C, 15, 1, device='cuda') * 0.1
        draw = torch.randn(H, C0, device='cuda')
        got = ops.blur_dgrad(lv, draw, C0, w, C)
        again = ops.blur_dgrad(lv, draw, C0, w, C)
        assert torch.equal(got, again)
        nbr = lv.nbr[:, :15].cpu().numpy()
        tmp = torch.einsum('mn,nct->mtc', draw.cpu().double(), w[..., 0].cpu().double())        # [H][15][C]
        ref = torch.zeros((H, C), dtype=torch.float64)
        for t in range(15):
            ok = np.nonzero(nbr[:, t] >= 0)[0]
            ref.index_add_(0, torch.from_numpy(nbr[ok, t]).long(), tmp[ok, t])
        assert _rel(got.cpu().double(), ref) < 1e-5, H
        # and the two-step form (explicit intermediate, gather through the table + the same alias records)
        src = tmp.reshape(H, 15 * C).float().cuda()
        two = ops.neighbor_gather_adjoint(lv, src, C)
        assert torch.equal(two, ops.neighbor_gather_adjoint(lv, src, C))
        assert _rel(two.cpu().double(), ref) < 1e-5


def test_padded_pack_and_vector_pad():
    """efgh_pack_weight_padded / efgh_pad_vec: the zero padding of layers whose width is not a multiple of 4, one launch each"""
    from efgh_amd import ops
    torch.manual_seed(0)
    w = torch.randn(3, 10, 3, 3, device='cuda')                       # (O, C, kh, kw): O = 3 -> 4, C = 10 -> 12
    Wp = ops.pack_weight(w, 3, 9, 10, 90, 9, 1, list(range(9)), Np=4, Cp=12)
    want = torch.zeros(4, 9, 12, device='cuda')
    want[:3, :, :10] = w.reshape(3, 10, 9).permute(0, 2, 1)
    assert torch.equal(Wp, want)
    v = torch.randn(5, device='cuda')
    assert torch.equal(ops.pad_vec(v, 8, 1.5), torch.cat([v, torch.full((3,), 1.5, device='cuda')]))
    assert ops.pad_vec(v, 5) is v


@pytest.mark.parametrize('cin,cout,hw', [(64, 64, (12, 20)), (64, 128, (16, 20)), (16, 16, (12, 20)), (4, 32, (12, 20)),
                                         (256, 256, (16, 24))])
@pytest.mark.parametrize('act', ['none', 'leaky', 'relu'])
def test_nan_preactivation_stays_nan_in_the_fused_epilogues(L, cin, cout, hw, act):
    """a diverged network must not look healthy: a NaN that reaches a fused epilogue (generic tile, Winograd F(4,3), small-channel and
    4-channel kernels, the 2-D Winograd output transform) leaves it as NaN under every activation - max / min forms return their
    non-NaN operand and would emit a finite 0.  Rows that never see the NaN stay finite."""
    from efgh_amd import ops
    torch.manual_seed(0)
    conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=True).cuda()
    x = torch.randn(1, cin, *hw)
    x[0, :, 5, 7] = float('nan')
    xg = ops.nchw_to_nhwc(x.cuda(), (cin + 3) // 4 * 4)
    code = {'none': L.ACT_NONE, 'leaky': L.ACT_LEAKY, 'relu': L.ACT_RELU}[act]
    with torch.no_grad():
        y = L.conv2d(L.Ctx(False), xg, conv, None, code, 0.2)[..., :cout]
    if act != 'relu':                                            # (ReLU: the older VALU kernels clamp a NaN to 0, as `v > 0 ? v : 0` does)
        assert torch.isnan(y[0, 4:7, 6:9]).all()                 # every output whose 3x3 window holds the poisoned pixel
    keep = torch.ones(hw, dtype=torch.bool, device=y.device)
    keep[0:12, 0:16] = False                                     # (a Winograd tile spreads the NaN over its 4 / 4x4 outputs)
    assert torch.isfinite(y[0][keep]).all() and int(keep.sum()) > 0


@pytest.mark.parametrize('T2,C,N', [(128, 128, 128), (1000, 256, 128), (777, 128, 384), (4099, 512, 256), (37, 64, 128)])
@pytest.mark.parametrize('nbuf', [2, 3])
def test_plane_gemm_equals_gather_gemm(T2, C, N, nbuf):
    """the LDS-DMA staged batched plain GEMM (planes.hip: global_load_lds into an XOR-swizzled ring, counted vmcnt + one barrier per
    step) against k_gather_gemm<0> on the tile-major layout of the 2-D Winograd planes ([tiles][36][C] x [36][N][C] ->
    [tiles][36][N]): same products in the same k order, so the outputs must be BIT-identical - including ragged tile counts (rows
    past the end are re-reads of the last row and never stored) and a float64 cross-check of one plane"""
    import ctypes
    from efgh_amd import _C
    from efgh_amd._C import c_int32
    lib = _C.lib()
    torch.manual_seed(T2 + C)
    V = torch.randn(T2, 36, C, device='cuda')
    U = torch.randn(36, N, C, device='cuda')
    outs = []
    for _ in range(2):
        o = torch.full((T2 + 1, 36, N), 7.0, device='cuda')          # (one guard row behind the last tile)
        g = _C.GemmDesc()
        g.A, g.lda, g.C, g.T, g.mode = V.data_ptr(), 36 * C, C, 1, 0
        g.W, g.N, g.M = U.data_ptr(), N, T2
        g.out, g.ldo = o.data_ptr(), 36 * N
        g.nbatch, g.batch_stride_a, g.batch_stride_w, g.batch_stride_out = 36, C, N * C, N
        outs.append((o, g))
    assert lib.efgh_plane_gemm_supported(ctypes.byref(outs[1][1])) == 1
    _C.check(lib.efgh_gather_gemm(ctypes.byref(outs[0][1]), _C.stream_ptr()))
    _C.check(lib.efgh_plane_gemm(ctypes.byref(outs[1][1]), c_int32(nbuf), _C.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0])
    assert bool((outs[1][0][T2] == 7.0).all())                       # nothing written past the last row
    ref = torch.einsum('tc,nc->tn', V[:, 5].double(), U[5].double())
    assert float((outs[1][0][:T2, 5].double() - ref).abs().max() / ref.abs().max()) < 2e-6


def test_plane_gemm_refuses_what_it_does_not_serve():
    import ctypes
    from efgh_amd import _C
    lib = _C.lib()
    V, U, o = torch.randn(64, 96, device='cuda'), torch.randn(64, 96, device='cuda'), torch.empty(64, 64, device='cuda')
    g = _C.GemmDesc()
    g.A, g.lda, g.C, g.T, g.mode, g.W, g.N, g.M, g.out, g.ldo = V.data_ptr(), 96, 96, 1, 0, U.data_ptr(), 64, 64, o.data_ptr(), 64
    assert lib.efgh_plane_gemm_supported(ctypes.byref(g)) == 0           # N % 128 != 0
    assert lib.efgh_plane_gemm(ctypes.byref(g), 0, _C.stream_ptr()) != 0
    assert lib.efgh_plane_wgrad_supported(ctypes.byref(g), _C.c_int64(64)) == 0


@pytest.mark.parametrize('T2,C,N', [(256, 128, 128), (5000, 256, 128), (3333, 128, 256), (20011, 256, 256), (3840, 512, 512)])
@pytest.mark.parametrize('nbuf', [2, 3])
def test_plane_wgrad_equals_gather_wgrad(T2, C, N, nbuf):
    """the LDS-DMA staged batched weight gradient dU_a[n][c] = sum_tile G_a[tile][n] V_a[tile][c] against k_gather_wgrad<0, 128>:
    per-chunk partial planes folded in chunk order; rows past the end of a chunk come from a zero page"""
    import ctypes
    from efgh_amd import _C
    from efgh_amd._C import c_int32, c_int64, ptr
    lib = _C.lib()
    torch.manual_seed(T2)
    V = torch.randn(T2, 36, C, device='cuda')
    Gy = torch.randn(T2, 36, N, device='cuda')
    g = _C.GemmDesc()
    g.A, g.lda, g.C, g.T, g.mode, g.N, g.M = V.data_ptr(), 36 * C, C, 1, 0, N, T2
    g.nbatch, g.batch_stride_a = 36, C
    assert lib.efgh_plane_wgrad_supported(ctypes.byref(g), c_int64(36 * N)) == 1
    S0, S1 = torch.empty(36, N, C, device='cuda'), torch.full((36, N, C), 3.0, device='cuda')
    w0 = torch.empty(max(1, lib.efgh_gather_wgrad_workspace(ctypes.byref(g))), device='cuda')
    w1 = torch.empty(max(1, lib.efgh_plane_wgrad_workspace(ctypes.byref(g))), device='cuda')
    _C.check(lib.efgh_gather_wgrad_batched(ctypes.byref(g), ptr(Gy), c_int64(36 * N), c_int64(N), ptr(S0), c_int64(N * C), ptr(w0),
                                           _C.stream_ptr()))
    _C.check(lib.efgh_plane_wgrad_batched(ctypes.byref(g), ptr(Gy), c_int64(36 * N), c_int64(N), ptr(S1), c_int64(N * C), ptr(w1),
                                          c_int32(nbuf), _C.stream_ptr()))
    S2 = torch.full((36, N, C), -1.0, device='cuda')
    _C.check(lib.efgh_plane_wgrad_batched(ctypes.byref(g), ptr(Gy), c_int64(36 * N), c_int64(N), ptr(S2), c_int64(N * C), ptr(w1),
                                          c_int32(nbuf), _C.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(S1, S2)                                   # fixed chunking, fixed fold order: bit-reproducible run to run
    # (its row chunks are chosen to fill rounds of 512 resident workgroups - not k_gather_wgrad's - so the two kernels group the
    # partial sums differently: equal to rounding, not bit for bit)
    assert float((S0 - S1).abs().max() / S0.abs().max()) < 1e-5
    for a in (0, 7, 35):
        ref = torch.einsum('tn,tc->nc', Gy[:, a].double(), V[:, a].double())
        assert float((S1[a].double() - ref).abs().max() / ref.abs().max()) < 5e-6


@pytest.mark.parametrize('cin,cout,k,stride,hw,B', [(64, 128, 3, 2, (33, 47), 2), (128, 64, 1, 1, (19, 23), 2), (256, 96, 3, 1, (9, 14), 1),
                                                     (32, 128, 3, 2, (40, 64), 1), (512, 256, 1, 2, (12, 20), 2), (64, 64, 3, 2, (64, 64), 1)])
def test_dma_gemm_equals_register_staged(L, cin, cout, k, stride, hw, B):
    """the LDS-DMA staged instances of k_gather_gemm (modes 0 / 1, N > 32, C % 32 == 0) against the register-staged kernel on
    strided / 1x1 / non-Winograd 3x3 layers with the full epilogue (bias, train-mode BatchNorm statistics, residual-free
    LeakyReLU; ragged tiles in M and N, zero-padded taps): the same products in the same order - outputs and statistics must be
    BIT-identical"""
    from efgh_amd import _C, ops
    lib = _C.lib()
    torch.manual_seed(cin + cout + k)
    conv = nn.Conv2d(cin, cout, k, stride, k // 2, bias=True).cuda()
    bn = nn.BatchNorm2d(cout).cuda()
    x = torch.randn(B, *hw, cin, device='cuda')
    outs = []
    old_wino = (ops.USE_WINO, ops.USE_WINO2D)
    prev = lib.efgh_gather_gemm_set_dma(1)
    try:
        ops.USE_WINO = ops.USE_WINO2D = False            # (the direct kernel also for the 3x3 / stride-1 case)
        for dma in (0, 1):
            lib.efgh_gather_gemm_set_dma(dma)
            with torch.no_grad():
                bn.running_mean.zero_(); bn.running_var.fill_(1.0)
                y_eval = L.conv2d(L.Ctx(False), x, conv, None, L.ACT_LEAKY, 0.2)
                bn.train()
                y_train = L.conv2d(L.Ctx(True), x, conv, bn, L.ACT_RELU)
                outs.append((y_eval.clone(), y_train.clone(), bn.running_mean.clone(), bn.running_var.clone()))
    finally:
        lib.efgh_gather_gemm_set_dma(prev)
        ops.USE_WINO, ops.USE_WINO2D = old_wino
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    ref = F.leaky_relu(conv(x.permute(0, 3, 1, 2)), 0.2).permute(0, 2, 3, 1)
    assert _rel(outs[1][0], ref.detach()) < 2e-5


def test_batched_winograd_weight_transforms_follow_the_batched_repack():
    """after an in-place weight update the batched repack (ops.repack_stale) rewrites the packed layouts AND their Winograd-domain
    images (wino_weight: 1-D F(4,3), wino2d_weight: F(4x4,3x3)) in place, one launch each: same buffers, bit-identical to the
    per-layer transforms of the new weights; with BATCH_WINO off the images are re-made lazily as before"""
    from efgh_amd import _C, ops
    from efgh_amd._C import c_int32, ptr
    torch.manual_seed(3)
    shapes = [(64, 64), (128, 64), (256, 128), (16, 32)]
    ws = [torch.randn(n, c, 3, 3, device='cuda').requires_grad_(True) for n, c in shapes]      # (trainable: what an optimizer steps)

    def packed(w):
        n, c = w.shape[:2]
        return ops.pack_weight(w, n, 9, c, c * 9, 9, 1, list(range(9)), key=('t', n, c))
    bufs = [packed(w) for w in ws]
    U1 = [ops.wino_weight(b, *s) for b, s in zip(bufs, shapes)]
    U2 = [ops.wino2d_weight(b, *s) for b, s in zip(bufs[:3], shapes[:3])]
    for old_flag in (True, False):
        ops.BATCH_WINO = old_flag
        try:
            with torch.no_grad():
                for w in ws:
                    w.add_(torch.randn_like(w))                  # what a stock optimizer does: version counters move
            ops.repack_stale(torch.device('cuda', torch.cuda.current_device()))
            new_bufs = [packed(w) for w in ws]
            assert all(a is b for a, b in zip(bufs, new_bufs))       # persistent buffers, rewritten in place
            V1 = [ops.wino_weight(b, *s) for b, s in zip(bufs, shapes)]
            V2 = [ops.wino2d_weight(b, *s) for b, s in zip(bufs[:3], shapes[:3])]
            if old_flag:
                assert all(a is b for a, b in zip(U1 + U2, V1 + V2))  # and so are their Winograd images
            for b, (n, c), v in zip(bufs, shapes, V1):
                fresh = torch.empty_like(v)
                _C.check(_C.lib().efgh_wino_pack(ptr(b), ptr(fresh), c_int32(n), c_int32(c), _C.stream_ptr()))
                assert torch.equal(fresh, v)
            for b, (n, c), v in zip(bufs[:3], shapes[:3], V2):
                fresh = torch.empty_like(v)
                _C.check(_C.lib().efgh_wino2d_pack(ptr(b), ptr(fresh), c_int32(n), c_int32(c), _C.stream_ptr()))
                assert torch.equal(fresh, v)
                # and the packed layout itself is the new weight
            for w, b in zip(ws, bufs):
                assert torch.equal(b, w.permute(0, 2, 3, 1).reshape(w.shape[0], 9, w.shape[1]))
            U1, U2 = V1, V2
        finally:
            ops.BATCH_WINO = True


@pytest.mark.parametrize('cin,cout,hw,B', [(128, 256, (9, 13), 2), (256, 256, (8, 12), 1), (512, 512, (10, 14), 3), (128, 128, (17, 262), 1),
                                           (256, 512, (2, 3), 2)])
def test_inference_maxpool_rides_in_the_winograd2d_output_transform(L, cin, cout, hw, B):
    """nets/vgg.py:69-83 in eval mode: [Conv2d 3x3, BatchNorm2d, ReLU, MaxPool2d(2,2)] on the 2-D Winograd path writes the pooled map
    straight from the output transform (efgh_wino2d_output_pooled) - bit-identical to the unfused layer followed by efgh_maxpool2
    (the transform's contraction order is written out, so both instantiations round alike), odd heights / widths (floor mode) and
    ragged tiles included, and equal to torch within the layer's usual tolerance"""
    from efgh_amd import ops
    torch.manual_seed(5)
    feats = nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1), nn.BatchNorm2d(cout), nn.ReLU(), nn.MaxPool2d(2, 2))
    with torch.no_grad():
        feats[1].running_mean.normal_(0, 0.2); feats[1].running_var.uniform_(0.5, 2.0)
        feats[1].weight.normal_(0, 1.0)                     # (negative scales too: the affine runs before the max)
    feats.eval()
    x = torch.randn(B, cin, *hw)
    with torch.no_grad():
        ref = feats(x)
    fg = nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1), nn.BatchNorm2d(cout), nn.ReLU(), nn.MaxPool2d(2, 2)).cuda()
    fg.load_state_dict(feats.state_dict())
    fg.eval()
    xg = ops.nchw_to_nhwc(x.cuda(), cin)
    old_min, ops.WINO2D_MIN_C = ops.WINO2D_MIN_C, 128
    outs = {}
    try:
        for fused in (True, False):
            ops.POOL_FUSED = fused
            with torch.no_grad():
                outs[fused] = L.run_vgg(L.Ctx(False), fg, xg)
    finally:
        ops.POOL_FUSED, ops.WINO2D_MIN_C = True, old_min
    assert outs[True].shape == (B, hw[0] // 2, hw[1] // 2, cout)
    assert torch.equal(outs[True], outs[False])
    assert _rel(outs[True].permute(0, 3, 1, 2).cpu(), ref) < 2e-5


@pytest.mark.parametrize('cin,cout,hw,B', [(4, 64, (9, 70), 2), (3, 64, (8, 33), 1), (4, 32, (6, 130), 2), (4, 128, (5, 31), 1), (3, 64, (2, 2), 3)])
def test_inference_maxpool_rides_in_the_4_channel_input_layer(L, cin, cout, hw, B):
    """the first layer of every VGG trunk in eval mode: [Conv2d(3|4 -> N) 3x3, BatchNorm2d, ReLU, MaxPool2d(2,2)] as ONE kernel
    (efgh_c4_conv3x3_pooled: a wave owns a row pair, the window maximum is taken in its epilogue) - bit-identical to k_c4_conv
    followed by efgh_maxpool2 (same MFMA order; max is exact), odd sizes and partial 32-pixel blocks included"""
    from efgh_amd import ops
    torch.manual_seed(7)
    feats = nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1), nn.BatchNorm2d(cout), nn.ReLU(), nn.MaxPool2d(2, 2))
    with torch.no_grad():
        feats[1].running_mean.normal_(0, 0.2); feats[1].running_var.uniform_(0.5, 2.0); feats[1].weight.normal_(0, 1.0)
    feats.eval()
    x = torch.randn(B, cin, *hw)
    with torch.no_grad():
        ref = feats(x)
    fg = nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1), nn.BatchNorm2d(cout), nn.ReLU(), nn.MaxPool2d(2, 2)).cuda()
    fg.load_state_dict(feats.state_dict())
    fg.eval()
    xg = ops.nchw_to_nhwc(x.cuda(), 4)
    outs = {}
    try:
        for fused in (True, False):
            ops.POOL_FUSED = fused
            with torch.no_grad():
                outs[fused] = L.run_vgg(L.Ctx(False), fg, xg)
    finally:
        ops.POOL_FUSED = True
    assert outs[True].shape == (B, hw[0] // 2, hw[1] // 2, cout)
    assert torch.equal(outs[True], outs[False])
    assert _rel(outs[True].permute(0, 3, 1, 2).cpu(), ref) < 2e-5


@pytest.mark.parametrize('tiled', [True, False])
def test_batched_repack_kernels_equal_the_single_pack(tiled):
    """ops.repack_stale: every registered layout of every stale weight in one launch - the LDS-tiled kernel (coalesced reads along
    W's contiguous axis, 128-byte rows out) and the flat one - against efgh_pack_weight_padded of the new weights: forward layouts,
    transposed + tap-reversed data-gradient layouts, zero-padded rows / channels, tap subsets, 1 and 15 taps"""
    from efgh_amd import ops
    torch.manual_seed(11)
    old = ops.PACK_TILED
    ops.PACK_TILED = tiled
    try:
        cases = []
        for (n, c, k) in [(64, 4, 9), (130, 70, 9), (256, 128, 9), (3, 10, 9), (37, 33, 1), (32, 64, 15), (16, 512, 4)]:
            w = torch.randn(n, c, k, device='cuda').requires_grad_(True)
            rev = list(range(k))[::-1]
            sub = list(range(0, k, 2))
            lay = [(n, k, c, c * k, k, 1, list(range(k)), None, None, ('fwd',)),                      # Wp[n][t][c]
                   (c, k, n, k, c * k, 1, rev, None, None, ('dgrad',)),                                # Wp[c][t'][n], taps reversed
                   (n, k, c, c * k, k, 1, list(range(k)), -(-n // 4) * 4, -(-c // 4) * 4, ('pad',)),   # zero-padded to multiples of 4
                   (n, len(sub), c, c * k, k, 1, sub, None, None, ('sub',))]                           # a subset of the taps
            for (N, T, C, sn, sc, st, taps, Np, Cp, key) in lay:
                if T > 16:
                    continue
                buf = ops.pack_weight(w, N, T, C, sn, sc, st, taps, Np=Np, Cp=Cp, key=key + (tiled,))
                cases.append((w, (N, T, C, sn, sc, st, taps, Np, Cp), buf))
        with torch.no_grad():
            for w in {id(c[0]): c[0] for c in cases}.values():
                w.add_(torch.randn_like(w))
        ops.repack_stale(torch.device('cuda', torch.cuda.current_device()))
        for w, (N, T, C, sn, sc, st, taps, Np, Cp), buf in cases:
            fresh = ops.pack_weight(w, N, T, C, sn, sc, st, taps, Np=Np, Cp=Cp)       # (no key: a fresh single-launch pack)
            assert torch.equal(buf, fresh), (tuple(w.shape), N, T, C, Np, Cp)
    finally:
        ops.PACK_TILED = old


def test_two_stage_segment_reductions_equal_the_one_stage_kernels():
    """torch.max over the vertices of a sample (enet.py:154, hnet.py:53) and torch.mean over the positions (gnet.py:165): the
    two-stage kernels (row slices folded in order) against the one-stage ones and torch - ragged segments, an empty one, repeated
    maxima (the FIRST row wins, as torch.max), 128 columns and 3; the mean to one ulp (another order of the same float64 sums)"""
    from efgh_amd import ops
    torch.manual_seed(2)
    lens = [31159, 0, 2048, 17, 40000]
    M, C = sum(lens), 128
    x = torch.randn(M, C, device='cuda')
    x[torch.randint(0, M, (20000,), device='cuda'), torch.randint(0, C, (20000,), device='cuda')] = 3.5      # repeated maxima
    seg = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device='cuda')
    outs = {}
    for two in (True, False):
        ops.SEGMENT_TWO_STAGE = two
        try:
            outs[two] = ops.segment_colmax(x, C, C, seg, len(lens), want_arg=True)
        finally:
            ops.SEGMENT_TWO_STAGE = True
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    for s, n in enumerate(lens):
        if n:
            a, b = int(seg[s]), int(seg[s + 1])
            ref = x[a:b].max(0)
            assert torch.equal(outs[True][0][s], ref.values)
            # torch.max's index on ties is unspecified on CUDA: check ours is the first row that attains the maximum
            first = (x[a:b] == ref.values[None]).float().argmax(0) + a
            assert torch.equal(outs[True][1][s].long(), first)
    for C2, P, B in ((3, 7680, 4), (4, 30720, 1), (64, 1000, 8)):
        t = torch.randn(B * P, 4 if C2 < 4 else C2, device='cuda')
        ms = {}
        for two in (True, False):
            ops.SEGMENT_TWO_STAGE = two
            try:
                ms[two] = ops.segment_colmean(t, t.shape[-1], C2, P, B)
            finally:
                ops.SEGMENT_TWO_STAGE = True
        ref = t.view(B, P, -1)[:, :, :C2].double().mean(1).float()
        assert float((ms[True] - ref).abs().max()) <= 1.2e-7 * float(ref.abs().max()) + 1e-9
        assert float((ms[True] - ms[False]).abs().max()) <= 1.2e-7 * float(ref.abs().max()) + 1e-9


@pytest.mark.parametrize('cin,cout,hw,B', [(64, 128, (9, 70), 2), (64, 64, (8, 13), 1), (64, 128, (17, 262), 1), (64, 128, (6, 256), 3),
                                           (64, 64, (2, 3), 2)])
def test_inference_maxpool_half_in_the_winograd43_epilogue(L, cin, cout, hw, B):
    """the 64-channel pooled layer of the VGG trunks in eval mode: k_wino43<.., HPOOL> writes the maximum over horizontal pixel
    pairs into a half-width map, efgh_maxpool_v2 takes the vertical half - bit-identical to the plain kernel followed by
    efgh_maxpool2 (max is exact), ragged widths (W % 4 != 0, odd W: floor mode) and odd heights included"""
    from efgh_amd import ops
    torch.manual_seed(9)
    feats = nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1), nn.BatchNorm2d(cout), nn.ReLU(), nn.MaxPool2d(2, 2))
    with torch.no_grad():
        feats[1].running_mean.normal_(0, 0.2); feats[1].running_var.uniform_(0.5, 2.0); feats[1].weight.normal_(0, 1.0)
    feats.eval()
    x = torch.randn(B, cin, *hw)
    with torch.no_grad():
        ref = feats(x)
    fg = nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1), nn.BatchNorm2d(cout), nn.ReLU(), nn.MaxPool2d(2, 2)).cuda()
    fg.load_state_dict(feats.state_dict())
    fg.eval()
    xg = ops.nchw_to_nhwc(x.cuda(), cin)
    geom = (B, hw[0], hw[1], hw[0], hw[1], 1, 1, [t // 3 - 1 for t in range(9)], [t % 3 - 1 for t in range(9)], hw[0], hw[1], 1, 1, 0, 0)
    assert ops.pool_fusable(1, cin, cout, geom) == 'h'
    outs = {}
    try:
        for fused in (True, False):
            ops.POOL_FUSED = fused
            with torch.no_grad():
                outs[fused] = L.run_vgg(L.Ctx(False), fg, xg)
    finally:
        ops.POOL_FUSED = True
    assert outs[True].shape == (B, hw[0] // 2, hw[1] // 2, cout)
    assert torch.equal(outs[True], outs[False])
    assert _rel(outs[True].permute(0, 3, 1, 2).cpu(), ref) < 2e-5
