"""Round 6: BatchNorm apply inside the 2-D Winograd input transform (ops.LazyAct: the producer's activation is never stored) and the
BatchNorm backward's apply pass inside the gradient-side transforms (efgh_wino2d_bwd_transforms: draw is never stored) - held to the
materialised two-pass forms of the same layers and to torch autograd (nets/resnet.py:55-71, nets/vgg.py:69-83, nets/net_utils.py:66-98)."""
import copy

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _relerr(a, b):
    return float((a - b).norm() / (b.norm() + 1e-20))


def _mk_bn(c):
    bn = nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
    return bn


class _Block(nn.Module):          # nets/resnet.py:38-71 BasicBlock (parameter container + torch forward as the reference)
    def __init__(self, inp, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(inp, planes, 3, stride, 1, bias=False)
        self.bn1 = _mk_bn(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = _mk_bn(planes)
        self.downsample = None
        if stride != 1 or inp != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inp, planes, 1, stride, bias=False), _mk_bn(planes))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        return F.relu(self.bn2(self.conv2(y)) + idt)


def _run(model_fn, params_of, x_nchw, gy_nchw, lazy, fused, sums2d=False):
    from efgh_amd import ops
    old = (ops.LAZY_ACT, ops.W2_BWD_FUSED, ops.BN_BWD_FUSED_2D)
    ops.LAZY_ACT, ops.W2_BWD_FUSED, ops.BN_BWD_FUSED_2D = lazy, fused, sums2d
    try:
        h0 = list(ops.LAZY_HITS)
        ops.TLS.train_step = True
        xg = _nhwc(x_nchw).cuda().requires_grad_(True)
        y = model_fn(xg)
        y.backward(_nhwc(gy_nchw).cuda())
        torch.cuda.synchronize()
        return (y.detach().clone(), xg.grad.clone(), [p.grad.clone() for p in params_of()],
                (ops.LAZY_HITS[0] - h0[0], ops.LAZY_HITS[1] - h0[1]))
    finally:
        ops.TLS.train_step = False
        ops.LAZY_ACT, ops.W2_BWD_FUSED, ops.BN_BWD_FUSED_2D = old
        for p in params_of():
            p.grad = None


@pytest.mark.parametrize('inp,planes,stride,hw', [(128, 128, 1, (24, 40)), (64, 128, 2, (36, 52)), (256, 256, 1, (9, 13)),
                                                  (128, 256, 2, (20, 28))])
def test_basic_block_lazy_and_fused_backward(inp, planes, stride, hw):
    from efgh_amd.nets import layers as L
    torch.manual_seed(3)
    blk = _Block(inp, planes, stride)
    x = torch.randn(2, inp, *hw)
    xr = x.clone().requires_grad_(True)
    yr = blk(xr)
    gy = torch.randn_like(yr)
    yr.backward(gy)
    ref = [p.grad.clone() for p in blk.parameters()]
    g = copy.deepcopy(blk).cuda()
    for p in g.parameters():
        p.grad = None
    fn = lambda xg: L.run_basic_block(L.Ctx(True), g, xg)
    prm = lambda: list(g.parameters())
    base = _run(fn, prm, x, gy, False, False)
    assert base[3] == (0, 0)
    lazy = _run(fn, prm, x, gy, True, False)
    both = _run(fn, prm, x, gy, True, True)
    assert lazy[3][0] >= 1 and lazy[3][1] == 0            # conv2 applied bn1 + ReLU itself
    # conv2's backward transforms (residual layer: mask from the sign bits) and, when conv1 is a stride-1 layer of >= 128 channels
    # itself, conv1's (mask re-derived from raw)
    assert both[3][1] >= (2 if (stride == 1 and inp >= 128) else 1)
    # forward: the same fma + ReLU on the way into the transform -> the same bits
    assert torch.equal(base[0], lazy[0]) and torch.equal(base[0], both[0])
    # the deferred activation changes nothing in backward either (the mask was always re-derived from raw)
    assert torch.equal(base[1], lazy[1])
    for a, b in zip(base[2], lazy[2]):
        assert torch.equal(a, b)
    # fused backward transforms: the same float64 expression, contraction left to the compiler in the old pass -> rounding level
    assert _relerr(both[1], base[1]) < 2e-6
    for a, b in zip(both[2], base[2]):
        assert _relerr(a, b) < 2e-6
    # and all of it against torch autograd (two ReLU kinks inside: an element within rounding of one takes the other branch on one side,
    # which is a 1e-3-level change of the gradients - the tight comparison is the one against the two-pass form above)
    assert _relerr(both[0].permute(0, 3, 1, 2).cpu(), yr.detach()) < 1e-5
    assert _relerr(both[1].permute(0, 3, 1, 2).cpu(), xr.grad) < 5e-3
    for (n, _), a, b in zip(blk.named_parameters(), both[2], ref):
        assert _relerr(a.cpu(), b) < 5e-3, n


def test_vgg_pairs_and_convt_pair_lazy():
    """vgg.py:69-83: conv-BN-ReLU chains with and without a pool between them; net_utils.py:66-98: convT-BN-LeakyReLU -> conv-BN-LeakyReLU"""
    from efgh_amd.nets import layers as L
    torch.manual_seed(5)
    feats = nn.Sequential(nn.Conv2d(64, 128, 3, 1, 1), _mk_bn(128), nn.ReLU(), nn.MaxPool2d(2, 2),
                          nn.Conv2d(128, 256, 3, 1, 1), _mk_bn(256), nn.ReLU(),
                          nn.Conv2d(256, 256, 3, 1, 1), _mk_bn(256), nn.ReLU(), nn.MaxPool2d(2, 2),
                          nn.Conv2d(256, 512, 3, 1, 1), _mk_bn(512), nn.ReLU(),
                          nn.Conv2d(512, 512, 3, 1, 1), _mk_bn(512), nn.ReLU())
    up = nn.Sequential(nn.ConvTranspose2d(512, 128, 3, 2, 1, 1, bias=False), _mk_bn(128), nn.LeakyReLU(0.2),
                       nn.Conv2d(128, 128, 3, 1, 1, bias=False), _mk_bn(128), nn.LeakyReLU(0.2))
    x = torch.randn(2, 64, 48, 64)
    xr = x.clone().requires_grad_(True)
    yr = up(feats(xr))
    gy = torch.randn_like(yr)
    yr.backward(gy)
    ref = [p.grad.clone() for p in list(feats.parameters()) + list(up.parameters())]
    fg, ug = copy.deepcopy(feats).cuda(), copy.deepcopy(up).cuda()
    prm = lambda: list(fg.parameters()) + list(ug.parameters())
    for p in prm():
        p.grad = None

    def fn(xg):
        c = L.Ctx(True)
        return L.run_convt_bn_relu(c, ug, L.run_vgg(c, fg, xg))
    base = _run(fn, prm, x, gy, False, False)
    both = _run(fn, prm, x, gy, True, True)
    # + the BatchNorm-backward column sums of a layer taken in the output transform of its consumer's 2-D data gradient (no reduction
    # pass over dy and raw for the un-pooled, residual-free layers in front of a 2-D Winograd layer)
    import efgh_amd.ops as ops_
    calls = []
    orig = ops_.act_bn_bwd_reduce
    ops_.act_bn_bwd_reduce = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        n0 = len(calls)
        allf = _run(fn, prm, x, gy, True, True, sums2d=True)
        n_all = len(calls) - n0
        _run(fn, prm, x, gy, True, True, sums2d=False)
        n_wo = len(calls) - n0 - n_all
    finally:
        ops_.act_bn_bwd_reduce = orig
    assert n_all <= n_wo - 3, (n_all, n_wo)                 # 128->256 (behind 256->256), 256->512 (behind 512->512), the convT (behind its conv)
    assert torch.equal(allf[0], base[0]) and _relerr(allf[1], base[1]) < 3e-5
    for a, b in zip(allf[2], base[2]):
        assert _relerr(a, b) < 3e-5
    assert both[3][0] >= 3 and both[3][1] >= 3, both[3]      # 256->256, 512->512, the convT's conv; fused backward in the >= 128-channel un-pooled layers
    assert torch.equal(base[0], both[0])
    assert _relerr(both[1], base[1]) < 3e-5
    names = ['f.' + n for n, _ in feats.named_parameters()] + ['u.' + n for n, _ in up.named_parameters()]
    conv_bias = {'f.%d.bias' % i for i, m in enumerate(feats) if isinstance(m, nn.Conv2d)}
    for n, a, b, r in zip(names, both[2], base[2], ref):
        if n in conv_bias:
            continue                                          # conv bias in front of a train-mode BatchNorm: exact zero here, rounding noise in torch
        assert _relerr(a, b) < 3e-5, n            # (seven layers deep: rounding-level differences of draw, summed)
        assert _relerr(a.cpu(), r) < 5e-3, n
    assert _relerr(both[0].permute(0, 3, 1, 2).cpu(), yr.detach()) < 1e-5
    assert _relerr(both[1].permute(0, 3, 1, 2).cpu(), xr.grad) < 5e-3


def test_lazy_tensor_reaching_an_unaware_consumer_is_materialised():
    """a deferred activation handed to a layer that cannot apply it (here: a 1x1 convolution) must be normalised first - never read raw"""
    from efgh_amd import ops
    from efgh_amd.nets import fn as FN, layers as L
    torch.manual_seed(7)
    c1, b1 = nn.Conv2d(128, 128, 3, 1, 1, bias=False).cuda(), _mk_bn(128).cuda()
    c2, b2 = nn.Conv2d(128, 64, 1, 1, 0, bias=False).cuda(), _mk_bn(64).cuda()
    x = torch.randn(2, 16, 24, 128, device='cuda', requires_grad=True)
    ops.TLS.train_step = True
    try:
        ctx = L.Ctx(True)
        y1 = L.conv2d(ctx, x, c1, b1, L.ACT_RELU, defer_act=True)         # (the caller lied about the consumer)
        assert getattr(y1, '_efgh_lazy', None) is not None
        out = L.conv2d(ctx, y1, c2, b2, L.ACT_RELU)
        y1m = L.conv2d(L.Ctx(True), x, c1, b1, L.ACT_RELU)
        ref = L.conv2d(L.Ctx(True), y1m, c2, b2, L.ACT_RELU)
        assert torch.equal(out, ref)
    finally:
        ops.TLS.train_step = False


@pytest.mark.parametrize('cin,cout,hw', [(4, 64, (36, 52)), (64, 128, (24, 40)), (256, 256, (16, 24)), (64, 64, (11, 15))])
def test_pooled_bn_backward_sums_from_pooled_tensors(cin, cout, hw):
    """conv + BatchNorm + ReLU + MaxPool2d(2,2) (vgg.py:69-83), training backward: the two column sums taken from the pooled gradient and
    the pooled ACTIVATION alone (efgh_pool_bn_bwd_reduce_pooled: xhat = (y - beta) / gamma at the winning element) against the pass
    that re-reads the full-resolution raw map - same gradients to rounding, also with a gamma that is exactly zero (that channel
    falls back to raw inside the kernel) and on odd map sizes"""
    from efgh_amd import ops
    from efgh_amd.nets import layers as L
    torch.manual_seed(11)
    conv, bn = nn.Conv2d(cin, cout, 3, 1, 1).cuda(), _mk_bn(cout).cuda()
    with torch.no_grad():
        bn.weight[3] = 0.0                       # gamma == 0: y is constant in that channel
        bn.weight[5] = -0.7                      # a negative gamma: the window's maximum is the raw minimum
    x = torch.randn(2, hw[0], hw[1], (cin + 3) // 4 * 4, device='cuda')
    if cin < 4:
        x[..., cin:] = 0
    res = {}
    for flag in (True, False):
        ops.POOL_REDUCE_FROM_Y = flag
        ops.TLS.train_step = True
        try:
            for p in list(conv.parameters()) + list(bn.parameters()):
                p.grad = None
            xg = x.clone().requires_grad_(True)
            y = L.conv2d(L.Ctx(True), xg, conv, bn, L.ACT_RELU, pool=True)
            g = torch.linspace(-1, 1, y.numel(), device='cuda').view_as(y)
            (y * g).sum().backward()
            res[flag] = (y.detach().clone(), xg.grad.clone(), conv.weight.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone())
        finally:
            ops.POOL_REDUCE_FROM_Y = True
            ops.TLS.train_step = False
    assert torch.equal(res[True][0], res[False][0])
    for a, b, n in zip(res[True][1:], res[False][1:], ('dx', 'dW', 'dgamma', 'dbeta')):
        assert _relerr(a, b) < 5e-6, (n, _relerr(a, b))
    assert torch.equal(res[True][4], res[False][4]) or _relerr(res[True][4], res[False][4]) < 1e-6       # (dbeta = sum dpre: the same terms)
    assert abs(float(res[True][3][3]) - float(res[False][3][3])) <= 2e-6 * float(res[False][3].abs().max())   # the gamma == 0 channel
