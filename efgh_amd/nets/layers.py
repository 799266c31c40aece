"""Layer executors: run torch.nn parameter containers (Conv2d, ConvTranspose2d, BatchNorm, Linear,
Conv1d) through the HIP gather-GEMM.  Activations are channels-last [B][H][W][C] fp32 tensors.

The nn.Module objects only hold parameters/buffers (so that state_dict keys and shapes equal the
reference's); their own forward() is never called.
"""
import itertools

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..ops import ACT_LEAKY, ACT_NONE, ACT_RELU, ceil4
from . import fn as FN


class Ctx:
    """Per-forward execution context."""

    def __init__(self, train):
        self.train = bool(train)
        self.grad = torch.is_grad_enabled()      # autograd path (hand-written backward kernels)


def _bn_eval_affine(bn, Np):
    """eval-mode BatchNorm as y = x*scale + shift, cached on the buffers' versions.  Train-mode forwards update
    running_mean / running_var through raw pointers (efgh_bn_finalize), which does not bump their version counters;
    `num_batches_tracked += 1` (in _bn_train, the only writer) does, so it is part of the key."""
    key = ('bn_eval', Np)
    vers = ops._ver(bn.weight, bn.bias, bn.running_mean, bn.running_var) + \
        ((bn.num_batches_tracked._version,) if bn.num_batches_tracked is not None else ())

    def make():
        with torch.no_grad():
            scale = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
            shift = bn.bias.detach() - bn.running_mean * scale
        return ops.pad_vec(scale, Np), ops.pad_vec(shift, Np)

    return ops._cached(bn, key, vers, make)


def _bias_vec(bias, Np):
    if bias is None:
        return None
    if bias.numel() == Np:
        return bias.detach()
    return ops._cached(bias, ('bias', Np), ops._ver(bias), lambda: ops.pad_vec(bias, Np))


def _bn_train(ctx, bn, stats, G, Np, count):
    """finalize batch statistics -> (scale, shift); updates running stats in place."""
    N = bn.num_features
    ops.bn_tick(bn)
    momentum = bn.momentum if bn.momentum is not None else 0.1
    if Np == N:
        scale, shift, _, _ = ops.bn_finalize(stats, G, N, count, bn.weight.detach(), bn.bias.detach(),
                                             bn.running_mean, bn.running_var, momentum, bn.eps)
        return scale, shift
    # padded channel count (tiny layers): run on padded temporaries, copy the real part back
    g, b = ops.pad_vec(bn.weight, Np), ops.pad_vec(bn.bias, Np)
    rm, rv = ops.pad_vec(bn.running_mean, Np), ops.pad_vec(bn.running_var, Np, 1.0)
    if rm is bn.running_mean:
        rm, rv = rm.clone(), rv.clone()
    scale, shift, _, _ = ops.bn_finalize(stats, G, Np, count, g, b, rm, rv, momentum, bn.eps)
    bn.running_mean.copy_(rm[:N])
    bn.running_var.copy_(rv[:N])
    return scale, shift


def _epilogue_plan(ctx, bias, bn, Np):
    """returns (bias, scale, shift, fused) for the GEMM epilogue; fused=False means BN needs batch stats."""
    b = _bias_vec(bias, Np)
    if bn is None:
        return b, None, None, True
    if not ctx.train:
        sc, sh = _bn_eval_affine(bn, Np)
        return b, sc, sh, True
    return b, None, None, False


def _alloc_out(x, shape_rows, N, out):
    """out = None -> fresh [rows..][Np] buffer; or (tensor, coff): write into channels coff.. of an
    existing channels-last buffer (concat fusion)."""
    Np = ceil4(N)
    if out is None:
        t = torch.empty(tuple(shape_rows) + (Np,), dtype=torch.float32, device=x.device)
        return t, Np, 0
    t, coff = out
    assert tuple(t.shape[:-1]) == tuple(shape_rows) and coff + N <= t.shape[-1] and N % 4 == 0
    return t, t.shape[-1], coff


def _run(ctx, x, lda, C, T, Wp, N, M, mode, geoms, out_t, ldo, coff, bias, bn, act, slope, residual=None,
         res_ld=0, res_off=0, table=None, a_off=0, count=None, c_real=None, pool=False):
    """One logical layer = one or more GEMM launches (`geoms`: list of (geom, Wp, M_launch)), then the
    BatchNorm finalize/apply pass when batch statistics are needed."""
    Np = N if N % 4 == 0 else ceil4(N)
    b, sc, sh, fused = _epilogue_plan(ctx, bias, bn, Np)
    cr = c_real if c_real is not None else C

    def fl(geom, m):       # algorithmic FLOPs of one launch: real (unpadded) channel counts
        return 2.0 * m * N * (T if geom is None else len(geom[7])) * cr
    if fused:
        for geom, wp, m in geoms:
            ops.gather_gemm(x, lda, C, T if geom is None else len(geom[7]), wp, Np, m, out_t, ldo, mode=mode,
                            geom=geom, table=table, bias=b, scale=sc, shift=sh, residual=residual, ldr=res_ld,
                            act=act, slope=slope, a_off=a_off, out_off=coff, res_off=res_off, flops=fl(geom, m), pool=pool)
        return
    assert not pool
    # train-mode BatchNorm: raw conv output + per-block column statistics, then normalise in place
    thin = all(ops.thin_eligible(mode, C, Np, T if geom is None else len(geom[7])) for geom, _, _ in geoms)
    if thin:       # VALU kernels carry no statistics epilogue: one extra streaming pass over the (small) output
        for geom, wp, m in geoms:
            ops.gather_gemm(x, lda, C, len(geom[7]), wp, Np, m, out_t, ldo, mode=mode, geom=geom, bias=b,
                            act=ACT_NONE, a_off=a_off, out_off=coff, flops=fl(geom, m))
        stats, _ = ops.col_stats(out_t, M, Np, ldo, x_off=coff)
        gs = [stats.shape[0]]
    else:
        gs = [ops.stats_rows(mode, C, Np, geom, m) for geom, _, m in geoms]
        stats = torch.empty((sum(gs), 2, Np), dtype=torch.float32, device=x.device)
        g0 = 0
        for (geom, wp, m), g in zip(geoms, gs):
            ops.gather_gemm(x, lda, C, T if geom is None else len(geom[7]), wp, Np, m, out_t, ldo, mode=mode,
                            geom=geom, table=table, bias=b, act=ACT_NONE, stats=stats[g0:g0 + g], a_off=a_off,
                            out_off=coff, flops=fl(geom, m))
            g0 += g
    cnt = count if count is not None else M
    scale, shift = _bn_train(ctx, bn, stats, sum(gs), Np, float(cnt))
    ops.scale_shift_act(out_t, ldo, scale, shift, out_t, ldo, M, Np, act, slope, res=residual, ldr=res_ld,
                        x_off=coff, y_off=coff, res_off=res_off)


def _same3x3_geom(B, H, W):
    return (B, H, W, H, W, 1, 1, [t // 3 - 1 for t in range(9)], [t % 3 - 1 for t in range(9)], H, W, 1, 1, 0, 0)


def lazy_consumer_ok(ctx, conv, B, H, W):
    """training step: will `conv` (applied to a [B][H][W][in_channels] map) run on the 2-D Winograd path, whose input transform can
    apply its producer's pending BatchNorm + activation (ops.LazyAct)?  The producer then skips its normalise pass (defer_act)."""
    return bool(ctx.grad and ctx.train and ops.LAZY_ACT and isinstance(conv, nn.Conv2d) and conv.kernel_size == (3, 3)
                and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.in_channels % 4 == 0
                and ops.lazy_capable(1, conv.in_channels, ceil4(conv.out_channels), _same3x3_geom(B, H, W)))


# ----------------------------------------------------------------------------------------------
def conv2d(ctx, x, conv, bn=None, act=ACT_NONE, slope=0.0, residual=None, out=None, in_ch=None, skip_out=False, pool=False,
           defer_act=False):
    """nn.Conv2d (+BatchNorm2d) (+residual) (+activation) on [B][H][W][Cx].
    in_ch: (coff, C) selects a channel slice of x.  Returns the output buffer [B][Ho][Wo][ld]."""
    B, H, W, ldx = x.shape
    Cw, O = conv.in_channels, conv.out_channels
    kh, kw = conv.kernel_size
    sh, sw = conv.stride
    ph, pw = conv.padding
    a_off, Cx = in_ch if in_ch is not None else (0, ldx)
    Cp = ceil4(Cw)
    assert Cx == Cp, (Cx, Cw)
    Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    T = kh * kw
    Np = ceil4(O)
    Wp = ops.pack_weight(conv.weight, O, T, Cw, Cw * T, T, 1, list(range(T)), Np=Np, Cp=Cp, key=('conv', Np, Cp))
    dh = [i // kw - ph for i in range(T)]
    dw = [i % kw - pw for i in range(T)]
    geom = (B, H, W, Ho, Wo, sh, sw, dh, dw, Ho, Wo, 1, 1, 0, 0)
    M = B * Ho * Wo
    if ctx.grad:
        assert in_ch is None or getattr(x, '_efgh_lazy', None) is None       # (a slice view would lose the pending activation)
        xs = x if in_ch is None else x[..., a_off:a_off + Cx]
        return _conv2d_grad(ctx, xs, conv, bn, act, slope, residual, geom, (B, H, W, Ho, Wo), Cp, passthrough=skip_out, pool=pool,
                            out=out, defer_act=defer_act)
    assert not defer_act
    if pool:
        # inference: the following MaxPool2d(2,2) rides in the layer's output transform (run_vgg asks pool_fusable first)
        assert out is None and residual is None and not ctx.train
        out_t, ldo, coff = _alloc_out(x, (B, Ho if pool == 'h' else Ho // 2, Wo // 2), O, None)
        _run(ctx, x, ldx, Cp, T, Wp, O, M, 1, [(geom, Wp, M)], out_t, ldo, coff, conv.bias, bn, act, slope, a_off=a_off, c_real=Cw,
             pool=pool)
        return ops.maxpool_v2(out_t) if pool == 'h' else out_t
    out_t, ldo, coff = _alloc_out(x, (B, Ho, Wo), O, out)
    res_ld = residual.shape[-1] if residual is not None else 0
    _run(ctx, x, ldx, Cp, T, Wp, O, M, 1, [(geom, Wp, M)], out_t, ldo, coff, conv.bias, bn, act, slope,
         residual=residual, res_ld=res_ld, a_off=a_off, c_real=Cw)
    return out_t


def _convt_wcol(convt):
    """[ceil4(9*O)][C] weight of the GEMM+col2im form: row (kh*3+kw)*O+o = W[:, o, kh, kw]"""
    Cw, O = convt.in_channels, convt.out_channels
    w = convt.weight

    def make():
        wd = w.detach().reshape(Cw, O, 9).permute(2, 1, 0).reshape(9 * O, Cw)
        full = torch.zeros((ceil4(9 * O), Cw), dtype=torch.float32, device=wd.device)
        full[:9 * O] = wd
        return full
    return ops._cached(w, ('wcol',), ops._ver(w), make)


def _convt_small_forward(x, convt, out_raw, scale=None, shift=None, act=ACT_NONE, slope=0.0):
    """x [B][H][W][C] -> out_raw [B][Ho][Wo][4]: one GEMM over the input pixels + col2im fold"""
    B, H, W, Cw = x.shape
    O = convt.out_channels
    Ho, Wo = out_raw.shape[1], out_raw.shape[2]
    wcol = _convt_wcol(convt)
    Y = torch.empty((B * H * W, wcol.shape[0]), dtype=torch.float32, device=x.device)
    ops.gather_gemm(x, FN.ld_of(x), Cw, 1, wcol, wcol.shape[0], B * H * W, Y, wcol.shape[0], mode=0,
                    flops=2.0 * B * H * W * 9 * O * Cw)
    ops.convt_col2im(Y, B, H, W, Ho, Wo, O, convt.padding[0], scale, shift, act, slope, out_raw)


def conv_transpose2d(ctx, x, convt, bn=None, act=ACT_NONE, slope=0.0, out=None, skip_out=False, defer_for=None):
    """nn.ConvTranspose2d(k=3, s=2) as four stride-1 sub-convolutions, one per output parity class
    (or, for <= 3 output channels, as ONE GEMM over the input pixels + a col2im fold)."""
    B, H, W, ldx = x.shape
    Cw, O = convt.in_channels, convt.out_channels
    assert convt.kernel_size == (3, 3) and convt.stride == (2, 2) and ldx == Cw and Cw % 4 == 0
    ph, pw = convt.padding
    oph, opw = convt.output_padding
    Ho, Wo = (H - 1) * 2 - 2 * ph + 3 + oph, (W - 1) * 2 - 2 * pw + 3 + opw
    Np = ceil4(O)
    if O <= 3 and ph == pw and convt.bias is None and bn is not None and out is None and not ctx.grad:
        out_t = torch.empty((B, Ho, Wo, 4), dtype=torch.float32, device=x.device)
        if not ctx.train:
            sc, sh = _bn_eval_affine(bn, 4)
            _convt_small_forward(x, convt, out_t, sc, sh, act, slope)
        else:
            _convt_small_forward(x, convt, out_t)
            stats, G = ops.col_stats(out_t, B * Ho * Wo, 4, 4)
            scale, shift = _bn_train(ctx, bn, stats, G, 4, float(B * Ho * Wo))
            ops.scale_shift_act(out_t, 4, scale, shift, out_t, 4, B * Ho * Wo, 4, act, slope)
        return out_t
    out_t, ldo, coff = (None, 0, 0) if ctx.grad else _alloc_out(x, (B, Ho, Wo), O, out)
    geoms = []
    for cy in range(2):
        for cx in range(2):
            khs = [k for k in range(3) if (cy + ph - k) % 2 == 0]
            kws = [k for k in range(3) if (cx + pw - k) % 2 == 0]
            taps = [(a, b) for a in khs for b in kws]
            Hv, Wv = (Ho - cy + 1) // 2, (Wo - cx + 1) // 2
            if Hv <= 0 or Wv <= 0:
                continue
            dh = [(cy + ph - a) // 2 for a, _ in taps]
            dw = [(cx + pw - b) // 2 for _, b in taps]
            tapidx = [a * 3 + b for a, b in taps]
            # ConvTranspose2d weight is (in, out, kh, kw)
            Wp = ops.pack_weight(convt.weight, O, len(taps), Cw, 9, O * 9, 1, tapidx, Np=Np,
                                 key=('convt', cy, cx, ph, pw, Np))
            geom = (B, H, W, Hv, Wv, 1, 1, dh, dw, Ho, Wo, 2, 2, cy, cx)
            geoms.append((geom, Wp, B * Hv * Wv, tapidx, (cy, cx)))
    if ctx.grad:
        assert out is None
        # defer_for: the Conv2d that is the ONLY consumer of this layer's activation (net_utils.py:66-98)
        return _convt_grad(ctx, x, convt, bn, act, slope, geoms, (B, H, W, Ho, Wo), passthrough=skip_out,
                           defer_act=defer_for is not None and lazy_consumer_ok(ctx, defer_for, B, Ho, Wo))
    _run(ctx, x, ldx, Cw, 0, None, O, B * Ho * Wo, 1, [g[:3] for g in geoms], out_t, ldo, coff, convt.bias, bn,
         act, slope)
    return out_t


def linear_rows(ctx, x, M, C, weight, bias, bn=None, act=ACT_NONE, slope=0.0, out=None, lda=None, a_off=0,
                count=None):
    """rows [M][lda] x (O, C[,1]) weight -> [M][ld]: nn.Linear / nn.Conv1d(k=1) (+BatchNorm1d) (+act)."""
    O = weight.shape[0]
    Cp = ceil4(C)
    lda = lda if lda is not None else x.shape[-1]
    Np = ceil4(O)
    if ctx.grad:
        assert out is None
        xs = x if (a_off == 0 and x.shape[-1] == Cp) else x.view(-1, x.shape[-1])[:, a_off:a_off + Cp]
        return _linear_grad(ctx, xs.reshape(M, -1) if xs.dim() != 2 else xs, M, C, weight, bias, bn, act, slope)
    Wp = ops.pack_weight(weight, O, 1, C, C, 1, 1, [0], Np=Np, Cp=Cp, key=('lin', Np, Cp))
    if out is None:
        out_t = torch.empty((M, Np), dtype=torch.float32, device=x.device)
        ldo, coff = Np, 0
    else:
        out_t, coff = out
        ldo = out_t.shape[-1]
    _run(ctx, x, lda, Cp, 1, Wp, O, M, 0, [(None, Wp, M)], out_t, ldo, coff, bias, bn, act, slope, a_off=a_off,
         count=count, c_real=C)
    return out_t


def blur_conv(ctx, splat, H, C, table, conv0, conv1, out=None, last_act=ACT_NONE, last_slope=0.0):
    """BCL blur: gather 15 neighbour rows + Conv2d(C,C0,(15,1)) + ReLU + Conv2d(C0,C1,1)
    (nets/bilateralNN.py:240-246).  splat [H][C]; `table` = the lattice level (efgh_amd.lattice.LatticeLevel: its neighbour
    table also serves the adjoint of the gather) or a bare [H][16] neighbour table -> [H][ld]."""
    C0, C1 = conv0.out_channels, conv1.out_channels
    lv = table if hasattr(table, 'nbr') else None
    if lv is not None:
        table = lv.nbr
    if ctx.grad:
        assert out is None
        mid = _blur_grad(ctx, splat, H, C, table, conv0, lv)
        return linear_rows(ctx, mid, H, C0, conv1.weight, conv1.bias, act=last_act, slope=last_slope)
    Wp0 = ops.pack_weight(conv0.weight, C0, 15, C, C * 15, 15, 1, list(range(15)), key=('blur0',))
    mid = torch.empty((H, C0), dtype=torch.float32, device=splat.device)
    ops.gather_gemm(splat, C, C, 15, Wp0, C0, H, mid, C0, mode=2, table=table, bias=conv0.bias.detach(),
                    act=ACT_RELU)
    return linear_rows(ctx, mid, H, C0, conv1.weight, conv1.bias, out=out, act=last_act, slope=last_slope)   # (last_relu, bilateralNN.py:121-135)


def maxpool2(ctx, x):
    if ctx.grad:
        return FN.MaxPool2Fn.apply(x)
    return ops.maxpool2(x)


# ----------------------------------------------------------------------------------------------
def _pool_fusable_eval(x, conv):
    """inference: can the MaxPool2d(2,2) behind this 3x3 / stride-1 / pad-1 convolution ride in its epilogue (ops.pool_fusable)?"""
    if conv.kernel_size != (3, 3) or conv.stride != (1, 1) or conv.padding != (1, 1):
        return False
    B, H, W, ldx = x.shape
    if ldx != ceil4(conv.in_channels):
        return False
    geom = (B, H, W, H, W, 1, 1, [t // 3 - 1 for t in range(9)], [t % 3 - 1 for t in range(9)], H, W, 1, 1, 0, 0)
    return ops.pool_fusable(1, ldx, ceil4(conv.out_channels), geom)


def run_vgg(ctx, features, x):
    """nets/vgg.py:69-83: [Conv2d, BatchNorm2d, ReLU]* with 'M' = MaxPool2d(2,2)."""
    mods = list(features.children())
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.MaxPool2d):
            x = maxpool2(ctx, x)
            i += 1
        else:
            assert isinstance(m, nn.Conv2d) and isinstance(mods[i + 1], nn.BatchNorm2d)
            # training path: a following MaxPool2d is folded into the layer (BatchNorm + ReLU + pool in one pass over raw);
            # inference: into the output transform of the 2-D Winograd layers (the pooled map is all that is ever written)
            pooled = i + 3 < len(mods) and isinstance(mods[i + 3], nn.MaxPool2d) and m.out_channels % 4 == 0
            fuse = pooled and (ctx.grad or (not ctx.train and _pool_fusable_eval(x, m)))
            # an un-pooled layer's activation goes to the next convolution only: when that one runs on the 2-D Winograd path it
            # normalises + activates inside its input transform and this layer writes no activation at all (training step)
            nxt = mods[i + 3] if i + 3 < len(mods) else None
            defer = (not pooled and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1)
                     and lazy_consumer_ok(ctx, nxt, x.shape[0], x.shape[1], x.shape[2]))
            x = conv2d(ctx, x, m, mods[i + 1], ACT_RELU, pool=fuse, defer_act=defer)
            i += 4 if fuse else 3
    return x


def run_conv_bn_relu(ctx, seq, x, out=None, in_ch=None, skip_out=False):
    """nets/net_utils.py:45-64: Conv2d(no bias) + BN + LeakyReLU(0.2).  skip_out: see run_convt_bn_relu."""
    if skip_out and not ctx.grad:
        return conv2d(ctx, x, seq[0], seq[1], ACT_LEAKY, 0.2, out=out, in_ch=in_ch), x
    return conv2d(ctx, x, seq[0], seq[1], ACT_LEAKY, 0.2, out=out, in_ch=in_ch, skip_out=skip_out)


def run_convt_bn_relu(ctx, seq, x, out=None, skip_out=False):
    """nets/net_utils.py:66-98: ConvT+BN+LeakyReLU(0.2) then Conv3x3+BN+LeakyReLU(0.2).
    skip_out (training path): also returns an alias of x for x's NEXT consumer; the gradient that consumer sends back is added
    in this layer's dgrad epilogue (see _conv2d_grad.dgrad)."""
    if skip_out and ctx.grad:
        y, alias = conv_transpose2d(ctx, x, seq[0], seq[1], ACT_LEAKY, 0.2, skip_out=True, defer_for=seq[3])
        return conv2d(ctx, y, seq[3], seq[4], ACT_LEAKY, 0.2, out=out), alias
    y = conv_transpose2d(ctx, x, seq[0], seq[1], ACT_LEAKY, 0.2, defer_for=seq[3] if ctx.grad else None)
    y = conv2d(ctx, y, seq[3], seq[4], ACT_LEAKY, 0.2, out=out)
    return (y, x) if skip_out else y


# ---- G's depth and mask heads (gnet.py:56-68, 121-124) as ONE three-channel pipeline -------------------------------------------
# convt_dimg (128 -> 1) and convt_mask (128 -> 2) are two convt_bn_relu stacks over the same input.  Every tensor behind their
# transposed convolutions is 16 bytes per pixel at full raw resolution whether it carries one, two or three channels (channels are
# padded to four), and every pass over it is HBM-bound - so the two stacks run as one: transposed weights concatenated along the
# output channel, the per-channel BatchNorms concatenated, the two 3x3 convolutions as one block-diagonal 3 -> 3 convolution (the
# off-diagonal weights are constant zeros: x + 0*y is exact).  Per channel the arithmetic is the one of the separate stacks; the 128-
# channel input is read once instead of twice (forward GEMM, weight gradient) and its gradient is written once instead of written,
# read and re-written.  The combined parameters are built with torch.cat / pad under autograd, so each module's own parameters
# receive their gradients through the usual accumulate hooks; the modules and the state_dict are untouched.
class _Shim:
    pass


class _NbtPair:
    """`bn.num_batches_tracked += 1` on a concatenated BatchNorm: forwarded to the real modules (or their FlatParams tick slots)"""

    def __init__(self, bns):
        self.bns = bns

    @property
    def _version(self):
        return tuple(b.num_batches_tracked._version for b in self.bns if b.num_batches_tracked is not None)

    def __iadd__(self, k):
        for b in self.bns:
            for _ in range(int(k)):
                ops.bn_tick(b)
        return self


class _FusedParamsFn(torch.autograd.Function):
    """the concatenated parameters of the fused heads: forward = the given `build` of the detached parts (torch.cat / pad),
    backward = that build's slices handed back to the parts.  Parts that live in a train.FlatParams get their slice written
    STRAIGHT into the flat gradient buffer on the stream of this backward (FN.claim_grad, as the layer Functions do): no
    CatBackward -> AccumulateGrad detour - those nodes were created on the first step's stream and made torch warn
    "AccumulateGrad node's stream does not match" on every later step of the image branch."""

    @staticmethod
    def forward(ctx, build, split, *parts):
        ctx.split, ctx.parts = split, parts
        return build(*[p.detach() for p in parts])

    @staticmethod
    def backward(ctx, g):
        outs = []
        for p, gs in zip(ctx.parts, ctx.split(g)):
            gv, done = FN.claim_grad(p)
            if gv is not None:
                gv.copy_(gs.reshape(gv.shape))
                done()
                outs.append(None)
            else:
                outs.append(gs.reshape(p.shape).contiguous())
        return (None, None) + tuple(outs)


def _fused(build, split, *parts):
    if any(p.requires_grad for p in parts) and torch.is_grad_enabled():
        FN.note_use(*parts)
        return _FusedParamsFn.apply(build, split, *parts)
    return build(*[p.detach() for p in parts])


def _bn_cat(bns, grad):
    a = bns[0]
    assert all(b.eps == a.eps and b.momentum == a.momentum and b.affine and b.track_running_stats for b in bns)
    bn = _Shim()
    sizes = [b.num_features for b in bns]

    def cat(ts):
        if grad:
            return _fused(lambda *t: torch.cat(t), lambda g: torch.split(g, sizes), *ts)
        return torch.cat([t.detach() for t in ts])
    bn.weight, bn.bias = cat([b.weight for b in bns]), cat([b.bias for b in bns])
    with torch.no_grad():
        bn.running_mean = torch.cat([b.running_mean for b in bns])
        bn.running_var = torch.cat([b.running_var for b in bns])
    bn.eps, bn.momentum, bn.num_features = a.eps, a.momentum, sum(b.num_features for b in bns)
    bn.num_batches_tracked = _NbtPair(bns)
    bn._efgh_nbt = None

    def sync():                              # train mode: the running statistics the kernels updated, back into the modules
        o = 0
        with torch.no_grad():
            for b in bns:
                b.running_mean.copy_(bn.running_mean[o:o + b.num_features])
                b.running_var.copy_(bn.running_var[o:o + b.num_features])
                o += b.num_features
    bn.sync = sync
    return bn


_HEADS_GEN = itertools.count(1)


def convt_heads_fusable(seq_d, seq_m):
    try:
        ct_d, ct_m, cv_d, cv_m = seq_d[0], seq_m[0], seq_d[3], seq_m[3]
        return (isinstance(ct_d, nn.ConvTranspose2d) and isinstance(ct_m, nn.ConvTranspose2d) and ct_d.out_channels + ct_m.out_channels <= 3
                and ct_d.in_channels == ct_m.in_channels and ct_d.bias is None and ct_m.bias is None
                and (ct_d.kernel_size, ct_d.stride, ct_d.padding, ct_d.output_padding) == (ct_m.kernel_size, ct_m.stride, ct_m.padding, ct_m.output_padding)
                and isinstance(cv_d, nn.Conv2d) and isinstance(cv_m, nn.Conv2d) and cv_d.bias is None and cv_m.bias is None
                and cv_d.kernel_size == cv_m.kernel_size == (3, 3) and cv_d.stride == cv_m.stride == (1, 1)
                and cv_d.padding == cv_m.padding == (1, 1) and cv_d.in_channels == cv_d.out_channels == ct_d.out_channels
                and cv_m.in_channels == cv_m.out_channels == ct_m.out_channels
                and all(isinstance(b, nn.BatchNorm2d) and b.affine and b.track_running_stats
                        for b in (seq_d[1], seq_m[1], seq_d[4], seq_m[4]))
                and seq_d[1].eps == seq_m[1].eps and seq_d[1].momentum == seq_m[1].momentum
                and seq_d[4].eps == seq_m[4].eps and seq_d[4].momentum == seq_m[4].momentum)
    except (IndexError, TypeError):
        return False


def _heads_modules(ctx, seq_d, seq_m):
    ct_d, ct_m, cv_d, cv_m = seq_d[0], seq_m[0], seq_d[3], seq_m[3]
    od, om = ct_d.out_channels, ct_m.out_channels

    def make():
        g = ctx.grad
        ct = _Shim()
        ct.in_channels, ct.out_channels = ct_d.in_channels, od + om
        ct.kernel_size, ct.stride, ct.padding, ct.output_padding, ct.bias = ct_d.kernel_size, ct_d.stride, ct_d.padding, ct_d.output_padding, None
        cv = _Shim()
        cv.in_channels = cv.out_channels = od + om
        cv.kernel_size, cv.stride, cv.padding, cv.bias = (3, 3), (1, 1), (1, 1), None

        def build_ct(a, b):                                                                 # (in, out, 3, 3)
            return torch.cat([a, b], 1)

        def build_cv(a, b):                                                                 # block-diagonal (out, in, 3, 3)
            return torch.cat([F.pad(a, (0, 0, 0, 0, 0, om)), F.pad(b, (0, 0, 0, 0, od, 0))], 0)
        if g:
            ct.weight = _fused(build_ct, lambda gr: (gr[:, :od], gr[:, od:]), ct_d.weight, ct_m.weight)
            cv.weight = _fused(build_cv, lambda gr: (gr[:od, :od], gr[od:, od:]), cv_d.weight, cv_m.weight)
            # the packed layouts of these per-step tensors live in persistent buffers: the cache dictionaries are kept on the depth
            # stack and handed to every step's tensor (ops.pack_weight then re-packs in place instead of allocating and registering
            # a new set of buffers each step)
            store = seq_d.__dict__.setdefault('_efgh_heads_pack', {'ct': {}, 'cv': {}})
            ct.weight.__dict__['_efgh_cache'] = store['ct']
            cv.weight.__dict__['_efgh_cache'] = store['cv']
            # every step's tensor is a NEW content generation of that shared store: a fresh torch.cat output has version 0, no
            # optimizer epoch of its own, and - once the previous step's graph has been freed - usually the previous step's
            # address, so without the stamp ops._ver() would equal last step's key and the step-1 packing would be served forever
            ct.weight._efgh_gen = cv.weight._efgh_gen = next(_HEADS_GEN)
        else:
            ct.weight = build_ct(ct_d.weight.detach(), ct_m.weight.detach())
            cv.weight = build_cv(cv_d.weight.detach(), cv_m.weight.detach())
        return ct, _bn_cat((seq_d[1], seq_m[1]), g), cv, _bn_cat((seq_d[4], seq_m[4]), g)
    if ctx.grad or ctx.train:
        return make()                        # (new autograd leaves / fresh running statistics every step)
    ps = [ct_d.weight, ct_m.weight, cv_d.weight, cv_m.weight]
    for b in (seq_d[1], seq_m[1], seq_d[4], seq_m[4]):
        ps += [b.weight, b.bias, b.running_mean, b.running_var]
    vers = ops._ver(*ps) + tuple(b.num_batches_tracked._version for b in (seq_d[1], seq_m[1], seq_d[4], seq_m[4])
                                  if b.num_batches_tracked is not None)
    ent = ops._cached(seq_d, ('heads',), vers, lambda: (seq_m, make()))      # (cached on the depth stack; the partner is part of the entry)
    if ent[0] is not seq_m:
        return make()
    return ent[1]


def run_convt_heads(ctx, seq_d, seq_m, x):
    """the two convt_bn_relu stacks `seq_d` (depth) and `seq_m` (mask) over the same input -> [B][2H][2W][4]: channel 0 = seq_d's
    output, channels 1.. = seq_m's (see above)"""
    ct, bn1, cv, bn2 = _heads_modules(ctx, seq_d, seq_m)
    y = conv_transpose2d(ctx, x, ct, bn1, ACT_LEAKY, 0.2)
    y = conv2d(ctx, y, cv, bn2, ACT_LEAKY, 0.2)
    if ctx.train:
        bn1.sync()
        bn2.sync()
    return y


def run_basic_block(ctx, blk, x, out=None, alias_in=False):
    """nets/resnet.py:55-71."""
    def defer1(xin):
        # conv1's activation feeds conv2 only: a 2-D Winograd conv2 applies bn1 + ReLU inside its input transform (ops.LazyAct)
        c1 = blk.conv1
        Ho = (xin.shape[1] + 2 * c1.padding[0] - c1.kernel_size[0]) // c1.stride[0] + 1
        Wo = (xin.shape[2] + 2 * c1.padding[1] - c1.kernel_size[1]) // c1.stride[1] + 1
        return lazy_consumer_ok(ctx, blk.conv2, xin.shape[0], Ho, Wo)
    if ctx.grad and blk.downsample is None:
        # the identity branch takes an alias of x handed out by conv1's Function: its gradient is added in conv1's dgrad
        # epilogue instead of by autograd's elementwise accumulation (one read-read-write pass over the activation)
        y, idt = conv2d(ctx, x, blk.conv1, blk.bn1, ACT_RELU, skip_out=True, defer_act=defer1(x))
    elif ctx.grad:
        # x has two consumers here (conv1, downsample) and possibly a third outside (a decoder concatenation, alias_in): they are
        # chained through passthrough aliases, so the gradients of x are accumulated in dgrad epilogues, not by autograd
        # (the 1x1 / stride-2 downsample is first in the chain: in backward it is the last to run and adds its one parity class
        # IN PLACE to the gradient conv1 has already produced)
        idt, a1 = conv2d(ctx, x, blk.downsample[0], blk.downsample[1], ACT_NONE, skip_out=True)
        y, a2 = conv2d(ctx, a1, blk.conv1, blk.bn1, ACT_RELU, skip_out=True, defer_act=defer1(a1))
        y = conv2d(ctx, y, blk.conv2, blk.bn2, ACT_RELU, residual=idt, out=out)
        return (y, a2) if alias_in else y
    else:
        y = conv2d(ctx, x, blk.conv1, blk.bn1, ACT_RELU)
        idt = conv2d(ctx, x, blk.downsample[0], blk.downsample[1], ACT_NONE) if blk.downsample is not None else x
    y = conv2d(ctx, y, blk.conv2, blk.bn2, ACT_RELU, residual=idt, out=out)
    return (y, x) if alias_in else y


def run_resnet_layer(ctx, layer, x, out=None, alias_in=False):
    """alias_in: also return an alias of the layer's input for a further consumer of it (see run_basic_block)"""
    blocks = list(layer.children())
    alias = x
    for i, blk in enumerate(blocks):
        last = i == len(blocks) - 1
        if i == 0 and alias_in:
            x, alias = run_basic_block(ctx, blk, x, out=out if last else None, alias_in=True)
        else:
            x = run_basic_block(ctx, blk, x, out=out if last else None)
    return (x, alias) if alias_in else x


# ==============================================================================================
# training path: the same layers as autograd Functions with hand-written HIP backward (fn.py)
# ==============================================================================================
def _bn_args(bn):
    return (None, None) if bn is None else (bn.weight, bn.bias)


def _conv2d_grad(ctx, x, conv, bn, act, slope, residual, geom, dims, Cp, passthrough=False, pool=False, out=None, defer_act=False):
    B, H, W, Ho, Wo = dims
    Cw, O = conv.in_channels, conv.out_channels
    kh, kw = conv.kernel_size
    sh, sw = conv.stride
    ph, pw = conv.padding
    T, Np = kh * kw, ceil4(O)

    def pack_fwd(w, i):
        return ops.pack_weight(w, O, T, Cw, Cw * T, T, 1, list(range(T)), Np=Np, Cp=Cp, key=('conv', Np, Cp))

    def unpack(dWp, i, dW):
        ops.unpack_weight(dWp, dW, O, T, Cw, Cp, Cw * T, T, 1, list(range(T)))
    unpack.args = lambda i: (O, T, Cw, Cp, Cw * T, T, 1, list(range(T)), False)       # (ops.gather_wgrad(unpack=...): fold + unpack in one launch)

    def dgrad(spec, w, draw, xin, add=None, bnsrc=None, pre_v=None):
        """pre_v: B^T draw B, already made by ops.wino2d_bwd_transforms (draw itself is None then: it was never stored).
        add: a gradient that reached x through ANOTHER consumer (handed over on this layer's passthrough alias); it is folded
        into the result in the kernels' epilogues instead of by a separate elementwise pass of autograd.
        bnsrc: the BatchNorm layer that produced x (fn.BnSrc): its backward column sums are taken in this launch's epilogue when
        the kernel supports it, and the returned gradient is tagged with them"""
        dev = (draw if draw is not None else pre_v).device
        assert pre_v is None or (sh == 1 and sw == 1)
        if sh == 1 and sw == 1:
            # taps in ascending (dh, dw) order = the canonical 3x3 order the Winograd kernel recognises
            order = sorted(range(T), key=lambda i: (ph - i // kw, pw - i % kw))
            dhs = [ph - i // kw for i in order]
            dws = [pw - i % kw for i in order]
            Wd = ops.pack_weight(w, Cw, T, O, T, Cw * T, 1, order, Np=Cp, Cp=Np, key=('conv_d', Np, Cp))
            dx = torch.empty((B, H, W, Cp), dtype=torch.float32, device=dev)
            g = (B, Ho, Wo, H, W, 1, 1, dhs, dws, H, W, 1, 1, 0, 0)
            st = ops.gather_gemm(draw, Np, Np, T, Wd, Cp, B * H * W, dx, Cp, mode=1, geom=g,
                                 residual=add, ldr=0 if add is None else FN.ld_of(add), flops=2.0 * B * H * W * Cw * T * O,
                                 bn_bwd=bnsrc, pre_v=pre_v)
            if st is not None:
                dx._efgh_bnsums = (st, bnsrc, dx._version)      # (an in-place accumulation by autograd moves the version on)
            return dx
        assert sh == 2 and sw == 2
        classes = []
        for cy in range(2):
            for cx in range(2):
                khs = [k for k in range(kh) if (cy + ph - k) % 2 == 0]
                kws = [k for k in range(kw) if (cx + pw - k) % 2 == 0]
                taps = [(a, b) for a in khs for b in kws]
                Hv, Wv = (H - cy + 1) // 2, (W - cx + 1) // 2
                classes.append((cy, cx, taps, Hv, Wv))
        full = all(len(c[2]) > 0 and c[3] > 0 and c[4] > 0 for c in classes)
        if add is not None and not full:
            # e.g. the 1x1 / stride-2 downsample of a ResNet block: only one of the four input-parity classes receives anything.
            # The other gradient IS the result there, so the launches accumulate into it in place (every element is read and
            # written by one thread) - no zero fill, no addition pass
            dx, ldx_ = add, FN.ld_of(add)
            assert add.shape[-1] == Cp
        elif full:
            dx, ldx_ = torch.empty((B, H, W, Cp), dtype=torch.float32, device=dev), Cp
        else:
            dx, ldx_ = torch.zeros((B, H, W, Cp), dtype=torch.float32, device=dev), Cp
        for cy, cx, taps, Hv, Wv in classes:
            if not taps or Hv <= 0 or Wv <= 0:
                continue
            dhs = [(cy + ph - a) // 2 for a, _ in taps]
            dws = [(cx + pw - b) // 2 for _, b in taps]
            tapidx = [a * kw + b for a, b in taps]
            Wd = ops.pack_weight(w, Cw, len(taps), O, T, Cw * T, 1, tapidx, Np=Cp, Cp=Np,
                                 key=('conv_d2', cy, cx, Np, Cp))
            g = (B, Ho, Wo, Hv, Wv, 1, 1, dhs, dws, H, W, 2, 2, cy, cx)
            ops.gather_gemm(draw, Np, Np, len(taps), Wd, Cp, B * Hv * Wv, dx, ldx_, mode=1, geom=g,
                            residual=add, ldr=0 if add is None else FN.ld_of(add),
                            flops=2.0 * B * Hv * Wv * Cw * len(taps) * O)
        return dx

    dgrad.takes_bnsrc = True
    # data AND weight gradient on the 2-D Winograd path: the BatchNorm backward's apply pass rides in their transforms
    fus = bool((kh, kw, sh, sw, ph, pw) == (3, 3, 1, 1, 1, 1) and bn is not None and ctx.train
               and ops.lazy_capable(1, Np, Cp, _same3x3_geom(B, H, W)) and ops.wgrad_lazy_capable(1, Cp, Np, geom))
    spec = FN.LayerSpec(O, Cp, T, 1, [(geom, B * Ho * Wo)], B * Ho * Wo, (B, Ho, Wo), pack_fwd, dgrad, unpack,
                        bn=bn, train=ctx.train, act=act, slope=slope, c_real=Cw,
                        passthrough=passthrough and x.requires_grad and x.shape[-1] == Cp, pool=pool,
                        defer_act=defer_act, bwd_fusable=fus)
    g_, b_ = _bn_args(bn)
    out = FN.GemmLayerFn.apply(x, conv.weight, conv.bias, g_, b_, residual, spec, out)
    if passthrough and not spec.passthrough:
        return out, x
    return out


def _convt_grad(ctx, x, convt, bn, act, slope, geoms, dims, passthrough=False, defer_act=False):
    B, H, W, Ho, Wo = dims
    Cw, O = convt.in_channels, convt.out_channels
    ph, pw = convt.padding
    Np = ceil4(O)

    def pack_fwd(w, i):
        return geoms[i][1]

    def unpack(dWp, i, dW):
        tapidx = geoms[i][3]
        ops.unpack_weight(dWp, dW, O, len(tapidx), Cw, Cw, 9, O * 9, 1, tapidx)
    unpack.args = lambda i: (O, len(geoms[i][3]), Cw, Cw, 9, O * 9, 1, geoms[i][3], False)

    def dgrad(spec, w, draw, xin, add=None):
        # dX[ci][ih][iw] = sum_{co,kh,kw} dY[co][2ih-ph+kh][2iw-pw+kw] * W[ci][co][kh][kw]  (stride-2 conv)
        dhs = [k // 3 - ph for k in range(9)]
        dws = [k % 3 - pw for k in range(9)]
        Wd = ops.pack_weight(w, Cw, 9, O, O * 9, 9, 1, list(range(9)), Cp=Np, key=('convt_d', Np))
        dx = torch.empty((B, H, W, Cw), dtype=torch.float32, device=draw.device)
        g = (B, Ho, Wo, H, W, 2, 2, dhs, dws, H, W, 1, 1, 0, 0)
        ops.gather_gemm(draw, Np, Np, 9, Wd, Cw, B * H * W, dx, Cw, mode=1, geom=g,
                        residual=add, ldr=0 if add is None else FN.ld_of(add),
                        flops=2.0 * B * H * W * Cw * 9 * O / 4 * 4)
        return dx

    custom_fwd = custom_wgrad = None
    if O <= 3 and ph == pw and convt.bias is None and bn is not None:
        def custom_fwd(xin, w, out_raw):
            _convt_small_forward(xin, convt, out_raw)

        def custom_wgrad(xin, w, draw):
            # dWcol[(tap,o)][c] = sum_pix im2col(draw)[pix][(tap,o)] * x[pix][c]
            n9 = ceil4(9 * O)
            ycol = torch.empty((B * H * W, n9), dtype=torch.float32, device=draw.device)
            ops.convt_im2col(draw, B, H, W, Ho, Wo, O, ph, ycol)
            dwcol = torch.empty((n9, 1, Cw), dtype=torch.float32, device=draw.device)
            ops.gather_wgrad(xin, FN.ld_of(xin), Cw, 1, n9, B * H * W, ycol, n9, dwcol, mode=0)
            return dwcol[:9 * O, 0].reshape(9, O, Cw).permute(2, 1, 0).reshape(Cw, O, 3, 3).contiguous()

    spec = FN.LayerSpec(O, Cw, 0, 1, [(g[0], g[2]) for g in geoms], B * Ho * Wo, (B, Ho, Wo), pack_fwd, dgrad,
                        unpack, bn=bn, train=ctx.train, act=act, slope=slope, c_real=Cw,
                        custom_forward=custom_fwd, custom_wgrad=custom_wgrad,
                        passthrough=passthrough and x.requires_grad, defer_act=defer_act and custom_fwd is None)
    g_, b_ = _bn_args(bn)
    out = FN.GemmLayerFn.apply(x, convt.weight, convt.bias, g_, b_, None, spec)
    if passthrough and not spec.passthrough:
        return out, x
    return out


def _linear_grad(ctx, x, M, C, weight, bias, bn, act, slope):
    O = weight.shape[0]
    Cp, Np = ceil4(C), ceil4(O)

    def pack_fwd(w, i):
        return ops.pack_weight(w, O, 1, C, C, 1, 1, [0], Np=Np, Cp=Cp, key=('lin', Np, Cp))

    def unpack(dWp, i, dW):
        ops.unpack_weight(dWp, dW, O, 1, C, Cp, C, 1, 1, [0])
    unpack.args = lambda i: (O, 1, C, Cp, C, 1, 1, [0], False)

    def dgrad(spec, w, draw, xin):
        Wd = ops.pack_weight(w, C, 1, O, 1, C, 1, [0], Np=Cp, Cp=Np, key=('lin_d', Np, Cp))
        dx = torch.empty((M, Cp), dtype=torch.float32, device=draw.device)
        ops.gather_gemm(draw, Np, Np, 1, Wd, Cp, M, dx, Cp, mode=0, flops=2.0 * M * C * O)
        return dx

    spec = FN.LayerSpec(O, Cp, 1, 0, [(None, M)], M, (M,), pack_fwd, dgrad, unpack, bn=bn, train=ctx.train, act=act,
                        slope=slope, c_real=C)
    g_, b_ = _bn_args(bn)
    return FN.GemmLayerFn.apply(x, weight, bias, g_, b_, None, spec)


def _blur_grad(ctx, splat, H, C, table, conv0, lv=None):
    C0 = conv0.out_channels

    def pack_fwd(w, i):
        return ops.pack_weight(w, C0, 15, C, C * 15, 15, 1, list(range(15)), key=('blur0',))

    def unpack(dWp, i, dW):
        ops.unpack_weight(dWp, dW, C0, 15, C, C, C * 15, 15, 1, list(range(15)))
    unpack.args = lambda i: (C0, 15, C, C, C * 15, 15, 1, list(range(15)), False)

    def dgrad(spec, w, draw, xin):
        if lv is not None and ops.BLUR_DGRAD_FUSED and C % 4 == 0 and C0 % 4 == 0:
            return ops.blur_dgrad(lv, draw, C0, w, C)        # one gather-GEMM through the lattice's symmetric table
        # tmp[m][t*C+c] = sum_n draw[m][n] * W0[n][c][t], then scattered through the neighbour table
        Wd = w.detach().squeeze(-1).permute(2, 1, 0).contiguous().view(15 * C, C0)
        tmp = torch.empty((H, 15 * C), dtype=torch.float32, device=draw.device)
        ops.gather_gemm(draw, C0, C0, 1, Wd, 15 * C, H, tmp, 15 * C, mode=0, flops=2.0 * H * 15 * C * C0)
        if lv is not None:              # through the lattice's own (symmetric) table: a gather, no atomics
            return ops.neighbor_gather_adjoint(lv, tmp, C)
        dsplat = torch.zeros((H, C), dtype=torch.float32, device=draw.device)
        ops.table_scatter_add(tmp, table, H, 15, C, dsplat)
        return dsplat

    spec = FN.LayerSpec(C0, C, 15, 2, [(None, H)], H, (H,), pack_fwd, dgrad, unpack, act=ACT_RELU, table=table,
                        c_real=C)
    return FN.GemmLayerFn.apply(splat, conv0.weight, conv0.bias, None, None, None, spec)
