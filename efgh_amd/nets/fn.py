"""torch.autograd.Function shims over the HIP forward/backward kernels (training path).

torch's autograd engine is used only as the tape (plumbing); every forward and backward
computation inside these Functions runs in libefgh_hip.so.  Tensors are "row matrices":
shape [..., C] with unit stride on the last axis and a uniform row stride (channel slices of a
contiguous channels-last buffer qualify), addressed as (data_ptr, ld).
"""
import torch

from .. import ops
from ..ops import ACT_NONE, ceil4


def rows_ok(t):
    if t.stride(-1) != 1:
        return False
    ld = t.stride(-2) if t.dim() >= 2 else t.shape[-1]
    exp = ld
    for d in range(t.dim() - 2, -1, -1):
        if t.shape[d] != 1 and t.stride(d) != exp:
            return False
        exp *= t.shape[d]
    return ld % 4 == 0 and t.data_ptr() % 16 == 0


def as_rows(t):
    return t if rows_ok(t) else t.contiguous()


def ld_of(t):
    return t.stride(-2) if t.dim() >= 2 else t.shape[-1]


class LayerSpec:
    """Static description of one GEMM layer (built by layers.py).

    launches: [(geom | None, M_launch)] forward launches (several for a transposed conv);
    pack_fwd(weight, i) -> packed forward weight of launch i;
    dgrad(spec, weight, draw, x) -> gradient w.r.t. x;
    wgrad_unpack(dWp, i, dW): scatter launch i's packed weight gradient into the reference layout."""

    def __init__(self, N, C, T, mode, launches, M, out_shape, pack_fwd, dgrad, wgrad_unpack, bn=None,
                 train=False, act=ACT_NONE, slope=0.0, table=None, c_real=None, custom_forward=None,
                 custom_wgrad=None, passthrough=False, pool=False, defer_act=False, bwd_fusable=False):
        self.N, self.C, self.T, self.mode, self.M = N, C, T, mode, M
        self.launches, self.out_shape = launches, out_shape
        self.pack_fwd, self.dgrad, self.wgrad_unpack = pack_fwd, dgrad, wgrad_unpack
        self.bn, self.train, self.act, self.slope, self.table = bn, train, act, slope, table
        self.c_real = c_real if c_real is not None else C
        # optional replacements of the launch loop / weight gradient (e.g. transposed conv as GEMM + col2im)
        self.custom_forward, self.custom_wgrad = custom_forward, custom_wgrad
        # passthrough: forward also returns an alias of x for a skip connection; the gradient arriving on that alias is added
        # in the dgrad kernel's epilogue (`dgrad(..., add=)`) instead of by a separate elementwise pass
        self.passthrough = passthrough
        # pool: the layer is followed by MaxPool2d(2,2); BatchNorm + activation + pooling run as one pass over the raw output and
        # the full-resolution activation is never stored (backward recomputes the window from raw)
        self.pool = pool
        # defer_act: the caller guarantees that this layer's activation has ONE consumer and that it is a 2-D Winograd layer
        # (ops.lazy_capable): the normalise + activate pass is not run, the raw output is returned carrying an ops.LazyAct
        self.defer_act = defer_act
        # bwd_fusable: the layer's data AND weight gradient both run on the 2-D Winograd path, so the BatchNorm backward's apply
        # pass can ride in their gradient-side transforms (ops.wino2d_bwd_transforms); its dgrad takes pre_v=
        self.bwd_fusable = bwd_fusable


def materialize(x, lazy):
    """act(x*scale + shift) of a raw BatchNorm output with a pending activation, for a consumer that cannot apply it itself"""
    M = x.numel() // x.shape[-1]
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    ops.scale_shift_act(x, ld_of(x), lazy.scale, lazy.shift, y, x.shape[-1], M, x.shape[-1], lazy.act, lazy.slope)
    return y


def claim_grad(p):
    """(flat gradient view, deliver()) when parameter `p` lives in a train.FlatParams whose slice may be overwritten by this
    backward, else (None, None): the backward then returns the gradient to autograd as usual."""
    slot = getattr(p, '_efgh_flat', None) if p is not None else None
    if slot is None:
        return None, None
    flat, i = slot
    g = flat.claim(p, i)
    if g is None:
        return None, None
    return g, (lambda stream=None: flat.deliver(i, stream))


def note_use(*params):
    """forward-side count of a FlatParams parameter's consumers in this step (train.FlatParams.uses)"""
    for p in params:
        slot = getattr(p, '_efgh_flat', None) if p is not None else None
        if slot is not None:
            slot[0].uses[slot[1]] += 1


def _bias_grad_behind_bn(ctx, p_bias, draw, M, Np, N, train_bn, delivered):
    """gradient of a conv / conv1d bias that feeds a BatchNorm.  Train mode: the batch mean absorbs the bias, so
    d/dbias = sum_m draw = coef * (sum dpre - M * mean(dpre) - mean(dpre * xhat) * sum xhat) = 0 exactly (sum xhat = 0); a column-sum
    pass over draw only measures rounding noise (which is what the reference's autograd returns, `tests/test_oracle_e2e.py`), so the
    exact zero is returned without touching draw.  Eval-mode statistics (frozen BatchNorm with autograd on): the real column sum."""
    if not ctx.needs_input_grad[2]:
        return None
    if not train_bn:
        return ops.col_sum(draw, M, Np)[:N]
    g, done = claim_grad(p_bias)
    if g is not None:                      # the flat gradient slice is already zero (FlatParams.zero_grad)
        delivered.append(done)
        return None
    return torch.zeros(N, dtype=torch.float32, device=ctx.saved_tensors[0].device)


class BnSrc:
    """what the BatchNorm backward of a layer needs besides dy, attached (`_efgh_bnsrc`) to the activation the layer returns: the
    consumer of that activation hands it to its data-gradient kernel, which then takes the layer's two column sums in its epilogue
    (ops.gather_gemm bn_bwd) and tags the gradient it returns (`_efgh_bnsums`); the layer's own backward finds the tag on its dy
    and skips the reduction pass over dy and raw.  Any detour of the gradient through autograd (several consumers summed by the
    engine, a view, a clone) arrives as another tensor object without the tag: the layer then reduces as before."""
    __slots__ = ('raw', 'y', 'psc', 'psh', 'mean', 'invstd', 'act', 'slope', 'M', 'N')

    def __init__(self, raw, y, psc, psh, mean, invstd, act, slope, M, N):
        self.raw, self.y, self.psc, self.psh, self.mean, self.invstd = raw, y, psc, psh, mean, invstd
        self.act, self.slope, self.M, self.N = act, slope, M, N

    def fits(self, M, N):
        return self.M == M and self.N == N


TRACE = None          # debug aid (tools/layer_census.py): list of per-layer records appended by GemmLayerFn.backward


class GemmLayerFn(torch.autograd.Function):
    """y = act(BN(gemm(x, W) + bias) + residual)   with hand-written backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, residual, spec, out_target=None):
        ctx.set_materialize_grads(False)      # an unused passthrough alias must arrive as None, not as a zero tensor to add
        ctx.xsrc = getattr(x, '_efgh_bnsrc', None)           # the BatchNorm layer that produced x (BnSrc), if any
        lazy = getattr(x, '_efgh_lazy', None)                # x is a RAW BatchNorm output whose activation this layer applies itself
        x = as_rows(x)
        if lazy is not None and not (spec.custom_forward is None and len(spec.launches) == 1 and not spec.passthrough
                                     and x.shape[-1] == spec.C
                                     and ops.lazy_capable(spec.mode, spec.C, ceil4(spec.N), spec.launches[0][0])):
            x, lazy = materialize(x, lazy), None              # (a consumer the producer was not told about: never wrong, only slower)
        ctx.lazy = lazy
        dev = x.device
        N, Np = spec.N, ceil4(spec.N)
        M = spec.M
        bn = spec.bn
        b = None if bias is None else ops.pad_vec(bias.detach(), Np)
        out = torch.empty(tuple(spec.out_shape) + (Np,), dtype=torch.float32, device=dev)
        res = None if residual is None else as_rows(residual)
        mean = invstd = raw = None
        need_stats = bn is not None and spec.train
        stats = None
        thin = spec.custom_forward is not None or all(
            ops.thin_eligible(spec.mode, spec.C, Np, spec.T if g is None else len(g[7])) for g, _ in spec.launches)
        if need_stats and not thin:
            gs = [ops.stats_rows(spec.mode, spec.C, Np, g, m) for (g, m) in spec.launches]
            stats = torch.empty((sum(gs), 2, Np), dtype=torch.float32, device=dev)
        fused_plain = bn is None            # bias (+residual) (+act) straight in the epilogue
        g0 = 0
        if spec.custom_forward is not None:
            assert bn is not None and bias is None
            spec.custom_forward(x, weight, out)
        ops.TLS.w2v_wanted = bool(ctx.needs_input_grad[1])
        for li, (geom, m) in enumerate(spec.launches if spec.custom_forward is None else []):
            wp = spec.pack_fwd(weight, li)
            T = spec.T if geom is None else len(geom[7])
            fl = 2.0 * m * N * T * spec.c_real
            if fused_plain:
                ops.gather_gemm(x, ld_of(x), spec.C, T, wp, Np, m, out, Np, mode=spec.mode, geom=geom,
                                table=spec.table, bias=b, residual=res, ldr=0 if res is None else ld_of(res),
                                act=spec.act, slope=spec.slope, flops=fl, lazy=lazy)
            else:
                st = None
                if need_stats and not thin:
                    st = stats[g0:g0 + gs[li]]
                    g0 += gs[li]
                ops.gather_gemm(x, ld_of(x), spec.C, T, wp, Np, m, out, Np, mode=spec.mode, geom=geom,
                                table=spec.table, bias=b, act=ACT_NONE, stats=st, flops=fl, lazy=lazy)
        ops.TLS.w2v_wanted = False
        y = out
        if bn is not None:
            raw = out
            if need_stats and thin:
                stats, _ = ops.col_stats(out, M, Np, Np)
            if need_stats:
                ops.bn_tick(bn)          # num_batches_tracked += 1, batched per forward / per step
                momentum = bn.momentum if bn.momentum is not None else 0.1
                g_, b_ = ops.pad_vec(gamma.detach(), Np), ops.pad_vec(beta.detach(), Np)
                if Np == N:
                    rm, rv = bn.running_mean, bn.running_var
                else:
                    rm, rv = ops.pad_vec(bn.running_mean, Np).clone(), ops.pad_vec(bn.running_var, Np, 1.0).clone()
                scale, shift, mean, invstd = ops.bn_finalize(stats, stats.shape[0], Np, float(M), g_, b_, rm, rv,
                                                             momentum, bn.eps, save=True)
                if Np != N:
                    bn.running_mean.copy_(rm[:N])
                    bn.running_var.copy_(rv[:N])
            else:
                with torch.no_grad():
                    invstd = ops.pad_vec(torch.rsqrt(bn.running_var + bn.eps), Np)
                    mean = ops.pad_vec(bn.running_mean.clone(), Np)
                    scale = ops.pad_vec(gamma.detach(), Np) * invstd
                    shift = ops.pad_vec(beta.detach(), Np) - mean * scale
            if spec.pool:
                assert res is None
                y = ops.maxpool2_affine(raw, scale, shift, spec.act, spec.slope)
            elif (spec.defer_act and ops.LAZY_ACT and need_stats and res is None and Np == N and out_target is None
                  and spec.custom_forward is None):
                # the single consumer (a 2-D Winograd layer) normalises and activates inside its input transform: no pass here,
                # no activation tensor; this layer's own backward re-derives the mask from raw*scale + shift as it always did
                # (an ALIAS object of raw, not raw itself: the BnSrc tag below refers to raw, and y referring to something that refers to
                # y would be a cycle - a step's activations would wait for the garbage collector instead of their reference counts)
                y = raw.detach()
                y._efgh_lazy = ops.LazyAct(scale, shift, spec.act, spec.slope)
            else:
                # out_target = (buffer [..][Ct], channel offset): the activation is written straight into that channel slice (the
                # training-path form of the decoder's torch.cat, see ConcatFn); the returned tensor is a view of the buffer
                y = None
                if out_target is not None and Np == N:
                    tb, toff = out_target
                    if tuple(tb.shape[:-1]) == tuple(raw.shape[:-1]) and toff % 4 == 0 and toff + N <= tb.shape[-1]:
                        y = tb[..., toff:toff + N]
                if y is None:
                    y = torch.empty_like(raw)
                # a residual layer's activation mask cannot be re-derived from raw alone: its backward used to read the activation
                # back in both passes; the forward pass leaves the SIGN BITS instead (M*N/32 words)
                ybits = None
                if (res is not None and need_stats and ops.BN_MASK_BITS and Np % 32 == 0 and spec.act != ACT_NONE
                        and any(ctx.needs_input_grad)):
                    ybits = torch.empty(M * Np // 32, dtype=torch.int32, device=dev)
                ops.scale_shift_act(raw, Np, scale, shift, y, ld_of(y), M, Np, spec.act, spec.slope, res=res,
                                    ldr=0 if res is None else ld_of(res), bits=ybits)
                ctx.ybits = ybits
        if TRACE is not None:
            y._efgh_src = ('bn' if bn is not None else 'plain', spec.N, spec.pool, residual is not None)
        ctx.bnsrc = None
        if need_stats and not spec.pool and Np == N and any(ctx.needs_input_grad):
            # (mask from y for residual layers, re-derived from raw * scale + shift otherwise - as the layer's own backward does)
            # (y.detach(): an alias object - y itself carrying a reference to something that refers to y would be a cycle, and a step's
            # activations would wait for the garbage collector instead of being freed by reference count)
            ctx.bnsrc = y._efgh_bnsrc = BnSrc(raw, y.detach() if residual is not None else None, scale, shift, mean, invstd,
                                              spec.act, spec.slope, M, Np)
        ctx.spec = spec
        if not hasattr(ctx, 'ybits'):
            ctx.ybits = None
        ctx.train_step = ops.TLS.train_step      # (backward runs on autograd's device thread: it restores the caller's switch)
        ctx.has = (bias is not None, gamma is not None, residual is not None)
        ctx.params = (weight, bias, gamma, beta)      # the Parameter objects themselves (claim_grad), not saved copies
        if any(ctx.needs_input_grad[1:5]):
            note_use(weight, bias, gamma, beta)
        # layers without a residual re-derive the activation mask from raw*scale+shift in backward
        psc, psh = (scale, shift) if (bn is not None and residual is None) else (None, None)
        # coef = gamma * invstd of the BatchNorm backward is the forward's `scale`
        ctx.save_for_backward(x, weight, y, raw, mean, invstd, scale if bn is not None else None, psc, psh)
        if spec.passthrough:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        old, ops.TLS.train_step = ops.TLS.train_step, ctx.train_step
        try:
            return GemmLayerFn._backward(ctx, dy, dskip)
        finally:
            ops.TLS.train_step = old

    @staticmethod
    def _backward(ctx, dy, dskip=None):
        spec = ctx.spec
        sums = getattr(dy, '_efgh_bnsums', None) if dy is not None else None
        if sums is not None and (sums[1] is not ctx.bnsrc or sums[2] != dy._version):
            sums = None
        if dy is None:                        # (only the alias was used downstream)
            dy = torch.zeros(tuple(spec.out_shape) + (ceil4(spec.N),), dtype=torch.float32, device=ctx.saved_tensors[0].device)
        x, weight, y, raw, mean, invstd, coef, psc, psh = ctx.saved_tensors
        ymask = None if psc is not None else y
        ybits = ctx.ybits
        has_bias, has_bn, has_res = ctx.has
        N, Np, M = spec.N, ceil4(spec.N), spec.M
        dev = x.device
        fused_pool = spec.pool and has_bn and spec.train and not has_res and Np == N
        if TRACE is not None:
            g0 = spec.launches[0][0] if spec.launches else None
            TRACE.append(dict(M=M, N=N, C=spec.C, T=spec.T, mode=spec.mode, bn=has_bn, res=has_res, pool=spec.pool, act=spec.act,
                              nl=len(spec.launches), wino=bool(g0 is not None and ops.wino_eligible(spec.mode, spec.C, Np, g0)),
                              wino2d=bool(g0 is not None and ops.wino2d_eligible(spec.mode, spec.C, Np, g0)),
                              c4=bool(g0 is not None and ops.c4_eligible(spec.mode, spec.C, Np, g0)),
                              x_from=getattr(x, '_efgh_src', None), dx=bool(ctx.needs_input_grad[0]),
                              in_elems=int(x.numel()), custom=spec.custom_forward is not None))
        if spec.pool and not fused_pool:    # pooled gradient -> full resolution (window recomputed from raw*scale+shift)
            dy = ops.maxpool2_bwd_affine(raw, psc, psh, spec.act, spec.slope, dy.contiguous())
        dy = as_rows(dy)
        dbias = dgamma = dbeta = dres = None
        need_pre = spec.act != ACT_NONE or has_bn or has_res or has_bias
        p_w, p_bias, p_gamma, p_beta = ctx.params
        delivered = []
        # per-channel sums go straight into the flat gradient slices when the parameters live in a FlatParams (no padding)
        gs1 = gs2 = None
        if Np == N and has_bn and ctx.needs_input_grad[3] and ctx.needs_input_grad[4]:
            gs1, d1 = claim_grad(p_beta)
            gs2, d2 = claim_grad(p_gamma) if gs1 is not None else (None, None)
            if gs2 is None:
                gs1 = None
            else:
                delivered += [d1, d2]
        elif Np == N and has_bias and not has_bn and ctx.needs_input_grad[2]:
            gs1, d1 = claim_grad(p_bias)
            if gs1 is not None:
                delivered.append(d1)
        own_wgrad = ctx.needs_input_grad[1] and spec.custom_wgrad is None
        pre_v = pre_gy = None
        # the apply pass of the BatchNorm backward inside the gradient-side transforms of a 2-D Winograd layer (draw is never stored)
        fuse_bwd = (ops.W2_BWD_FUSED and spec.bwd_fusable and has_bn and spec.train and Np == N
                    and ctx.needs_input_grad[0] and own_wgrad and (ybits is not None or (psc is not None and not has_res))
                    and len(spec.out_shape) == 3 and ((fused_pool and ops.W2_BWD_FUSED_POOL) or not spec.pool))
        if fused_pool:
            # BatchNorm backward straight from the pooled gradient (no full-resolution dy is ever written)
            draw, s1, s2 = ops.pool_bn_bwd(dy.contiguous(), raw, mean, invstd, coef, psc, psh, spec.act, spec.slope,
                                           s1=gs1, s2=gs2, transforms=fuse_bwd, y_pool=y)
            if fuse_bwd:
                (pre_v, pre_gy), draw = draw, None
            if gs1 is None:
                dbeta, dgamma = s1, s2
            if has_bias:
                dbias = _bias_grad_behind_bn(ctx, p_bias, draw, M, Np, N, True, delivered)
        elif not need_pre:
            draw = dy
        else:
            G = ops.bwd_groups(M)
            part = torch.empty((G, 2, Np), dtype=torch.float64, device=dev)      # float64 column sums (see backward.hip)
            s1 = gs1 if gs1 is not None else torch.empty(Np, dtype=torch.float32, device=dev)
            s2 = gs2 if gs2 is not None else torch.empty(Np, dtype=torch.float32, device=dev)
            train_bn = has_bn and spec.train
            m1 = torch.empty(Np, dtype=torch.float64, device=dev) if train_bn else None
            m2 = torch.empty(Np, dtype=torch.float64, device=dev) if train_bn else None
            ldy = Np if ymask is None else ld_of(ymask)
            if ybits is not None:                     # (the mask as sign bits: `y` with ldy = 0, see efgh_act_bn_bwd_reduce)
                ymask, ldy = ybits, 0
            if sums is not None and train_bn:
                # the kernel that produced dy took the two column sums in its epilogue: fold its per-block partials
                f1, f2, m1, m2 = ops.bwd_finalize_f32(sums[0], Np, float(M))
                if gs1 is not None:
                    gs1.copy_(f1)
                    gs2.copy_(f2)
                s1, s2 = (gs1, gs2) if gs1 is not None else (f1, f2)
            else:
                ops.act_bn_bwd_reduce(dy, ld_of(dy), ymask, ldy, raw, Np, mean if has_bn else None,
                                      invstd if has_bn else None, M, Np, spec.act, spec.slope, part, s1, s2, m1, m2,
                                      pscale=psc, pshift=psh)
            if has_bn and gs1 is None:
                # (s1 / s2 are this backward's own tensors: handed to autograd as they are when there is no padding to cut off)
                dbeta, dgamma = (s1, s2) if Np == N else (s1[:N].clone(), s2[:N].clone())
            if has_bias and not has_bn and gs1 is None:
                dbias = s1 if Np == N else s1[:N].clone()
            if fuse_bwd:
                Bo, Ho, Wo = spec.out_shape
                pre_v, pre_gy, dres = ops.wino2d_bwd_transforms(dy, ld_of(dy), raw, Np, ybits, None if ybits is not None else psc,
                                                                None if ybits is not None else psh, mean, invstd, coef, m1, m2, Np,
                                                                Bo, Ho, Wo, spec.act, spec.slope, has_res)
                draw = None
            else:
                draw = torch.empty(tuple(spec.out_shape) + (Np,), dtype=torch.float32, device=dev)
                if has_res:
                    dres = torch.empty(tuple(spec.out_shape) + (Np,), dtype=torch.float32, device=dev)
                ops.act_bn_bwd_apply(dy, ld_of(dy), ymask, ldy, raw, Np, mean if train_bn else None,
                                     invstd if train_bn else None, coef, m1, m2, M, Np, spec.act, spec.slope, draw, Np,
                                     dres, Np, pscale=psc, pshift=psh)
            if has_bias and has_bn:          # bias in front of BatchNorm: d/dbias = column sums of draw
                dbias = _bias_grad_behind_bn(ctx, p_bias, draw, M, Np, N, train_bn, delivered)
        # ---- dgrad is on the critical path of backward and is enqueued first; the weight gradient only needs draw and x, and
        # nothing in this backward pass waits for it: when it can be written straight into the flat gradient buffer it goes to the
        # weight-gradient stream BEHIND this layer's dgrad, so that it runs underneath the HBM-bound BatchNorm passes of the next
        # layer instead of competing with the dgrad for the matrix pipes (measured: waiting only for draw costs 3 ms per step)
        gW = dw_done = side = None
        if own_wgrad:
            gW, dw_done = claim_grad(p_w)
            side = ops.wgrad_stream(dev) if (gW is not None and ops.WGRAD_SIDE) else None
        dx = None
        if ctx.needs_input_grad[0]:
            kw = {}
            if dskip is not None:
                kw['add'] = as_rows(dskip)
            if ctx.xsrc is not None and getattr(spec.dgrad, 'takes_bnsrc', False):
                kw['bnsrc'] = ctx.xsrc
            if pre_v is not None:
                kw['pre_v'] = pre_v
            dx = spec.dgrad(spec, weight, draw, x, **kw)
        # ---- wgrad
        dW = None
        if ctx.needs_input_grad[1] and spec.custom_wgrad is not None:
            dW = spec.custom_wgrad(x, weight, draw)
        elif own_wgrad:
            dW = gW if gW is not None else torch.empty_like(weight)

            xw, lazy_w = x, ctx.lazy
            if lazy_w is not None and not ops.wgrad_lazy_capable(spec.mode, spec.C, Np, spec.launches[0][0]):
                xw, lazy_w = materialize(x, lazy_w), None      # (a switch flipped between forward and backward: never wrong, only slower)

            def run_wgrad():
                x = xw
                for li, (geom, m) in enumerate(spec.launches):
                    T = spec.T if geom is None else len(geom[7])
                    dWp = torch.empty((Np, T, spec.C), dtype=torch.float32, device=dev)
                    ua = getattr(spec.wgrad_unpack, 'args', None)
                    done = ops.gather_wgrad(x, ld_of(x), spec.C, T, Np, m, draw, Np, dWp, mode=spec.mode, geom=geom,
                                            table=spec.table, unpack=None if ua is None else (dW,) + tuple(ua(li)),
                                            lazy=lazy_w, pre_gy=pre_gy)
                    if not done:             # (a single row chunk, or a path with a transform behind its fold)
                        spec.wgrad_unpack(dWp, li, dW)
            if side is None:
                run_wgrad()
            else:
                side.wait_stream(torch.cuda.current_stream())
                x.record_stream(side)
                (draw if draw is not None else pre_gy).record_stream(side)
                with torch.cuda.stream(side):
                    run_wgrad()
            if gW is not None:
                dW = None
                delivered.append(dw_done if side is None else (lambda: dw_done(side)))
        if dres is not None and Np != N:
            dres = dres[..., :N]
        for done in delivered:               # after the writes are enqueued: the all-reduce bucket countdown
            done()
        return dx, dW, dbias, dgamma, dbeta, dres, None, None


# ---------------------------------------------------------------------------------------------
class ConcatFn(torch.autograd.Function):
    """torch.cat(parts, -1) of activations that their producers ALREADY wrote into adjacent channel slices of `buf`
    (GemmLayerFn's out_target): no copy forward, channel-slice views of the gradient backward.  `concat()` falls back to
    torch.cat when a part lives elsewhere (cropped decoder outputs of odd-height maps, layers without an out_target)."""

    @staticmethod
    def forward(ctx, buf, *parts):
        ctx.widths = [p.shape[-1] for p in parts]
        return buf.view_as(buf)

    @staticmethod
    def backward(ctx, g):
        outs, off = [], 0
        for w in ctx.widths:
            outs.append(g[..., off:off + w])
            off += w
        return (None,) + tuple(outs)


def concat(buf, parts):
    off, ok = 0, buf is not None
    for p in parts:
        ok = ok and p.shape[:-1] == buf.shape[:-1] and p.stride() == buf.stride() and \
            p.data_ptr() == buf.data_ptr() + 4 * off
        off += p.shape[-1]
    if ok and off == buf.shape[-1]:
        return ConcatFn.apply(buf, *parts)
    return torch.cat(parts, -1)


class MaxPool2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.maxpool2(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        B, H, W, C = x.shape
        dx = torch.zeros_like(x) if (H % 2 or W % 2) else torch.empty_like(x)
        ops.maxpool2_bwd(x, dy.contiguous(), dx)
        return dx


class SegmentColMaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, seg, nseg, C):
        x = as_rows(x)
        y, arg = ops.segment_colmax(x, ld_of(x), C, seg, nseg, want_arg=True)
        ctx.save_for_backward(arg)
        ctx.shape = (tuple(x.shape), C, nseg)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        shape, C, nseg = ctx.shape
        dx = torch.zeros(shape, dtype=torch.float32, device=dy.device)
        ops.segment_colmax_bwd(dy.contiguous(), arg, nseg, C, dx, shape[-1])
        return dx, None, None, None


class SegmentColMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, P, nseg, C):
        x = as_rows(x)
        ctx.meta = (tuple(x.shape), P, nseg, C)
        return ops.segment_colmean(x, ld_of(x), C, P, nseg)

    @staticmethod
    def backward(ctx, dy):
        shape, P, nseg, C = ctx.meta
        dx = torch.zeros(shape, dtype=torch.float32, device=dy.device)
        ops.segment_colmean_bwd(dy.contiguous(), P, nseg, C, dx, shape[-1])
        return dx, None, None, None


class SplatFn(torch.autograd.Function):
    """BCL splat + density normalisation of one lattice level; the level's rows are [el_minus_gr | feat[:, :Cf]]; the lattice
    arrays carry no gradient (generate_data.py:119)."""

    @staticmethod
    def forward(ctx, feat, lv, Cf, use_emg=True, normalize=True):
        feat = as_rows(feat)
        splat, wsum = ops.splat_fwd(lv, feat, Cf, use_emg, normalize)
        ctx.save_for_backward(wsum)
        ctx.meta = (lv, Cf, feat.shape[-1], use_emg, normalize)
        return splat

    @staticmethod
    def backward(ctx, g):
        (wsum,) = ctx.saved_tensors
        lv, Cf, ldf, use_emg, normalize = ctx.meta
        gfeat = torch.zeros((lv.n_in, ldf), dtype=torch.float32, device=g.device) if ldf != Cf else \
            torch.empty((lv.n_in, Cf), dtype=torch.float32, device=g.device)
        ops.splat_bwd(lv, g.contiguous(), wsum, Cf, gfeat, use_emg, normalize)
        return gfeat, None, None, None, None


class Softmax2ToNchwFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = ops.softmax2_to_nchw(x)
        ctx.save_for_backward(y)
        ctx.ld = x.shape[-1]
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        B, _, H, W = y.shape
        dx = torch.empty((B, H, W, ctx.ld), dtype=torch.float32, device=dy.device)
        ops.softmax2_bwd(y, dy.contiguous(), dx)
        return dx


class HeadsToNchwFn(torch.autograd.Function):
    """G's depth and mask heads carried as one 4-channel map (layers.run_convt_heads) -> g_depth, g_mask (gnet.py:121-124)"""
    @staticmethod
    def forward(ctx, x):
        depth, mask = ops.heads_to_nchw(x.contiguous())
        ctx.save_for_backward(mask)
        ctx.set_materialize_grads(False)
        return depth, mask

    @staticmethod
    def backward(ctx, gd, gm):
        (mask,) = ctx.saved_tensors
        return ops.heads_bwd(mask, None if gm is None else gm.contiguous(), None if gd is None else gd.contiguous())


class NhwcToNchwFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Cs):
        ctx.meta = (tuple(x.shape), Cs)
        return ops.nhwc_to_nchw(x.contiguous(), Cs)

    @staticmethod
    def backward(ctx, dy):
        shape, Cs = ctx.meta
        return ops.nchw_to_nhwc(dy.contiguous(), shape[-1]), None


class RangeImageFn(torch.autograd.Function):
    """range image of e_l.[pc;1]; gradient w.r.t. e_l through the rasterised VALUES (x,y,z,r) only
    (indices are truncated integers), every rasterised point receives grad_img[u,v] (oracle note)."""

    @staticmethod
    def forward(ctx, pc, e_l, H, W, fov_up, fov_down):
        img, pix = ops.range_image(pc, e_l, H, W, fov_up, fov_down)
        ctx.save_for_backward(pc, e_l, pix)
        ctx.hw = (H, W)
        return img

    @staticmethod
    def backward(ctx, g):
        pc, e_l, pix = ctx.saved_tensors
        H, W = ctx.hw
        B, _, N = pc.shape
        ge = ops.raster_pose_bwd(pix, g.contiguous(), pc, e_l.detach(), B, N, H * W, 0).view(B, 4, 4)
        return None, ge, None, None, None, None


class DepthImageFn(torch.autograd.Function):
    """depth image; only the 4th channel (w) depends on cam_T_velo (row 2)."""

    @staticmethod
    def forward(ctx, pc, P, H, W):
        img, pix = ops.depth_image(pc, P, H, W)
        ctx.save_for_backward(pc, pix)
        ctx.hw = (H, W)
        return img

    @staticmethod
    def backward(ctx, g):
        pc, pix = ctx.saved_tensors
        H, W = ctx.hw
        B, _, N = pc.shape
        gP = ops.raster_pose_bwd(pix, g.contiguous(), pc, None, B, N, H * W, 1).view(B, 3, 4)
        return None, gP, None, None


class CorrHeadFn(torch.autograd.Function):
    """fnet.py:57,64,78-81 incl. the (max-min) normalisation and the mirror/circular pad."""

    @staticmethod
    def forward(ctx, cam, rng):
        cam, rng = cam.contiguous(), rng.contiguous()
        score, logit, rp, cam_mm, rng_mm = ops.corr_head(cam, rng, want_logit=True, want_aux=True)
        ctx.save_for_backward(cam, rng, rp, cam_mm, rng_mm, score)
        return score

    @staticmethod
    def backward(ctx, ds):
        cam, rng, rp, cam_mm, rng_mm, score = ctx.saved_tensors
        B, h, wc, C = cam.shape
        wr = rng.shape[2]
        off = int(wr / 8)
        dl = (ds * score * (1 - score) / 16.0).contiguous()                  # d/dlogit, incl. the 1/C scale
        dcam_n, drp = ops.corr1d_bwd(rp, cam, cam_mm, dl, B, h, wc, wr + 2 * off)
        drng_n = ops.corr_unpad(drp, B, h, wr, C, off)

        return ops.norm_bwd(cam, dcam_n.contiguous(), cam_mm), ops.norm_bwd(rng, drng_n.contiguous(), rng_mm)


class GImageLossFn(torch.autograd.Function):
    """(l_depth, l_mask) of Gloss over the full-resolution depth / mask images (loss_utils.py:186-199), one HIP sweep each way"""

    @staticmethod
    def forward(ctx, pred_depth, pred_mask, gdep4, img_mask):
        pred_depth, pred_mask = pred_depth.contiguous(), pred_mask.contiguous()
        out3, gt_depth, gt_mask = ops.gimg_loss_fwd(pred_depth, pred_mask, gdep4, img_mask)
        ctx.save_for_backward(pred_depth, pred_mask, gt_depth, img_mask, out3)
        n_valid = out3[2].detach().clone()             # pixels the masked depth mean was taken over (data-parallel weighting)
        ctx.mark_non_differentiable(gt_depth, gt_mask, n_valid)
        return out3[0], out3[1], gt_depth, gt_mask, n_valid

    @staticmethod
    def backward(ctx, g_dep, g_msk, _a, _b, _c):
        pred_depth, pred_mask, gt_depth, img_mask, out3 = ctx.saved_tensors
        d_depth, d_mask = ops.gimg_loss_bwd(pred_depth, pred_mask, gt_depth, img_mask, out3,
                                            g_dep.reshape(1).float().contiguous(), g_msk.reshape(1).float().contiguous())
        return d_depth, d_mask, None, None
