"""Batched pose-head algebra without host synchronisation (replaces the per-sample python loops
with .item()/.tolist() of common/torch_utils.py:105-146, 170-233, 256-296): one HIP launch per head
(csrc/pose.hip), forward and - on the training path - a hand-written backward.  The same formulas as
device tensor expressions are kept below as the `USE_KERNELS = False` path (what the kernels are tested
against).  Differentiability mirrors the reference: the skew matrix K is built from detached values,
only (1-c)/s^2 carries gradient (torch_utils.py:184,194); translation matrices are detached (:229)."""
import ctypes
import math
import os

import torch

from .. import _C

_CONST = {}
USE_KERNELS = True      # the heads are csrc/pose.hip kernels.  False (set by tests/test_gpu_pose.py only): the same formulas as batched tensor
                        # expressions - the independent reference the kernels and their hand-written backward are held to, not a product path


def _fused():
    return USE_KERNELS


class PoseHeadFn(torch.autograd.Function):
    """head_normal with the backward of csrc/pose.hip (k_head_normal_bwd)"""

    @staticmethod
    def forward(ctx, abs_logits, sgn_logits, dest):
        ctx.save_for_backward(abs_logits, sgn_logits)
        ctx.dest = dest
        return _head_normal_launch(abs_logits, sgn_logits, dest)

    @staticmethod
    def backward(ctx, ga, gn, gR):
        abs_logits, sgn_logits = ctx.saved_tensors
        B, nd = abs_logits.shape
        d = ctx.dest
        g = torch.empty((B, nd), dtype=torch.float32, device=abs_logits.device)
        ga, gn, gR = (None if t is None else t.contiguous() for t in (ga, gn, gR))
        _C.check(_C.lib().efgh_pose_head_normal_bwd(
            _C.ptr(abs_logits), ctypes.c_int64(abs_logits.stride(0)), _C.ptr(sgn_logits), ctypes.c_int64(sgn_logits.stride(0)),
            ctypes.c_int32(B), ctypes.c_int32(nd), ctypes.c_float(d[0]), ctypes.c_float(d[1]), ctypes.c_float(d[2]),
            _C.ptr(ga), _C.ptr(gn), _C.ptr(gR), _C.ptr(g), _C.stream_ptr()))
        return g, None, None


def head_normal(abs_logits, sgn_logits, dest):
    """softmax_l2 + normal_from_abs_sign + rotation_between in ONE launch: abs_logits (B,nd) and sgn_logits (B,2^nd) row views
    -> (abs (B,nd,1), normal (B,nd,1), R (B,4,4))"""
    B, nd = abs_logits.shape
    if not _fused():
        a = softmax_l2(abs_logits)
        n = normal_from_abs_sign(a, sgn_logits, nd)
        n3 = n if nd == 3 else torch.cat([n, torch.zeros(B, 1, 1, device=n.device)], 1)
        return a, n, rotation_between(n3, const(dest, n.device))
    if torch.is_grad_enabled() and abs_logits.requires_grad:
        return PoseHeadFn.apply(abs_logits, sgn_logits, tuple(dest))
    return _head_normal_launch(abs_logits.detach(), sgn_logits.detach(), dest)


def _head_normal_launch(abs_logits, sgn_logits, dest):
    B, nd = abs_logits.shape
    _C.require_cuda(abs_logits, sgn_logits)
    assert abs_logits.stride(1) == 1 and sgn_logits.stride(1) == 1 and sgn_logits.shape[1] == 1 << nd
    dev = abs_logits.device
    a = torch.empty((B, nd, 1), dtype=torch.float32, device=dev)
    n = torch.empty((B, nd, 1), dtype=torch.float32, device=dev)
    R = torch.empty((B, 4, 4), dtype=torch.float32, device=dev)
    _C.check(_C.lib().efgh_pose_head_normal(_C.ptr(abs_logits), ctypes.c_int64(abs_logits.stride(0)), _C.ptr(sgn_logits),
                                            ctypes.c_int64(sgn_logits.stride(0)), ctypes.c_int32(B), ctypes.c_int32(nd),
                                            ctypes.c_float(dest[0]), ctypes.c_float(dest[1]), ctypes.c_float(dest[2]),
                                            _C.ptr(a), _C.ptr(n), _C.ptr(R), _C.stream_ptr()))
    return a, n, R


def const(values, device, dtype=torch.float32):
    """small constant tensors, uploaded once per device: `torch.tensor(..., device=cuda)` is a blocking host-to-device copy,
    i.e. a stream synchronisation in the middle of the forward"""
    key = (tuple(values), str(device), dtype)
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = torch.tensor(list(values), device=device, dtype=dtype)
    return t


def inv(m):
    """batched matrix inverse WITHOUT torch.inverse's host-side singularity check (a device-to-host copy and a stream
    synchronisation per call): same LU factorisation, same values"""
    return torch.linalg.inv_ex(m)[0]


def softmax_l2(x):
    """softmax(dim=1) then L2-normalise (enet.py:161-164, hnet.py:59-63) -> (B,C,1)"""
    a = torch.softmax(x, 1)
    return (a / torch.sqrt(torch.sum(a * a, 1, keepdim=True))).unsqueeze(-1)


def normal_from_abs_sign(abs_, sign_logits, ndim):
    """torch_utils.py:105-146: class = first argmax; bits MSB-first -> +-1; normal = abs*sign"""
    cls = torch.argmax(sign_logits, dim=1)                              # softmax is monotone
    shifts = torch.arange(ndim - 1, -1, -1, device=cls.device)
    bits = (cls[:, None] >> shifts[None, :]) & 1
    sgn = (bits * 2 - 1).to(abs_.dtype)
    return abs_ * sgn[:, :, None]


def rotation_between(srce, dest):
    """torch_utils.py:170-200.  srce (B,3,1), dest (3,) constant -> (B,4,4)"""
    B = srce.size(0)
    v1 = srce[:, :, 0]
    v2 = dest.to(v1)[None, :].expand(B, -1)
    v = torch.linalg.cross(v1, v2, dim=1)
    c = torch.sum(v1 * v2, 1)
    s2 = torch.sum(v * v, 1)                                            # s**2
    vd = v.detach()
    z = torch.zeros_like(vd[:, 0])
    K = torch.stack([torch.stack([z, -vd[:, 2], vd[:, 1]], 1),
                     torch.stack([vd[:, 2], z, -vd[:, 0]], 1),
                     torch.stack([-vd[:, 1], vd[:, 0], z], 1)], 1)      # (B,3,3)
    eye3 = torch.eye(3, device=v1.device, dtype=v1.dtype)[None]
    s = torch.sqrt(s2)
    coef = (1 - c) / (s * s)
    rot3 = eye3 + K + torch.bmm(K, K) * coef[:, None, None]
    same = (1 - c) == 0
    opp = (1 + c) == 0
    neg = -eye3.expand(B, -1, -1).clone()
    fix0 = (v1[:, 0] == 0) & (v2[:, 0] == 0)
    fix2 = (v1[:, 2] == 0) & (v2[:, 2] == 0) & ~fix0
    neg[:, 0, 0] = torch.where(fix0, torch.ones_like(c), neg[:, 0, 0])
    neg[:, 2, 2] = torch.where(fix2, torch.ones_like(c), neg[:, 2, 2])
    rot3 = torch.where(opp[:, None, None], neg, rot3)
    rot3 = torch.where(same[:, None, None], eye3.expand(B, -1, -1), rot3)
    R = torch.zeros((B, 4, 4), device=v1.device, dtype=v1.dtype)
    R[:, 3, 3] = 1
    # the reference's -I case also negates [3,3]
    R[:, 3, 3] = torch.where(opp & ~same, -torch.ones_like(c), R[:, 3, 3])
    R = R.clone()
    R[:, :3, :3] = rot3
    return R


def translation_matrix(vec):
    """torch_utils.py:220-233 (detached)"""
    B = vec.size(0)
    t = torch.eye(4, device=vec.device, dtype=vec.dtype)[None].repeat(B, 1, 1)
    t[:, :3, 3] = vec[:, :3, 0].detach()
    return t


class CamTVeloFn(torch.autograd.Function):
    """A^-1 c_T A calib l_T with gradients w.r.t. c_T and l_T (csrc/pose.hip k_cam_T_velo / k_cam_T_velo_bwd)"""

    @staticmethod
    def forward(ctx, c_T, l_T, calib, A):
        _C.require_cuda(c_T, l_T, calib, A)
        B = l_T.shape[0]
        c, l, k, a = c_T.contiguous(), l_T.contiguous(), calib.contiguous(), A.contiguous()
        out = torch.empty((B, 3, 4), dtype=torch.float32, device=l_T.device)
        _C.check(_C.lib().efgh_pose_cam_T_velo(_C.ptr(c), ctypes.c_int64(9), _C.ptr(l), _C.ptr(k), _C.ptr(a), ctypes.c_int32(B),
                                               _C.ptr(out), _C.stream_ptr()))
        ctx.save_for_backward(c, l, k, a)
        return out

    @staticmethod
    def backward(ctx, g):
        c, l, k, a = ctx.saved_tensors
        B = l.shape[0]
        gc = torch.empty((B, 3, 3), dtype=torch.float32, device=l.device) if ctx.needs_input_grad[0] else None
        gl = torch.empty((B, 4, 4), dtype=torch.float32, device=l.device) if ctx.needs_input_grad[1] else None
        if gc is not None or gl is not None:
            _C.check(_C.lib().efgh_pose_cam_T_velo_bwd(_C.ptr(c), ctypes.c_int64(9), _C.ptr(l), _C.ptr(k), _C.ptr(a),
                                                       _C.ptr(g.contiguous()), ctypes.c_int32(B), _C.ptr(gc), _C.ptr(gl),
                                                       _C.stream_ptr()))
        return gc, gl, None, None


def _mat44(a, b, ta=0, tb=0):
    B = a.shape[0]
    out = torch.empty((B, 4, 4), dtype=torch.float32, device=a.device)
    _C.check(_C.lib().efgh_pose_mat44_mul(_C.ptr(a), _C.ptr(b), ctypes.c_int32(B), ctypes.c_int32(ta), ctypes.c_int32(tb),
                                          _C.ptr(out), _C.stream_ptr()))
    return out


class ComposeFn(torch.autograd.Function):
    """a @ b for (B,4,4) poses"""

    @staticmethod
    def forward(ctx, a, b):
        _C.require_cuda(a, b)
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        return _mat44(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        return (_mat44(g, b, 0, 1) if ctx.needs_input_grad[0] else None,
                _mat44(a, g, 1, 0) if ctx.needs_input_grad[1] else None)


def compose(a, b):
    """torch.bmm(a, b) for (B,4,4) poses without a library GEMM launch"""
    if not _fused():
        return torch.bmm(a, b)
    return ComposeFn.apply(a, b)


def compute_cam_T_velo(c_T, l_T, calib, A):
    """torch_utils.py:256-269"""
    if _fused():
        if torch.is_grad_enabled() and (c_T.requires_grad or l_T.requires_grad):
            return CamTVeloFn.apply(c_T, l_T, calib, A)
        with torch.no_grad():
            return CamTVeloFn.apply(c_T, l_T, calib, A)
    m = torch.bmm(calib, l_T)
    m = torch.bmm(A, m)
    m = torch.bmm(c_T, m)
    return torch.bmm(inv(A), m)


def yaw_rotation_from_scores(f_score):
    """fnet.py:87-91: argmax -> yaw -> (cos,sin,0) -> rotation onto e1"""
    n = f_score.size(-1)
    if _fused():
        _C.require_cuda(f_score)
        assert f_score.stride(1) == 1
        R = torch.empty((f_score.shape[0], 4, 4), dtype=torch.float32, device=f_score.device)
        _C.check(_C.lib().efgh_pose_head_yaw(_C.ptr(f_score), ctypes.c_int64(f_score.stride(0)), ctypes.c_int32(f_score.shape[0]),
                                             ctypes.c_int32(n), _C.ptr(R), _C.stream_ptr()))
        return R
    f_idx = torch.argmax(f_score, dim=1, keepdim=True).float()
    f_rad = -(f_idx / (n - 1)) * 2 * math.pi + math.pi
    rad = f_rad[:, 0].double()                       # python math.cos/sin operate in double
    f_fwd = torch.stack([torch.cos(rad), torch.sin(rad), torch.zeros_like(rad)], 1).float()[:, :, None]
    e1 = const((1., 0., 0.), f_score.device)
    return rotation_between(f_fwd, e1)
