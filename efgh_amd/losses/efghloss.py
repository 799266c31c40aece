"""EFGHCriterion with the reference's API (losses/efghloss.py:3-38, losses/loss_utils.py):
`loss_name`, `compute_loss(pc, img, calib, A, gt, pred) -> (losses, gt)`.

No host synchronisation (the reference loops over the batch with .item()/int() on every sample).  The pose terms (E/H cosine +
sign cross-entropy, F mined BCE, G translation) and all ground-truth poses are ONE HIP kernel forward and one backward
(csrc/pose.hip, PoseLossFn); the image terms are another (csrc/loss.hip, GImageLossFn); the GT depth image goes through the HIP
rasteriser.  `_compute_loss_expressions` keeps the same loss as batched device tensor expressions (`common.pose.USE_KERNELS = False`; what the
kernels are tested against).  Quirks
reproduced on purpose: `total` sums every dict entry (so e_gn / h_hrzn count twice,
efghloss.py:33-36), g_mask is scaled by lambda_g_mask AND lambda_g_depth (loss_utils.py:199,204),
gt['g_trs'] is built from un-detached predictions (:170-175)."""
import ctypes
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..common import pose
from ..nets import fn as FN


_GT = {'e_gn': (0, 3), 'e_l': (3, 19), 'h_hrzn': (19, 22), 'h_c': (22, 31), 'f_l': (31, 47), 'g_trs': (47, 50), 'g_l': (50, 66),
       'e_gn_abs': (66, 69), 'h_hrzn_abs': (69, 71)}          # columns of efgh_pose_loss_fwd's gt72 (include/efgh_hip.h)


class _PoseLossDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ('e_gn_abs', 'e_gn_sgn', 'h_hrzn_abs', 'h_hrzn_sgn', 'f_score', 'g_trs', 'e_l', 'f_l')] + \
               [(n, ctypes.c_int64) for n in ('ld_e_gn_sgn', 'ld_h_hrzn_sgn', 'ld_f_score')] + \
               [(n, ctypes.c_void_p) for n in ('rand_init_l', 'rand_init_c', 'sensor2_T_sensor1')] + \
               [(n, ctypes.c_int32) for n in ('rand_init_l_dim', 'rand_init_c_dim', 'B', 'W', 'fov_pos_num')] + \
               [(n, ctypes.c_float) for n in ('fov_neg_ratio', 'lambda_e_gn', 'lambda_h_hrzn', 'lambda_fov', 'lambda_g_trs',
                                              'lambda_g_depth', 'lambda_g_mask')]


def _pose_desc(t, cfg):
    e_abs, e_sgn, h_abs, h_sgn, fs, g_trs, e_l, f_l, rl, rc, T4 = t
    lam, pos, neg = cfg
    p = lambda x: ctypes.c_void_p(x.data_ptr())
    return _PoseLossDesc(p(e_abs), p(e_sgn), p(h_abs), p(h_sgn), p(fs), p(g_trs), p(e_l), p(f_l), e_sgn.stride(0), h_sgn.stride(0),
                         fs.stride(0), p(rl), p(rc), p(T4), rl.shape[-1], rc.shape[-1], fs.shape[0], fs.shape[1], int(pos), float(neg),
                         lam['e_gn'],
                         lam['h_hrzn'], lam['fov'], lam['g_trs'], lam['g_depth'], lam['g_mask'])


class PoseLossFn(torch.autograd.Function):
    """(predictions, l_depth, l_mask, ground truth) -> (L[11] in loss_name order, gt72, gt classes, gt f_score)"""

    @staticmethod
    def forward(ctx, e_abs, e_sgn, h_abs, h_sgn, f_score, g_trs, e_l, f_l, l_dep, l_msk, rand_l, rand_c, T4, cfg):
        dev = f_score.device
        t = [x.detach() for x in (e_abs, e_sgn, h_abs, h_sgn, f_score, g_trs, e_l, f_l, rand_l, rand_c, T4)]
        t = [x if (x.dim() == 2 and x.stride(1) == 1) else x.contiguous() for x in t]
        ops._C.require_cuda(*t)
        ops._C.require_f32(*t)
        B, W = t[4].shape
        assert t[0].shape[:2] == (B, 3) and t[1].shape[1] >= 8 and t[2].shape[:2] == (B, 2) and t[3].shape[1] >= 4
        assert t[8].shape[1:] in ((3, 3), (4, 4)) and t[9].shape[1:] in ((3, 3), (4, 4)) and t[10].shape[1:] == (4, 4)
        f32 = dict(dtype=torch.float32, device=dev)
        gtbuf, gtcls = torch.empty((B, 72), **f32), torch.empty((B, 2), dtype=torch.int64, device=dev)
        gtfs, sel = torch.empty((B, W), **f32), torch.empty((B, W), **f32)
        part = torch.empty(B * (7 + 2 * ((W + 255) // 256)), **f32)
        L, nsel = torch.empty(11, **f32), torch.empty(1, **f32)
        desc = _pose_desc(t, cfg)
        ops._C.check(ops._C.lib().efgh_pose_loss_fwd(ctypes.byref(desc), ops.ptr(l_dep.detach()), ops.ptr(l_msk.detach()),
                                                     ops.ptr(gtbuf), ops.ptr(gtcls), ops.ptr(gtfs), ops.ptr(sel), ops.ptr(part),
                                                     ops.ptr(L), ops.ptr(nsel), ops._C.stream_ptr()))
        ctx.save_for_backward(*t, sel, nsel)
        ctx.cfg = cfg
        ctx.shapes = [x.shape for x in (e_abs, e_sgn, h_abs, h_sgn, f_score, g_trs, e_l)]
        ctx.mark_non_differentiable(gtbuf, gtcls, gtfs)
        return L, gtbuf, gtcls, gtfs

    @staticmethod
    def backward(ctx, gL, _a, _b, _c):
        *t, sel, nsel = ctx.saved_tensors
        dev = sel.device
        B, W = sel.shape
        f32 = dict(dtype=torch.float32, device=dev)
        g = [torch.empty(sh, **f32) for sh in ((B, 3), (B, 8), (B, 2), (B, 4), (B, W), (B, 3), (B, 4, 4))]
        g2 = torch.empty(2, **f32)
        desc = _pose_desc(t, ctx.cfg)
        ops._C.check(ops._C.lib().efgh_pose_loss_bwd(ctypes.byref(desc), ops.ptr(gL.contiguous()), ops.ptr(sel), ops.ptr(nsel),
                                                     *[ops.ptr(x) for x in g], ops.ptr(g2), ops._C.stream_ptr()))
        sh = ctx.shapes
        if sh[1][1] != 8 or sh[3][1] != 4:      # sign logits handed over with their padding columns
            raise ops._C.EfghError('sign logits must be (B, 8) / (B, 4)')
        return (g[0].view(sh[0]), g[1], g[2].view(sh[2]), g[3], g[4], g[5].view(sh[5]), g[6], None, g2[0], g2[1], None, None,
                None, None)


class EFGHCriterion(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.device = args['DEVICE']
        self.lam = dict(args['lambda'])
        self.positive_num = args['fov_pos_num']
        self.neg_ratio = args['fov_neg_ratio']
        self.raw_cam_img_size = args['raw_cam_img_size']
        self.loss_name = ['total', 'e_gn', 'e_gn_sgn', 'e_gn_abs', 'h_hrzn', 'h_hrzn_abs', 'h_hrzn_sgn', 'fov',
                          'g_trs', 'g_depth', 'g_mask']                  # efghloss.py:13-17

    # ---- loss_utils.py:25-58 / :227-262 ------------------------------------------------------
    @staticmethod
    def _abs_sign(pred_abs, pred_sgn, gt_vec, nd):
        gt_abs = torch.abs(gt_vec)[:, :nd, :]
        s = torch.sign(gt_vec)[:, :, 0]
        s = torch.where(s == -1, torch.zeros_like(s), s)
        w = pose.const(tuple(float(2 ** (nd - 1 - i)) for i in range(nd)), s.device, s.dtype)
        cls = (s[:, :nd] * w[None]).sum(1).long()
        cos = F.cosine_similarity(pred_abs, gt_abs, dim=1)
        l_abs = torch.mean(1 - cos) * 10.0
        l_sgn = F.cross_entropy(pred_sgn, cls) * 1.0
        return l_abs, l_sgn, gt_abs, cls

    def _gt_fov(self, axis, width):
        """Floss.gt_fov, loss_utils.py:119-144: `positive_num` ones centred on the GT yaw, wrapping"""
        yaw = torch.atan2(axis[:, 1, 0], axis[:, 0, 0]).detach()
        f_idx = ((-yaw + math.pi) / (2 * math.pi)) * width
        xmin = f_idx.long() - int(self.positive_num / 2)
        j = torch.arange(width, device=axis.device)[None, :]
        return (torch.remainder(j - xmin[:, None], width) < self.positive_num).float()

    dp_exact = True       # weight the masked depth mean by the ranks' valid-pixel counts under data parallelism (see below)

    def _dp_weight_masked_mean(self, l_dep, n_valid):
        """`torch.nn.DataParallel` (main.py:127) gathers the replicas' outputs and computes ONE loss over the global batch
        (iterater.py:35-42).  Every term of efghloss is a plain batch mean - so the mean over ranks of the per-rank terms IS the
        global term - except `g_depth`, a mean over the VALID pixels of the whole batch (loss_utils.py:186-190: sum of squares of
        all samples / valid pixels of all samples): the mean of the ranks' own masked means weights rank r by 1 / n_r instead of
        1 / mean(n).  With a process group of more than one rank the per-rank term is therefore rescaled by n_r / mean_r(n_r)
        (one 4-byte all-reduce, issued here in the forward, before any gradient bucket): the all-reduced mean of the ranks' losses
        and gradients then equals the single global loss EXACTLY, whatever the samples' valid-pixel counts
        (tests/test_gpu_dp.py::test_two_ranks_equal_dataparallel_on_the_real_net measures both forms).  The F term selects the same
        number of scores per sample (positive_num * (1 + neg_ratio), loss_utils.py:96-115), so its mean needs no weight."""
        import torch.distributed as dist
        if not (self.dp_exact and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return l_dep
        n_mean = n_valid.detach().clone().reshape(1)
        dist.all_reduce(n_mean, op=dist.ReduceOp.SUM)
        n_mean = n_mean / dist.get_world_size()
        scale = torch.where(n_mean > 0, n_valid.reshape(1) / n_mean.clamp_min(1.0), torch.ones_like(n_mean))
        return l_dep * scale.reshape(()).detach()

    def compute_loss(self, pc, img, calib, A, gt, pred):
        if not pose.USE_KERNELS:
            return self._compute_loss_expressions(pc, img, calib, A, gt, pred)
        dev = pred['f_score'].device
        ops._C.require_cuda(pred['f_score'], pc, calib, A)
        ops._C.require_f32(pc, calib, A)
        gt = dict(gt)
        B = pc.size(0)
        f32 = lambda t: torch.as_tensor(t).to(dev).float()
        rawH, rawW = self.raw_cam_img_size
        with torch.no_grad():
            gdep, _ = ops.depth_image(pc, f32(gt['cam_T_velo']), rawH, rawW)   # [B][H][W][4], depth = channel 3
        gt['img_mask'] = torch.as_tensor(gt['img_mask']).to(dev)
        imask = gt['img_mask'].to(torch.uint8).contiguous()
        # masked L2 on the depth image + BCE on the mask image: one HIP sweep forward, one backward (loss_utils.py:186-199)
        l_dep, l_msk_mean, gt['g_depth'], gt['g_mask'], n_valid = FN.GImageLossFn.apply(pred['g_depth'], pred['g_mask'], gdep, imask)
        l_dep = self._dp_weight_masked_mean(l_dep, n_valid)
        cfg = (self.lam, self.positive_num, self.neg_ratio)
        Lv, gtbuf, gtcls, gtfs = PoseLossFn.apply(pred['e_gn_abs'], pred['e_gn_sgn'], pred['h_hrzn_abs'], pred['h_hrzn_sgn'],
                                                  pred['f_score'], pred['g_trs'], pred['e_l'], pred['f_l'], l_dep, l_msk_mean,
                                                  f32(gt['rand_init_l']), f32(gt['rand_init_c']),
                                                  f32(gt['sensor2_T_sensor1']), cfg)
        col = lambda k: gtbuf[:, _GT[k][0]:_GT[k][1]]
        for k in ('e_gn', 'h_hrzn', 'g_trs', 'e_gn_abs', 'h_hrzn_abs'):
            gt[k] = col(k).unsqueeze(-1)
        for k in ('e_l', 'f_l', 'g_l'):
            gt[k] = col(k).view(B, 4, 4)
        gt['h_c'] = col('h_c').view(B, 3, 3)
        gt['e_gn_sgn'], gt['h_hrzn_sgn'] = gtcls[:, 0], gtcls[:, 1]
        gt['f_score'] = gtfs
        idx = {n: i for i, n in enumerate(self.loss_name)}
        L = {k: Lv[idx[k]] for k in ('e_gn', 'e_gn_abs', 'e_gn_sgn', 'h_hrzn', 'h_hrzn_abs', 'h_hrzn_sgn', 'fov', 'g_trs',
                                     'g_depth', 'g_mask', 'total')}
        return L, gt

    def _compute_loss_expressions(self, pc, img, calib, A, gt, pred):
        dev = pred['f_score'].device
        ops._C.require_cuda(pred['f_score'], pc, calib, A)
        ops._C.require_f32(pc, calib, A)
        lam = self.lam
        f32 = lambda t: torch.as_tensor(t).to(dev).float()
        L, gt = self._pose_terms_expressions(dict(gt), pred)
        rawH, rawW = self.raw_cam_img_size
        with torch.no_grad():
            gdep, _ = ops.depth_image(pc, f32(gt['cam_T_velo']), rawH, rawW)   # [B][H][W][4], depth = channel 3
        gt['img_mask'] = torch.as_tensor(gt['img_mask']).to(dev)
        imask = gt['img_mask'].to(torch.uint8).contiguous()
        # masked L2 on the depth image + BCE on the mask image: one HIP sweep forward, one backward (loss_utils.py:186-199)
        l_dep, l_msk_mean, gt['g_depth'], gt['g_mask'], n_valid = FN.GImageLossFn.apply(pred['g_depth'], pred['g_mask'], gdep, imask)
        l_dep = self._dp_weight_masked_mean(l_dep, n_valid)
        l_msk = l_msk_mean * lam['g_mask']
        L['g_depth'] = l_dep * lam['g_depth']
        L['g_mask'] = l_msk * lam['g_depth']
        total = 0
        for k in L:
            total = total + L[k]
        L['total'] = total
        return L, gt

    def _pose_terms_expressions(self, gt, pred):
        """E / H / F / G-translation entries of the loss dictionary and the ground-truth poses, as tensor expressions"""
        dev = pred['f_score'].device
        lam = self.lam
        B = pred['f_score'].size(0)
        f32 = lambda t: torch.as_tensor(t).to(dev).float()
        e1, e2, e3 = pose.const((1., 0., 0.), dev), pose.const((0., 1., 0.), dev), pose.const((0., 0., 1.), dev)
        L = {}
        # ---- E
        R = f32(gt['rand_init_l'])[:, :3, :3]
        g = torch.bmm(R, e3[None, :, None].expand(B, -1, -1))
        g = g / torch.sqrt(torch.sum(g ** 2, 1, keepdim=True))
        gt['e_gn'] = g
        gt['e_l'] = pose.rotation_between(g, e3)
        la, ls, gt['e_gn_abs'], gt['e_gn_sgn'] = self._abs_sign(pred['e_gn_abs'], pred['e_gn_sgn'], g, 3)
        L['e_gn'] = (la + ls) * lam['e_gn']
        L['e_gn_abs'] = la * lam['e_gn']
        L['e_gn_sgn'] = ls * lam['e_gn']
        # ---- H
        R = f32(gt['rand_init_c'])[:, :3, :3]
        g = torch.bmm(R, e2[None, :, None].expand(B, -1, -1))
        g = g / torch.sqrt(torch.sum(g ** 2, 1, keepdim=True))
        gt['h_hrzn'] = g
        gt['h_c'] = pose.rotation_between(g, e2)[:, :3, :3]
        la, ls, gt['h_hrzn_abs'], gt['h_hrzn_sgn'] = self._abs_sign(pred['h_hrzn_abs'], pred['h_hrzn_sgn'], g, 2)
        L['h_hrzn'] = (la + ls) * lam['h_hrzn']
        L['h_hrzn_abs'] = la * lam['h_hrzn']
        L['h_hrzn_sgn'] = ls * lam['h_hrzn']
        # ---- F (loss_utils.py:77-117)
        T4 = f32(gt['sensor2_T_sensor1'])
        Tinv = pose.inv(T4[:, :3, :3])
        pe = pred['e_l'][:, :3, :3].detach()
        axis = torch.bmm(torch.bmm(pe, Tinv), e1[None, :, None].expand(B, -1, -1))
        W = pred['f_score'].size(-1)
        gt['f_score'] = self._gt_fov(axis, W)
        ge = gt['e_l'][:, :3, :3].detach()
        fl = torch.zeros((B, 4, 4), device=dev)
        fl[:, :3, :3] = pose.inv(torch.bmm(ge, Tinv))
        fl[:, 3, 3] = 1
        gt['f_l'] = fl
        pos = gt['f_score'] > 0
        lc = F.binary_cross_entropy(pred['f_score'], gt['f_score'], reduction='none').detach().clone()
        lc[pos] = 0
        _, idx = lc.sort(1, descending=True)
        _, rank = idx.sort(1)
        num_pos = pos.long().sum(1, keepdim=True)
        num_neg = torch.clamp(self.neg_ratio * num_pos, max=pos.size(1) - 1)
        wsel = pos | (rank < num_neg)
        # every sample selects positive_num*(1+neg_ratio) entries, so the masked mean equals the
        # reference's mean over the (B, -1) view
        lf = F.binary_cross_entropy(pred['f_score'], gt['f_score'], reduction='none')
        L['fov'] = (lf * wsel).sum() / wsel.sum() * lam['fov']
        # ---- G (loss_utils.py:165-207)
        origin = pose.const((0., 0., 0., 1.), dev)[None, :, None].expand(B, -1, -1)
        pef = torch.bmm(pred['f_l'], pred['e_l'])
        gt['g_trs'] = torch.bmm(torch.bmm(T4, pose.inv(pef)), origin)[:, :3, :]
        gef = torch.bmm(gt['f_l'], gt['e_l'])
        gcp = torch.bmm(torch.bmm(T4, pose.inv(gef)), origin)
        gt['g_l'] = pose.translation_matrix(gcp)
        l_trs = F.smooth_l1_loss(gt['g_trs'], pred['g_trs'])
        L['g_trs'] = l_trs * lam['g_trs']
        return L, gt
