"""EFGHCriterion with the reference's API (losses/efghloss.py:3-38, losses/loss_utils.py):
`loss_name`, `compute_loss(pc, img, calib, A, gt, pred) -> (losses, gt)`.

Batched device tensor expressions without host synchronisation (the reference loops over the batch
with .item()/int() on every sample); the GT depth image goes through the HIP rasteriser.  Quirks
reproduced on purpose: `total` sums every dict entry (so e_gn / h_hrzn count twice,
efghloss.py:33-36), g_mask is scaled by lambda_g_mask AND lambda_g_depth (loss_utils.py:199,204),
gt['g_trs'] is built from un-detached predictions (:170-175)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..common import pose
from ..nets import fn as FN


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

    def compute_loss(self, pc, img, calib, A, gt, pred):
        dev = pred['f_score'].device
        ops._C.require_cuda(pred['f_score'], pc, calib, A)
        ops._C.require_f32(pc, calib, A)
        lam = self.lam
        gt = dict(gt)
        B = pc.size(0)
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
        rawH, rawW = self.raw_cam_img_size
        with torch.no_grad():
            gdep, _ = ops.depth_image(pc, f32(gt['cam_T_velo']), rawH, rawW)   # [B][H][W][4], depth = channel 3
        gt['img_mask'] = torch.as_tensor(gt['img_mask']).to(dev)
        imask = gt['img_mask'].to(torch.uint8).contiguous()
        l_trs = F.smooth_l1_loss(gt['g_trs'], pred['g_trs'])
        # masked L2 on the depth image + BCE on the mask image: one HIP sweep forward, one backward (loss_utils.py:186-199)
        l_dep, l_msk_mean, gt['g_depth'], gt['g_mask'] = FN.GImageLossFn.apply(pred['g_depth'], pred['g_mask'], gdep, imask)
        l_msk = l_msk_mean * lam['g_mask']
        L['g_trs'] = l_trs * lam['g_trs']
        L['g_depth'] = l_dep * lam['g_depth']
        L['g_mask'] = l_msk * lam['g_depth']
        total = 0
        for k in L:
            total = total + L[k]
        L['total'] = total
        return L, gt
