"""H-net: horizon from the camera image (reference nets/hnet.py) on the HIP path."""
import torch
import torch.nn as nn

from .. import ops
from ..common import pose
from ..ops import ACT_RELU
from . import fn as FN
from . import layers as L
from .builders import VGGFeatures


class Hnet(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.device = args['DEVICE']
        self.vgg = VGGFeatures('A')
        self.conv_hrzn_1 = nn.Conv1d(512, 256, 1)
        self.conv_hrzn_2 = nn.Conv1d(256, 128, 1)
        self.conv_hrzn_3 = nn.Conv1d(128, 128, 1)
        self.bn_hrzn_1 = nn.BatchNorm1d(256)
        self.bn_hrzn_2 = nn.BatchNorm1d(128)
        self.bn_hrzn_3 = nn.BatchNorm1d(128)
        self.lin_hrzn_1 = nn.Linear(128, 128)
        self.lin_hrzn_2 = nn.Linear(128, 128)
        self.lin_hrzn_3 = nn.Linear(128, 32)
        self.lin_hrzn_abs = nn.Linear(32, 2)
        self.lin_hrzn_sgn = nn.Linear(32, 4)

    def forward(self, img, check=False, img_nhwc=None, keep=None):
        ops._C.require_cuda(img)
        ctx = L.Ctx(self.training)
        B = img.size(0)
        dev = img.device
        x = img_nhwc if img_nhwc is not None else ops.nchw_to_nhwc(img, 4)
        x = L.run_vgg(ctx, self.vgg.features, x)                        # (B,h,w,512)
        P = x.shape[1] * x.shape[2]
        M = B * P
        x = x.reshape(M, 512)
        for conv, bn in ((self.conv_hrzn_1, self.bn_hrzn_1), (self.conv_hrzn_2, self.bn_hrzn_2),
                         (self.conv_hrzn_3, self.bn_hrzn_3)):
            x = L.linear_rows(ctx, x, M, conv.in_channels, conv.weight, conv.bias, bn=bn, act=ACT_RELU)
        seg = torch.arange(0, M + 1, P, dtype=torch.int32, device=dev)
        if ctx.grad:
            x = FN.SegmentColMaxFn.apply(x, seg, B, 128)
        else:
            x, _ = ops.segment_colmax(x, x.shape[-1], 128, seg, B)
        for lin in (self.lin_hrzn_1, self.lin_hrzn_2, self.lin_hrzn_3):
            x = L.linear_rows(ctx, x, B, lin.in_features, lin.weight, lin.bias, act=ACT_RELU)
        sgn = L.linear_rows(ctx, x, B, 32, self.lin_hrzn_sgn.weight, self.lin_hrzn_sgn.bias)[:, :4]
        abs0 = L.linear_rows(ctx, x, B, 32, self.lin_hrzn_abs.weight, self.lin_hrzn_abs.bias)[:, :2]
        habs, h, h_T4 = pose.head_normal(abs0, sgn, (0., 1., 0.))                     # hnet.py:59-77 (z = 0)
        h_T = h_T4[:, :3, :3]
        rot_deg = torch.rad2deg(torch.atan2(h_T[:, 1, 0], h_T[:, 0, 0])).detach()      # torch_utils.py:245
        h_img, h_img_nhwc = ops.rotate_nearest_u8(img, rot_deg)
        if keep is not None:
            keep['h_img_nhwc'] = h_img_nhwc
            keep['rot_deg'] = rot_deg
        ret = {'h_hrzn_abs': habs, 'h_hrzn_sgn': sgn.contiguous(), 'h_hrzn': h, 'h_img': h_img, 'h_c': h_T,
               'intrinsic_sensor2': h_T, 'network': 'H'}
        ret['_h_img_nhwc'] = h_img_nhwc
        return ret
