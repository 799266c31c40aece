"""F-net: yaw by cross-modal correlation (reference nets/fnet.py) on the HIP path."""
import math
import os

import torch
import torch.nn as nn

from .. import ops
from ..common import pose
from . import fn as FN
from . import layers as L
from .builders import VGGFeatures, conv_bn_relu, convt_bn_relu


class Fnet(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.device = args['DEVICE']
        raw = args['raw_cam_img_size']
        self.range_img_size = (int(raw[0] / 2), int(raw[1] * 2))
        self.lidar_fov_rad = args['lidar_fov_rad']
        self.vgg_camera = VGGFeatures('C')
        self.vgg_5_1_camera = convt_bn_relu(512, 128, (3, 3), (2, 2), 1)
        self.vgg_5_2_camera = convt_bn_relu(128, 32, (3, 3), (2, 2), 0)
        self.vgg_5_3_camera = convt_bn_relu(32, 16, (3, 3), (2, 2), 1)
        self.conv_range = conv_bn_relu(4, 3, (1, 2), (1, 1), 0)
        self.vgg_range = VGGFeatures('C')
        self.vgg_5_1_range = convt_bn_relu(512, 128, (3, 3), (2, 2), 1)
        self.vgg_5_2_range = convt_bn_relu(128, 32, (3, 3), (2, 2), 0)
        self.vgg_5_3_range = convt_bn_relu(32, 16, (3, 3), (2, 2), 1)

    def _trunk(self, ctx, x, side):
        x = L.run_vgg(ctx, getattr(self, 'vgg_' + side).features, x)
        x = L.run_convt_bn_relu(ctx, getattr(self, 'vgg_5_1_' + side), x)
        x = L.run_convt_bn_relu(ctx, getattr(self, 'vgg_5_2_' + side), x)
        return L.run_convt_bn_relu(ctx, getattr(self, 'vgg_5_3_' + side), x)

    def forward(self, pc, ret, check=False, keep=None, cam_stream=None):
        """cam_stream: the stream H ran on (EFGHBackbone): the camera trunk is enqueued there, behind H, while the range trunk -
        which needs E's rotation only - runs on the current stream; joined before the correlation head"""
        ctx = L.Ctx(self.training)
        H, W = self.range_img_size
        fov = (self.lidar_fov_rad[0] * math.pi, self.lidar_fov_rad[1] * math.pi)
        if ctx.grad:
            e_range = FN.RangeImageFn.apply(pc, ret['e_l'], H, W, fov[0], fov[1])  # fnet.py:43-45
        else:
            e_range, _ = ops.range_image(pc, ret['e_l'], H, W, fov[0], fov[1])
        from . import efghbackbone as bb
        main = torch.cuda.current_stream() if pc.is_cuda else None

        def camera():
            h_img = ret.get('_h_img_nhwc')
            if h_img is None:
                h_img = ops.nchw_to_nhwc(ret['h_img'], 4)
            return self._trunk(ctx, h_img, 'camera')                              # (B,h,wc,16)

        def rng_branch():
            r0 = L.run_conv_bn_relu(ctx, self.conv_range, e_range)                # (B,H,W-1,4)
            return self._trunk(ctx, r0, 'range')                                  # (B,h,wr,16)
        if cam_stream is not None:
            rng = rng_branch()
            with torch.cuda.stream(cam_stream):
                cam = camera()
            main.wait_stream(cam_stream)
            cam.record_stream(main)
        else:
            cam = camera()
            rng = rng_branch()
        if ctx.grad:
            f_score, logit = FN.CorrHeadFn.apply(cam, rng), None                  # fnet.py:57-81
        else:
            f_score, logit = ops.corr_head(cam, rng, want_logit=keep is not None)
        f_T = pose.yaw_rotation_from_scores(f_score)                              # :87-91
        if keep is not None:
            keep.update({'e_range': e_range, 'cam3': cam, 'rng3': rng, 'f_logit': logit})
        ret = dict(ret)
        ret['f_score'] = f_score
        ret['f_l'] = f_T
        ret['sensor2_T_sensor1'] = pose.compose(f_T, ret['sensor2_T_sensor1'])       # :101
        ret['network'] = ret['network'] + 'F'
        return ret
