"""EFGHBackbone with the reference's module API (nets/efghbackbone.py:11-43): same constructor,
same forward(pc, img, calib, A, check) -> dict with the reference's 22 keys, same state_dict."""
import torch.nn as nn

from .. import ops
from ..common import pose
from .enet import Enet
from .fnet import Fnet
from .gnet import Gnet
from .hnet import Hnet

__all__ = ['EFGHBackbone']


class EFGHBackbone(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.E = Enet(args)
        self.H = Hnet(args)
        self.F = Fnet(args)
        self.G = Gnet(args)
        self.device = args['DEVICE']

    def forward(self, pc, img, calib, A, check=False, keep=None):
        ops._C.require_cuda(pc, img, calib, A)
        ops._C.require_f32(pc, img, calib, A)
        img_nhwc = ops.nchw_to_nhwc(img, 4)             # shared by H and G
        rete = self.E(pc, check, keep=keep)
        reth = self.H(img, check, img_nhwc=img_nhwc, keep=keep)
        ret = {}
        ret.update(rete)
        ret.update(reth)
        ret['network'] = rete['network'] + reth['network']
        ret['eh_cam_T_velo'] = pose.compute_cam_T_velo(ret['intrinsic_sensor2'], ret['sensor2_T_sensor1'], calib, A)
        ret = self.F(pc, ret, check, keep=keep)
        ret['efh_cam_T_velo'] = pose.compute_cam_T_velo(ret['intrinsic_sensor2'], ret['sensor2_T_sensor1'], calib, A)
        ret = self.G(pc, img, ret, check, img_nhwc=img_nhwc, keep=keep)
        ret['efgh_cam_T_velo'] = pose.compute_cam_T_velo(ret['intrinsic_sensor2'], ret['sensor2_T_sensor1'], calib, A)
        ret['cam_T_velo'] = ret['efgh_cam_T_velo']
        ret.pop('_h_img_nhwc', None)
        return ret
