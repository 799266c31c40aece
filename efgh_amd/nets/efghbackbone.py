"""EFGHBackbone with the reference's module API (nets/efghbackbone.py:11-43): same constructor,
same forward(pc, img, calib, A, check) -> dict with the reference's 22 keys, same state_dict."""
import os

import torch
import torch.nn as nn

from .. import ops
from ..common import pose
from .enet import Enet
from .fnet import Fnet
from .gnet import Gnet
from .hnet import Hnet

__all__ = ['EFGHBackbone']

SIDE_STREAM = os.environ.get('EFGH_SIDE_STREAM', '1') != '0'
_SIDE = {}


def _side_stream(device):
    s = _SIDE.get(device.index)
    if s is None:
        s = _SIDE[device.index] = torch.cuda.Stream(device=device)
    return s


def _stamp_pose(state, key, calib, A):
    """`<key>_cam_T_velo` = A^-1 * intrinsic_sensor2 * A * calib * sensor2_T_sensor1 for the poses accumulated so far
    (common/torch_utils.py:256-269)"""
    state[key + '_cam_T_velo'] = pose.compute_cam_T_velo(state['intrinsic_sensor2'], state['sensor2_T_sensor1'], calib, A)
    return state


class EFGHBackbone(nn.Module):
    """The four stages are sub-modules named E, H, F, G (the checkpoint keys depend on that).  E (point cloud) and H (image) are
    independent; F refines the yaw from their outputs, G the translation; after each refinement the camera-from-LiDAR
    projection is re-derived and kept under its own key."""

    def __init__(self, args):
        super().__init__()
        for name, cls in (('E', Enet), ('H', Hnet), ('F', Fnet), ('G', Gnet)):
            setattr(self, name, cls(args))
        self.device = args['DEVICE']

    def forward(self, pc, img, calib, A, check=False, keep=None):
        ops._C.require_cuda(pc, img, calib, A)
        ops._C.require_f32(pc, img, calib, A)
        shared_img = ops.nchw_to_nhwc(img, 4)                # channels-last copy used by both H and G
        if SIDE_STREAM and pc.is_cuda:
            # E (point branch: ~450 small launches, many of them grids of a few workgroups) and H (image branch: full-chip MFMA
            # kernels) are independent: H is enqueued first on the current stream, E runs on a side stream underneath it - the
            # host-side read-back of the lattice sizes then waits for the side stream only.  Autograd runs each node's backward on
            # the stream of its forward, so the two branches overlap in backward as well.
            main = torch.cuda.current_stream()
            side = _side_stream(pc.device)
            side.wait_stream(main)
            pc.record_stream(side)
            image_part = self.H(img, check, img_nhwc=shared_img, keep=keep)
            with torch.cuda.stream(side):
                point_part = self.E(pc, check, keep=keep)
            main.wait_stream(side)
            for v in point_part.values():
                if torch.is_tensor(v):
                    v.record_stream(main)
        else:
            point_part = self.E(pc, check, keep=keep)
            image_part = self.H(img, check, img_nhwc=shared_img, keep=keep)
        state = {**point_part, **image_part, 'network': point_part['network'] + image_part['network']}
        _stamp_pose(state, 'eh', calib, A)
        state = _stamp_pose(self.F(pc, state, check, keep=keep), 'efh', calib, A)
        state = _stamp_pose(self.G(pc, img, state, check, img_nhwc=shared_img, keep=keep), 'efgh', calib, A)
        state['cam_T_velo'] = state['efgh_cam_T_velo']
        state.pop('_h_img_nhwc', None)
        return state
