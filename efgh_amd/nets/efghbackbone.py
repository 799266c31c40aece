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


G_SIDE = True         # G's image part next to H / F (needs SIDE_STREAM)
F_SIDE = True         # H and F's camera trunk on a side stream, E and F's range trunk on the current one


def _side_stream(device, i=0):
    s = _SIDE.get((device.index, i))
    if s is None:
        with ops._LOCK:                      # (two threads may ask for the same stream first at the same time)
            s = _SIDE.get((device.index, i))
            if s is None:
                # (default priority: a raised priority for G's stream cost 4.5 ms per step)
                s = _SIDE[(device.index, i)] = torch.cuda.Stream(device=device)
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

    def _epoch_holders(self):
        """the distinct content-epoch owners (ops.Epoch) among this model's parameters; recomputed only when some parameters
        changed owner (a train.FlatParams was built: ops.HOLDER_GEN) - walking 353 parameters costs a millisecond per forward"""
        ent = self.__dict__.get('_efgh_holders')
        if ent is None or ent[0] != ops.HOLDER_GEN[0]:
            seen, out = set(), []
            for p in self.parameters():
                h = ops.epoch_of(p)
                if id(h) not in seen:
                    seen.add(id(h))
                    out.append(h)
            ent = self.__dict__['_efgh_holders'] = (ops.HOLDER_GEN[0], out)
        return ent[1]

    def forward(self, pc, img, calib, A, check=False, keep=None):
        # BatchNorm's `num_batches_tracked += 1` of every layer that runs in training mode: one multi-tensor add per forward
        own = self.training and ops.nbt_collect()
        try:
            return self._forward(pc, img, calib, A, check, keep)
        finally:
            if own:
                ops.nbt_flush()

    def _forward(self, pc, img, calib, A, check=False, keep=None):
        if getattr(self, '_is_replica', False):
            # torch.nn.DataParallel over more than one device (main.py:127 with several GPUs visible): its replicas are shallow
            # per-forward copies whose parameters are broadcast outputs, driven by one Python thread per device.  This path is
            # one process per GPU: refuse loudly instead of running a slow, half-supported schedule silently
            raise ops._C.EfghError(
                'EFGHBackbone was entered from a torch.nn.DataParallel replica (more than one device in device_ids).  This '
                'framework runs ONE PROCESS PER GPU: start the unchanged entry script with `python -m efgh_amd.run main.py '
                '<config.yaml>` (pins the process to one device), or '
                'restrict the process to one device (HIP_VISIBLE_DEVICES=0 / DataParallel(model, device_ids=[0])).  '
                'See INTEGRATION.md section 4.')
        ops._C.require_cuda(pc, img, calib, A)
        ops._C.require_f32(pc, img, calib, A)
        # packed weights that went stale with the last optimizer step are rewritten in place by ONE launch; every branch below
        # reads them, so that launch goes out here, on the current stream, before the streams fork
        # (every distinct owner: with frozen sub-networks the frozen parameters stay with GLOBAL_EPOCH while the trainable ones carry
        # their FlatParams' epoch - the first parameter alone would name only one of them)
        for holder in self._epoch_holders():
            ops.repack_stale(pc.device, holder)
        ops.TLS.train_step = bool(self.training and torch.is_grad_enabled())   # (GemmLayerFn carries it over to its backward)
        if self.training:
            ops.w2v_clear()
        shared_img = ops.nchw_to_nhwc(img, 4)                # channels-last copy used by both H and G
        g_pre = None
        s_h = None
        if SIDE_STREAM and pc.is_cuda:
            # Three independent pieces of work start here: H (image branch), E (point branch: ~450 small launches) and the part of G
            # that only needs the camera image (encoder, decoder, depth / mask heads).  H and G's image part are enqueued on side
            # streams first; E runs on the current stream (its host-side read-back of the lattice sizes then only waits for E), and
            # F's range trunk - which needs nothing but E's rotation - follows it there, while F's camera trunk follows H on H's
            # stream.  Kernels of one branch fill the tails and the HBM-bound phases of the others; autograd runs each node's
            # backward on the stream of its forward, so the branches overlap in backward as well.
            main = torch.cuda.current_stream()
            if F_SIDE:
                s_h = _side_stream(pc.device, 0)
                s_h.wait_stream(main)
                shared_img.record_stream(s_h)
                img.record_stream(s_h)
                with torch.cuda.stream(s_h):
                    image_part = self.H(img, check, img_nhwc=shared_img, keep=keep)
            else:
                image_part = self.H(img, check, img_nhwc=shared_img, keep=keep)
            if G_SIDE:
                s_g = _side_stream(pc.device, 1)
                s_g.wait_stream(main)
                shared_img.record_stream(s_g)
                img.record_stream(s_g)
                with torch.cuda.stream(s_g):
                    g_pre = self.G.image_part(img, shared_img)
            point_part = self.E(pc, check, keep=keep)
        else:
            point_part = self.E(pc, check, keep=keep)
            image_part = self.H(img, check, img_nhwc=shared_img, keep=keep)
        state = {**point_part, **image_part, 'network': point_part['network'] + image_part['network']}
        if s_h is None:
            _stamp_pose(state, 'eh', calib, A)
            state = self.F(pc, state, check, keep=keep)
        else:
            state['eh_cam_T_velo'] = None                     # (keeps the reference's key order; filled in after the join below)
            pre_f = state
            state = self.F(pc, state, check, keep=keep, cam_stream=s_h)      # joins s_h before the correlation head
            for v in image_part.values():
                if torch.is_tensor(v):
                    v.record_stream(torch.cuda.current_stream())
            state['eh_cam_T_velo'] = _stamp_pose(dict(pre_f), 'eh', calib, A)['eh_cam_T_velo']
        state = _stamp_pose(state, 'efh', calib, A)
        if g_pre is not None:
            # the rest of G needs conv_i1's output only; the depth / mask heads keep running on their stream next to it
            torch.cuda.current_stream().wait_event(g_pre['ready'])
            for k in ('ci1', 'cat0'):
                if torch.is_tensor(g_pre[k]):
                    g_pre[k].record_stream(torch.cuda.current_stream())
        state = _stamp_pose(self.G(pc, img, state, check, img_nhwc=shared_img, keep=keep, pre=g_pre), 'efgh', calib, A)
        if g_pre is not None:
            torch.cuda.current_stream().wait_stream(_side_stream(pc.device, 1))
            for k in ('g_depth', 'g_mask'):
                g_pre[k].record_stream(torch.cuda.current_stream())
        state['cam_T_velo'] = state['efgh_cam_T_velo']
        state.pop('_h_img_nhwc', None)
        return state
