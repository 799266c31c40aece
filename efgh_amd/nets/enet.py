"""E-net: ground normal from the point cloud (reference nets/enet.py) on the HIP path."""
import torch
import torch.nn as nn

from .. import lattice, ops
from ..common import pose
from ..ops import ACT_LEAKY, ACT_RELU
from . import fn as FN
from . import layers as L
from .builders import BilateralConvFlex, conv_1x1


class Enet(nn.Module):
    def __init__(self, args):
        super().__init__()
        dim = args['dim']
        self.scale_map = args['scale_map']
        assert dim == 3 and all(int(r) == 1 for _, r in self.scale_map), 'd=3, radius-1 BCL only'
        # the reference's E-net switches (enet.py:25-83 -> net_utils.py:6-11, bilateralNN.py:121-135,196): the shipped configurations
        # set use_leaky / bcn_use_norm and clear last_relu (configs/train_rellis.yaml:8-12); the other values are honoured as well
        self.use_leaky, self.use_norm, self.last_relu = bool(args['use_leaky']), bool(args['bcn_use_norm']), bool(args['last_relu'])
        if not args.get('bcn_use_bias', True):
            # (use_bias only adds a parameter behind the SLICE step, which enet.py:37-81 never enables: do_slice=False)
            pass
        self.device = args['DEVICE']
        self.conv_in = nn.Sequential(conv_1x1(dim, 32, True), conv_1x1(32, 32, True), conv_1x1(32, 32, True))
        self.bcn1 = BilateralConvFlex(32 + dim + 1, [32, 32])          # enet.py:30-83
        self.bcn2 = BilateralConvFlex(32 + dim + 1, [64, 64])
        self.bcn3 = BilateralConvFlex(64 + dim + 1, [128, 128])
        self.bcn4 = BilateralConvFlex(128 + dim + 1, [256, 256])
        self.bcn5 = BilateralConvFlex(256 + dim + 1, [256, 256])
        self.conv_gn_1 = nn.Conv1d(256, 128, 1)
        self.conv_gn_2 = nn.Conv1d(128, 128, 1)
        self.conv_gn_3 = nn.Conv1d(128, 128, 1)
        self.bn_gn_1 = nn.BatchNorm1d(128)
        self.bn_gn_2 = nn.BatchNorm1d(128)
        self.bn_gn_3 = nn.BatchNorm1d(128)
        self.lin_gn_1 = nn.Linear(128, 128)
        self.lin_gn_2 = nn.Linear(128, 128)
        self.lin_gn_3 = nn.Linear(128, 32)
        self.lin_gn_abs = nn.Linear(32, 3)
        self.lin_gn_sgn = nn.Linear(32, 8)

    def forward(self, pc, check=False, keep=None):
        """pc (B,3,N) -> dict as reference enet.py:179-187 (every sample gets its own lattice)."""
        ops._C.require_cuda(pc)
        ctx = L.Ctx(self.training)
        B, _, N = pc.shape
        dev = pc.device
        bcns = [self.bcn1, self.bcn2, self.bcn3, self.bcn4, self.bcn5]
        scales = [s for s, _ in self.scale_map]
        cins = [m.num_input for m in bcns]
        # all samples in one launch sequence per level, all five levels enqueued before the one read-back of their sizes
        # (every sample keeps its own lattice)
        lv = lattice.build_pyramid_batched(pc, scales, need_off=ctx.grad or keep is not None)      # (`off` serves the splat's backward only)
        if keep is not None:
            keep['lattice'] = lv
        # conv_in on [B*N][4] (x,y,z,0)
        x = ops.nchw_to_nhwc(pc, 4).view(B * N, 4)
        for i in range(3):
            conv = self.conv_in[i][0]
            x = L.linear_rows(ctx, x, B * N, conv.in_channels, conv.weight, conv.bias, act=ACT_LEAKY if self.use_leaky else ACT_RELU,
                              slope=0.1 if self.use_leaky else 0.0)                 # net_utils.py:11: LEAKY_RATE 0.1
        # level-l input rows = [el_minus_gr (4, in the lattice's own array) | previous features]: read in place by the splat
        cur = x
        for l in range(5):
            d = lv[l]
            cf = cins[l] - 4
            if ctx.grad:
                splat = FN.SplatFn.apply(cur, d, cf, True, self.use_norm)
            else:
                splat, _ = ops.splat_fwd(d, cur, cf, normalize=self.use_norm)
            # last_relu (bilateralNN.py:121-135): an activation behind the second convolution - LeakyReLU(0.1) / ReLU by use_leaky
            la = (ACT_LEAKY if self.use_leaky else ACT_RELU) if self.last_relu else ops.ACT_NONE
            cur = L.blur_conv(ctx, splat, d.H, cins[l], d, bcns[l].blur_conv[0], bcns[l].blur_conv[2], last_act=la,
                              last_slope=0.1 if (self.last_relu and self.use_leaky) else 0.0)
        x = cur                                                          # (sum_b H5_b, 256)
        segs = lv[4].seg
        M = x.shape[0]
        for conv, bn in ((self.conv_gn_1, self.bn_gn_1), (self.conv_gn_2, self.bn_gn_2),
                         (self.conv_gn_3, self.bn_gn_3)):
            x = L.linear_rows(ctx, x, M, conv.in_channels, conv.weight, conv.bias, bn=bn, act=ACT_RELU)
        seg = torch.tensor(segs, dtype=torch.int32).pin_memory().to(dev, non_blocking=True)      # (no stream synchronisation)
        if ctx.grad:
            x = FN.SegmentColMaxFn.apply(x, seg, B, 128)
        else:
            x, _ = ops.segment_colmax(x, x.shape[-1], 128, seg, B)       # torch.max over vertices (:154)
        for lin in (self.lin_gn_1, self.lin_gn_2, self.lin_gn_3):
            x = L.linear_rows(ctx, x, B, lin.in_features, lin.weight, lin.bias, act=ACT_RELU)
        gn_sgn = L.linear_rows(ctx, x, B, 32, self.lin_gn_sgn.weight, self.lin_gn_sgn.bias)[:, :8]
        gn_abs0 = L.linear_rows(ctx, x, B, 32, self.lin_gn_abs.weight, self.lin_gn_abs.bias)[:, :3]
        gn_abs, e_gn, e_T = pose.head_normal(gn_abs0, gn_sgn, (0., 0., 1.))          # enet.py:161-176
        return {'e_gn_abs': gn_abs, 'e_gn_sgn': gn_sgn.contiguous(), 'e_gn': e_gn, 'e_l': e_T,
                'sensor2_T_sensor1': e_T, 'network': 'E'}
