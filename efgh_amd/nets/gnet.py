"""G-net: translation + dense depth / mask (reference nets/gnet.py) on the HIP path."""
import torch
import torch.nn as nn

from .. import ops
from ..common import pose
from . import fn as FN
from . import layers as L
from .builders import conv_bn_relu, convt_bn_relu, resnet18_layers

import os as _os
HEADS_FUSED = True    # depth + mask heads as one 3-channel pipeline (layers.run_convt_heads)


class Gnet(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.device = args['DEVICE']
        self.raw_cam_img_size = args['raw_cam_img_size']
        self.conv_i0 = conv_bn_relu(3, 64, 3, 1, 1)
        self.conv_img2, self.conv_img3, self.conv_img4, self.conv_img5 = resnet18_layers()
        self.convt_img4 = convt_bn_relu(512, 256, 3, 2, 1, 1)
        self.convt_img3 = convt_bn_relu(512, 128, 3, 2, 1, 1)
        self.convt_img2 = convt_bn_relu(256, 64, 3, 2, 1, 1)
        self.convt_dimg = convt_bn_relu(128, 1, 3, 2, 1, 1)
        self.convt_mask = convt_bn_relu(128, 2, 3, 2, 1, 1)
        self.conv_i1 = conv_bn_relu(64, 32, 1, 1, 0)
        self.conv_d1 = conv_bn_relu(4, 32, 3, 2, 1)
        self.conv2, self.conv3, self.conv4, self.conv5 = resnet18_layers()
        self.conv_trs_1 = conv_bn_relu(512, 512, 1)
        self.conv_trs_2 = conv_bn_relu(512, 512, 1)
        self.conv_trs_3 = conv_bn_relu(512, 512, 1)
        self.conv_trs_4 = nn.Conv1d(512, 3, 1)

    def image_part(self, img, img_nhwc=None):
        """everything of G that depends on the camera image only (encoder, decoder, the depth / mask heads, conv_i1): the backbone
        runs it on a side stream next to F, which it does not depend on (gnet.py:103-134)"""
        ctx = L.Ctx(self.training)
        dev = img.device
        x = img_nhwc if img_nhwc is not None else ops.nchw_to_nhwc(img, 4)
        B, H, W, _ = x.shape

        def buf(h, w, c):
            return torch.empty((B, h, w, c), dtype=torch.float32, device=dev)

        def tgt(b, off):
            return (b, off)

        def cat(b, parts):
            """torch.cat / concat_tensors of gnet.py:117-121: channel slices of one buffer the producers wrote into; on the
            training path an autograd node over the same buffer (FN.concat: no copy when every part already lives in its slice)"""
            return FN.concat(b, parts) if ctx.grad else b
        # widths must halve exactly three times (the reference's torch.cat fails otherwise); heights only need
        # H even: a stride-2 layer gives ceil(h/2) rows, the transposed convolution 2*ceil(h/2), and concat_tensors
        # (common/torch_utils.py:309-319) crops the surplus row: p1 = int((2*ceil(h/2) - h) / 2) = 0, i.e. the first h rows
        assert H % 2 == 0 and W % 8 == 0, 'reference needs raw H % 4 == 0 and (W/2) % 8 == 0 (SURVEY 8a-17)'
        h2, h3 = (H + 1) // 2, ((H + 1) // 2 + 1) // 2

        def crop(t, h, b, off):
            """concat_tensors' crop of a decoder output to the skip connection's height"""
            if t.shape[1] == h:
                return t
            t = t[:, :h]
            if not ctx.grad:
                b[..., off:off + t.shape[-1]].copy_(t)
            return t
        cat0_box = []

        def cat0_early():
            if not cat0_box:
                cat0_box.append(buf(H, W, 64))
            return cat0_box[0]
        cat3 = buf(h3, W // 4, 512)           # [conv_img4 | convt_img4]
        cat2 = buf(h2, W // 2, 256)           # [conv_img3 | convt_img3]
        cat1 = buf(H, W, 128)                 # [convt_img2 | conv_img2]
        c1 = L.run_conv_bn_relu(ctx, self.conv_i0, x)                              # gnet.py:103
        c2 = L.run_resnet_layer(ctx, self.conv_img2, c1, out=tgt(cat1, 64))
        if ctx.grad:
            # c2 / c3 / c4 each feed the next encoder layer AND a decoder concatenation: the concatenation takes the alias the
            # encoder layer hands out, so the two gradients meet in a dgrad epilogue (L.run_basic_block)
            c3, c2 = L.run_resnet_layer(ctx, self.conv_img3, c2, out=tgt(cat2, 0), alias_in=True)
            c4, c3 = L.run_resnet_layer(ctx, self.conv_img4, c3, out=tgt(cat3, 0), alias_in=True)
            c5, c4 = L.run_resnet_layer(ctx, self.conv_img5, c4, alias_in=True)
        else:
            c3 = _layer_from_slice(ctx, self.conv_img3, cat1, 64, 64, out=(cat2, 0))
            c4 = _layer_from_slice(ctx, self.conv_img4, cat2, 0, 128, out=(cat3, 0))
            c5 = _layer_from_slice(ctx, self.conv_img5, cat3, 0, 256, out=None)
        t4 = L.run_convt_bn_relu(ctx, self.convt_img4, c5, out=tgt(cat3, 256) if 2 * c5.shape[1] == h3 else None)     # :116
        t4 = crop(t4, h3, cat3, 256)
        t3 = L.run_convt_bn_relu(ctx, self.convt_img3, cat(cat3, [c4, t4]), out=tgt(cat2, 128) if 2 * h3 == h2 else None)
        t3 = crop(t3, h2, cat2, 128)
        t2 = L.run_convt_bn_relu(ctx, self.convt_img2, cat(cat2, [c3, t3]), out=tgt(cat1, 0))
        ci1 = None
        if ctx.grad:                          # t2 feeds conv_i1 (below) and the concatenation: same chaining
            ci1, t2 = L.run_conv_bn_relu(ctx, self.conv_i1, t2, out=tgt(cat0_early(), 0), skip_out=True)
        else:
            L.run_conv_bn_relu(ctx, self.conv_i1, cat1, out=(cat0_early(), 0), in_ch=(0, 64))
        # everything the rest of G needs from this part exists now; the two heads below can run next to it
        ready = None
        if x.is_cuda:
            ready = torch.cuda.Event()
            ready.record()
        cv = cat(cat1, [t2, c2])
        cat0 = cat0_early()                   # [conv_i1 | conv_d1]
        if HEADS_FUSED and x.is_cuda and L.convt_heads_fusable(self.convt_dimg, self.convt_mask):
            # both heads as one 3-channel pipeline (layers.run_convt_heads): channel 0 depth, channels 1-2 mask logits
            hm = L.run_convt_heads(ctx, self.convt_dimg, self.convt_mask, cv)      # (B,2H,2W,4)
            g_depth, g_mask = FN.HeadsToNchwFn.apply(hm) if ctx.grad else ops.heads_to_nchw(hm)
        else:
            dimg, cv = L.run_convt_bn_relu(ctx, self.convt_dimg, cv, skip_out=True)    # (B,2H,2W,4) ch0
            mask = L.run_convt_bn_relu(ctx, self.convt_mask, cv)                       # (B,2H,2W,4) ch0,1
            if ctx.grad:
                g_depth = FN.NhwcToNchwFn.apply(dimg, 1)
                g_mask = FN.Softmax2ToNchwFn.apply(mask)
            else:
                g_depth = ops.nhwc_to_nchw(dimg, 1)
                g_mask = ops.softmax2_to_nchw(mask)
        return {'g_depth': g_depth, 'g_mask': g_mask, 'ci1': ci1, 'cat0': cat0, 'ready': ready}

    def forward(self, pc, img, ret, check=False, img_nhwc=None, keep=None, pre=None):
        """pre: the result of image_part() when the caller has already run it (EFGHBackbone overlaps it with F)"""
        ctx = L.Ctx(self.training)
        if pre is None:
            pre = self.image_part(img, img_nhwc)
        g_depth, g_mask, ci1, cat0 = pre['g_depth'], pre['g_mask'], pre['ci1'], pre['cat0']
        B = cat0.shape[0]

        def tgt(b, off):
            return (b, off)
        rawH, rawW = self.raw_cam_img_size
        if ctx.grad:
            f_depth = FN.DepthImageFn.apply(pc, ret['efh_cam_T_velo'], rawH, rawW)  # :136
        else:
            f_depth, _ = ops.depth_image(pc, ret['efh_cam_T_velo'], rawH, rawW)
        if ctx.grad:
            cd1 = L.run_conv_bn_relu(ctx, self.conv_d1, f_depth, out=tgt(cat0, 32))
            y = FN.concat(cat0, [ci1, cd1])
        else:
            L.run_conv_bn_relu(ctx, self.conv_d1, f_depth, out=(cat0, 32))
            y = cat0
        y = L.run_resnet_layer(ctx, self.conv2, y)
        y = L.run_resnet_layer(ctx, self.conv3, y)
        y = L.run_resnet_layer(ctx, self.conv4, y)
        y = L.run_resnet_layer(ctx, self.conv5, y)
        for seq in (self.conv_trs_1, self.conv_trs_2, self.conv_trs_3):
            y = L.run_conv_bn_relu(ctx, seq, y)
        P = y.shape[1] * y.shape[2]
        t4r = L.linear_rows(ctx, y.reshape(B * P, 512), B * P, 512, self.conv_trs_4.weight, self.conv_trs_4.bias)
        if ctx.grad:
            trs = FN.SegmentColMeanFn.apply(t4r, P, B, 3)[:, :, None]              # :165
        else:
            trs = ops.segment_colmean(t4r, t4r.shape[-1], 3, P, B)[:, :, None]
        g_T = pose.translation_matrix(trs)
        if keep is not None:
            keep.update({'f_depth': f_depth})
        ret = dict(ret)
        ret.update({'g_depth': g_depth, 'g_mask': g_mask, 'g_trs': trs, 'g_l': g_T})
        ret['sensor2_T_sensor1'] = pose.compose(g_T, ret['sensor2_T_sensor1'])        # :180
        ret['network'] = ret['network'] + 'G'
        return ret


def _layer_from_slice(ctx, layer, src, coff, C, out):
    """run a resnet layer whose input is the channel slice [coff, coff+C) of a concat buffer"""
    blocks = list(layer.children())
    blk = blocks[0]
    y = L.conv2d(ctx, src, blk.conv1, blk.bn1, L.ACT_RELU, in_ch=(coff, C))
    idt = L.conv2d(ctx, src, blk.downsample[0], blk.downsample[1], L.ACT_NONE, in_ch=(coff, C))
    x = L.conv2d(ctx, y, blk.conv2, blk.bn2, L.ACT_RELU, residual=idt)
    return L.run_basic_block(ctx, blocks[1], x, out=out)
