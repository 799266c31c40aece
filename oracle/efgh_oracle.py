"""TEST INFRASTRUCTURE ONLY — CPU (torch fp32) restatement of the reference EFGHNet forward,
loss and (through torch autograd) backward, written functionally over a flat ``state_dict``.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  It is the
checker the HIP path is compared with on the GPU box (where /root/reference does not exist);
it is itself pinned against the reference by tests/test_oracle_e2e.py using the golden outputs
in tests/golden/e2e_small.npz, rotate_cases.npz and raster_cases.npz (all produced by the
unmodified reference, tests/golden/make_golden.py).

Every function cites the reference lines it restates (paths relative to /root/reference).
Per-sample python loops of the reference are kept as loops over B; semantics for B>1 are
"B independent B=1 evaluations" (SURVEY.md §8a-0).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import lattice as _lattice

PI = math.pi


# =========================================================================================
# small building blocks
# =========================================================================================
def _bn(P, pre, x, train, momentum=0.1, eps=1e-5):
    """nn.BatchNorm{1,2}d; running statistics in P are updated in place when train."""
    if train and (pre + '.num_batches_tracked') in P:
        P[pre + '.num_batches_tracked'] += 1
    return F.batch_norm(x, P[pre + '.running_mean'], P[pre + '.running_var'],
                        P[pre + '.weight'], P[pre + '.bias'], train, momentum, eps)


def conv_bn_relu(P, pre, x, train, stride=1, padding=0):
    """nets/net_utils.py:45-64 — Conv2d(no bias) + BN + LeakyReLU(0.2)."""
    x = F.conv2d(x, P[pre + '.0.weight'], None, stride, padding)
    return F.leaky_relu(_bn(P, pre + '.1', x, train), 0.2)


def convt_bn_relu(P, pre, x, train, padding, output_padding=0):
    """nets/net_utils.py:66-98 — ConvT(k3,s2)+BN+LeakyReLU(0.2) + Conv3x3+BN+LeakyReLU(0.2)."""
    x = F.conv_transpose2d(x, P[pre + '.0.weight'], None, 2, padding, output_padding)
    x = F.leaky_relu(_bn(P, pre + '.1', x, train), 0.2)
    x = F.conv2d(x, P[pre + '.3.weight'], None, 1, 1)
    return F.leaky_relu(_bn(P, pre + '.4', x, train), 0.2)


VGG_A = [64, 'M', 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M']   # nets/vgg.py:87
VGG_C = [64, 'M', 128, 'M', 256, 256, 'M', 512, 512, 'M']                  # nets/vgg.py:89


def vgg_features(P, pre, x, cfg, train):
    """nets/vgg.py:69-83 make_layers(batch_norm=True): conv3x3(bias)+BN+ReLU, 'M' = maxpool2."""
    i = 0
    for v in cfg:
        if v == 'M':
            x = F.max_pool2d(x, 2, 2)
            i += 1
        else:
            x = F.conv2d(x, P[f'{pre}.{i}.weight'], P[f'{pre}.{i}.bias'], 1, 1)
            x = F.relu(_bn(P, f'{pre}.{i + 1}', x, train))
            i += 3
    return x


def basic_block(P, pre, x, train, stride):
    """nets/resnet.py:55-71."""
    out = F.conv2d(x, P[pre + '.conv1.weight'], None, stride, 1)
    out = F.relu(_bn(P, pre + '.bn1', out, train))
    out = F.conv2d(out, P[pre + '.conv2.weight'], None, 1, 1)
    out = _bn(P, pre + '.bn2', out, train)
    if (pre + '.downsample.0.weight') in P:
        idt = F.conv2d(x, P[pre + '.downsample.0.weight'], None, stride, 0)
        idt = _bn(P, pre + '.downsample.1', idt, train)
    else:
        idt = x
    return F.relu(out + idt)


def resnet_layer(P, pre, x, train, stride):
    """nets/resnet.py:171-193 _make_layer with 2 BasicBlocks (resnet18)."""
    x = basic_block(P, pre + '.0', x, train, stride)
    return basic_block(P, pre + '.1', x, train, 1)


# =========================================================================================
# geometry helpers (common/torch_utils.py)
# =========================================================================================
def normal_from_abs_sign(abs_, sign, ndim):
    """torch_utils.py:105-146: class = argmax (first max), bits MSB-first, 0 -> -1."""
    out = []
    for b in range(abs_.size(0)):
        cls = int(torch.argmax(torch.softmax(sign[b], 0)).item())
        bits = [(cls >> (ndim - 1 - i)) & 1 for i in range(ndim)]
        sgn = torch.tensor([1.0 if v else -1.0 for v in bits])
        out.append((abs_[b, :, 0] * sgn)[None])
    return torch.cat(out, 0)[..., None]


def rotation_between(srce, dest):
    """torch_utils.py:170-200 Rodrigues; K from detached scalars, (1-c)/s^2 attached."""
    mats = []
    for b in range(srce.size(0)):
        v1, v2 = srce[b, :, 0], dest[0, :, 0]
        v = torch.linalg.cross(v1, v2)
        c = torch.dot(v1, v2)
        s = torch.sqrt(torch.sum(v ** 2))
        vd = v.detach()
        K = torch.tensor([[0, -vd[2], vd[1]], [vd[2], 0, -vd[0]], [-vd[1], vd[0], 0]])
        if (1 - c) == 0:
            R = torch.eye(4)
        elif (1 + c) == 0:
            R = -torch.eye(4)
            if v1[0].item() == 0.0 and v2[0].item() == 0.0:
                R[0, 0] = 1
            elif v1[2].item() == 0.0 and v2[2].item() == 0.0:
                R[2, 2] = 1
        else:
            rot3 = torch.eye(3) + K + torch.mm(K, K) * ((1 - c) / (s ** 2))
            R = torch.eye(4)
            R[:3, :3] = rot3
        mats.append(R[None])
    return torch.cat(mats, 0)


def translation_matrix(vec):
    """torch_utils.py:220-233 (built with torch.tensor([...]) -> detached)."""
    out = []
    for b in range(vec.size(0)):
        t = torch.eye(4, dtype=vec.dtype)          # (follows the input: float32 on the parity path, float64 in the conditioning checks)
        t[:3, 3] = vec[b, :3, 0].detach()
        out.append(t[None])
    return torch.cat(out, 0)


def compute_cam_T_velo(c_T, l_T, calib, A):
    """torch_utils.py:256-269: A^-1 · c_T · A · calib · l_T."""
    m = torch.bmm(calib, l_T)
    m = torch.bmm(A, m)
    m = torch.bmm(c_T, m)
    return torch.bmm(torch.inverse(A), m)


def pil_rotate_nearest_u8(img_u8, angle_f32):
    """PIL.Image.rotate(angle) NEAREST / no expand / zero fill, restated in integers
    (call site torch_utils.py:250; algorithm: SURVEY.md §8a-12).  img_u8: (H,W,C) uint8;
    angle_f32: np.float32 degrees exactly as torch_utils.py:245 produces it."""
    h, w = img_u8.shape[:2]
    # PIL: angle = angle % 360.0 evaluated on the float32 0-dim tensor -> python-style fmod in f32
    ang32 = np.float32(angle_f32)
    m = np.fmod(ang32, np.float32(360.0))
    if m != 0 and (m < 0):
        m = np.float32(m + np.float32(360.0))
    angle = float(np.float32(m))
    if angle == 0.0:
        return img_u8.copy()
    a = -math.radians(angle)
    m0, m1 = round(math.cos(a), 15), round(math.sin(a), 15)
    m3, m4 = round(-math.sin(a), 15), round(math.cos(a), 15)
    cx, cy = w / 2.0, h / 2.0
    m2 = m0 * (-cx) + m1 * (-cy) + cx
    m5 = m3 * (-cx) + m4 * (-cy) + cy

    def fix(v):
        return int(math.floor(v * 65536.0 + 0.5))

    a0, a1, a3, a4 = fix(m0), fix(m1), fix(m3), fix(m4)
    a2 = fix(m2 + m0 * 0.5 + m1 * 0.5)
    a5 = fix(m5 + m3 * 0.5 + m4 * 0.5)
    xs = np.arange(w, dtype=np.int64)[None, :]
    ys = np.arange(h, dtype=np.int64)[:, None]
    xin = (a2 + a0 * xs + a1 * ys) >> 16
    yin = (a5 + a3 * xs + a4 * ys) >> 16
    ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    out = np.zeros_like(img_u8)
    out[ok] = img_u8[yin[ok], xin[ok]]
    return out


def rotate_image(img, mat):
    """torch_utils.py:235-254 (non-differentiable: goes through uint8 / PIL)."""
    outs = []
    for b in range(img.size(0)):
        rot_deg = torch.rad2deg(torch.atan2(mat[b, 1, 0], mat[b, 0, 0])).detach()
        u8 = np.array(img[b].detach().numpy().transpose(1, 2, 0), dtype='uint8')
        r = pil_rotate_nearest_u8(u8, np.float32(rot_deg.item()))
        outs.append(torch.from_numpy(r.transpose(2, 0, 1).copy())[None])
    return torch.cat(outs, 0).float()


class _IndexPutLastWins(torch.autograd.Function):
    """`img[u, v] = values` (torch_utils.py:53, :93-96) as torch executes it single-threaded:
    forward — duplicate pixels keep the LAST (largest index) point; backward — index_put's
    autograd formula hands grad_img[u_i, v_i] to EVERY point i, overwritten ones included."""

    @staticmethod
    def forward(ctx, values, u, v, h, w):
        n = u.numel()
        img = torch.zeros((h, w, values.size(-1)))
        if n:
            pix = u * w + v
            win = torch.full((h * w,), -1, dtype=torch.long)
            win = win.scatter_reduce(0, pix, torch.arange(n), 'amax', include_self=True)
            sel = win[win >= 0]
            img[u[sel], v[sel]] = values[sel]
        ctx.save_for_backward(u, v)
        return img

    @staticmethod
    def backward(ctx, g):
        u, v = ctx.saved_tensors
        return g[u, v], None, None, None, None


def _last_wins_scatter(h, w, u, v, values):
    return _IndexPutLastWins.apply(values, u, v, h, w)


def range_image(pc4, size, fov):
    """torch_utils.py:11-59.  pc4: (B,4,N) homogeneous (the w=1 row enters the norm, :29)."""
    fov_up, fov_down = fov[0] * PI, fov[1] * PI
    xyz = pc4.float()
    x_, y_, z_ = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    r_ = torch.sqrt(torch.sum(xyz ** 2, 1))
    pitch_ = torch.asin(z_ / r_)
    yaw_ = torch.atan2(y_, x_)
    imgs = []
    for b in range(pc4.size(0)):
        mask = (pitch_[b] < fov_up) & (pitch_[b] > fov_down)
        vals = torch.stack([x_[b][mask], y_[b][mask], z_[b][mask], r_[b][mask]], 1)
        u = ((fov_up - pitch_[b][mask]) / (fov_up - fov_down)) * (size[0] - 1)
        v = ((-yaw_[b][mask] + PI) / (2 * PI)) * (size[1] - 1)
        img = _last_wins_scatter(size[0], size[1], u.long(), v.long(), vals)
        imgs.append(img.permute(2, 0, 1)[None])
    return torch.cat(imgs, 0)


def depth_image(pc, cam_T_velo, size):
    """torch_utils.py:61-103.  channels (px,py,pz,w); strict bounds (:81)."""
    imgs = []
    for b in range(pc.size(0)):
        pcb = torch.cat([pc[b, :3].float(), torch.ones((1, pc.size(-1)))], 0)
        xyw = torch.mm(cam_T_velo[b].float(), pcb)
        w = xyw[2]
        x = xyw[0] / w
        y = xyw[1] / w
        mask = (x < size[1]) & (x > 0) & (y < size[0]) & (y > 0) & (w > 0)
        vals = torch.stack([pcb[0][mask], pcb[1][mask], pcb[2][mask], w[mask]], 1)
        img = _last_wins_scatter(size[0], size[1], y[mask].long(), x[mask].long(), vals)
        imgs.append(img.permute(2, 0, 1)[None])
    return torch.cat(imgs, 0)


# =========================================================================================
# E-net (nets/enet.py) with BCL (nets/bilateralNN.py)
# =========================================================================================
def bcl(P, pre, feat, bary, off, nbr, use_norm=True, last_relu=False, use_leaky=True):
    """BilateralConvFlex.forward, nets/bilateralNN.py:148-249 (do_splat, no slice).  use_norm: density normalisation (:196-211);
    last_relu: an activation behind the last convolution, LeakyReLU(0.1) / ReLU by use_leaky (:121-135).
    feat (C,N) f32; bary (4,N) f32; off (4,N) i64; nbr (15,H) i64  ->  (C_out, H)."""
    C, N = feat.shape
    H = nbr.size(1)
    idx = (off + 1).reshape(-1)                                   # :186
    tmp = (bary[None, :, :] * feat[:, None, :]).reshape(C, -1).t()  # :182-184  (4N, C)
    splat = torch.zeros((H + 1, C)).index_add(0, idx, tmp)        # SparseSum :6-27
    if use_norm:
        ones = torch.zeros((H + 1,)).index_add(0, idx, bary.reshape(-1))  # :193-206
        splat = splat * (1.0 / (ones + 1e-5))[:, None]                # :209-211
    spread = splat[(nbr + 1)]                                     # (15,H,C)  :240-242
    x = spread.permute(2, 0, 1)[None]                             # (1,C,15,H)
    x = F.conv2d(x, P[pre + '.blur_conv.0.weight'], P[pre + '.blur_conv.0.bias'])   # :103-114
    x = F.relu(x)
    x = F.conv2d(x, P[pre + '.blur_conv.2.weight'], P[pre + '.blur_conv.2.bias'])   # :123
    if last_relu:
        x = F.leaky_relu(x, 0.1) if use_leaky else F.relu(x)      # :131-134
    return x[0, :, 0, :]


def enet(P, pc, train, lattice=None, args=None):
    """nets/enet.py:103-187.  The reference uses batch element 0 only (:107) and is hard-wired to batch 1; for
    B > 1 every sample gets its own lattice and BCL chain (B independent evaluations) and the only coupling is train-mode
    BatchNorm1d of the head, whose statistics run over the vertices of ALL samples - the (1, C, sum_b H5_b) tensor the head's
    Conv1d / BatchNorm1d see when the samples' vertex rows are laid end to end (B = 1: exactly the reference)."""
    B = pc.size(0)
    feats = []
    # the E-net switches of the configuration (enet.py:25-83; the shipped yamls: use_leaky, bcn_use_norm set, last_relu clear)
    use_leaky = True if args is None else bool(args.get('use_leaky', True))
    use_norm = True if args is None else bool(args.get('bcn_use_norm', True))
    last_relu = False if args is None else bool(args.get('last_relu', False))
    for b in range(B):
        lat = lattice if (lattice is not None and B == 1) else _lattice.generate_data(pc[b].detach().numpy())
        x = pc[b:b + 1, :3, :]
        for i in range(3):                                        # conv_in :24-28, LeakyReLU(0.1) / ReLU (net_utils.py:11)
            x = F.conv1d(x, P[f'E.conv_in.{i}.0.weight'], P[f'E.conv_in.{i}.0.bias'])
            x = F.leaky_relu(x, 0.1) if use_leaky else F.relu(x)
        feat = x[0]
        for l in range(5):                                        # :113-141
            g = lat[l]
            emg = torch.from_numpy(g['emg'])
            feat = bcl(P, f'E.bcn{l + 1}', torch.cat((emg, feat), 0), torch.from_numpy(g['bary']),
                       torch.from_numpy(g['off']), torch.from_numpy(g['nbr']), use_norm, last_relu, use_leaky)
        feats.append(feat)
    seg = [0]
    for f_ in feats:
        seg.append(seg[-1] + f_.size(1))
    x = torch.cat(feats, 1)[None]
    for i in (1, 2, 3):                                           # :150-152
        x = F.conv1d(x, P[f'E.conv_gn_{i}.weight'], P[f'E.conv_gn_{i}.bias'])
        x = F.relu(_bn(P, f'E.bn_gn_{i}', x, train))
    x = torch.cat([torch.max(x[:, :, seg[b]:seg[b + 1]], 2)[0] for b in range(B)], 0)      # :154-155, per sample
    for i in (1, 2, 3):
        x = F.relu(F.linear(x, P[f'E.lin_gn_{i}.weight'], P[f'E.lin_gn_{i}.bias']))
    sgn = F.linear(x, P['E.lin_gn_sgn.weight'], P['E.lin_gn_sgn.bias'])
    a = torch.softmax(F.linear(x, P['E.lin_gn_abs.weight'], P['E.lin_gn_abs.bias']), 1)
    a = (a / torch.sqrt(torch.sum(a ** 2, 1, keepdim=True)))[..., None]      # :161-164
    e_gn = normal_from_abs_sign(a, sgn, 3)
    e_T = rotation_between(e_gn, torch.tensor([0., 0., 1.])[None, :, None])
    return {'e_gn_abs': a, 'e_gn_sgn': sgn, 'e_gn': e_gn, 'e_l': e_T,
            'sensor2_T_sensor1': e_T, 'network': 'E'}


# =========================================================================================
# H-net (nets/hnet.py)
# =========================================================================================
def hnet(P, img, train):
    x = vgg_features(P, 'H.vgg.features', img, VGG_A, train)                # :41
    x = x.view(x.size(0), x.size(1), -1)
    for i in (1, 2, 3):                                                    # :49-51
        x = F.conv1d(x, P[f'H.conv_hrzn_{i}.weight'], P[f'H.conv_hrzn_{i}.bias'])
        x = F.relu(_bn(P, f'H.bn_hrzn_{i}', x, train))
    x = torch.max(x, 2)[0]
    for i in (1, 2, 3):
        x = F.relu(F.linear(x, P[f'H.lin_hrzn_{i}.weight'], P[f'H.lin_hrzn_{i}.bias']))
    sgn = F.linear(x, P['H.lin_hrzn_sgn.weight'], P['H.lin_hrzn_sgn.bias'])
    a = torch.softmax(F.linear(x, P['H.lin_hrzn_abs.weight'], P['H.lin_hrzn_abs.bias']), 1)
    a = (a / torch.sqrt(torch.sum(a ** 2, 1, keepdim=True)))[..., None]
    h = normal_from_abs_sign(a, sgn, 2)                                    # :75
    h3 = torch.cat([h, torch.zeros(h.size(0), 1, 1)], 1)
    h_T = rotation_between(h3, torch.tensor([0., 1., 0.])[None, :, None])[:, :3, :3]
    h_img = rotate_image(img, h_T)                                         # :79
    return {'h_hrzn_abs': a, 'h_hrzn_sgn': sgn, 'h_hrzn': h, 'h_img': h_img, 'h_c': h_T,
            'intrinsic_sensor2': h_T, 'network': 'H'}


# =========================================================================================
# F-net (nets/fnet.py)
# =========================================================================================
def f_trunk(P, side, x, train):
    x = vgg_features(P, f'F.vgg_{side}.features', x, VGG_C, train)
    x = convt_bn_relu(P, f'F.vgg_5_1_{side}', x, train, padding=1)         # :23,29
    x = convt_bn_relu(P, f'F.vgg_5_2_{side}', x, train, padding=0)
    x = convt_bn_relu(P, f'F.vgg_5_3_{side}', x, train, padding=1)
    return x


def circular_assign(feat, offset):
    """torch_utils.py:271-284: [mirror(last `offset` cols), feat, first `offset` cols]."""
    left = torch.flip(feat[..., -offset:], dims=[-1])
    return torch.cat([left, feat, feat[..., :offset]], -1)


def fnet(P, pc, ret, args, train, keep=None):
    raw = args['raw_cam_img_size']
    size = (int(raw[0] / 2), int(raw[1] * 2))                              # :19
    pc1 = torch.cat([pc, torch.ones(pc.size(0), 1, pc.size(2))], 1)
    e_pc = torch.bmm(ret['e_l'], pc1)                                      # :44
    e_range = range_image(e_pc, size, args['lidar_fov_rad'])               # :45
    scores, fls = [], []
    cam3_all = f_trunk(P, 'camera', ret['h_img'], train)                   # :53-56 (BN over the batch)
    r0 = conv_bn_relu(P, 'F.conv_range', e_range, train)                   # :59  (1,2) kernel
    rng3_all = f_trunk(P, 'range', r0, train)
    for b in range(pc.size(0)):                                            # B independent evaluations
        cam3, rng3 = cam3_all[b:b + 1], rng3_all[b:b + 1]
        cam_feat = cam3 / (torch.max(cam3) - torch.min(cam3))              # :57
        rng_feat = rng3 / (torch.max(rng3) - torch.min(rng3))              # :64
        rng_pad = circular_assign(rng_feat, int(rng_feat.size(-1) / 8))    # :78
        sc = F.conv2d(rng_pad, cam_feat)                                   # :79
        sc = sc / (cam_feat.size(0) * cam_feat.size(1))                    # :80
        if keep is not None:
            keep.setdefault('f_logit', []).append(sc.view(1, -1))
            keep.setdefault('cam_feat', []).append(cam_feat)
            keep.setdefault('rng_feat', []).append(rng_feat)
        sc = torch.sigmoid(sc.view(-1)).view(1, -1)                        # :81
        scores.append(sc)
        f_idx = torch.argmax(sc, dim=1, keepdim=True).float()              # :87
        f_rad = -(f_idx / (sc.size(-1) - 1)) * 2 * PI + PI                 # :88
        rad = f_rad[0]
        f_fwd = torch.tensor([math.cos(rad), math.sin(rad), 0.])[None, :, None]   # :89
        fls.append(rotation_between(f_fwd, torch.tensor([1., 0., 0.])[None, :, None]))
    ret = dict(ret)
    ret['f_score'] = torch.cat(scores, 0)
    ret['f_l'] = torch.cat(fls, 0)
    ret['sensor2_T_sensor1'] = torch.bmm(ret['f_l'], ret['sensor2_T_sensor1'])   # :101
    ret['network'] = ret['network'] + 'F'
    if keep is not None:
        keep['e_range'] = e_range
    return ret


# =========================================================================================
# G-net (nets/gnet.py)
# =========================================================================================
def concat_tensors(t1, t2):
    """torch_utils.py:309-319 (centre-crop H of t2 when taller)."""
    if t2.size(2) != t1.size(2):
        p1 = int((t2.size(2) - t1.size(2)) / 2)
        t2 = t2[:, :, p1:p1 + t1.size(2), :]
    return torch.cat((t1, t2), 1)


def gnet(P, pc, img, ret, args, train, keep=None):
    c1 = conv_bn_relu(P, 'G.conv_i0', img, train, 1, 1)                    # :103
    c2 = resnet_layer(P, 'G.conv_img2', c1, train, 1)
    c3 = resnet_layer(P, 'G.conv_img3', c2, train, 2)
    c4 = resnet_layer(P, 'G.conv_img4', c3, train, 2)
    c5 = resnet_layer(P, 'G.conv_img5', c4, train, 2)
    t4 = convt_bn_relu(P, 'G.convt_img4', c5, train, 1, 1)                 # :116
    t3 = convt_bn_relu(P, 'G.convt_img3', concat_tensors(c4, t4), train, 1, 1)
    t2 = convt_bn_relu(P, 'G.convt_img2', concat_tensors(c3, t3), train, 1, 1)
    cv = torch.cat((t2, c2), 1)
    dimg = convt_bn_relu(P, 'G.convt_dimg', cv, train, 1, 1)
    mask = torch.softmax(convt_bn_relu(P, 'G.convt_mask', cv, train, 1, 1), 1)
    f_depth = depth_image(pc, ret['efh_cam_T_velo'], args['raw_cam_img_size'])   # :136
    ci1 = conv_bn_relu(P, 'G.conv_i1', t2, train, 1, 0)                    # :142
    cd1 = conv_bn_relu(P, 'G.conv_d1', f_depth, train, 2, 1)
    x = torch.cat((ci1, cd1), 1)
    x = resnet_layer(P, 'G.conv2', x, train, 1)
    x = resnet_layer(P, 'G.conv3', x, train, 2)
    x = resnet_layer(P, 'G.conv4', x, train, 2)
    x = resnet_layer(P, 'G.conv5', x, train, 2)
    for i in (1, 2, 3):                                                    # :160-162
        x = conv_bn_relu(P, f'G.conv_trs_{i}', x, train, 1, 0)
    x = x.view(x.size(0), x.size(1), -1)
    x = F.conv1d(x, P['G.conv_trs_4.weight'], P['G.conv_trs_4.bias'])
    trs = torch.mean(x, 2, keepdim=True)                                   # :165
    g_T = translation_matrix(trs)
    ret = dict(ret)
    ret.update({'g_depth': dimg, 'g_mask': mask, 'g_trs': trs, 'g_l': g_T})
    ret['sensor2_T_sensor1'] = torch.bmm(g_T, ret['sensor2_T_sensor1'])    # :180
    ret['network'] = ret['network'] + 'G'
    if keep is not None:
        keep['f_depth'] = f_depth
    return ret


# =========================================================================================
# backbone (nets/efghbackbone.py:23-43)
# =========================================================================================
def forward(P, pc, img, calib, A, args, train=False, keep=None, lattice=None):
    rete = enet(P, pc, train, lattice, args)
    reth = hnet(P, img, train)
    ret = {}
    ret.update(rete)
    ret.update(reth)
    ret['network'] = rete['network'] + reth['network']
    ret['eh_cam_T_velo'] = compute_cam_T_velo(ret['intrinsic_sensor2'], ret['sensor2_T_sensor1'], calib, A)
    ret = fnet(P, pc, ret, args, train, keep)
    ret['efh_cam_T_velo'] = compute_cam_T_velo(ret['intrinsic_sensor2'], ret['sensor2_T_sensor1'], calib, A)
    ret = gnet(P, pc, img, ret, args, train, keep)
    ret['efgh_cam_T_velo'] = compute_cam_T_velo(ret['intrinsic_sensor2'], ret['sensor2_T_sensor1'], calib, A)
    ret['cam_T_velo'] = ret['efgh_cam_T_velo']
    return ret


# =========================================================================================
# losses (losses/loss_utils.py, losses/efghloss.py)
# =========================================================================================
def _abs_sign_loss(pred_abs, pred_sgn, gt_vec, ncls_dims, lam):
    """Eloss.compute / Hloss.compute, loss_utils.py:25-58 and :227-262."""
    gt_abs = torch.abs(gt_vec)[:, :ncls_dims, :]
    s = torch.sign(gt_vec)
    s = torch.where(s == -1, torch.zeros_like(s), s)
    cls = []
    for b in range(s.size(0)):
        v = s[b, :, 0]
        c = 0
        for i in range(ncls_dims):
            c = c + v[i] * (2 ** (ncls_dims - 1 - i))
        cls.append(c.long()[None])
    cls = torch.cat(cls, 0)
    cos = F.cosine_similarity(pred_abs, gt_abs, dim=1)
    l_abs = torch.mean(1 - cos) * 10.0
    l_sgn = F.cross_entropy(pred_sgn, cls) * 1.0
    return l_abs, l_sgn, gt_abs, cls


def gt_fov(gt_f_axis, width, positive_num=30):
    """Floss.gt_fov, loss_utils.py:119-144."""
    zz = torch.zeros((gt_f_axis.size(0), width))
    for b in range(gt_f_axis.size(0)):
        yaw = torch.atan2(gt_f_axis[b, 1, 0], gt_f_axis[b, 0, 0]).detach()
        f_idx = ((-yaw + PI) / (2 * PI)) * width
        xmin = int(f_idx) - int(positive_num / 2)
        xmax = xmin + positive_num
        if xmin >= 0 and xmax < width:
            zz[b, xmin:xmax] = 1
        elif xmin < 0:
            zz[b, 0:xmax] = 1
            zz[b, xmin:] = 1
        else:
            zz[b, xmin:] = 1
            zz[b, 0:xmax - width] = 1
    return zz


def compute_loss(pc, gt, pred, args):
    """EFGHCriterion.compute_loss, losses/efghloss.py:21-38.  Returns (losses, gt)."""
    lam = args['lambda']
    gt = dict(gt)
    L = {}
    e3 = torch.tensor([0., 0., 1.])[None, :, None]
    e2 = torch.tensor([0., 1., 0.])[None, :, None]
    e1 = torch.tensor([1., 0., 0.])[None, :, None]
    B = pc.size(0)
    # ---- E (loss_utils.py:25-58)
    R = gt['rand_init_l'][:, :3, :3].clone().detach().float()
    g = torch.bmm(R, e3.expand(B, -1, -1))
    g = g / torch.sqrt(torch.sum(g ** 2, 1, keepdim=True))
    gt['e_gn'] = g
    gt['e_l'] = rotation_between(g, e3)
    la, ls, gt['e_gn_abs'], gt['e_gn_sgn'] = _abs_sign_loss(pred['e_gn_abs'], pred['e_gn_sgn'], g, 3, lam)
    L['e_gn'] = (la + ls) * lam['e_gn']
    L['e_gn_abs'] = la * lam['e_gn']
    L['e_gn_sgn'] = ls * lam['e_gn']
    # ---- H (:227-262)
    R = gt['rand_init_c'][:, :3, :3].clone().detach().float()
    g = torch.bmm(R, e2.expand(B, -1, -1))
    g = g / torch.sqrt(torch.sum(g ** 2, 1, keepdim=True))
    gt['h_hrzn'] = g
    gt['h_c'] = rotation_between(g, e2)[:, :3, :3]
    la, ls, gt['h_hrzn_abs'], gt['h_hrzn_sgn'] = _abs_sign_loss(pred['h_hrzn_abs'], pred['h_hrzn_sgn'], g, 2, lam)
    L['h_hrzn'] = (la + ls) * lam['h_hrzn']
    L['h_hrzn_abs'] = la * lam['h_hrzn']
    L['h_hrzn_sgn'] = ls * lam['h_hrzn']
    # ---- F (:77-117)
    Tgt = gt['sensor2_T_sensor1'][:, :3, :3].clone().detach().float()
    Tinv = torch.inverse(Tgt)
    pe = pred['e_l'][:, :3, :3].clone().detach().float()
    axis = torch.bmm(torch.bmm(pe, Tinv), e1.expand(B, -1, -1))
    W = pred['f_score'].size(-1)
    gt['f_score'] = gt_fov(axis, W, args['fov_pos_num'])
    ge = gt['e_l'][:, :3, :3].clone().detach().float()
    fl = torch.zeros((B, 4, 4))
    fl[:, :3, :3] = torch.inverse(torch.bmm(ge, Tinv))
    fl[:, 3, 3] = 1
    gt['f_l'] = fl
    pos = gt['f_score'] > 0
    lc = F.binary_cross_entropy(pred['f_score'], gt['f_score'], reduction='none')
    lc = lc.clone()
    lc[pos] = 0
    _, idx = lc.sort(1, descending=True)
    _, rank = idx.sort(1)
    num_pos = pos.long().sum(1, keepdim=True)
    num_neg = torch.clamp(args['fov_neg_ratio'] * num_pos, max=pos.size(1) - 1)
    neg = rank < num_neg.expand_as(rank)
    wsel = (pos | neg)
    lf = F.binary_cross_entropy(pred['f_score'][wsel].view(B, -1), gt['f_score'][wsel].view(B, -1),
                                reduction='none')
    L['fov'] = torch.mean(lf) * lam['fov']
    # ---- G (:165-207)
    T4 = gt['sensor2_T_sensor1'].clone().detach().float()
    origin = torch.tensor([0., 0., 0., 1.])[None, :, None].expand(B, -1, -1)
    pef = torch.bmm(pred['f_l'], pred['e_l'])
    gt['g_trs'] = torch.bmm(torch.bmm(T4, torch.inverse(pef)), origin)[:, :3, :]
    gef = torch.bmm(gt['f_l'], gt['e_l'])
    gcp = torch.bmm(torch.bmm(T4, torch.inverse(gef)), origin)
    gt['g_l'] = translation_matrix(gcp)
    gdep = depth_image(pc, gt['cam_T_velo'].float(), args['raw_cam_img_size'])
    gt['g_depth'] = gdep[:, -1:, :, :]
    gt['g_mask'] = (gt['g_depth'] > 0).float()
    valid = (gt['g_depth'] > 0) & (gt['img_mask'] > 0)
    l_trs = F.smooth_l1_loss(gt['g_trs'], pred['g_trs'])
    diff = (gt['g_depth'] - pred['g_depth'])[valid]
    l_dep = (diff ** 2).mean()
    l_msk = F.binary_cross_entropy(pred['g_mask'][:, 0].reshape(B, -1), gt['g_mask'].view(B, -1)) * lam['g_mask']
    L['g_trs'] = l_trs * lam['g_trs']
    L['g_depth'] = l_dep * lam['g_depth']
    L['g_mask'] = l_msk * lam['g_depth']                                   # :204 (double scaling)
    total = 0
    for k in L:                                                            # efghloss.py:33-36
        total = total + L[k]
    L['total'] = total
    return L, gt
