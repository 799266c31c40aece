"""efgh_amd/common/summary.py (csrc/summary.hip + the Pillow kernels) against the fixtures produced by the unmodified reference
(tests/golden/summary_cases.npz) and against the oracle: image_draw / eval_image_draw byte for byte."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, 'golden', 'summary_cases.npz'))


def case(name):
    raw = (int(G[name + '.meta'][0]), int(G[name + '.meta'][1]))
    px = int(G[name + '.meta'][3])
    pick = lambda pre: {k[len(name) + len(pre) + 2:]: G[k] for k in G.files if k.startswith('%s.%s.' % (name, pre))}
    dev = lambda d: {k: torch.from_numpy(v)[None].cuda() for k, v in d.items()}
    inp, gt, pred = dev(pick('in')), dev(pick('gt')), dev(pick('pred'))
    pred['network'] = 'EHFG'
    return raw, px, inp, gt, pred, pick('draw'), pick('eval'), pick('prim')


@pytest.mark.parametrize('name', ['a', 'b'])
def test_rasters_vs_reference(name):
    from efgh_amd.common import summary as S
    raw, px, inp, gt, pred, _, _, prim = case(name)
    d = S.depth_image_last(inp['pc'][0], pred['eh_cam_T_velo'][0].cpu().numpy(), raw).cpu().numpy()
    r = S.range_image_last(inp['pc'][0], pred['e_l'][0].cpu().numpy(), (raw[0] // 2, raw[1] * 2), G['fov']).cpu().numpy()
    assert np.array_equal(d, prim['depth'])
    # float64 asin / atan2 of the device library vs the host's: identical pixels, values to the last bits
    assert np.array_equal(r != 0, prim['range'] != 0) and np.abs(r - prim['range']).max() < 1e-12


@pytest.mark.parametrize('name', ['a', 'b'])
def test_image_draw_vs_reference(name):
    from efgh_amd.common import summary as S
    raw, px, inp, gt, pred, draw, _, _ = case(name)
    got = S.image_draw(inp['pc'], inp['img'], inp['calib'], inp['A'], gt, pred, raw, list(G['fov']), cmap='plasma')
    assert set(got) == set(draw)
    for k in draw:
        a = got[k].cpu().numpy()
        assert a.shape == draw[k].shape and a.dtype == np.uint8, k
        assert np.array_equal(a, draw[k]), (k, int((a != draw[k]).sum()))


@pytest.mark.parametrize('name', ['a', 'b'])
def test_eval_image_draw_vs_reference(name):
    from efgh_amd.common import summary as S
    raw, px, inp, gt, pred, _, ev, _ = case(name)
    got = S.eval_image_draw(inp['pc'], inp['img'], inp['calib'], inp['A'], gt, pred, raw, list(G['fov']), px, cmap='jet')
    assert set(got) == set(ev)
    for k in ev:
        a = got[k].cpu().numpy()
        assert np.array_equal(a, ev[k]), (k, int((a != ev[k]).sum()))


def test_paint_dense_image_vs_oracle():
    """a DENSE float32 image (the g_depth prediction case: every pixel paints or is refused) at a size with many wavefront steps"""
    from efgh_amd.common import summary as S
    from oracle import summary_oracle as SO
    rs = np.random.RandomState(0)
    lut = np.load(os.path.join(HERE, '..', 'efgh_amd', 'common', 'colormaps.npz'))['jet']
    for (h, w, px) in ((37, 53, 2), (64, 40, 1), (30, 70, 3)):
        img = np.float32(rs.randn(h, w) * 2 + 5)
        p = S._Painter(torch.from_numpy(lut).cuda())
        p.add(torch.from_numpy(img).cuda(), px)
        rgb, mask = p.run()[0]
        ref_rgb, ref_mask = SO.minmax_color(img, lut, px)
        assert np.array_equal(rgb.cpu().numpy(), ref_rgb) and np.array_equal(mask.cpu().numpy() != 0, ref_mask), (h, w, px)


def test_update_summary_feeds_a_writer():
    from efgh_amd.common import summary as S

    class W:
        def __init__(self): self.sc, self.im = {}, {}
        def add_scalar(self, k, v, it): self.sc[k] = v
        def add_image(self, k, a, it): self.im[k] = a

    class Avg:
        avg = 1.5
    raw, px, inp, gt, pred, draw, _, _ = case('a')
    w = W()
    S.update_summary(w, 'train', 7, {'total': Avg()}, {'rot': 0.25}, inp['pc'], inp['img'], inp['calib'], inp['A'], gt, pred, raw,
                     list(G['fov']))
    assert w.sc == {'train_loss/total': 1.5, 'train_error/rot': 0.25}
    assert set(w.im) == {'train_image/' + k for k in draw}
    assert all(a.shape[0] == 3 and a.dtype == np.uint8 for a in w.im.values())
    assert np.array_equal(np.transpose(w.im['train_image/depth'], (1, 2, 0)), draw['depth'])


def test_image_draw_on_real_network_outputs_vs_oracle(manifest):
    """the whole chain on one synthetic frame: GPU sample preparation -> EFGHBackbone (eval) -> EFGHCriterion (its gt dict) ->
    image_draw / eval_image_draw, against the oracle on the same tensors (dense float32 g_depth / g_mask predictions included)"""
    import sys
    sys.path.insert(0, os.path.join(HERE, '..'))
    from examples.train_synthetic import collate, raw_frame
    from efgh_amd import synthetic as syn
    from efgh_amd.common import summary as S
    from efgh_amd.data import ProcessKITTIODOM
    from efgh_amd.losses import EFGHCriterion
    from efgh_amd.nets import EFGHBackbone
    from oracle import summary_oracle as SO
    raw, npts = (128, 256), 2048
    args = syn.default_args(raw, 'cuda')
    args.update({'lidar_line': None, 'num_points': npts, 'test': False,
                 'dclb': {'l_rot_range': 1 / 12., 'l_trs_range': 1.0, 'c_rot_range': 1 / 12.}})
    m = EFGHBackbone(args)
    m.load_state_dict(syn.synthetic_state_dict(manifest['state_dict'], 1))
    m = m.cuda().eval()
    calib0, _ = syn.calib_and_A(raw)
    P2 = np.eye(4); P2[:3] = calib0
    sweep = np.concatenate([syn.lidar_sweep(npts * 2, 5).T, np.ones((npts * 2, 1), np.float32)], 1)
    sample = ProcessKITTIODOM(args)(sweep, raw_frame(raw, 5), {'P2': P2, 'Tr': np.eye(4)}, np.eye(4), 'f5')[:5]
    pc, img, calib, A, gt = collate([sample], 'cuda')
    with torch.no_grad():
        pred = m(pc, img, calib, A)
        _, gt2 = EFGHCriterion(args).compute_loss(pc, img, calib, A, gt, pred)
    fov = args['lidar_fov_rad']
    luts = np.load(os.path.join(HERE, '..', 'efgh_amd', 'common', 'colormaps.npz'))
    n0 = lambda d: {k: (v[0].detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in d.items()}
    gtn, prn = n0(gt2), n0(pred)
    prn['network'] = pred['network']
    a = [t[0].cpu().numpy() for t in (pc, img, calib, A)]
    got = S.image_draw(pc, img, calib, A, gt2, pred, raw, fov)
    ref = SO.image_draw(a[0], a[1], a[2], a[3], gtn, prn, raw, fov, luts['plasma'])
    assert set(got) == set(ref) == {'cam', 'score', 'dimage', 'mask', 'range', 'depth'}
    for k in ref:
        g = got[k].cpu().numpy()
        assert g.shape == ref[k].shape, k
        # rasters: libm vs device asin / atan2 can move a point across a pixel edge; everything else is byte-exact
        assert (g != ref[k]).any(-1).mean() < (2e-3 if k in ('range', 'depth') else 1e-12), (k, float((g != ref[k]).any(-1).mean()))
    got_e = S.eval_image_draw(pc, img, calib, A, gt2, pred, raw, fov, 2)
    ref_e = SO.eval_image_draw(a[0], a[1], a[2], a[3], gtn, prn, raw, fov, 2, luts['jet'])
    for k in ref_e:
        g = got_e[k].cpu().numpy()
        assert g.shape == ref_e[k].shape and (g != ref_e[k]).any(-1).mean() < 2e-3, k
