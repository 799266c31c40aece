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
