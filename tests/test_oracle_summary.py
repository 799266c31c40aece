"""oracle/summary_oracle.py (numpy restatement of common/numpy_utils.py:8-413) against fixtures produced by the unmodified
reference (tests/golden/make_golden_summary.py): every image of image_draw and eval_image_draw, byte for byte."""
import os

import numpy as np
import pytest

from oracle import summary_oracle as SO

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, 'golden', 'summary_cases.npz'))
LUT = np.load(os.path.join(HERE, '..', 'efgh_amd', 'common', 'colormaps.npz'))


def case(name):
    raw = (int(G[name + '.meta'][0]), int(G[name + '.meta'][1]))
    px = int(G[name + '.meta'][3])
    pick = lambda pre: {k[len(name) + len(pre) + 2:]: G[k] for k in G.files if k.startswith('%s.%s.' % (name, pre))}
    inp, gt, pred = pick('in'), pick('gt'), pick('pred')
    pred['network'] = 'EHFG'
    return raw, px, inp, gt, pred, pick('draw'), pick('eval'), pick('prim')


@pytest.mark.parametrize('name', ['a', 'b'])
def test_rasters_last_point_wins(name):
    raw, px, inp, gt, pred, _, _, prim = case(name)
    assert np.array_equal(SO.depth_image_last(inp['pc'], pred['eh_cam_T_velo'], raw), prim['depth'])
    assert np.array_equal(SO.range_image_last(inp['pc'], pred['e_l'], (raw[0] // 2, raw[1] * 2), G['fov']), prim['range'])


@pytest.mark.parametrize('name', ['a', 'b'])
def test_image_draw_matches_reference(name):
    raw, px, inp, gt, pred, draw, _, _ = case(name)
    got = SO.image_draw(inp['pc'], inp['img'], inp['calib'], inp['A'], gt, pred, raw, G['fov'], LUT['plasma'])
    assert set(got) == set(draw)
    for k in draw:
        assert got[k].shape == draw[k].shape and got[k].dtype == np.uint8, k
        assert np.array_equal(got[k], draw[k]), (k, int((got[k] != draw[k]).sum()))


@pytest.mark.parametrize('name', ['a', 'b'])
def test_eval_image_draw_matches_reference(name):
    raw, px, inp, gt, pred, _, ev, _ = case(name)
    got = SO.eval_image_draw(inp['pc'], inp['img'], inp['calib'], inp['A'], gt, pred, raw, G['fov'], px, LUT['jet'])
    assert set(got) == set(ev)
    for k in ev:
        assert np.array_equal(got[k], ev[k]), (k, int((got[k] != ev[k]).sum()))


def test_paint_edge_rules():
    """the last row / column are never painted; ties do not repaint; a constant image stays empty"""
    img = np.zeros((6, 7)); img[5, 6] = 3.0; img[0, 0] = 1.5; img[2, 2] = 1.5
    out = SO.minmax_paint(img, 2)
    assert out[5, :].max() == 0 and out[:, 6].max() == 0 and out[4, 5] == 1.0
    assert out[0, 0] == 0.5 and out[3, 3] == 0.0          # (2,2) ties with the 0.5 already in its window: it does not paint
    assert out[3, 4] == 1.0 and out[4, 4] == 1.0          # (5,6) paints rows 3-4, columns 4-5 only
    assert SO.minmax_paint(np.full((4, 4), 2.0), 1).max() == 0
