"""Pins oracle/prep_oracle.py (CPU restatement of the reference's sample preparation) against the
fixtures produced by the unmodified reference (tests/golden/make_golden_prep.py) and against Pillow itself."""
import numpy as np
import pytest

from oracle import prep_oracle as PO


def _cases(golden_dir):
    import os
    G = np.load(os.path.join(str(golden_dir), 'prep_cases.npz'))
    names = sorted({k.split('.')[0] for k in G.files if '.' in k})
    return G, names


def drawn_indices(G, name):
    """the index list the reference's np.random.choice drew (same seed, same population size)"""
    raw_h, raw_w, npts, ll, seed, rellis = [int(v) for v in G[name + '.meta']]
    pcd = G[name + '.pcd']
    if rellis:
        pcd = pcd * np.array([-1, -1, 1, 1], np.float32)
    if ll >= 0:
        pcd = pcd[PO.lidar_line_indices(pcd.shape[0], ll)]
    keep = (pcd[:, 0] >= -50.) & (pcd[:, 0] < 50.) & (pcd[:, 1] >= -50.) & (pcd[:, 1] < 50.)
    n = int(keep.sum())
    if npts >= n:
        return None
    np.random.seed(seed)
    return np.random.choice(range(n), size=npts, replace=False)


def test_process_sample_equals_reference(golden_dir):
    G, names = _cases(golden_dir)
    assert len(names) == 5
    for name in names:
        raw_h, raw_w, npts, ll, seed, rellis = [int(v) for v in G[name + '.meta']]
        calib = (G['P'] @ G['Tr'] @ (np.linalg.inv(np.diag([-1., -1., 1., 1.])) if rellis else np.eye(4)))[:3]
        pc, img, calib_o, A, gts = PO.process_sample(
            G[name + '.pcd'], G[name + '.img'], calib, G[name + '.pose'], tuple(G[name + '.rand_init']), (raw_h, raw_w),
            npts, None if ll < 0 else ll, bool(rellis), drawn_indices(G, name))
        assert np.array_equal(img, G[name + '.out.img']), name                     # uint8-valued: exact
        assert img.dtype == np.float32 and img.shape == (3, raw_h // 2, raw_w // 2)
        for k in ('img_raw', 'img_rot', 'img_mask'):
            assert np.array_equal(gts[k], G[name + '.gt.' + k]), (name, k)
        assert np.array_equal(pc, G[name + '.out.pc']), name                        # same float64 operations
        assert np.allclose(calib_o, G[name + '.out.calib'], rtol=0, atol=1e-12)
        assert np.array_equal(A, G[name + '.out.A'])
        for k in ('rand_init_l', 'rand_init_c', 'sensor2_T_sensor1', 'intrinsic_sensor2', 'cam_T_velo'):
            assert np.allclose(gts[k], G[name + '.gt.' + k], rtol=0, atol=1e-12), (name, k)


@pytest.mark.parametrize('hw', [(37, 53), (64, 64), (50, 121)])
def test_rotate_expand_equals_pillow(hw):
    from PIL import Image
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)
    for ang in [0.0, 90.0, 180.0, 270.0, 360.0, -90.0, 4.010704, -17.5, 33.3, 123.456, 269.999, -0.0001, 45.0, 719.5]:
        ref = np.array(Image.fromarray(img).rotate(ang, expand=True))
        got = PO.pil_rotate_nearest_u8(img, ang, expand=True)
        assert got.shape == ref.shape and np.array_equal(got, ref), (hw, ang)


@pytest.mark.parametrize('hw,thw', [((48, 160), (24, 80)), ((75, 120), (60, 100)), ((60, 100), (30, 50)),
                                     ((33, 47), (50, 20)), ((20, 31), (20, 62)), ((64, 64), (7, 9))])
def test_bicubic_resize_equals_pillow(hw, thw):
    from PIL import Image
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)
    img[:5, :7] = 255
    img[-4:, -9:] = 0
    ref = np.array(Image.fromarray(img).resize((thw[1], thw[0])))
    got = PO.pil_resize_bicubic_u8(img, thw)
    assert np.array_equal(got, ref)


def test_lidar_line_indices_match_python_negative_indexing():
    n = 64 * 100 + 13
    x = np.arange(n)
    idx = PO.lidar_line_indices(n, 32)
    line_num = int(n / 64)
    ref = [x[i * line_num + j] for i in range(64) if i % 2 == 0 for j in range(int(-line_num / 2), int(line_num / 2))]
    assert np.array_equal(x[idx], np.array(ref))
