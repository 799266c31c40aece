"""GPU-side sample preparation (efgh_amd/data/prepare.py over csrc/prep.hip) against the fixtures produced by the
unmodified reference loaders, the oracle, and Pillow: uint8 images bit-exact, points exact after the float32 cast."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _load(golden_dir):
    G = np.load(os.path.join(str(golden_dir), 'prep_cases.npz'))
    return G, sorted({k.split('.')[0] for k in G.files if '.' in k})


def test_process_classes_equal_reference(golden_dir):
    from efgh_amd.data import ProcessKITTIODOM, ProcessRELLIS
    from tests.test_oracle_prep import drawn_indices
    G, names = _load(golden_dir)
    for name in names:
        raw_h, raw_w, npts, ll, seed, rellis = [int(v) for v in G[name + '.meta']]
        args = {'raw_cam_img_size': [raw_h, raw_w], 'lidar_line': None if ll < 0 else ll, 'num_points': npts, 'test': True}
        proc = (ProcessRELLIS if rellis else ProcessKITTIODOM)(args)
        calibs = {'P': G['P'], 'Tr': G['Tr']} if rellis else {'P2': G['P'], 'Tr': G['Tr']}
        pc, img, calib, A, gts, fname = proc(G[name + '.pcd'], G[name + '.img'], calibs, G[name + '.pose'], name,
                                             rand_init=tuple(G[name + '.rand_init']), sampled_indices=drawn_indices(G, name))
        assert fname == name and pc.is_cuda and img.is_cuda
        assert torch.equal(img.cpu(), torch.from_numpy(G[name + '.out.img'])), name
        for k in ('img_raw', 'img_rot', 'img_mask'):
            assert np.array_equal(gts[k].cpu().numpy(), G[name + '.gt.' + k]), (name, k)
        ref32 = G[name + '.out.pc'].astype(np.float32)
        got = pc.cpu().numpy()
        assert got.shape == ref32.shape
        ulp = np.abs(got.view(np.int32).astype(np.int64) - ref32.view(np.int32).astype(np.int64))
        assert ulp.max() <= 1 and (ulp > 0).mean() < 1e-3, (name, ulp.max())
        assert np.allclose(calib, G[name + '.out.calib'], rtol=0, atol=1e-12) and np.array_equal(A, G[name + '.out.A'])
        for k in ('rand_init_l', 'rand_init_c', 'sensor2_T_sensor1', 'intrinsic_sensor2', 'cam_T_velo'):
            assert np.allclose(gts[k], G[name + '.gt.' + k], rtol=0, atol=1e-12), (name, k)


def test_image_ops_equal_pillow_and_oracle_at_loader_sizes():
    """a 1200x1920 camera frame as RELLIS delivers it (raw 900x1600) and a KITTI-sized 376x1241 one (raw 352x1216)"""
    from PIL import Image
    from efgh_amd.data import prepare as P
    from oracle import prep_oracle as PO
    rng = np.random.default_rng(11)
    for (h, w), raw, rellis, rt in (((1200, 1920), (900, 1600), True, 0.21), ((376, 1241), (352, 1216), False, -0.07),
                                    ((376, 1241), (352, 1216), False, 0.0)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        gts = P.preproc_gt(0, 0, 0, 0, 0, 0, rt)
        got = P.preproc_img(img, gts, raw, rellis)
        ref = PO.preproc_img(img, gts, raw, rellis)
        for k in ('in', 'raw', 'rot', 'img_mask'):
            assert np.array_equal(got[k].cpu().numpy(), ref[k]), (h, w, k)
        deg = PO.rot_deg_of(gts['rand_init_c'])
        pil = np.array(Image.fromarray(img).rotate(deg, expand=True))
        dev = P.rotate_expand(torch.from_numpy(img).cuda(), deg)
        assert np.array_equal(dev.cpu().numpy(), pil)
        pil_small = np.array(Image.fromarray(img).resize((w // 2, h // 2)))
        assert np.array_equal(P.resize_image(torch.from_numpy(img).cuda(), (h // 2, w // 2)).cpu().numpy(), pil_small)


def test_points_full_size_properties():
    """131072-point draw from a 220k sweep: every output column is the transform of a distinct surviving input point"""
    from efgh_amd.data import prepare as P
    rng = np.random.default_rng(2)
    n = 220000
    pcd = np.empty((n, 4), np.float32)
    pcd[:, :2] = rng.uniform(-60, 60, (n, 2))          # ~69 % survive the 50 m box: more than requested
    pcd[:, 2] = rng.uniform(-3, 3, n)
    pcd[:, 3] = 0.5
    gts = P.preproc_gt(0.1, -0.05, 0.3, 0.2, 0.1, -0.3, 0.0)
    out32, out64 = P.preproc_pcd(pcd, gts, 131072, want_float64=True)
    Tinv = np.linalg.inv(gts['rand_init_l'])
    back = (Tinv[:3, :3] @ out64.cpu().numpy() + Tinv[:3, 3:4]).T                 # recovered source coordinates
    assert np.abs(back[:, :2]).max() <= 50.0 + 1e-6
    keys = np.round(back * 1e4).astype(np.int64)
    src = {tuple(r) for r in np.round(pcd[:, :3].astype(np.float64) * 1e4).astype(np.int64)}
    assert len({tuple(r) for r in keys}) == 131072 and all(tuple(r) in src for r in keys[::997])
    assert np.array_equal(out32.cpu().numpy(), out64.cpu().numpy().astype(np.float32))
    # fewer survivors than requested: zero padding, transformed like the reference does (loader_utils.py:190-199)
    few = P.preproc_pcd(pcd[:1000], gts, 4096)
    k = int(((np.abs(pcd[:1000, 0]) < 50) & (np.abs(pcd[:1000, 1]) < 50)).sum())
    tail = few[:, k:].cpu().numpy()
    assert np.allclose(tail, gts['rand_init_l'][:3, 3:4].astype(np.float32))


def test_cpu_tensors_are_refused_by_prep():
    from efgh_amd._C import EfghError
    from efgh_amd.data import prepare as P
    with pytest.raises(EfghError):
        P.preproc_img(np.zeros((8, 8, 3), np.uint8), P.preproc_gt(0, 0, 0, 0, 0, 0, 0.1), (8, 8), device='cpu')


def test_points_edge_cases():
    """no survivor of the radius box; exactly num_points survivors; lidar-line reduction with python's negative indexing"""
    from efgh_amd.data import prepare as P
    from oracle import prep_oracle as PO
    gts = P.preproc_gt(0.2, 0.1, -0.3, 1.0, -2.0, 0.5, 0.0)
    far = np.full((300, 4), 80.0, np.float32)
    out = P.preproc_pcd(far, gts, 128).cpu().numpy()
    assert np.allclose(out, gts['rand_init_l'][:3, 3:4].astype(np.float32))             # all columns = T @ (0,0,0,1)
    rng = np.random.default_rng(0)
    pts = rng.uniform(-40, 40, (256, 4)).astype(np.float32)
    got = P.preproc_pcd(pts, gts, 256).cpu().numpy()                                     # n_keep == num_points: no sampling
    ref = PO.preproc_pcd(pts, gts, 256)[:3].astype(np.float32)
    assert np.abs(got - ref).max() <= 1e-5
    sweep = rng.uniform(-45, 45, (64 * 50 + 7, 4)).astype(np.float32)
    idx = np.arange(1000)
    got = P.preproc_pcd(sweep, gts, 1000, lidar_line=16, sampled_indices=idx).cpu().numpy()
    ref = PO.preproc_pcd(sweep, gts, 1000, lidar_line=16, sampled_indices=idx)[:3].astype(np.float32)
    assert np.abs(got - ref).max() <= 1e-5
