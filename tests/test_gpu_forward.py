"""End-to-end EFGHNet forward on the HIP path vs (a) the reference's golden outputs and (b) the
oracle run on the same inputs.  Tolerance: 1e-4 relative on the pose logits (north star)."""
import json
import os

import numpy as np
import pytest
import torch

from efgh_amd import synthetic as syn

pytestmark = pytest.mark.gpu
RAW, NPTS = (128, 256), 2048
LOGITS = ('e_gn_sgn', 'e_gn_abs', 'h_hrzn_sgn', 'h_hrzn_abs', 'f_score', 'g_trs')


def _model(manifest, train):
    from efgh_amd.nets import EFGHBackbone
    m = EFGHBackbone(syn.default_args(RAW, 'cuda'))
    m.load_state_dict(syn.synthetic_state_dict(manifest['state_dict'], 1), strict=True)
    m = m.cuda()
    m.train(train)
    return m


def _inputs(B=1, first=0):
    b = syn.make_batch(RAW, NPTS, B, first)
    return b, [torch.from_numpy(b[k]).cuda() for k in ('pc', 'img', 'calib', 'A')]


def _rel(got, ref):
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12))


@pytest.mark.parametrize('train', [False, True])
def test_forward_vs_reference_golden(golden_dir, manifest, train):
    G = np.load(os.path.join(golden_dir, 'e2e_small.npz'))
    m = _model(manifest, train)
    _, inp = _inputs()
    with torch.no_grad():
        out = m(*inp)
    tag = 'train.' if train else 'eval.'
    assert out['network'] == 'EHFG'
    keys = [k[len(tag):] for k in G.files if k.startswith(tag) and k.count('.') == 1]
    assert len(keys) >= 18
    for k in keys:
        got = out[k].cpu().numpy()
        assert got.shape == G[tag + k].shape, k
        tol = 1e-4 if k in LOGITS else 5e-4
        if k == 'h_img':
            assert np.array_equal(got, G[tag + k])
            continue
        assert _rel(got, G[tag + k]) < tol, (k, _rel(got, G[tag + k]))
    if train:
        sd = m.state_dict()
        for k in [k for k in G.files if k.startswith('train.buf.')]:
            assert np.abs(sd[k[len('train.buf.'):]].cpu().numpy() - G[k]).max() < 1e-5, k


def test_batch2_equals_two_singles(manifest):
    """B>1 == B independent B=1 evaluations in eval mode."""
    m = _model(manifest, False)
    _, inp2 = _inputs(2, 3)
    with torch.no_grad():
        o2 = m(*inp2)
        for b in range(2):
            o1 = m(*[t[b:b + 1] for t in inp2])
            for k in LOGITS + ('g_depth', 'cam_T_velo'):
                assert _rel(o2[k][b:b + 1].cpu().numpy(), o1[k].cpu().numpy()) < 1e-5, (k, b)


def test_cpu_tensors_are_refused(manifest):
    from efgh_amd import _C
    m = _model(manifest, False)
    _, inp = _inputs()
    with pytest.raises(_C.EfghError):
        m(*[t.cpu() for t in inp])
