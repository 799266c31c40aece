"""The C restatement (oracle/lattice_oracle.c) is pinned bit-for-bit against arrays produced by
the unmodified reference (nets/generate_data.py + nets/transforms.py), tests/golden/make_golden.py."""
import hashlib
import json
import os

import numpy as np
import pytest

from efgh_amd import synthetic as syn
from oracle import lattice


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def test_constants(golden_dir):
    c = np.load(os.path.join(golden_dir, 'lattice_consts.npz'))
    import re
    src = open(os.path.join(os.path.dirname(lattice.__file__), 'lattice_oracle.c')).read()
    bits = [int(x, 16) for x in re.findall(r'0x([0-9A-F]{8})u', src)][:12]
    assert np.array_equal(np.array(bits, np.uint32).reshape(4, 3), c['elevate_mat'].view(np.uint32))
    assert c['offsets_r1'].shape == (15, 4)
    assert abs(float(c['expected_std']) - 4 * np.sqrt(2 / 3)) < 1e-15


@pytest.mark.parametrize('n', [512, 4096])
def test_levels_bit_exact(golden_dir, n):
    g = np.load(os.path.join(golden_dir, f'lattice_n{n}.npz'))
    out = lattice.generate_data(syn.lidar_sweep(n, 0))
    for l, d in enumerate(out):
        assert d['H'] == int(g[f'H{l}'])
        assert np.array_equal(d['bary'].view(np.uint32), g[f'bary{l}'].view(np.uint32))
        assert np.array_equal(d['emg'].view(np.uint32), g[f'emg{l}'].view(np.uint32))
        assert np.array_equal(d['off'], g[f'off{l}'])
        assert np.array_equal(d['nbr'], g[f'nbr{l}'])


@pytest.mark.parametrize('n', [65536, 131072])
def test_known_answers_full_size(golden_dir, n):
    kat = json.load(open(os.path.join(golden_dir, 'lattice_kat.json')))[str(n)]
    pc = syn.lidar_sweep(n, 0)
    assert sha16(pc) == kat['pc_sha16']
    out = lattice.generate_data(pc)
    for d, k in zip(out, kat['levels']):
        assert d['H'] == k['H']
        assert sha16(d['off'][None]) == k['offset_sha16'] and int(d['off'].sum()) == k['offset_sum']
        assert sha16(d['nbr'][None]) == k['neighbors_sha16'] and int(d['nbr'].sum()) == k['neighbors_sum']
        assert sha16(d['bary'][None]) == k['bary_sha16']
        assert sha16(d['emg'][None]) == k['emg_sha16']


def test_edge_cases():
    # one point, coincident points, and points exactly on lattice vertices
    for pc in (np.zeros((3, 1), np.float32), np.zeros((3, 7), np.float32),
               np.float32([[1.5, -2.25, 0.0], [0.0, 3.0, -3.0], [7.0, 7.0, 7.0]]).T.copy()):
        out = lattice.generate_data(pc)
        for d in out:
            assert d['off'].min() >= 0 and d['off'].max() < d['H']
            assert d['nbr'].min() >= -1 and d['nbr'].max() < d['H']
            assert np.array_equal(d['nbr'][0], np.arange(d['H']))      # offset 0 = the vertex itself
            assert np.allclose(d['bary'].sum(0), 1.0, atol=1e-5)


DEGENERATE = ('one_point', 'coincident7', 'origin7', 'line', 'plane', 'blob_1cm', 'two_clusters_80m', 'on_vertices')


@pytest.mark.parametrize('scene', DEGENERATE)
def test_degenerate_scenes_bit_exact_vs_reference(golden_dir, scene):
    """one point, coincident points, a line, a plane, a 1-cm blob, two clusters 80 m apart, points on lattice vertices: the
    reference's own outputs (tests/golden/make_golden_degenerate.py) - key ranges of extent 1 and `key2int` without a range check
    (transforms.py:62-77) are exercised here, not in the LiDAR-like fixtures"""
    g = np.load(os.path.join(golden_dir, 'lattice_degenerate.npz'))
    out = lattice.generate_data(g[scene + '/pc'])
    assert len(out) == 5
    for l, d in enumerate(out):
        assert d['H'] == int(g[f'{scene}/H{l}']), l
        assert np.array_equal(d['bary'].view(np.uint32), g[f'{scene}/bary{l}'].view(np.uint32)), l
        assert np.array_equal(d['emg'].view(np.uint32), g[f'{scene}/emg{l}'].view(np.uint32)), l
        assert np.array_equal(d['off'], g[f'{scene}/off{l}']), l
        assert np.array_equal(d['nbr'], g[f'{scene}/nbr{l}']), l
