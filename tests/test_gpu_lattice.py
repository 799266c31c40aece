"""HIP lattice build (through the C-ABI) vs the oracle and the reference's golden arrays: bit-exact."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from efgh_amd import synthetic as syn

pytestmark = pytest.mark.gpu
SCALES = (1.0, 0.75, 0.5, 0.25, 0.125)


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def _gpu(pc):
    from efgh_amd import lattice
    lv = lattice.build_pyramid(torch.from_numpy(pc).cuda(), SCALES)
    out = []
    for d in lv:
        out.append({'H': d.H, 'bary': d.bary.cpu().numpy(), 'emg': d.emg.cpu().numpy(),
                    'off': d.off.cpu().numpy().astype(np.int64),
                    'nbr': d.nbr.cpu().numpy()[:, :15].T.astype(np.int64).copy(),
                    'pts_next': d.pts_next[:, :d.H].cpu().numpy()})
    return out


@pytest.mark.parametrize('n', [512, 4096])
def test_vs_golden(golden_dir, n):
    g = np.load(os.path.join(golden_dir, f'lattice_n{n}.npz'))
    out = _gpu(syn.lidar_sweep(n, 0))
    for l, d in enumerate(out):
        assert d['H'] == int(g[f'H{l}'])
        assert np.array_equal(d['bary'].view(np.uint32), g[f'bary{l}'].view(np.uint32)), l
        assert np.array_equal(d['emg'].view(np.uint32), g[f'emg{l}'].view(np.uint32)), l
        assert np.array_equal(d['off'], g[f'off{l}']), l
        assert np.array_equal(d['nbr'], g[f'nbr{l}']), l


@pytest.mark.parametrize('n', [65536, 131072])
def test_full_size_known_answers(golden_dir, n):
    kat = json.load(open(os.path.join(golden_dir, 'lattice_kat.json')))[str(n)]
    out = _gpu(syn.lidar_sweep(n, 0))
    for d, k in zip(out, kat['levels']):
        assert d['H'] == k['H']
        assert sha16(d['off'][None]) == k['offset_sha16']
        assert sha16(d['nbr'][None]) == k['neighbors_sha16']
        assert sha16(d['bary'][None]) == k['bary_sha16']
        assert sha16(d['emg'][None]) == k['emg_sha16']


@pytest.mark.parametrize('seed,n', [(1, 1024), (5, 20000), (9, 131072)])
def test_vs_oracle_random_scenes(seed, n):
    from oracle import lattice as olat
    rs = np.random.RandomState(seed)
    pc = (rs.randn(3, n) * np.array([[20.], [20.], [2.]])).astype(np.float32)
    ref = olat.generate_data(pc)
    out = _gpu(pc)
    for l, (d, r) in enumerate(zip(out, ref)):
        assert d['H'] == r['H']
        for k in ('bary', 'emg', 'pts_next'):
            assert np.array_equal(d[k].view(np.uint32), r[k].view(np.uint32)), (l, k)
        assert np.array_equal(d['off'], r['off']) and np.array_equal(d['nbr'], r['nbr']), l


def test_degenerate_inputs():
    from oracle import lattice as olat
    for pc in (np.zeros((3, 1), np.float32), np.zeros((3, 7), np.float32)):
        ref = olat.generate_data(pc)
        out = _gpu(pc)
        for d, r in zip(out, ref):
            assert d['H'] == r['H'] and np.array_equal(d['off'], r['off']) and np.array_equal(d['nbr'], r['nbr'])


DEGENERATE = ('one_point', 'coincident7', 'origin7', 'line', 'plane', 'blob_1cm', 'two_clusters_80m', 'on_vertices')


@pytest.mark.parametrize('scene', DEGENERATE)
def test_degenerate_scenes_bit_exact_vs_reference(golden_dir, scene):
    """the HIP build against the REFERENCE's outputs (not only the oracle's) on the degenerate scenes of
    tests/golden/make_golden_degenerate.py: every level's vertex count, barycentric / el_minus_gr bit patterns, lattice_offset and
    blur_neighbors"""
    g = np.load(os.path.join(golden_dir, 'lattice_degenerate.npz'))
    out = _gpu(g[scene + '/pc'])
    assert len(out) == 5
    for l, d in enumerate(out):
        assert d['H'] == int(g[f'{scene}/H{l}']), l
        assert np.array_equal(d['bary'].view(np.uint32), g[f'{scene}/bary{l}'].view(np.uint32)), l
        assert np.array_equal(d['emg'].view(np.uint32), g[f'{scene}/emg{l}'].view(np.uint32)), l
        assert np.array_equal(d['off'], g[f'{scene}/off{l}']), l
        assert np.array_equal(d['nbr'], g[f'{scene}/nbr{l}']), l


def test_degenerate_scenes_batched_together(golden_dir):
    """all eight degenerate scenes padded to one point count and built as ONE batch (the shape the training path uses): every
    sample's lattice equals its own reference fixture - the samples' key ranges, hashes and numberings never interact.  Padding
    repeats a scene's last point (coincident points change neither the vertex set nor its first-seen order)"""
    from efgh_amd import lattice
    g = np.load(os.path.join(golden_dir, 'lattice_degenerate.npz'))
    n = max(g[s + '/pc'].shape[1] for s in DEGENERATE)
    pcs = []
    for s in DEGENERATE:
        p = g[s + '/pc']
        pcs.append(np.concatenate([p, np.repeat(p[:, -1:], n - p.shape[1], 1)], 1))
    lv = lattice.build_pyramid_batched(torch.from_numpy(np.stack(pcs)).cuda(), SCALES)
    for b, s in enumerate(DEGENERATE):
        n0 = g[s + '/pc'].shape[1]
        for l in range(5):
            d = lv[l].sample(b)
            assert d.H == int(g[f'{s}/H{l}']), (s, l)
            assert np.array_equal(d.nbr.cpu().numpy()[:, :15].T.astype(np.int64), g[f'{s}/nbr{l}']), (s, l)
            if l == 0:      # (deeper levels have the fixture's own point count; level 0 carries the padding)
                assert np.array_equal(d.off.cpu().numpy().astype(np.int64)[:, :n0], g[f'{s}/off{l}']), (s, l)
            else:
                assert np.array_equal(d.off.cpu().numpy().astype(np.int64), g[f'{s}/off{l}']), (s, l)
                assert np.array_equal(d.bary.cpu().numpy().view(np.uint32), g[f'{s}/bary{l}'].view(np.uint32)), (s, l)


def test_batched_build_equals_per_sample():
    """the batched lattice (one launch sequence for all samples) reproduces every sample's own lattice:
    local offsets / neighbours / barycentric weights bit-exact vs the oracle"""
    from efgh_amd import lattice
    from oracle import lattice as olat
    pcs = [syn.lidar_sweep(4096, 0), syn.lidar_sweep(4096, 7),
           (np.random.RandomState(3).randn(3, 4096) * np.array([[15.], [15.], [1.5]])).astype(np.float32)]
    pc = torch.from_numpy(np.stack(pcs)).cuda()
    lv = lattice.build_pyramid_batched(pc, SCALES)
    for b, p in enumerate(pcs):
        ref = olat.generate_data(p)
        for l, r in enumerate(ref):
            d = lv[l].sample(b)
            assert d.H == r['H'], (b, l)
            assert np.array_equal(d.off.cpu().numpy().astype(np.int64), r['off']), (b, l)
            assert np.array_equal(d.nbr.cpu().numpy()[:, :15].T.astype(np.int64), r['nbr']), (b, l)
            assert np.array_equal(d.bary.cpu().numpy().view(np.uint32), r['bary'].view(np.uint32)), (b, l)
            assert np.array_equal(d.emg.cpu().numpy().view(np.uint32), r['emg'].view(np.uint32)), (b, l)
            assert np.array_equal(d.pts_next.cpu().numpy().view(np.uint32), r['pts_next'].view(np.uint32)), (b, l)


def test_speculative_pyramid_equals_level_by_level():
    """second build of the same (batch, points, scales) signature: all five levels enqueued from the previous sizes with device-side
    counts and ONE read-back == the level-by-level build; capacities that turn out too small are detected and rebuilt"""
    from efgh_amd import lattice
    pcs = [syn.lidar_sweep(4096, 0), syn.lidar_sweep(4096, 7)]
    pc = torch.from_numpy(np.stack(pcs)).cuda()
    lattice._SIZES.clear()
    a = lattice.build_pyramid_batched(pc, SCALES)                  # level by level (no sizes known)
    key = next(iter(lattice._SIZES))
    assert lattice._SIZES[key] == [d.H for d in a]
    b = lattice.build_pyramid_batched(pc, SCALES)                  # speculative
    lattice._SIZES[key] = [8, 8, 8, 8, 8]                          # far too small: must fall back
    c = lattice.build_pyramid_batched(pc, SCALES)
    pc2 = torch.from_numpy(np.stack([syn.lidar_sweep(4096, 3), syn.lidar_sweep(4096, 4)])).cuda()
    d2 = lattice.build_pyramid_batched(pc2, SCALES)                # speculative with another scene's sizes
    lattice._SIZES.clear()
    e2 = lattice.build_pyramid_batched(pc2, SCALES)
    for x, y in ((a, b), (a, c), (e2, d2)):
        for l, (u, v) in enumerate(zip(x, y)):
            assert u.H == v.H and u.seg == v.seg and u.n_in == v.n_in, l
            for name in ('bary', 'emg', 'off', 'pts_next'):
                assert torch.equal(getattr(u, name), getattr(v, name)), (l, name)
            assert torch.equal(u.nbr, v.nbr), l
            assert torch.equal(u.vseg[:u.H, 1], v.vseg[:v.H, 1]), l
            # same lists per vertex (the segments may sit at different places)
            for h in (0, u.H // 3, u.H - 1):
                su, sv = u.vseg[h].tolist(), v.vseg[h].tolist()
                assert torch.equal(u.list[su[0]:su[0] + su[1]], v.list[sv[0]:sv[0] + sv[1]]), (l, h)


def test_small_hash_table_overflow_is_flagged_not_fatal():
    """hash_slots smaller than the number of distinct lattice keys: the build terminates, flags bit 2 of info[ERR], writes nothing
    out of bounds; the same level with the default table is correct (this is what the speculative path falls back to)"""
    from efgh_amd import _C, lattice
    L = _C.lib()
    pc = torch.from_numpy((np.random.RandomState(0).randn(3, 8192) * np.array([[30.], [30.], [3.]])).astype(np.float32)).cuda()
    n = pc.shape[1]
    st = _C.stream_ptr()
    infos = []
    for hs in (4096, 0):
        lv = lattice._level_arrays(L, pc.device, n, 4 * n, 1, ('hash', hs))
        lattice._launch_build(L, lv, pc.contiguous(), n, None, None, n, 1, 1.0, st)
        torch.cuda.synchronize()
        infos.append(lv.info.cpu().tolist())
    assert infos[0][lattice.INFO_ERR] & 4 and not infos[1][lattice.INFO_ERR]
    assert infos[1][lattice.INFO_H] > 4096                       # (more vertices than the small table had slots)
    from oracle import lattice as olat
    assert infos[1][lattice.INFO_H] == olat.generate_data(pc.cpu().numpy())[0]['H']


def _build_level(pc, mode, scale=1.0):
    """one level of one or more samples through the C-ABI with an explicit build plan; returns (LatticeLevel, host info)"""
    from efgh_amd import _C, lattice
    L = _C.lib()
    B, _, N = pc.shape
    n = B * N
    pts = pc.permute(1, 0, 2).reshape(3, n).contiguous()
    st = _C.stream_ptr()
    lv = lattice._level_arrays(L, pc.device, n, 4 * n, B, mode)
    lattice._launch_build(L, lv, pts, n, None, None, N, B, scale, st)
    H = int(lv.info[lattice.INFO_H].item())
    lattice._launch_neighbors(L, lv, B, max(H, 1), st)
    torch.cuda.synchronize()
    return lv, lv.info.cpu().tolist()


@pytest.mark.parametrize('B,n,scale', [(1, 1, 1.0), (1, 777, 1.0), (3, 4096, 0.5), (2, 40000, 1.0), (8, 16384, 0.125),
                                       (1, 2200000, 1.0)])           # (the last one: 4096-entry buckets, 98 KB of LDS)
def test_partitioned_build_equals_hash_build(B, n, scale):
    """the two builds of lattice.hip (buckets grouped in LDS vs global hash insert) give the same level: vertex numbering, offsets,
    neighbours + alias marks, next-level points, and the same ascending list per vertex"""
    from efgh_amd import _C, lattice
    rs = np.random.RandomState(B * 1000 + n)
    pc = torch.from_numpy((rs.randn(B, 3, n) * np.array([[25.], [25.], [2.5]])).astype(np.float32)).cuda()
    L = _C.lib()
    nb = L.efgh_lattice_part_buckets(_C.c_int32(B * n))
    assert nb >= 8
    a, ia = _build_level(pc, ('hash', 0), scale)
    for mode in (('part', nb, 2048), ('part', max(2, nb // 4), 1024)):
        b, ib = _build_level(pc, mode, scale)
        if ib[lattice.INFO_ERR] & 4:
            # only the deliberately coarse plan may overflow - or a scale at which single vertices hold thousands of entries
            # (what build_pyramid then does is covered by test_partitioned_build_overflow_is_flagged_not_fatal)
            assert mode[1] < nb or scale < 0.25
            continue
        H = ia[lattice.INFO_H]
        assert ib[:3] == ia[:3] and ib[lattice.INFO_SEG:] == ia[lattice.INFO_SEG:]
        for name in ('bary_pm', 'emg_pm', 'off_pm'):
            assert torch.equal(getattr(a, name), getattr(b, name)), name
        assert torch.equal(a.nbr[:H], b.nbr[:H])
        assert torch.equal(a.pts_next_buf[:, :H], b.pts_next_buf[:, :H]) and torch.equal(a.vsid[:H], b.vsid[:H])
        assert torch.equal(a.vseg[:H, 1], b.vseg[:H, 1])
        # the lists, vertex by vertex (segments sit at different places): gather both into vertex-major order and compare
        la, lb = a.list.cpu().numpy(), b.list.cpu().numpy()
        sa, sb_ = a.vseg[:H].cpu().numpy(), b.vseg[:H].cpu().numpy()
        idx = np.repeat(np.arange(H), sa[:, 1])
        within = np.arange(idx.size) - np.repeat(np.cumsum(sa[:, 1]) - sa[:, 1], sa[:, 1])
        assert np.array_equal(la[sa[idx, 0] + within], lb[sb_[idx, 0] + within])
        assert idx.size == 4 * B * n


def test_partitioned_build_overflow_is_flagged_not_fatal():
    """too few table slots for a bucket's vertices, and more entries than a bucket holds (every point in one lattice cell): bit 2 of
    info[ERR], nothing out of bounds; build_pyramid falls back to the hash build and stays exact"""
    from efgh_amd import lattice
    from oracle import lattice as olat
    rs = np.random.RandomState(0)
    pc = torch.from_numpy((rs.randn(1, 3, 8192) * np.array([[30.], [30.], [3.]])).astype(np.float32)).cuda()
    _, info = _build_level(pc, ('part', 8, 16))
    assert info[lattice.INFO_ERR] & 4
    same = torch.zeros(1, 3, 5000).cuda() + torch.tensor([1.0, 2.0, 0.5]).view(1, 3, 1).cuda()       # 5000 entries per vertex
    _, info = _build_level(same, ('part', 8, 64))
    assert info[lattice.INFO_ERR] & 4
    lattice._SIZES.clear()
    for _ in range(2):                 # level by level, then speculative
        lv = lattice.build_pyramid(same[0], SCALES)
        ref = olat.generate_data(same[0].cpu().numpy())
        for d, r in zip(lv, ref):
            assert d.H == r['H'] and np.array_equal(d.off.cpu().numpy().astype(np.int64), r['off'])
            assert np.array_equal(d.nbr.cpu().numpy()[:, :15].T.astype(np.int64), r['nbr'])


def test_inference_build_without_lattice_offset():
    """need_off=False (what the eval forward asks for): everything but `off` is produced and identical; asking for `off` is an
    error, not garbage"""
    from efgh_amd import _C, lattice
    pc = torch.from_numpy(np.stack([syn.lidar_sweep(4096, 0), syn.lidar_sweep(4096, 7)])).cuda()
    lattice._SIZES.clear()
    for _ in range(2):                                             # level by level, then speculative
        a = lattice.build_pyramid_batched(pc, SCALES)
        b = lattice.build_pyramid_batched(pc, SCALES, need_off=False)
        for u, v in zip(a, b):
            assert u.H == v.H and u.seg == v.seg and v.off_pm is None
            assert torch.equal(u.nbr, v.nbr) and torch.equal(u.bary, v.bary) and torch.equal(u.pts_next, v.pts_next)
            assert torch.equal(u.vseg[:u.H, 1], v.vseg[:v.H, 1])
            for h in (0, u.H // 2, u.H - 1):
                su, sv = u.vseg[h].tolist(), v.vseg[h].tolist()
                assert torch.equal(u.list[su[0]:su[0] + su[1]], v.list[sv[0]:sv[0] + sv[1]])
            with pytest.raises(_C.EfghError):
                v.off


@pytest.mark.parametrize('n,spread', [(6000, 3.0), (30000, 2.0), (60000, 1.2), (60000, 0.6), (200000, 1.0), (200000, 0.3)])
def test_partitioned_build_long_vertex_lists(n, spread):
    """spatially dense scenes (what a real sweep looks like near the sensor): vertex lists of hundreds to thousands of entries go
    through the in-LDS bitonic sorts (one wave per list, the whole workgroup on lists above 1 024 entries) and, for buckets above
    2 048 entries, through k_lat_bucket_big (8 192 entries); beyond that the level is flagged.  Whatever was built must equal the
    hash build: numbering, offsets, neighbours and every ascending list."""
    from efgh_amd import _C, lattice
    rs = np.random.RandomState(n)
    pc = torch.from_numpy((rs.randn(1, 3, n) * spread).astype(np.float32)).cuda()
    L = _C.lib()
    nb = L.efgh_lattice_part_buckets(_C.c_int32(n))
    a, ia = _build_level(pc, ('hash', 0))
    b, ib = _build_level(pc, ('part', nb, 512, True))
    H = ia[lattice.INFO_H]
    longest = int(a.vseg[:H, 1].max())
    print('n', n, 'spread', spread, 'H', H, 'longest list', longest, 'flagged', bool(ib[lattice.INFO_ERR] & 4))
    if ib[lattice.INFO_ERR] & 4:
        assert longest > 2048          # only an over-long list (or several of them in one bucket) may be refused
        return
    assert ib[:3] == ia[:3]
    for name in ('bary_pm', 'emg_pm', 'off_pm'):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert torch.equal(a.nbr[:H], b.nbr[:H]) and torch.equal(a.vseg[:H, 1], b.vseg[:H, 1])
    la, lb = a.list.cpu().numpy(), b.list.cpu().numpy()
    sa, sb_ = a.vseg[:H].cpu().numpy(), b.vseg[:H].cpu().numpy()
    idx = np.repeat(np.arange(H), sa[:, 1])
    within = np.arange(idx.size) - np.repeat(np.cumsum(sa[:, 1]) - sa[:, 1], sa[:, 1])
    assert np.array_equal(la[sa[idx, 0] + within], lb[sb_[idx, 0] + within])


def test_escalations_expire_after_clean_builds(monkeypatch):
    """an outlier frame must not pin a signature to the slow plans for the rest of the process: an escalated level (big-bucket
    kernel, then the hash build) is taken back one step after ESCALATION_DECAY clean speculative builds; every plan gives the
    same lattice, and STATS counts the paths"""
    from efgh_amd import lattice
    pc = torch.from_numpy(np.stack([syn.lidar_sweep(4096, 0), syn.lidar_sweep(4096, 7)])).cuda()
    lattice._SIZES.clear()
    ref = lattice.build_pyramid_batched(pc, SCALES)                          # level by level; sizes known from here on
    key = next(iter(lattice._SIZES))
    monkeypatch.setattr(lattice, 'ESCALATION_DECAY', 2)
    lattice._HASH_LEVELS[key] = {1}
    lattice._BIG_LEVELS[key] = {0}
    lattice._CLEAN[key] = 0
    before = dict(lattice.STATS)
    seen = []
    for _ in range(7):
        lv = lattice.build_pyramid_batched(pc, SCALES)
        seen.append((sorted(lattice._HASH_LEVELS[key]), sorted(lattice._BIG_LEVELS[key]), [d._mode[0] if d._mode else None for d in lv]))
        for u, v in zip(ref, lv):
            assert u.H == v.H and torch.equal(u.nbr, v.nbr) and torch.equal(u.off, v.off) and torch.equal(u.bary, v.bary)
    # hash level 1 -> big after two clean builds, then the big levels go one by one
    assert seen[0][:2] == ([1], [0]) and seen[1][:2] == ([], [0, 1]) and seen[3][:2] == ([], [0]) and seen[5][:2] == ([], [])
    assert lattice.STATS['speculative'] - before['speculative'] == 7 and lattice.STATS['reenqueued'] == before['reenqueued']


def _same_levels(a, b):
    for l, (x, y) in enumerate(zip(a, b)):
        assert x.H == y.H and x.seg == y.seg, l
        for name in ('bary_pm', 'emg_pm', 'off_pm'):
            assert torch.equal(getattr(x, name)[:x.n_in], getattr(y, name)[:y.n_in]), (l, name)
        assert torch.equal(x.nbr[:x.H], y.nbr[:y.H]), l
        assert torch.equal(x.pts_next, y.pts_next) and torch.equal(x.vsid[:x.H], y.vsid[:y.H]), l
        # the lists: same vertices, same ascending entries (the windows they live in differ between the builds)
        vx, vy = x.vseg[:x.H].cpu().numpy(), y.vseg[:y.H].cpu().numpy()
        assert np.array_equal(vx[:, 1], vy[:, 1]), l
        lx, ly = x.list.cpu().numpy(), y.list.cpu().numpy()
        for h in list(range(0, x.H, max(1, x.H // 97))) + [x.H - 1]:
            assert np.array_equal(lx[vx[h, 0]:vx[h, 0] + vx[h, 1]], ly[vy[h, 0]:vy[h, 0] + vy[h, 1]]), (l, h)
        al = lambda v: sorted(map(tuple, v.alist[:v.n_alias].cpu().tolist()))
        assert x.n_alias == y.n_alias and al(x) == al(y), l


@pytest.mark.parametrize('B,N,seeds', [(3, 4096, (0, 7, 11)), (8, 2048, tuple(range(8))), (2, 20480, (3, 4)), (1, 131072, (0,))])
def test_one_launch_tail_equals_the_per_level_kernels(B, N, seeds):
    """levels whose samples fit one workgroup's LDS are built by ONE launch (k_lat_tail: one workgroup per sample walks down the
    remaining levels, the workgroups exchange their vertex counts once per level) - against the per-level kernels array by array,
    on clouds where the tail starts at level 0 (small N), at level 1-3 (the bench sizes), with several samples per batch"""
    from efgh_amd import lattice
    pcs = [syn.lidar_sweep(N, s) if s % 2 == 0 else (np.random.RandomState(s).randn(3, N) * np.array([[15.], [15.], [1.5]])).astype(np.float32)
           for s in seeds]
    pc = torch.from_numpy(np.stack(pcs)).cuda()
    old = lattice.TAIL
    try:
        lattice.TAIL = False
        lattice._SIZES.clear(); lattice._PER_SAMPLE.clear(); lattice._NO_TAIL.clear()
        lattice.build_pyramid_batched(pc, SCALES)
        ref = lattice.build_pyramid_batched(pc, SCALES)                 # speculative, per-level kernels
        lattice.TAIL = True
        before = dict(lattice.STATS)
        got = lattice.build_pyramid_batched(pc, SCALES)                 # speculative, tail in one launch
        assert lattice.STATS['speculative'] == before['speculative'] + 1 and lattice.STATS['reenqueued'] == before['reenqueued']
        modes = [lv._mode[0] for lv in got]
        assert modes[-1] == 'tail', modes
        _same_levels(got, ref)
    finally:
        lattice.TAIL = old


def test_tail_overflow_falls_back_to_the_per_level_kernels(monkeypatch):
    """a vertex with more entries than the tail kernel's rank sort takes (3 000 coincident points: one list of 3 001) flags the level;
    the pyramid is re-enqueued with the per-level kernels for that level and the result is still the oracle's.  (The plan is forced
    to start the tail at level 0: left alone it would start below the level with the long list.)"""
    from efgh_amd import lattice
    from oracle import lattice as olat
    rs = np.random.RandomState(5)
    pc = (rs.randn(3, 3200) * np.array([[10.], [10.], [1.]])).astype(np.float32)
    pc[:, 200:] = pc[:, 199:200]
    t = torch.from_numpy(pc[None]).cuda()
    lattice._SIZES.clear(); lattice._PER_SAMPLE.clear(); lattice._NO_TAIL.clear()
    lattice.build_pyramid_batched(t, SCALES)
    lattice._BIG_LEVELS.clear(); lattice._HASH_LEVELS.clear()
    real = lattice._tail_plan

    def forced(L, key, B, N, nlev):
        bad = lattice._NO_TAIL.get(key, set())
        l0 = max(bad) + 1 if bad else 0
        return (l0, 2048) if l0 < nlev else None
    monkeypatch.setattr(lattice, '_tail_plan', forced)
    before = dict(lattice.STATS)
    lv = lattice.build_pyramid_batched(t, SCALES)
    monkeypatch.setattr(lattice, '_tail_plan', real)
    assert lattice.STATS['reenqueued'] > before['reenqueued'], [x._mode for x in lv]
    assert 0 in next(iter(lattice._NO_TAIL.values())) and lv[0]._mode[0] != 'tail' and lv[-1]._mode[0] == 'tail', [x._mode for x in lv]
    ref = olat.generate_data(pc)
    for l, r in enumerate(ref):
        d = lv[l].sample(0)
        assert d.H == r['H'], l
        assert np.array_equal(d.off.cpu().numpy().astype(np.int64), r['off']), l
        assert np.array_equal(d.nbr.cpu().numpy()[:, :15].T.astype(np.int64), r['nbr']), l
