"""Two data-parallel ranks sharing one GPU (gloo rendezvous on 127.0.0.1): Trainer with the hook-driven overlapped
all-reduce keeps the replicas bit-identical and trains."""
import json
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, manifest_path):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from efgh_amd import synthetic as syn
    from efgh_amd.losses import EFGHCriterion
    from efgh_amd.nets import EFGHBackbone
    from efgh_amd.train import Trainer
    raw, npts = (128, 256), 2048
    manifest = json.load(open(manifest_path))
    args = syn.default_args(raw, 'cuda')
    m = EFGHBackbone(args)
    m.load_state_dict(syn.synthetic_state_dict(manifest['state_dict'], 1 + rank))      # deliberately different: rank 0 is broadcast
    tr = Trainer(m.cuda(), EFGHCriterion(args), lr=1e-3)
    w0 = tr.flat.w.clone()
    b = syn.make_batch(raw, npts, 1, first_seed=rank)                                    # each rank: its own sample
    inp = [torch.from_numpy(b[k]).cuda() for k in ('pc', 'img', 'calib', 'A')]
    gt = {k: torch.from_numpy(v) for k, v in b['gt'].items()}
    for _ in range(2):
        losses, _ = tr.step(*inp, gt)
    ws = [torch.zeros_like(tr.flat.w) for _ in range(world)]
    dist.all_gather(ws, tr.flat.w)
    same = all(torch.equal(ws[0], w) for w in ws[1:])
    moved = float((tr.flat.w - w0).abs().max()) > 0
    q.put((rank, bool(same), bool(moved), float(losses['total'].detach()), len(tr.comm.buckets)))
    dist.destroy_process_group()


def test_two_ranks_stay_identical_and_train():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    mpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'state_dict_manifest.json')
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, mpath)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=600) for _ in ps]
    for p in ps:
        p.join(120)
    assert all(same and moved for _, same, moved, _, _ in res), res
    assert res[0][4] >= 5                       # 191 MB of gradients in ~32 MB buckets


def _dp_semantics_worker(rank, world, port, q, manifest_path):
    """rank r: ONE sample (seed r), forward + efghloss + backward, gradients averaged over the ranks (what Trainer / the launcher's
    ProcessDataParallel do).  Rank 0 also plays torch.nn.DataParallel in one process: the two samples as two per-replica-BatchNorm
    forwards, outputs gathered along the batch axis, ONE efghloss over the batch of 2 (main.py:127, iterater.py:35-42)."""
    import numpy as np
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from efgh_amd import synthetic as syn
    from efgh_amd.losses import EFGHCriterion
    from efgh_amd.nets import EFGHBackbone
    raw, npts = (128, 256), 2048
    manifest = json.load(open(manifest_path))
    args = syn.default_args(raw, 'cuda')

    def fresh():
        m = EFGHBackbone(args)
        m.load_state_dict(syn.synthetic_state_dict(manifest['state_dict'], 1))
        return m.cuda().train()

    def sample(i):
        b = syn.make_batch(raw, npts, 1, first_seed=i)
        return [torch.from_numpy(b[k]).cuda() for k in ('pc', 'img', 'calib', 'A')], {k: torch.from_numpy(v) for k, v in b['gt'].items()}

    def grads(m):
        return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).double() for p in m.parameters()])

    names = [n for n, _ in fresh().named_parameters()]
    out = {}
    for exact in (True, False):
        m = fresh()
        crit = EFGHCriterion(args)
        crit.dp_exact = exact
        inp, gt = sample(rank)
        pred = m(*inp)
        L, _ = crit.compute_loss(*inp, gt, pred)
        L['total'].backward()
        torch.cuda.synchronize()
        g = grads(m)
        dist.all_reduce(g)
        g /= world
        terms = torch.tensor([float(L[k]) for k in crit.loss_name + ['total']], dtype=torch.float64, device='cuda')
        dist.all_reduce(terms)
        terms /= world
        out[exact] = (g, terms, crit.loss_name + ['total'])
    if rank == 0:
        m = fresh()
        crit = EFGHCriterion(args)
        crit.dp_exact = False                      # ONE process computes the global loss: nothing to weight
        (i0, g0), (i1, g1) = sample(0), sample(1)
        p0, p1 = m(*i0), m(*i1)                    # two replicas' forwards: BatchNorm statistics per replica
        pred = {k: (torch.cat([p0[k], p1[k]], 0) if torch.is_tensor(p0[k]) else p0[k]) for k in p0}
        inp = [torch.cat([a, b], 0) for a, b in zip(i0, i1)]
        gt = {k: torch.cat([g0[k], g1[k]], 0) for k in g0}
        L, _ = crit.compute_loss(*inp, gt, pred)
        L['total'].backward()
        torch.cuda.synchronize()
        g_ref = grads(m)
        ref_terms = {k: float(L[k]) for k in crit.loss_name + ['total']}
        sizes = [p.numel() for p in m.parameters()]
        offs = np.concatenate([[0], np.cumsum(sizes)])
        res = {'terms_ref': ref_terms}
        for exact in (True, False):
            g, terms, keys = out[exact]
            res['terms_%s' % exact] = {k: float(v) for k, v in zip(keys, terms.tolist())}
            per_net = {}
            for net in 'EHFG':
                idx = [i for i, n in enumerate(names) if n.startswith(net + '.')]
                sel = torch.cat([torch.arange(int(offs[i]), int(offs[i + 1])) for i in idx]).cuda()
                d = float((g[sel] - g_ref[sel]).norm() / (g_ref[sel].norm() + 1e-300))
                per_net[net] = d
            res['grad_rel_%s' % exact] = per_net
        q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_dataparallel_on_the_real_net(capsys):
    """Data-parallel semantics on the REAL net (round-4 verdict): two ranks x one sample each, gradients and loss terms averaged
    over the ranks, against what `torch.nn.DataParallel` computes - two per-replica-BatchNorm forwards, outputs gathered, ONE
    efghloss over the batch of two.  Every term is a batch mean except `g_depth` (a mean over the valid pixels of the whole
    batch): with the valid-pixel weighting of EFGHCriterion._dp_weight_masked_mean every term and every sub-network's gradient
    agree to rounding; without it `g_depth` deviates by the imbalance of the samples' valid-pixel counts (printed, and recorded in
    DESIGN 6)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    mpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'state_dict_manifest.json')
    ps = [ctx.Process(target=_dp_semantics_worker, args=(r, 2, port, q, mpath)) for r in range(2)]
    for p in ps:
        p.start()
    res = q.get(timeout=900)
    for p in ps:
        p.join(120)
    ref = res['terms_ref']
    print('\nDP semantics: reference terms', {k: round(v, 6) for k, v in ref.items()})
    for exact in (True, False):
        t = res['terms_%s' % exact]
        dev = {k: abs(t[k] - ref[k]) / (abs(ref[k]) + 1e-12) for k in ref}
        print('  count-weighted=%s: worst term deviation %.2e (%s); g_depth %.3e; gradient deviation per sub-net %s' % (
            exact, max(dev.values()), max(dev, key=dev.get), dev['g_depth'], {k: '%.1e' % v for k, v in res['grad_rel_%s' % exact].items()}))
    t = res['terms_True']
    for k in ref:
        assert abs(t[k] - ref[k]) <= 2e-5 * abs(ref[k]) + 1e-7, (k, t[k], ref[k])
    g = res['grad_rel_True']
    assert all(v < 1e-5 for v in g.values()), g          # measured 2.7e-8 on every sub-network (round 5)
    assert res['grad_rel_False']['G'] > 10 * g['G']        # (the weighting is what closes G's gap: 6.7e-3 without it)
    # the unweighted form is what rounds 1-4 did: the same except for g_depth
    u = res['terms_False']
    for k in ref:
        if k not in ('g_depth', 'total'):
            assert abs(u[k] - ref[k]) <= 2e-5 * abs(ref[k]) + 1e-7, (k, u[k], ref[k])
