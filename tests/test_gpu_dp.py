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
