"""Data-parallel plumbing (flat parameter buffer + bucketed gradient all-reduce) on 2 CPU ranks over gloo."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from efgh_amd.train import FlatParams, allreduce_mean_
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.ReLU(), torch.nn.Linear(19, 5))
    flat = FlatParams(model)
    assert flat.n == sum(p.numel() for p in model.parameters())
    names_before = list(model.state_dict().keys())
    torch.manual_seed(100 + rank)                       # each rank: its own samples
    x = torch.randn(8, 37)
    flat.zero_grad()
    model(x).pow(2).mean().backward()
    local = flat.g.clone()
    allreduce_mean_(flat.g, world, bucket_elems=100)    # several buckets
    flat.g.div_(world)
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    ok = torch.allclose(flat.g, sum(gathered) / world, atol=1e-7)
    # every parameter's .grad still aliases the flat buffer; names unchanged
    ok = ok and all(p.grad.data_ptr() == flat.g.data_ptr() + 4 * off for p, (off, _) in zip(flat.params, flat.offsets))
    ok = ok and names_before == list(model.state_dict().keys())
    # the overlapped form (hooks launch the bucket all-reduces during backward) gives the same averaged gradient
    from efgh_amd.train import OverlappedAllReduce
    torch.manual_seed(0)
    model2 = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.ReLU(), torch.nn.Linear(19, 5))
    flat2 = FlatParams(model2)
    comm = OverlappedAllReduce(flat2, world, bucket_elems=300)
    assert len(comm.buckets) >= 2 and sum(e - s for s, e in comm.buckets) == flat2.n
    for step in range(2):                                # hooks re-arm every step
        flat2.zero_grad()
        comm.start_step()
        model2(x).pow(2).mean().backward()
        comm.finish()
        assert comm.order == sorted(comm.order, reverse=True) and len(comm.order) == len(comm.buckets)      # fixed issue order
        flat2.g.div_(world)
        ok = ok and torch.allclose(flat2.g, flat.g, atol=1e-7)
        ok = ok and all(p.grad.data_ptr() == flat2.g.data_ptr() + 4 * off for p, (off, _) in zip(flat2.params, flat2.offsets))
    q.put((rank, bool(ok), flat.g.sum().item()))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
    assert all(ok for _, ok, _ in res), res
    assert abs(res[0][2] - res[1][2]) < 1e-5            # both ranks hold the same averaged gradient
