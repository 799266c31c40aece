"""Training step around the hot path (reference iterater.py:35-43, main.py:127,181-183):
forward -> efghloss -> backward -> gradient all-reduce over RCCL -> fused Adam.

One process per GPU.  All parameters (and their gradients) are views into ONE flat fp32 buffer each,
so the data-parallel exchange is a bucketed all-reduce of 191 MB whose buckets are launched from
parameter hooks while backward is still running, and the optimizer is one kernel launch instead of 353.
`DataParallel` semantics (one loss over the global batch, batch-mean terms) == mean of the per-rank
gradients for equal per-rank batches."""
import os

import torch
import torch.distributed as dist

from . import _C, ops


class FlatParams:
    """Re-homes every parameter of `model` (and its .grad) inside flat buffers, keeping names/shapes."""

    def __init__(self, model):
        self.frozen = tuple(bool(p.requires_grad) for p in model.parameters())      # checked by Trainer.step
        self.all_params = list(model.parameters())
        ps = [p for p in model.parameters() if p.requires_grad]
        dev, n = ps[0].device, sum(p.numel() for p in ps)
        self.n = n
        self.w = torch.cat([p.data.reshape(-1).float() for p in ps])          # (one launch, not one copy per parameter)
        self.g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.offsets = []
        off = 0
        for p in ps:
            k = p.numel()
            p.data = self.w[off:off + k].view(p.shape)
            p.grad = self.g[off:off + k].view(p.shape)
            self.offsets.append((off, k))
            off += k
        self.params = ps
        # gradients written straight into the flat slices by the layer backward (nets/fn.py `claim_grad`) instead of handed to
        # autograd's AccumulateGrad (which costs an allocation, a copy and an `add` launch per parameter).  `arrived[i]` is set
        # by whichever route delivers parameter i's gradient first; later deliveries in the same step go through autograd and
        # accumulate, so shared parameters and gradient accumulation over several backward calls keep their meaning.
        self.direct = True
        # BatchNorm's `num_batches_tracked` counters re-homed as views of ONE int64 vector: a training forward notes which layers
        # ran (GemmLayerFn.forward -> tick) and Trainer.step adds the whole step's counts with one launch instead of one per layer
        bns = [m for m in model.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)
               and m.num_batches_tracked is not None]
        self.nbt = torch.stack([m.num_batches_tracked.to(dev) for m in bns]).clone() if bns else None
        self.ticks, self.collect_ticks = [0] * len(bns), False
        self._bns = bns
        for j, m in enumerate(bns):
            m.num_batches_tracked = self.nbt[j]
            m._efgh_nbt = (self, j)
        self.arrived = [False] * len(ps)
        # uses[i]: how many layers of the current forward consume parameter i (nets/fn.py note_use).  A parameter with more than
        # one consumer is never claimed: all of its contributions go through autograd, which sums them and fires the
        # post-accumulate hook once - a direct write by the first consumer (on the weight-gradient stream) would be unordered
        # with the accumulation of the second, and would count the all-reduce bucket down before the second has arrived.
        self.uses = [0] * len(ps)
        self.listeners = []               # callables(index), e.g. the overlapped all-reduce's bucket countdown
        # content epoch of the flat weight buffer (FusedAdam rewrites it through raw pointers): the packed-weight / folded-BN caches
        # of THESE parameters key on it, and their batched repack is registered on it (ops.Epoch)
        self.epoch = ops.Epoch()
        ops.HOLDER_GEN[0] += 1
        for i, p in enumerate(ps):
            p._efgh_flat = (self, i)
            p._efgh_epoch = self.epoch
            p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _make_hook(self, i):
        def hook(param):
            self.deliver(i)
        return hook

    def deliver(self, i, stream=None):
        """parameter i's gradient for this backward pass is (enqueued to be) in the flat buffer; `stream`: the stream the write was
        enqueued on when that is not the current one (weight gradients on the weight-gradient stream, nets/fn.py)"""
        first, self.arrived[i] = not self.arrived[i], True
        if first:
            for fn in self.listeners:
                fn(i, stream)

    def claim(self, p, i):
        """the flat gradient view of parameter i if a backward may overwrite it now (zeroed, nothing delivered yet), else None"""
        if not self.direct or self.arrived[i] or self.uses[i] > 1:
            return None
        off, k = self.offsets[i]
        if p.grad is None or p.grad.data_ptr() != self.g.data_ptr() + 4 * off:
            return None
        return p.grad

    def tick(self, j):
        self.ticks[j] += 1

    def flush_ticks(self):
        self.collect_ticks = False
        if any(self.ticks):
            # a model.to() / .double() / rebinding after the Trainer was built breaks the aliasing of num_batches_tracked with
            # self.nbt silently: count on the module's own buffer then
            base, item = self.nbt.data_ptr(), self.nbt.element_size()
            for j, m in enumerate(self._bns):
                if m.num_batches_tracked.data_ptr() != base + j * item:
                    if self.ticks[j]:
                        m.num_batches_tracked += self.ticks[j]
                    m._efgh_nbt = None
                    self.ticks[j] = 0
            if not any(self.ticks):
                return
            k = self.ticks[0]
            if all(t == k for t in self.ticks):
                self.nbt += k
            else:
                self.nbt += torch.tensor(self.ticks, dtype=self.nbt.dtype).to(self.nbt.device)
            self.ticks = [0] * len(self.ticks)

    def zero_grad(self):
        self.g.zero_()
        self.arrived = [False] * len(self.params)
        for p, (off, k) in zip(self.params, self.offsets):      # autograd may have replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.g.data_ptr() + 4 * off:
                p.grad = self.g[off:off + k].view(p.shape)


def allreduce_mean_(flat_g, world, bucket_elems=8 * 1024 * 1024):
    """sum all-reduce in ~32 MB buckets (fully connected xGMI: large messages, few of them); the 1/world
    factor is folded into the optimizer kernel.  No-op for world == 1."""
    if world <= 1:
        return []
    works = []
    for s in range(0, flat_g.numel(), bucket_elems):
        works.append(dist.all_reduce(flat_g[s:s + bucket_elems], op=dist.ReduceOp.SUM, async_op=True))
    for w in works:
        w.wait()
    return works


class OverlappedAllReduce:
    """Gradient all-reduce overlapped with backward: the flat gradient buffer is cut into ~32 MB buckets of
    whole parameters; every delivered parameter gradient (FlatParams.deliver: from
    autograd's post-accumulate hook or from a layer backward that wrote the flat slice itself) counts its bucket down and launches the bucket's asynchronous
    sum all-reduce (RCCL on its own stream) as soon as the last gradient of the bucket has been written - backward produces
    the gradients back to front, so the tail buckets are on the wire while the front of the network is still being
    differentiated.  `finish()` launches whatever is left (parameters without a gradient) and waits."""

    def __init__(self, flat, world, bucket_elems=8 * 1024 * 1024):
        self.flat, self.world = flat, world
        self.buckets = []                 # [start, end) element ranges of the flat buffer
        self.bucket_of = []               # parameter index -> bucket index
        s = 0
        for i, (off, k) in enumerate(flat.offsets):
            if off + k - s > bucket_elems and off > s:
                self.buckets.append((s, off))
                s = off
            self.bucket_of.append(len(self.buckets))
        self.buckets.append((s, flat.n))
        self.sizes = [0] * len(self.buckets)
        for b in self.bucket_of:
            self.sizes[b] += 1
        self.pending, self.works, self.launched = list(self.sizes), [], [False] * len(self.buckets)
        self.next_b, self.order = len(self.buckets) - 1, []
        self.main_stream = None           # the stream backward() is called on (set by start_step)
        self.written = [[] for _ in self.buckets]         # per bucket: events recorded behind the gradient writes into it
        if world > 1:
            flat.listeners.append(self._arrived)

    def _arrived(self, i, stream=None):
        b = self.bucket_of[i]
        if self.flat.g.is_cuda:
            # an event right behind the write, on the stream that carries it: the bucket's all-reduce waits for exactly the work
            # that produced its gradients, not for everything else that happens to be enqueued on the branch streams
            ev = torch.cuda.Event()
            ev.record(stream if stream is not None else torch.cuda.current_stream())
            self.written[b].append(ev)
        self.pending[b] -= 1
        self._launch_ready()

    def _launch_ready(self):
        """collectives must be issued in the SAME order on every rank (RCCL matches them by issue order, not by buffer): buckets
        go out strictly from the last to the first - the order backward completes them in - and a bucket that becomes complete
        early waits for its successors, so the order cannot depend on how autograd happens to schedule a rank's hooks"""
        while self.next_b >= 0 and self.pending[self.next_b] <= 0:
            self._launch(self.next_b)
            self.next_b -= 1

    def _launch(self, b):
        if self.launched[b]:
            return
        self.launched[b] = True
        if self.flat.g.is_cuda:
            cur = torch.cuda.current_stream()     # (a hook may run with any branch stream current)
            if self.pending[b] <= 0:
                for ev in self.written[b]:        # every parameter of the bucket was delivered: wait for those writes only
                    cur.wait_event(ev)
            else:                                 # finish(): parameters without a gradient this step - join everything
                for st in ops.side_streams() + ([self.main_stream] if self.main_stream is not None else []):
                    if st != cur:
                        cur.wait_stream(st)
        s, e = self.buckets[b]
        self.order.append(b)
        self.works.append(dist.all_reduce(self.flat.g[s:e], op=dist.ReduceOp.SUM, async_op=True))

    def start_step(self):
        self.main_stream = torch.cuda.current_stream() if self.flat.g.is_cuda else None
        self.pending, self.works, self.launched = list(self.sizes), [], [False] * len(self.buckets)
        self.next_b, self.order = len(self.buckets) - 1, []
        self.written = [[] for _ in self.buckets]

    def finish(self):
        if self.world <= 1:
            return
        for b in range(len(self.buckets) - 1, -1, -1):          # whatever is left (parameters without a gradient), same order
            self._launch(b)
        for w in self.works:
            w.wait()


class FusedAdam:
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay) on a FlatParams, one HIP launch."""

    def __init__(self, flat, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.flat, self.lr, self.betas, self.eps, self.wd = flat, lr, betas, eps, weight_decay
        self.m = torch.zeros_like(flat.w)
        self.v = torch.zeros_like(flat.w)
        self.t = 0

    def step(self, grad_scale=1.0):
        _C.require_cuda(self.flat.w)
        self.t += 1
        f = self.flat
        _C.check(_C.lib().efgh_adam_step(_C.ptr(f.w), _C.ptr(f.g), _C.ptr(self.m), _C.ptr(self.v), _C.c_int64(f.n),
                                         _C.c_float(self.lr), _C.c_float(self.betas[0]), _C.c_float(self.betas[1]),
                                         _C.c_float(self.eps), _C.c_float(self.wd), _C.c_int32(self.t),
                                         _C.c_float(grad_scale), _C.stream_ptr()))
        ops.bump_epoch(self.flat.epoch)            # packed-weight / folded-BN caches are stale now


def adjust_learning_rate(base_lr, it, every=50000, gamma=0.7):
    """common/helper.py:28-38"""
    return base_lr * (gamma ** (it // every))


class Trainer:
    """one data-parallel training step; `world`/`rank` from torch.distributed when initialised."""

    def __init__(self, model, criterion, lr=1e-4, weight_decay=0.0):
        self.model, self.criterion = model, criterion
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.flat = FlatParams(model)
        if self.world > 1:                                  # identical start on every rank: trainable, FROZEN and buffers
            dist.broadcast(self.flat.w, 0)
            flat_ids = {id(p) for p in self.flat.params}
            for t in list(model.parameters()) + list(model.buffers()):
                if id(t) not in flat_ids:
                    dist.broadcast(t.data, 0)
        self.opt = FusedAdam(self.flat, lr=lr, weight_decay=weight_decay)
        self.comm = OverlappedAllReduce(self.flat, self.world)
        if self.world > 1:
            ops.reserve_comm_queue()
        self.base_lr, self.it = lr, 0

    def load_checkpoint(self, ckpt):
        """resume from a checkpoint in the reference's layout (common/helper.py:40-61, main.py:149-160,190-198): model state,
        Adam moments and the iteration counter, so that the step-wise decay 0.7^(iter // 50000) continues where it stopped.
        (The reference re-evaluates the schedule once per iterater() call, iterater.py:21, i.e. per epoch; here it is evaluated
        every step from the same formula - the learning rate changes at iteration 50000*k exactly instead of at the next epoch
        boundary.)"""
        from .io import checkpoint as ck
        if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, '__fspath__'):
            ckpt = torch.load(ckpt, map_location='cpu')
        ck.load_model_state(self.model, ckpt)
        if 'optimizer' in ckpt:
            ck.load_adam_state(self.opt, ckpt['optimizer'])
        self.it = int(ckpt.get('iter', -1)) + 1
        ops.bump_epoch(self.flat.epoch)
        ops.bump_epoch()
        return self.it

    def step(self, pc, img, calib, A, gt):
        if tuple(bool(p.requires_grad) for p in self.flat.all_params) != self.flat.frozen:
            raise _C.EfghError('the set of trainable parameters changed after the Trainer was built (FlatParams snapshots '
                               'requires_grad): freeze parameters first, then construct the Trainer')
        self.opt.lr = adjust_learning_rate(self.base_lr, self.it)
        ops.w2v_clear()
        self.flat.uses = [0] * len(self.flat.params)
        self.model.train()
        self.flat.collect_ticks = True
        try:
            pred = self.model(pc, img, calib, A)
        finally:
            self.flat.flush_ticks()
        losses, gt = self.criterion.compute_loss(pc, img, calib, A, gt, pred)
        self.flat.zero_grad()
        self.comm.start_step()
        losses['total'].backward()            # bucket all-reduces start from the parameter hooks during this call
        # the point branch runs on a side stream (nets/efghbackbone.py) and so does its backward; autograd joins the streams of the
        # AccumulateGrad nodes it ran, but gradients written directly into the flat buffer have no such node: join explicitly
        for s in ops.side_streams():
            torch.cuda.current_stream().wait_stream(s)
        self.comm.finish()
        self.opt.step(grad_scale=1.0 / self.world)
        self.it += 1
        return losses, pred
