"""Zero-edit drop-in launcher: runs the reference's UNMODIFIED entry script on the MI355X path.

    python -m efgh_amd.run [--device I | --all-devices] main.py configs/train_rellis.yaml        # one GPU
    python -m efgh_amd.run --gpus N main.py configs/train_rellis.yaml                            # N GPUs, one process each

The reference binds its model and criterion by module name (`import nets`, `import losses`, main.py:14-15;
`nets.__dict__[arch + 'Backbone']`, `losses.__dict__[arch + 'Criterion']`, main.py:126,129).  This launcher installs
`efgh_amd.nets` / `efgh_amd.losses` under those two names in `sys.modules`, puts the script's directory at the front of
`sys.path` (what `python main.py` does, so `data_loader`, `common`, `iterater`, `valid`, `test` stay the reference's own)
and `runpy`-runs the script IN THIS PROCESS as `__main__` - before anything has touched the GPU, never an exec.

`torch.nn.DataParallel(model)` (main.py:127) spans every visible device; this path is one process per GPU, so unless
`--all-devices` is given the process is pinned to ONE device first (`--device I`, default 0, counted in the list that is
visible now): DataParallel then has `device_ids == [0]` and calls the module directly.  With several devices left visible
the backbone refuses a replica forward loudly (nets/efghbackbone.py).

`--gpus N` is the data-parallel form of the same drop-in (the reference's multi-GPU mode IS `DataParallel`): the
parent starts N children (no GPU touched in the parent), child r is pinned to device r, joins a process group (RCCL; gloo when
ranks share a GPU) and runs the unmodified script with three names rebound for the duration of the run:
  * `torch.nn.DataParallel` -> `ProcessDataParallel` (below): same `.module` / `module.`-prefixed state_dict, identical start
    on every rank (broadcast), and the gradients of all ranks averaged in buckets right before every `optimizer.step()`;
  * `torch.utils.data.DataLoader`: a loader without a sampler gets a DistributedSampler (its rank's share of every epoch,
    reshuffled per epoch when the script asked for shuffling) and batch_size / N samples per step (at least 1), so that the
    GLOBAL batch is what the config says - what DataParallel's scatter does;
  * `torch.save` on ranks > 0 is a no-op (rank 0 writes the checkpoints, as device 0's replica does under DataParallel).
BatchNorm statistics stay per rank, as per replica under DataParallel.

What makes the reference's REAL main.py safe under N processes (round 5; every rank runs the whole script):
  * only the loader the script asked to SHUFFLE - the training loader, main.py:84-91 - is sharded; validation / test loaders
    (`shuffle=False`, main.py:98-121) stay whole on every rank, so `is_val_best` and every printed metric are computed from all
    of the data, as under DataParallel;
  * rank 0 goes through the script's pre-model section FIRST (the ckpt_dir prompt, wipe and config copy, main.py:45-75): the
    other ranks wait on the process group's store until rank 0 reaches its `DataParallel(model)` call, then run the same lines
    with `input()` answering yes and every file operation under `ckpt_dir` turned into a no-op - so nobody races on
    `os.path.exists(ckpt_dir)`, prompts on the shared stdin or deletes what rank 0 just wrote;
  * on ranks > 0 `shutil.copyfile/copy/copy2/move/rmtree` and `os.remove/unlink/rmdir` skip paths under the config's `ckpt_dir`
    (`save_checkpoint`, common/helper.py:40-61, copies and prunes checkpoint files that only rank 0 writes) and
    `tensorboardX.SummaryWriter` is a null writer;
  * a `test:` configuration is refused with `--gpus N` (test.py writes ONE prediction CSV / prints metrics over its own loader:
    that is a single-process job);
  * the parent polls its children: the first non-zero exit terminates the siblings and becomes the exit code, and the process
    group has a timeout (`EFGH_RUN_TIMEOUT_S`, default 1800 s), so a rank that skips a step's `optimizer.step()` - the
    reference's "CUDA out of memory: continue" path, iterater.py:108-116, taken on one rank only - ends the job with an error
    instead of hanging it: a skipped step must be skipped on ALL ranks.
"""
import os
import runpy
import sys

ALIASES = {'nets': 'efgh_amd.nets', 'losses': 'efgh_amd.losses'}


def pin_one_device(index=0):
    """restrict this process to one GPU.  Must run before the HIP runtime is initialised (importing torch does not)."""
    for var in ('HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        cur = os.environ.get(var)
        if cur:
            ids = [v for v in cur.split(',') if v.strip() != '']
            if index >= len(ids):
                raise SystemExit('efgh_amd.run: --device %d but %s=%s' % (index, var, cur))
            os.environ[var] = ids[index]
            return var, ids[index]
    os.environ['HIP_VISIBLE_DEVICES'] = str(index)
    return 'HIP_VISIBLE_DEVICES', str(index)


def install_aliases():
    """`import nets` / `import losses` resolve to the MI355X packages from here on (and `from nets.x import y` to their
    submodules of the same name, where they exist)"""
    import importlib
    out = {}
    for name, target in ALIASES.items():
        mod = importlib.import_module(target)
        sys.modules[name] = mod
        prefix = target + '.'
        for k, v in list(sys.modules.items()):
            if k.startswith(prefix) and v is not None:
                sys.modules[name + '.' + k[len(prefix):]] = v
        out[name] = mod
    return out


def install_loop_rebinds():
    """`torch.cuda.empty_cache()` at the end of EVERY iteration (iterater.py:106, valid.py:57, test.py:81,161) becomes a no-op for
    the duration of the run.  On the reference's eager path it papers over fragmentation; here a training step keeps ~10 GB of
    activations per sample in the caching allocator's pool on purpose (sized for 288 GB) and handing them back to
    hipFree / hipMalloc every step costs more than the step's own enqueue time (`bench.py` -> `config_r`: measured both ways).
    `EFGH_RUN_KEEP_EMPTY_CACHE=1` leaves the call alone.  The out-of-memory handlers (iterater.py:108-116) call it too: an OOM
    under this launcher is a real capacity limit, not fragmentation - the batch is skipped as in the reference."""
    if os.environ.get('EFGH_RUN_KEEP_EMPTY_CACHE', '0') == '1':
        return
    import torch
    torch.cuda.empty_cache = lambda: None


# ---------------------------------------------------------------------------------------------------------------------
# --gpus N: one process per GPU behind the reference's DataParallel call
# ---------------------------------------------------------------------------------------------------------------------
def _process_data_parallel_class():
    import torch
    import torch.distributed as dist
    import torch.nn as nn

    class ProcessDataParallel(nn.Module):
        """stands in for `torch.nn.DataParallel(model)` (main.py:127) when every GPU has its own process: holds the model as
        `.module` (so checkpoints keep their `module.` prefix, main.py:136,153), starts from rank 0's parameters and buffers,
        and averages the gradients over the ranks - flattened into ~32 MB buckets, one all-reduce each - right before every
        optimizer step (a global optimizer pre-step hook: the reference's loop calls `optimizer.step()` itself,
        iterater.py:41-43)."""
        BUCKET = 8 * 1024 * 1024

        def __init__(self, module, device_ids=None, output_device=None, dim=0):
            super().__init__()
            self.module = module
            self.device_ids, self.output_device, self.dim = [0], 0, dim
            self.world = dist.get_world_size() if dist.is_initialized() else 1
            if self.world > 1:
                if dist.get_rank() == 0:
                    _premodel_done()          # the ranks waiting in child_setup() may run the script's pre-model section now
                with torch.no_grad():
                    for t in list(module.parameters()) + list(module.buffers()):
                        dist.broadcast(t.data, 0)
                from torch.optim.optimizer import register_optimizer_step_pre_hook
                self._hook = register_optimizer_step_pre_hook(self._average_gradients)

        def forward(self, *args, **kwargs):
            return self.module(*args, **kwargs)

        def _average_gradients(self, optimizer, args, kwargs):
            mine = {id(p) for p in self.module.parameters()}
            ps = [p for g in optimizer.param_groups for p in g['params'] if id(p) in mine and p.grad is not None]
            if not ps:
                return
            if ps[0].is_cuda:
                # the branches of the network run their backward on side streams (nets/efghbackbone.py): join them first
                from . import ops
                for s in ops.side_streams():
                    torch.cuda.current_stream().wait_stream(s)
            i = 0
            while i < len(ps):
                j, n = i, 0
                while j < len(ps) and (n == 0 or n + ps[j].numel() <= self.BUCKET):
                    n += ps[j].numel()
                    j += 1
                flat = torch.cat([p.grad.reshape(-1) for p in ps[i:j]])
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                flat /= self.world
                o = 0
                for p in ps[i:j]:
                    p.grad.copy_(flat[o:o + p.numel()].view_as(p.grad))
                    o += p.numel()
                i = j

    return ProcessDataParallel


def install_process_parallel(rank, world, protected=()):
    """rebinds DataParallel / DataLoader / torch.save for a rank of a `--gpus N` run (see the module docstring)"""
    import torch
    import torch.utils.data as tud
    PDP = _process_data_parallel_class()
    torch.nn.DataParallel = PDP
    torch.nn.parallel.DataParallel = PDP
    RealLoader = tud.DataLoader

    class _EpochSampler(tud.distributed.DistributedSampler):
        """a DistributedSampler that moves to the next epoch's permutation by itself (the script never calls set_epoch)"""

        def __iter__(self):
            it = super().__iter__()
            self.set_epoch(self.epoch + 1)
            return it

    class ShardedLoader(RealLoader):
        def __init__(self, dataset, batch_size=1, shuffle=None, sampler=None, batch_sampler=None, **kw):
            # (only the loader the script asked to shuffle is the training loader, main.py:84-91; evaluation loaders stay whole)
            if world > 1 and shuffle and sampler is None and batch_sampler is None and not isinstance(dataset, tud.IterableDataset):
                sampler = _EpochSampler(dataset, num_replicas=world, rank=rank, shuffle=bool(shuffle), seed=0, drop_last=False)
                shuffle = None
                if batch_size is not None:
                    batch_size = max(1, int(batch_size) // world)
            super().__init__(dataset, batch_size=batch_size, shuffle=shuffle, sampler=sampler, batch_sampler=batch_sampler, **kw)

    tud.DataLoader = ShardedLoader
    torch.utils.data.DataLoader = ShardedLoader
    if rank != 0:
        torch.save = lambda *a, **k: None
        _guard_file_ops(protected)
        _null_summary_writer()
        import builtins
        builtins.input = lambda *a, **k: 'y'          # (query_yes_no, common/helper.py:63-93: rank 0 asked the user already)
    return PDP


_PREMODEL_KEY = 'efgh_run/premodel_done'


def _premodel_done():
    import torch.distributed as dist
    try:
        dist.distributed_c10d._get_default_store().set(_PREMODEL_KEY, '1')
    except Exception as e:                            # (a store without set/wait: nothing waits on it either)
        sys.stderr.write('efgh_amd.run: could not signal the waiting ranks: %r\n' % (e,))


def _wait_for_rank0_premodel(timeout_s):
    import datetime
    import torch.distributed as dist
    dist.distributed_c10d._get_default_store().wait([_PREMODEL_KEY], datetime.timedelta(seconds=timeout_s))


def _inside(path, roots):
    try:
        a = os.path.abspath(os.fspath(path))
    except TypeError:
        return False
    return any(a == r or a.startswith(r + os.sep) for r in roots)


def _guard_file_ops(protected):
    """ranks > 0: file operations whose TARGET lies under a protected directory (the config's ckpt_dir) do nothing - rank 0 is the
    only writer there (save_checkpoint's copies and pruning, common/helper.py:40-61; the ckpt_dir wipe, main.py:45-70)"""
    import shutil
    roots = [os.path.abspath(p) for p in protected if p]
    if not roots:
        return

    def skip_on(fn, which):
        def guarded(*a, **k):
            tgt = a[which] if len(a) > which else None
            if tgt is not None and _inside(tgt, roots):
                return tgt if which == 1 else None
            return fn(*a, **k)
        guarded.__name__ = getattr(fn, '__name__', 'guarded')
        return guarded
    for name in ('copyfile', 'copy', 'copy2', 'move'):
        setattr(shutil, name, skip_on(getattr(shutil, name), 1))        # (src, dst): the destination decides
    shutil.rmtree = skip_on(shutil.rmtree, 0)
    for name in ('remove', 'unlink', 'rmdir'):
        setattr(os, name, skip_on(getattr(os, name), 0))


def _null_summary_writer():
    try:
        import tensorboardX
    except Exception:
        return

    class NullWriter:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None
    tensorboardX.SummaryWriter = NullWriter


def script_config(rest):
    """the reference passes ONE yaml file as the script's first argument (main.py:31-32): {} when that is not what `rest` holds"""
    if not rest or not str(rest[0]).lower().endswith(('.yaml', '.yml')) or not os.path.isfile(rest[0]):
        return {}
    try:
        import yaml
        with open(rest[0]) as f:
            cfg = yaml.safe_load(f)
        return cfg if isinstance(cfg, dict) else {}
    except Exception:
        return {}


def spawn(gpus, argv_tail):
    """the parent of a `--gpus N` run: N children, nothing here touches the GPU"""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    # the devices this job may use, as the children will see them: rank r gets ids[r % len(ids)] as its ONLY device
    cur = os.environ.get('HIP_VISIBLE_DEVICES') or os.environ.get('CUDA_VISIBLE_DEVICES')
    if cur:
        ids = [v for v in cur.split(',') if v.strip() != '']
    else:
        import torch
        ids = [str(i) for i in range(torch.cuda.device_count())]          # (counting devices does not initialise the GPU)
    procs = []
    for r in range(gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), EFGH_RUN_CHILD='1', EFGH_RUN_SHARED='1' if gpus > len(ids) else '0')
        env.pop('CUDA_VISIBLE_DEVICES', None)
        if ids:
            env['HIP_VISIBLE_DEVICES'] = ids[r % len(ids)]
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC only on this pool (RCCL needs it)
        env.setdefault('OMP_NUM_THREADS', '8')
        procs.append(subprocess.Popen([sys.executable, '-m', 'efgh_amd.run'] + argv_tail, env=env))
    # poll: a rank that dies (an exception, sys.exit(1) in the script's loop) leaves the others blocked in a collective - the
    # first non-zero exit ends the job with that code
    import time
    rc, live = 0, list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0 and rc == 0:
                rc = r
                for q in live:
                    q.terminate()
                deadline = time.time() + 10.0
                for q in live:
                    try:
                        q.wait(max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
    return rc


def child_setup(rest=()):
    """a rank of a `--gpus N` run: pin the device, join the group, rebind the three names.  Returns (rank, world)"""
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    import torch
    import torch.distributed as dist
    # (the parent made this rank's device the only visible one; more ranks than devices = ranks share GPUs, a plumbing run)
    shared = os.environ.get('EFGH_RUN_SHARED') == '1' or not os.environ.get('HIP_VISIBLE_DEVICES')
    backend = os.environ.get('EFGH_DIST_BACKEND', 'gloo' if shared else 'nccl')
    import datetime
    timeout = datetime.timedelta(seconds=float(os.environ.get('EFGH_RUN_TIMEOUT_S', '1800')))
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', 0), timeout=timeout)
    else:
        dist.init_process_group(backend, timeout=timeout)
    cfg = script_config(rest)
    install_process_parallel(rank, world, protected=[cfg.get('ckpt_dir')] if isinstance(cfg.get('ckpt_dir'), str) else [])
    if rank != 0 and world > 1:
        # rank 0 first through the script's pre-model section (prompt, ckpt_dir wipe, config copy: main.py:45-75); released by its
        # DataParallel(model) call.  The wait covers a user thinking about the prompt; rank 0 dying ends the job from the parent.
        _wait_for_rank0_premodel(float(os.environ.get('EFGH_RUN_PREMODEL_WAIT_S', '86400')))
    return rank, world


def parse(argv):
    device, pin, gpus = 0, True, 1
    i = 0
    while i < len(argv):
        a = argv[i]
        if a in ('--device', '--gpus') and i + 1 < len(argv):
            if a == '--device':
                device = int(argv[i + 1])
            else:
                gpus = int(argv[i + 1])
            i += 2
        elif a.startswith('--device='):
            device = int(a.split('=', 1)[1])
            i += 1
        elif a.startswith('--gpus='):
            gpus = int(a.split('=', 1)[1])
            i += 1
        elif a == '--all-devices':
            pin = False
            i += 1
        elif a in ('-h', '--help'):
            print(__doc__)
            raise SystemExit(0)
        else:
            break
    if i >= len(argv):
        raise SystemExit('usage: python -m efgh_amd.run [--gpus N | --device I | --all-devices] <script.py> [script arguments...]')
    return device, pin, gpus, argv[i], argv[i + 1:]


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    device, pin, gpus, script, rest = parse(argv)
    if not os.path.isfile(script):
        raise SystemExit('efgh_amd.run: no such script: %s' % script)
    child = os.environ.get('EFGH_RUN_CHILD') == '1' and 'RANK' in os.environ
    if gpus > 1 and not child:
        if script_config(rest).get('test', False) not in (False, None):
            raise SystemExit('efgh_amd.run: --gpus %d with a `test:` configuration: evaluation (test.py:13-167) writes one prediction '
                             'CSV and prints metrics over its own loader - run it as a single process (drop --gpus)' % gpus)
        raise SystemExit(spawn(gpus, argv))
    if child:
        child_setup(rest)
    elif pin:
        pin_one_device(device)
    install_aliases()
    install_loop_rebinds()
    script = os.path.abspath(script)
    sys.argv = [script] + list(rest)
    sys.path.insert(0, os.path.dirname(script))
    try:
        runpy.run_path(script, run_name='__main__')
    finally:
        if child:
            import torch.distributed as dist
            if dist.is_initialized():
                if dist.get_rank() == 0:
                    _premodel_done()          # (a script that never wrapped a model: release the ranks that waited for it)
                dist.destroy_process_group()


if __name__ == '__main__':
    main()
