"""Which host lines launch the non-HIP-extension kernels of a training step?
    python tools/glue_census.py [--small] [--batch B]
One training step under a TorchDispatchMode: every aten op on device tensors that is not a pure view / allocation is charged to
the innermost frame inside efgh_amd/ that issued it (ops of autograd's own backward formulas have no such frame).  Output:
calls per (file:line, aten op), most frequent first, and the total.  (The extension's own kernels go through ctypes and do not
appear as aten ops.)  `census()` is also used by tests/test_gpu_train.py::test_training_step_is_not_torch_glue."""
import argparse
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_KERNEL = ('view', 'reshape', 'slice', 'select', 'expand', 'permute', 'transpose', 'unsqueeze', 'squeeze', 'detach',
             'alias', 'as_strided', 'empty', 'new_empty', 'unbind', 'split', 'narrow', 't.default', 'size', 'stride',
             'is_', 'lift_fresh', '_local_scalar_dense', 'unfold', 'chunk', 'diagonal', 'numel', '_version',
             'record_stream', 'resize_', 'set_')


SHAPES = collections.defaultdict(list)
BIG = collections.Counter()      # (where, op) -> MB of the largest operand, summed over calls (ops on >= 16 M elements only)


def census(step_fn):
    """run step_fn() under a TorchDispatchMode; Counter{(innermost efgh_amd frame, aten op): calls} of the ops on device tensors
    that are not pure views / allocations"""
    from torch.utils._python_dispatch import TorchDispatchMode
    count = collections.Counter()

    class Census(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = str(func)
            flat = [t for t in torch.utils._pytree.tree_leaves((args, kwargs, out)) if isinstance(t, torch.Tensor)]
            if any(t.is_cuda for t in flat) and not any(k in name for k in NO_KERNEL):
                where = 'autograd engine / outside efgh_amd'
                for fr in reversed(traceback.extract_stack()):
                    if '/efgh_amd/' in fr.filename:
                        where = '%s:%d %s' % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.name)
                        break
                count[(where, name)] += 1
                big = max((t.numel() for t in flat), default=0)
                if big >= (1 << 24):
                    BIG[(where, name)] += big * 4 / 1e6
                    SHAPES[(where, name)].append(tuple(max(flat, key=lambda t: t.numel()).shape))
            return out

    with Census():
        step_fn()
    torch.cuda.synchronize()
    return count


def main():
    sys.path.insert(0, ROOT)
    from efgh_amd import synthetic as syn
    from efgh_amd.losses import EFGHCriterion
    from efgh_amd.nets import EFGHBackbone
    from efgh_amd.train import Trainer
    ap = argparse.ArgumentParser()
    ap.add_argument('--small', action='store_true')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--top', type=int, default=60)
    a = ap.parse_args()
    raw, npts = ((128, 256), 2048) if a.small else ((768, 2560), 131072)
    dev = torch.device('cuda', 0)
    args = syn.default_args(raw, 'cuda')
    torch.manual_seed(0)
    model = EFGHBackbone(args).to(dev)
    batch = syn.make_batch(raw, npts, a.batch, first_seed=0)
    inp = [torch.from_numpy(batch[k]).to(dev) for k in ('pc', 'img', 'calib', 'A')]
    gt = {k: torch.from_numpy(v).to(dev) for k, v in batch['gt'].items()}
    tr = Trainer(model, EFGHCriterion(args), lr=1e-4)
    for _ in range(2):
        tr.step(*inp, gt)
    torch.cuda.synchronize()
    count = census(lambda: tr.step(*inp, gt))
    total = sum(count.values())
    print('aten ops on device tensors in one training step (about one launch each): %d' % total)
    for (where, name), n in count.most_common(a.top):
        print('%5d  %-36s %s' % (n, name.replace('aten.', ''), where))
    print('ops on large tensors (MB of the largest operand, summed):')
    for (where, name), mb in BIG.most_common(30):
        print('%9.0f MB  x%-3d %-30s %s' % (mb, count[(where, name)], name.replace('aten.', ''), where))
        print('              shapes:', collections.Counter(SHAPES[(where, name)]).most_common(12))


if __name__ == '__main__':
    main()
