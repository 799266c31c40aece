"""Which host lines launch the non-HIP-extension kernels of a training step?
    python tools/glue_census.py [--small] [--batch B]
One training step under a TorchDispatchMode: every aten op on device tensors that is not a pure view / allocation is charged to
the innermost frame inside efgh_amd/ that issued it (ops of autograd's own backward formulas have no such frame).  Output: launches per (file:line, aten op), most frequent first, and the total.
(The extension's own kernels go through ctypes and do not appear as aten ops.)"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from efgh_amd import synthetic as syn                       # noqa: E402
from efgh_amd.losses import EFGHCriterion                   # noqa: E402
from efgh_amd.nets import EFGHBackbone                      # noqa: E402
from efgh_amd.train import Trainer                          # noqa: E402


def main():
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
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    count = collections.Counter()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    NO_KERNEL = ('view', 'reshape', 'slice', 'select', 'expand', 'permute', 'transpose', 'unsqueeze', 'squeeze', 'detach',
                 'alias', 'as_strided', 'empty', 'new_empty', 'unbind', 'split', 'narrow', 't.default', 'size', 'stride',
                 'is_', 'lift_fresh', '_local_scalar_dense', 'unfold', 'chunk', 'diagonal', 'numel', '_version',
                 'record_stream', 'resize_', 'set_', '_to_copy.default_cpu')

    class Census(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = str(func)
            flat = [t for t in torch.utils._pytree.tree_leaves((args, kwargs, out)) if isinstance(t, torch.Tensor)]
            if any(t.is_cuda for t in flat) and not any(k in name for k in NO_KERNEL):
                where = 'autograd engine / outside efgh_amd'
                for fr in reversed(traceback.extract_stack()):
                    if '/efgh_amd/' in fr.filename:
                        where = '%s:%d %s' % (fr.filename.replace(root + '/', ''), fr.lineno, fr.name)
                        break
                count[(where, name)] += 1
            return out

    with Census():
        tr.step(*inp, gt)
    torch.cuda.synchronize()
    dur = collections.Counter()
    total = sum(count.values())
    print('aten ops on device tensors in one training step (about one launch each): %d' % total)
    for (where, name), n in count.most_common(a.top):
        print('%5d  %-36s %s' % (n, name.replace('aten.', ''), where))


if __name__ == '__main__':
    main()
