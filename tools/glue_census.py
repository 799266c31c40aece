"""Which host lines launch the non-HIP-extension kernels of a training step?
    python tools/glue_census.py [--small] [--batch B]
One training step under torch.profiler (with_stack): every aten op that launches a device kernel is charged to the innermost
frame inside efgh_amd/ that issued it.  Output: launches per (file:line, aten op), most frequent first, and the total.
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
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        tr.step(*inp, gt)
        torch.cuda.synchronize()
    count = collections.Counter()
    dur = collections.Counter()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
            continue
        if ev.cpu_parent is not None and ev.cpu_parent.kernels:       # count the kernel once, at the outermost aten op
            continue
        where = 'autograd engine / outside efgh_amd'
        for fr in ev.stack:
            if 'efgh_amd/' in fr or 'bench.py' in fr or 'tools/' in fr:
                where = fr.replace(root + '/', '').strip()
                break
        key = (where, ev.name)
        count[key] += len(ev.kernels)
        dur[key] += sum(k.duration for k in ev.kernels)
    total = sum(count.values())
    print('torch-launched kernels in one training step: %d  (%.2f ms of device time)' % (total, sum(dur.values()) / 1e3))
    for (where, name), n in count.most_common(a.top):
        print('%5d  %8.1f us  %-28s %s' % (n, dur[(where, name)], name, where))


if __name__ == '__main__':
    main()
