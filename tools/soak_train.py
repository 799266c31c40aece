"""soak: full-size training with DIFFERENT synthetic frame-pairs every step (the lattice sizes change, so the speculative pyramid
build has to cope: capacities from the previous step + 25 %, overflow -> exact rebuild), checks finiteness, memory and step time
    python tools/soak_train.py [--steps 40] [--batch 8]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efgh_amd import lattice, synthetic as syn               # noqa: E402
from efgh_amd.losses import EFGHCriterion                   # noqa: E402
from efgh_amd.nets import EFGHBackbone                      # noqa: E402
from efgh_amd.train import Trainer                          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=40)
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--npts', type=int, nargs='+', default=[131072])
a = ap.parse_args()
raw = (768, 2560)
args = syn.default_args(raw, 'cuda')
torch.manual_seed(0)
model = EFGHBackbone(args).cuda()
tr = Trainer(model, EFGHCriterion(args), lr=1e-4)
times, mem0 = [], None
for it in range(a.steps):
    npts = a.npts[it % len(a.npts)]
    b = syn.make_batch(raw, npts, a.batch, first_seed=it * a.batch)
    inp = [torch.from_numpy(b[k]).cuda() for k in ('pc', 'img', 'calib', 'A')]
    gt = {k: torch.from_numpy(v).cuda() for k, v in b['gt'].items()}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    L, _ = tr.step(*inp, gt)
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
    tot = float(L['total'])
    assert tot == tot and abs(tot) < 1e9, (it, tot)
    if it == 3:
        mem0 = torch.cuda.memory_allocated()
    if os.environ.get('EFGH_SOAK_EXACT'):
        print('exact %d %r' % (it, tot))
    if it % 5 == 0 or it == a.steps - 1:
        print('step %3d  npts %6d  total %.2f  %.1f ms  alloc %.1f GB  peak %.1f GB  lattice sizes %s' % (
            it, npts, tot, times[-1], torch.cuda.memory_allocated() / 1e9, torch.cuda.max_memory_allocated() / 1e9,
            list(lattice._SIZES.values())[-1]))
print('median step %.1f ms, max after warm-up %.1f ms; allocated now vs step 3: %+.2f GB' % (
    sorted(times[3:])[len(times[3:]) // 2], max(times[3:]), (torch.cuda.memory_allocated() - mem0) / 1e9))
