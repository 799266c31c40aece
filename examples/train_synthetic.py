"""End-to-end use of the pieces on synthetic frames: GPU sample preparation (efgh_amd.data, the reference's
`ProcessKITTIODOM` call), EFGHBackbone, EFGHCriterion, the fused-Adam Trainer and the device-side error meter — the
loop of the reference's iterater.py:25-60 with nothing on the CPU between the decoded frame and the optimizer step.

    python examples/train_synthetic.py --iters 3 --raw 128 256 --points 2048
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efgh_amd import synthetic as syn  # noqa: E402
from efgh_amd.common.metrics import Err  # noqa: E402
from efgh_amd.data import ProcessKITTIODOM  # noqa: E402
from efgh_amd.losses import EFGHCriterion  # noqa: E402
from efgh_amd.nets import EFGHBackbone  # noqa: E402
from efgh_amd.train import Trainer  # noqa: E402


def raw_frame(raw_hw, seed):
    """a decoded camera frame a bit larger than the network's raw size, and an unfiltered sweep (n,4)"""
    rs = np.random.RandomState(seed)
    img = rs.randint(0, 256, size=(raw_hw[0] + 8, raw_hw[1] + 20, 3)).astype(np.uint8)
    return img


def collate(samples, device):
    pc = torch.stack([s[0] for s in samples]).float()
    img = torch.stack([s[1] for s in samples]).float()
    calib = torch.from_numpy(np.stack([s[2] for s in samples])).float().to(device)
    A = torch.from_numpy(np.stack([s[3] for s in samples])).float().to(device)
    gt = {}
    for k in samples[0][4]:
        v = [s[4][k] for s in samples]
        gt[k] = torch.stack(v) if torch.is_tensor(v[0]) else torch.from_numpy(np.stack(v))
    return pc, img, calib, A, gt


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=3)
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--raw', type=int, nargs=2, default=[128, 256])
    ap.add_argument('--points', type=int, default=2048)
    a = ap.parse_args(argv)
    raw = tuple(a.raw)
    args = syn.default_args(raw, 'cuda')
    args.update({'lidar_line': None, 'num_points': a.points, 'test': False,
                 'dclb': {'l_rot_range': 1 / 12., 'l_trs_range': 1.0, 'c_rot_range': 1 / 12.}})
    torch.manual_seed(0)
    model = EFGHBackbone(args).cuda()
    trainer = Trainer(model, EFGHCriterion(args), lr=1e-4)
    prep = ProcessKITTIODOM(args)
    err = Err(args['dataset'])
    calib0, _ = syn.calib_and_A(raw)
    P2 = np.eye(4); P2[:3] = calib0
    hist = []
    for it in range(a.iters):
        samples = []
        for b in range(a.batch):
            seed = it * a.batch + b
            sweep = np.concatenate([syn.lidar_sweep(a.points * 2, seed).T, np.ones((a.points * 2, 1), np.float32)], 1)
            samples.append(prep(sweep, raw_frame(raw, seed), {'P2': P2, 'Tr': np.eye(4)}, np.eye(4), 'f%06d' % seed)[:5])
        pc, img, calib, A, gt = collate(samples, 'cuda')
        losses, pred = trainer.step(pc, img, calib, A, gt)
        err.update({'sensor2_T_sensor1': gt['sensor2_T_sensor1'].float().cuda()}, pred)
        hist.append(float(losses['total'].detach()))
        print('iter %d  total %.4f  %s' % (it, hist[-1], '  '.join('%s %.3f' % kv for kv in err.dict.items())))
    return hist


if __name__ == '__main__':
    main()
