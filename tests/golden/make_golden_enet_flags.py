"""Golden outputs of the reference's E net (nets/enet.py, nets/bilateralNN.py, UNMODIFIED, run on the CPU of this container through
ref_harness.py) for the values of its three switches that the shipped configurations do not use: use_leaky = False (ReLU instead of
LeakyReLU(0.1) in conv_in and behind last_relu, net_utils.py:11), bcn_use_norm = False (no density normalisation of the splat,
bilateralNN.py:196-211), last_relu = True (an activation behind the last blur convolution, bilateralNN.py:121-135).  Eval forward,
train-mode forward and the gradient norms of a fixed scalar loss.  Run:  python tests/golden/make_golden_enet_flags.py
-> tests/golden/enet_flags.npz (data only; inputs and weights are regenerated from seeds by efgh_amd.synthetic)."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import ref_harness as rh            # noqa: E402
from efgh_amd import synthetic as syn  # noqa: E402

torch.set_num_threads(1)
nets, losses, tu = rh.import_reference()
from nets.enet import Enet          # noqa: E402

VARIANTS = {'relu': {'use_leaky': False}, 'nonorm': {'bcn_use_norm': False}, 'lastrelu': {'last_relu': True},
            'all': {'use_leaky': False, 'bcn_use_norm': False, 'last_relu': True}}
N = 2048


def main():
    man = json.load(open(os.path.join(HERE, 'state_dict_manifest.json')))['state_dict']
    sd = {k[2:]: v for k, v in syn.synthetic_state_dict(man, seed=1).items() if k.startswith('E.')}
    pc = torch.from_numpy(syn.lidar_sweep(N, 3))[None]
    store = {}
    for tag, over in VARIANTS.items():
        args = dict(rh.default_args((128, 256)), **over)
        m = Enet(args)
        m.load_state_dict(sd, strict=True)
        m.eval()
        with torch.no_grad():
            r = m(pc)
        for k in ('e_gn_abs', 'e_gn_sgn', 'e_l'):
            store[f'{tag}.eval.{k}'] = r[k].numpy()
        m.load_state_dict(sd, strict=True)
        m.train()
        r = m(pc)
        for k in ('e_gn_abs', 'e_gn_sgn'):
            store[f'{tag}.train.{k}'] = r[k].detach().numpy()
        w1 = torch.linspace(-1, 1, r['e_gn_sgn'].numel()).view_as(r['e_gn_sgn'])
        w2 = torch.linspace(1, 2, r['e_gn_abs'].numel()).view_as(r['e_gn_abs'])
        m.zero_grad()
        ((r['e_gn_sgn'] * w1).sum() + (r['e_gn_abs'] * w2).sum()).backward()
        names, gn = [], []
        for name, p in m.named_parameters():
            names.append(name)
            gn.append(0.0 if p.grad is None else p.grad.double().norm().item())
            if name in ('conv_in.0.0.weight', 'bcn1.blur_conv.0.weight', 'bcn3.blur_conv.2.bias', 'lin_gn_abs.weight'):
                store[f'{tag}.grad.{name}'] = p.grad.numpy()
        store[f'{tag}.grad_norm'] = np.array(gn)
        print(tag, {k: float(np.abs(v).max()) for k, v in store.items() if k.startswith(tag + '.eval')})
    store['param_names'] = np.array(names)
    np.savez_compressed(os.path.join(HERE, 'enet_flags.npz'), **store)


if __name__ == '__main__':
    main()
