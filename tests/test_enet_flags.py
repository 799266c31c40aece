"""The E net's three switches at the values the shipped configurations do NOT use (use_leaky = False, bcn_use_norm = False,
last_relu = True; nets/enet.py:25-83, nets/bilateralNN.py:121-135,196-211, nets/net_utils.py:11): the oracle and the HIP path
against outputs of the unmodified reference (tests/golden/make_golden_enet_flags.py -> enet_flags.npz)."""
import os

import numpy as np
import pytest
import torch

from efgh_amd import synthetic as syn

N = 2048
VARIANTS = {'relu': {'use_leaky': False}, 'nonorm': {'bcn_use_norm': False}, 'lastrelu': {'last_relu': True},
            'all': {'use_leaky': False, 'bcn_use_norm': False, 'last_relu': True}}


def _rel(got, ref):
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12))


def _loss(r):
    w1 = torch.linspace(-1, 1, r['e_gn_sgn'].numel(), device=r['e_gn_sgn'].device).view_as(r['e_gn_sgn'])
    w2 = torch.linspace(1, 2, r['e_gn_abs'].numel(), device=r['e_gn_abs'].device).view_as(r['e_gn_abs'])
    return (r['e_gn_sgn'] * w1).sum() + (r['e_gn_abs'] * w2).sum()


@pytest.mark.parametrize('tag', list(VARIANTS))
def test_oracle_enet_flag_variants(golden_dir, manifest, tag):
    from oracle import efgh_oracle as O
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    G = np.load(os.path.join(golden_dir, 'enet_flags.npz'))
    args = dict(syn.default_args((128, 256), 'cpu'), **VARIANTS[tag])
    P = syn.synthetic_state_dict(manifest['state_dict'], 1)
    pc = torch.from_numpy(syn.lidar_sweep(N, 3))[None]
    with torch.no_grad():
        r = O.enet(P, pc, False, args=args)
    for k in ('e_gn_abs', 'e_gn_sgn', 'e_l'):
        assert _rel(r[k].numpy(), G[f'{tag}.eval.{k}']) < 2e-5, k
    names = [str(n) for n in G['param_names']]
    for n in names:
        P['E.' + n].requires_grad_(True)
    r = O.enet(P, pc, True, args=args)
    for k in ('e_gn_abs', 'e_gn_sgn'):
        assert _rel(r[k].detach().numpy(), G[f'{tag}.train.{k}']) < 5e-5, k
    _loss(r).backward()
    gn = np.array([0.0 if P['E.' + n].grad is None else P['E.' + n].grad.double().norm().item() for n in names])
    ref = G[f'{tag}.grad_norm']
    assert np.abs(gn - ref).max() <= 2e-4 * ref.max()
    for n in ('conv_in.0.0.weight', 'bcn1.blur_conv.0.weight', 'bcn3.blur_conv.2.bias', 'lin_gn_abs.weight'):
        assert _rel(P['E.' + n].grad.numpy(), G[f'{tag}.grad.{n}']) < 2e-3, n       # (first-layer gradients through five BCL levels: thread-count-dependent summation orders of torch's CPU kernels move them at the 5e-4 level)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', list(VARIANTS))
def test_hip_enet_flag_variants(golden_dir, manifest, tag):
    """the HIP E net honours the switches (round 5 refused everything but the shipped values): eval logits <= 1e-4 against the
    reference, train-mode logits and gradients against the reference's"""
    from efgh_amd.nets.enet import Enet
    G = np.load(os.path.join(golden_dir, 'enet_flags.npz'))
    args = dict(syn.default_args((128, 256), 'cuda'), **VARIANTS[tag])
    sd = {k[2:]: v for k, v in syn.synthetic_state_dict(manifest['state_dict'], 1).items() if k.startswith('E.')}
    m = Enet(args)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    pc = torch.from_numpy(syn.lidar_sweep(N, 3))[None].cuda()
    with torch.no_grad():
        r = m(pc)
    for k in ('e_gn_abs', 'e_gn_sgn', 'e_l'):
        assert _rel(r[k].cpu().numpy().reshape(G[f'{tag}.eval.{k}'].shape), G[f'{tag}.eval.{k}']) < 1e-4, k
    m.load_state_dict(sd, strict=True)
    m.train()
    r = m(pc)
    for k in ('e_gn_abs', 'e_gn_sgn'):
        assert _rel(r[k].detach().cpu().numpy().reshape(G[f'{tag}.train.{k}'].shape), G[f'{tag}.train.{k}']) < 2e-4, k
    _loss(r).backward()
    names = [str(n) for n in G['param_names']]
    params = dict(m.named_parameters())
    gn = np.array([0.0 if params[n].grad is None else params[n].grad.double().norm().item() for n in names])
    ref = G[f'{tag}.grad_norm']
    # (conv biases in front of a train-mode BatchNorm: exact zero here, rounding noise in the reference)
    skip = np.array([n.startswith('conv_gn_') and n.endswith('.bias') for n in names])
    assert np.abs(gn - ref)[~skip].max() <= 2e-3 * ref.max(), (np.abs(gn - ref)[~skip].max(), ref.max())
    for n in ('conv_in.0.0.weight', 'bcn1.blur_conv.0.weight', 'lin_gn_abs.weight'):
        assert _rel(params[n].grad.cpu().numpy(), G[f'{tag}.grad.{n}']) < 2e-3, n
