"""Checkpoint interchange with the reference.

The reference saves `{'iter', 'state_dict', 'min_loss', 'optimizer'}` with `torch.save`
(common/helper.py:40-61, iterater.py:82-89); `state_dict` comes from the DataParallel-wrapped model,
so every key carries a `module.` prefix (main.py:127,136), and `optimizer` is
`torch.optim.Adam.state_dict()` over `named_parameters()` order (main.py:178-183).  These helpers read
and write exactly that layout, so checkpoints move freely between the reference and this path."""
import os
import shutil

import torch


def strip_module_prefix(sd):
    return {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}


def load_model_state(model, ckpt, strict=True):
    """ckpt: path | checkpoint dict | bare state_dict (with or without the `module.` prefix)"""
    if isinstance(ckpt, (str, os.PathLike)):
        ckpt = torch.load(ckpt, map_location='cpu')
    sd = ckpt['state_dict'] if isinstance(ckpt, dict) and 'state_dict' in ckpt else ckpt
    return model.load_state_dict(strip_module_prefix(sd), strict=strict)


def update_dict_filter(pretrained_dict, convert_dict, model_dict):
    """main.py:212-225: rename keys (every `convert_dict` key that occurs as a substring is replaced; a key matched by several
    entries yields one renamed copy per entry, as in the reference), then keep what the model has"""
    update = {}
    for k, v in pretrained_dict.items():
        converted = False
        for old, new in convert_dict.items():
            if old in k:
                update[k.replace(old, new)] = v
                converted = True
        if not converted:
            update[k] = v
    return {k: v for k, v in update.items() if k in model_dict}


def grad_false_keys_filter(model, grad_false_keys):
    """main.py:227-235: parameters whose name contains one of the keys are frozen (requires_grad = False)"""
    for k, p in model.named_parameters():
        if any(key in k for key in grad_false_keys):
            p.requires_grad = False
    return model


def load_pretrained(model, ckpt, convert_dict=None, grad_false_keys=None):
    """the `pretrained_path` branch of main.py:162-176: partial, renamed, non-strict load + freezing.  The optimizer must be
    built afterwards over the parameters that still require a gradient (main.py:178-183; train.Trainer / FlatParams do)."""
    if isinstance(ckpt, (str, os.PathLike)):
        ckpt = torch.load(ckpt, map_location='cpu')
    sd = ckpt['state_dict'] if isinstance(ckpt, dict) and 'state_dict' in ckpt else ckpt
    wrapped = all(k.startswith('module.') for k in sd)
    model_dict = model.state_dict()
    if wrapped:                                      # a checkpoint of the DataParallel-wrapped reference model
        model_dict = {'module.' + k: v for k, v in model_dict.items()}
    update = update_dict_filter(sd, convert_dict or {}, model_dict)
    res = model.load_state_dict(strip_module_prefix(update) if wrapped else update, strict=False)
    grad_false_keys_filter(model, grad_false_keys or [])
    return res


def adam_state_dict(opt, lr=None):
    """`torch.optim.Adam.state_dict()`-compatible view of a train.FusedAdam (one entry per parameter)."""
    flat = opt.flat
    state = {}
    for i, (p, (off, k)) in enumerate(zip(flat.params, flat.offsets)):
        if opt.t > 0:
            state[i] = {'step': torch.tensor(float(opt.t)),
                        'exp_avg': opt.m[off:off + k].view(p.shape).detach().clone(),
                        'exp_avg_sq': opt.v[off:off + k].view(p.shape).detach().clone()}
    group = {'lr': opt.lr if lr is None else lr, 'betas': tuple(opt.betas), 'eps': opt.eps,
             'weight_decay': opt.wd, 'amsgrad': False, 'maximize': False, 'foreach': None, 'capturable': False,
             'differentiable': False, 'fused': None, 'params': list(range(len(flat.params)))}
    return {'state': state, 'param_groups': [group]}


def load_adam_state(opt, sd):
    """inverse of adam_state_dict: accepts the reference's optimizer state (main.py:190-198)"""
    flat = opt.flat
    g = sd['param_groups'][0]
    opt.lr, opt.betas, opt.eps, opt.wd = g['lr'], tuple(g['betas']), g['eps'], g.get('weight_decay', 0.0)
    steps = []
    for i, (p, (off, k)) in enumerate(zip(flat.params, flat.offsets)):
        st = sd['state'].get(i)
        if st is None:
            continue
        opt.m[off:off + k].copy_(st['exp_avg'].reshape(-1))
        opt.v[off:off + k].copy_(st['exp_avg_sq'].reshape(-1))
        steps.append(int(st['step']))
    opt.t = max(steps) if steps else 0


def save_checkpoint(ckpt_dir, model, opt, it, min_loss, is_best=False, iter_interval=1000,
                    filename='checkpoint.pth.tar'):
    """common/helper.py:40-61 semantics: rolling file, periodic copies, best copy, pruning after 5 intervals"""
    os.makedirs(ckpt_dir, exist_ok=True)
    state = {'iter': it, 'state_dict': {'module.' + k: v.detach().cpu() for k, v in model.state_dict().items()},
             'min_loss': min_loss, 'optimizer': adam_state_dict(opt)}
    path = os.path.join(ckpt_dir, filename)
    torch.save(state, path)
    if it % iter_interval == 0:
        shutil.copyfile(path, os.path.join(ckpt_dir, 'checkpoint_%d.pth.tar' % it))
    if is_best:
        shutil.copyfile(path, os.path.join(ckpt_dir, 'model_best.pth.tar'))
    if it > 5 * iter_interval:
        old = os.path.join(ckpt_dir, 'checkpoint_%d.pth.tar' % (it - 5 * iter_interval))
        if os.path.exists(old):
            os.remove(old)
    return path
