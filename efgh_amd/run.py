"""Zero-edit drop-in launcher: runs the reference's UNMODIFIED entry script on the MI355X path.

    python -m efgh_amd.run [--device I | --all-devices] main.py configs/train_rellis.yaml

The reference binds its model and criterion by module name (`import nets`, `import losses`, main.py:14-15;
`nets.__dict__[arch + 'Backbone']`, `losses.__dict__[arch + 'Criterion']`, main.py:126,129).  This launcher installs
`efgh_amd.nets` / `efgh_amd.losses` under those two names in `sys.modules`, puts the script's directory at the front of
`sys.path` (what `python main.py` does, so `data_loader`, `common`, `iterater`, `valid`, `test` stay the reference's own)
and `runpy`-runs the script IN THIS PROCESS as `__main__` - before anything has touched the GPU, never an exec.

`torch.nn.DataParallel(model)` (main.py:127) spans every visible device; this path is one process per GPU, so unless
`--all-devices` is given the process is pinned to ONE device first (`--device I`, default 0, counted in the list that is
visible now): DataParallel then has `device_ids == [0]` and calls the module directly.  With several devices left visible
the backbone refuses a replica forward loudly (nets/efghbackbone.py).
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


def parse(argv):
    device, pin = 0, True
    i = 0
    while i < len(argv):
        a = argv[i]
        if a == '--device' and i + 1 < len(argv):
            device = int(argv[i + 1])
            i += 2
        elif a.startswith('--device='):
            device = int(a.split('=', 1)[1])
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
        raise SystemExit('usage: python -m efgh_amd.run [--device I | --all-devices] <script.py> [script arguments...]')
    return device, pin, argv[i], argv[i + 1:]


def main(argv=None):
    device, pin, script, rest = parse(list(sys.argv[1:] if argv is None else argv))
    if not os.path.isfile(script):
        raise SystemExit('efgh_amd.run: no such script: %s' % script)
    if pin:
        pin_one_device(device)
    install_aliases()
    script = os.path.abspath(script)
    sys.argv = [script] + list(rest)
    sys.path.insert(0, os.path.dirname(script))
    runpy.run_path(script, run_name='__main__')


if __name__ == '__main__':
    main()
