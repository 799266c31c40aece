"""Builds efgh_amd/lib/libefgh_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
SO = os.path.join(LIBDIR, 'libefgh_hip.so')

# per-file extra flags: the lattice float recipe must not be contracted / reassociated
EXTRA = {'lattice.hip': ['-ffp-contract=off'], 'pose.hip': ['-ffp-contract=off']}
COMMON = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function']
# A/B builds (tools): EFGH_BUILD_FLAGS='-DSOMETHING=1' EFGH_BUILD_DIR=lib_ab python -m efgh_amd.build -> efgh_amd/lib_ab/libefgh_hip.so (EFGH_LIB selects it)
COMMON += os.environ.get('EFGH_BUILD_FLAGS', '').split()
LIBDIR = os.path.join(HERE, os.environ.get('EFGH_BUILD_DIR', 'lib'))
SO = os.path.join(LIBDIR, 'libefgh_hip.so')


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp')))


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, 'obj')
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hdrs.append(os.path.join(os.path.dirname(HERE), 'include', 'efgh_hip.h'))
    hdr_m = max(os.path.getmtime(h) for h in hdrs)
    objs, rebuilt = [], False
    procs = []
    for f in sources():
        src = os.path.join(CSRC, f)
        obj = os.path.join(objdir, f + '.o')
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m):
            cmd = [hipcc, '-x', 'hip'] + COMMON + EXTRA.get(f, []) + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd))
            procs.append((f, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            rebuilt = True
    for f, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError('hipcc failed on ' + f)
        if verbose and out:
            print(out.decode())
    if rebuilt or not os.path.exists(SO):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', SO] + objs
        subprocess.check_call(cmd)
    return SO


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
