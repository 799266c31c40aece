"""Import the *reference* EFGH python (read-only, /root/reference) in THIS container.

Fixture-generation tooling only: never imported by the product, by `-m gpu` tests, by smoke() or
by bench.py, and never shipped to the GPU box in any useful form (the reference tree does not
exist there).  It installs stand-ins for the reference's missing *third-party* imports
(numba, cffi-built `_khash_ffi`, open3d, pyquaternion, tensorboardX, nuscenes) so that the
reference's own, unmodified sources run on the CPU:

* ``numba.njit``      -> identity decorator (build_it, nets/transforms.py:125-184, runs as python)
* ``_khash_ffi.lib``  -> ctypes binding of oracle/_ref/libkhash_ref.so, i.e. the reference's own
                         khash.h / khash_int2int.h compiled in place (oracle/khash_ref_shim.c)
* ``torch.cuda.LongTensor/FloatTensor`` -> grad-preserving casts (torch_utils.py:50-51)
* ``np.long``         -> np.int64 (generate_data.py:49, transforms.py:110)
"""
import ctypes
import os
import sys
import types

import numpy as np
import torch

REF_ROOT = '/root/reference'
_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(os.path.dirname(_HERE))
_KHASH_SO = os.path.join(_REPO, 'oracle', '_ref', 'libkhash_ref.so')


def _install_stubs():
    if 'numba' in sys.modules and getattr(sys.modules['numba'], '_efgh_stub', False):
        return
    # ---- numba -------------------------------------------------------------------------
    nb = types.ModuleType('numba')
    nb._efgh_stub = True

    class _Ty:
        def __call__(self, *a, **k):
            return self

        def __getitem__(self, item):
            return self

    for name in ('int64', 'int32', 'float32', 'float64', 'void', 'boolean'):
        setattr(nb, name, _Ty())

    def njit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not isinstance(args[0], _Ty):
            return args[0]
        return lambda f: f

    nb.njit = njit
    nb.jit = njit
    cffi_support = types.ModuleType('numba.cffi_support')
    cffi_support.register_module = lambda m: None
    nb.cffi_support = cffi_support
    sys.modules['numba'] = nb
    sys.modules['numba.cffi_support'] = cffi_support

    # ---- _khash_ffi (the reference's own khash, compiled in place) ----------------------
    if not os.path.exists(_KHASH_SO):
        raise RuntimeError('build oracle/_ref first: make -C oracle ref')
    so = ctypes.CDLL(_KHASH_SO)
    so.ref_khash_int2int_init.restype = ctypes.c_void_p
    so.ref_khash_int2int_destroy.argtypes = [ctypes.c_void_p]
    so.ref_khash_int2int_get.restype = ctypes.c_longlong
    so.ref_khash_int2int_get.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong]
    so.ref_khash_int2int_set.restype = ctypes.c_int
    so.ref_khash_int2int_set.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong]
    kh = types.ModuleType('_khash_ffi')
    lib = types.SimpleNamespace(
        khash_int2int_init=lambda: so.ref_khash_int2int_init(),
        khash_int2int_destroy=lambda h: so.ref_khash_int2int_destroy(h),
        khash_int2int_get=lambda h, k, d: so.ref_khash_int2int_get(h, int(k), int(d)),
        khash_int2int_set=lambda h, k, v: so.ref_khash_int2int_set(h, int(k), int(v)),
    )
    kh.lib = lib
    sys.modules['_khash_ffi'] = kh

    # ---- other absent third-party modules (only imported, never called on this path) ---
    for name in ('open3d', 'tensorboardX', 'nuscenes', 'nuscenes.nuscenes', 'nuscenes.utils',
                 'nuscenes.utils.data_classes', 'nuscenes.utils.geometry_utils', 'cv2',
                 'torchgeometry'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['tensorboardX'].SummaryWriter = object
    # pyquaternion is absent from the build container.  The reference uses four things of it (common/helper.py:185-190):
    # Quaternion(w, x, y, z), the Hamilton product `*`, `.inverse` and indexing [0..3] = (w, x, y, z) - plain quaternion
    # algebra, stood in for here so that the reference's OWN raw-mode error routine (calc_error_raw_np / quaternion_distance)
    # can produce the fixtures of make_golden_metrics.py.  Fixture generation only; nothing under tests/ or the product uses it.
    class Quaternion:
        def __init__(self, w, x, y, z):
            self.q = np.array([w, x, y, z], dtype=np.float64)

        def __mul__(self, o):
            w1, x1, y1, z1 = self.q
            w2, x2, y2, z2 = o.q
            return Quaternion(w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                              w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2)

        @property
        def inverse(self):
            w, x, y, z = self.q
            n = float(np.dot(self.q, self.q))
            return Quaternion(w / n, -x / n, -y / n, -z / n)

        def __getitem__(self, i):
            return self.q[i]

    pq = types.ModuleType('pyquaternion')
    pq.Quaternion = Quaternion
    sys.modules.setdefault('pyquaternion', pq)
    sys.modules['nuscenes.nuscenes'].NuScenes = object
    sys.modules['nuscenes.utils.data_classes'].LidarPointCloud = object
    sys.modules['nuscenes.utils.data_classes'].Box = object

    if not hasattr(np, 'long'):
        np.long = np.int64
    torch.cuda.LongTensor = lambda t: t.long()
    torch.cuda.FloatTensor = lambda t: t.float()


def import_reference():
    """Returns (nets, losses, common.torch_utils) modules of the reference."""
    _install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # our own repo root must not shadow `nets` / `losses` / `common`
    import importlib
    nets = importlib.import_module('nets')
    losses = importlib.import_module('losses')
    tu = importlib.import_module('common.torch_utils')
    assert nets.__file__.startswith(REF_ROOT), nets.__file__
    return nets, losses, tu


def default_args(raw_hw=(768, 2560), device='cpu'):
    """configs/train_rellis.yaml hot-path keys, with a configurable image size."""
    return {
        'dim': 3,
        'scale_map': [[1., 1], [0.75, 1], [0.5, 1], [0.25, 1], [0.125, 1]],
        'DEVICE': device,
        'use_leaky': True, 'bcn_use_bias': True, 'bcn_use_norm': True, 'last_relu': False,
        'raw_cam_img_size': [int(raw_hw[0]), int(raw_hw[1])],
        'lidar_fov_rad': [0.125, -0.125],
        'dataset': 'RELLIS_3D',
        'lambda': {'e_gn': 100., 'h_hrzn': 100., 'fov': 100., 'g_trs': 1000., 'g_depth': 0.1,
                   'g_mask': 1000.},
        'fov_pos_num': 30, 'fov_neg_ratio': 5,
    }
