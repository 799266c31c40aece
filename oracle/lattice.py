"""ctypes front-end of oracle/lattice_oracle.c (test infrastructure only)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liboracle_lattice.so')
_lib = None

DEFAULT_SCALES = (1.0, 0.75, 0.5, 0.25, 0.125)      # configs/train_rellis.yaml:30-35


def build():
    src = os.path.join(_HERE, 'lattice_oracle.c')
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '_build/liboracle_lattice.so'],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.oracle_gd_run.restype = ctypes.c_void_p
        L.oracle_gd_run.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
        L.oracle_gd_N.restype = ctypes.c_int64
        L.oracle_gd_N.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.oracle_gd_H.restype = ctypes.c_int64
        L.oracle_gd_H.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.oracle_gd_copy.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.oracle_gd_free.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


def generate_data(pc, scales=DEFAULT_SCALES):
    """pc: (3,N) float32 -> list of per-level dicts (numpy), as GenerateData.__call__
    (nets/generate_data.py:117-193): bary (4,N) f32, emg (4,N) f32, off (4,N) i64,
    nbr (15,H) i64, H, plus pts_next (3,H) f32 (the next level's input points)."""
    L = lib()
    pc = np.ascontiguousarray(pc, dtype=np.float32)
    assert pc.ndim == 2 and pc.shape[0] == 3
    sc = np.asarray(scales, dtype=np.float64)
    h = L.oracle_gd_run(pc.ctypes.data, pc.shape[1], len(sc), sc.ctypes.data)
    out = []
    try:
        for l in range(len(sc)):
            n, H = L.oracle_gd_N(h, l), L.oracle_gd_H(h, l)
            d = {'N': n, 'H': H,
                 'bary': np.empty((4, n), np.float32), 'emg': np.empty((4, n), np.float32),
                 'off': np.empty((4, n), np.int64), 'nbr': np.empty((15, H), np.int64),
                 'pts_next': np.empty((3, H), np.float32),
                 'mins': np.empty(4, np.int64), 'maxs': np.empty(4, np.int64)}
            for what, key in enumerate(('bary', 'emg', 'off', 'nbr', 'pts_next', 'mins', 'maxs')):
                L.oracle_gd_copy(h, l, what, d[key].ctypes.data)
            out.append(d)
    finally:
        L.oracle_gd_free(h)
    return out
