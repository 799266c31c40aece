"""ctypes loader for libefgh_hip.so (the C-ABI declared in include/efgh_hip.h).

There is no CPU fallback: if the library is missing, or a call fails, this raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get('EFGH_LIB') or os.path.join(_HERE, 'lib', 'libefgh_hip.so')      # EFGH_LIB: A/B runs of two builds
_lib = None
ABI_VERSION = 3          # EFGH_ABI_VERSION of include/efgh_hip.h this binding was written against

c_void_p, c_int, c_int32, c_int64, c_float = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int32,
                                              ctypes.c_int64, ctypes.c_float)


class EfghError(RuntimeError):
    pass


class GemmDesc(ctypes.Structure):
    """mirror of efgh_gemm_desc (include/efgh_hip.h)"""
    _fields_ = [
        ('A', c_void_p), ('lda', c_int64), ('C', c_int32), ('T', c_int32), ('mode', c_int32),
        ('B', c_int32), ('Hin', c_int32), ('Win', c_int32), ('Hv', c_int32), ('Wv', c_int32),
        ('sh', c_int32), ('sw', c_int32),
        ('dh', ctypes.c_int8 * 16), ('dw', ctypes.c_int8 * 16),
        ('Ho', c_int32), ('Wo', c_int32), ('osh', c_int32), ('osw', c_int32), ('oh0', c_int32),
        ('ow0', c_int32),
        ('table', c_void_p), ('W', c_void_p), ('N', c_int32), ('M', c_int64), ('M_dev', c_void_p),
        ('bias', c_void_p), ('scale', c_void_p), ('shift', c_void_p),
        ('residual', c_void_p), ('ldr', c_int64),
        ('act', c_int32), ('slope', c_float), ('out', c_void_p), ('ldo', c_int64),
        ('stats', c_void_p),
        ('nbatch', c_int32), ('batch_stride_a', c_int64), ('batch_stride_w', c_int64), ('batch_stride_out', c_int64),
        ('batch_stride_table', c_int32), ('table_alias_mask', c_int32),
        ('stats_mode', c_int32), ('bn_raw', c_void_p), ('bn_ldraw', c_int64), ('bn_y', c_void_p), ('bn_ldy', c_int64),
        ('bn_pscale', c_void_p), ('bn_pshift', c_void_p), ('bn_mean', c_void_p), ('bn_invstd', c_void_p),
        ('bn_act', c_int32), ('bn_slope', c_float),
    ]


class WgradOutDesc(ctypes.Structure):
    """mirror of efgh_wgrad_out_desc (include/efgh_hip.h): where a weight gradient belongs in the caller's own layout"""
    _fields_ = [('W', c_void_p), ('N', c_int32), ('T', c_int32), ('C', c_int32), ('Cp', c_int32),
                ('sn', c_int64), ('sc', c_int64), ('st', c_int64), ('taps', c_int32 * 16), ('accumulate', c_int32)]


WROTE_OUT = 1            # EFGH_WROTE_OUT


def check_wrote(rc):
    """return code of a weight-gradient entry point -> True when the launch wrote the caller's layout itself (EFGH_WROTE_OUT)"""
    if rc == WROTE_OUT:
        return True
    check(rc)
    return False


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise EfghError(f'{SO_PATH} is missing: run `python -c "import __graft_entry__ as g; '
                            f'g.build()"` (hipcc, gfx950). There is no CPU fallback.')
        import torch  # noqa: F401  (load torch's HIP runtime first so both share one runtime)
        _lib = ctypes.CDLL(SO_PATH)
        _lib.efgh_last_error.restype = ctypes.c_char_p
        _lib.efgh_version.restype = c_int
        if _lib.efgh_version() != ABI_VERSION:
            got, _lib = _lib.efgh_version(), None
            raise EfghError(f'{SO_PATH} has ABI version {got}, this package binds version {ABI_VERSION} of include/efgh_hip.h: '
                            f'rebuild it (`python -m efgh_amd.build`)')
        _lib.efgh_wino2d_tiles.restype = c_int64
        _lib.efgh_segment_workspace.restype = c_int64
        _lib.efgh_gather_wgrad_workspace.restype = c_int64
        _lib.efgh_plane_wgrad_workspace.restype = c_int64
        _lib.efgh_wino_wgrad_workspace.restype = c_int64
        _lib.efgh_sc_wgrad_workspace.restype = c_int64
        _lib.efgh_c4n4_wgrad_workspace.restype = c_int64
        _lib.efgh_c4_wgrad_workspace.restype = c_int64
        _lib.efgh_thin_wgrad_workspace.restype = c_int64
        _lib.efgh_lattice_hash_capacity.restype = c_int64
        _lib.efgh_lattice_hash_capacity.argtypes = [c_int32]
        _lib.efgh_lattice_workspace_bytes.restype = c_int64
        _lib.efgh_lattice_workspace_bytes.argtypes = [c_int32, c_int32, c_int32]
        _lib.efgh_lattice_part_max_entries.argtypes = [c_int32]
        _lib.efgh_lattice_part_buckets.argtypes = [c_int32]
        _lib.efgh_lattice_part_list_len.restype = c_int64
        _lib.efgh_lattice_part_list_len.argtypes = [c_int32, c_int32]
        _lib.efgh_lattice_part_zeroed_bytes.restype = c_int64
        _lib.efgh_lattice_part_zeroed_bytes.argtypes = [c_int32]
        _lib.efgh_lattice_part_workspace_bytes.restype = c_int64
        _lib.efgh_lattice_part_workspace_bytes.argtypes = [c_int32, c_int32, c_int32, c_int32, c_int32]
    return _lib


def check(rc):
    if rc != 0:
        raise EfghError(f'libefgh_hip: rc={rc}: {lib().efgh_last_error().decode()}')


def ptr(t):
    """device pointer of a torch tensor (None -> NULL)"""
    return c_void_p(0 if t is None else t.data_ptr())


_raw_stream = None


def stream_ptr():
    """the current HIP stream of the current device as a raw pointer.  `torch.cuda.current_stream().cuda_stream` builds a Python
    Stream object per call (device-index resolution, `os.environ` look-ups: 2 us x ~600 launches of a batch-1 forward = 15 % of
    its host time, tools/host_profile.py); the private raw getter is the same value without the object"""
    global _raw_stream
    import torch
    if _raw_stream is None:
        get, dev = getattr(torch._C, '_cuda_getCurrentRawStream', None), getattr(torch._C, '_cuda_getDevice', None)
        if get is not None and dev is not None:
            _raw_stream = lambda: get(dev())
        else:                                   # (a torch without the private getters)
            _raw_stream = lambda: torch.cuda.current_stream().cuda_stream
    return c_void_p(_raw_stream())


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise EfghError('efgh_amd runs on the GPU through libefgh_hip.so only (got a CPU tensor); '
                            'there is no CPU fallback')


def require_f32(*tensors):
    """the kernels read raw float32 memory: anything else would be misread silently (the reference's loop casts its inputs
    with .float(), iterater.py:29-32)"""
    import torch
    for t in tensors:
        if t is not None and t.dtype != torch.float32:
            raise EfghError('efgh_amd expects float32 tensors, got %s (cast with .float() as iterater.py:29-32 does)' % t.dtype)
