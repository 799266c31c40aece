"""resident workgroups per CU of the contraction kernels as the runtime reports them (efgh_debug_occupancy)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes
from efgh_amd import _C
L = _C.lib()
buf = ctypes.create_string_buffer(4096)
L.efgh_debug_occupancy(buf, 4096)
print(buf.value.decode())
