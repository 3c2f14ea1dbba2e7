"""ctypes front-end of oracle/nets_oracle.c (TEST INFRASTRUCTURE ONLY): runs a lowered Program on the CPU."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so, src = os.path.join(_HERE, "liboracle_nets.so"), os.path.join(_HERE, "nets_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "liboracle_nets.so"], stdout=subprocess.DEVNULL)
        _LIB = ctypes.CDLL(so)
        _LIB.orc_sizeof_op.restype = ctypes.c_size_t
        _LIB.orc_sizeof_tensor.restype = ctypes.c_size_t
    return _LIB


def run_program(prog, ext_arrays, want_views=()):
    """ext_arrays: list of contiguous float32 numpy arrays (inputs are read, outputs written in place).
    returns {view: ndarray [n,h,w,c]} for want_views"""
    from cartoonsegmentation_amd import program as P
    L = lib()
    assert L.orc_sizeof_op() == ctypes.sizeof(P.CsmOp) and L.orc_sizeof_tensor() == ctypes.sizeof(P.CsmTensorDesc)
    ops, tens, w = prog.serialise(oracle=True)
    ws = np.zeros(max(prog.workspace_floats, 64), np.float32)
    ext = (ctypes.c_void_p * max(len(ext_arrays), 1))()
    for i, a in enumerate(ext_arrays):
        assert a.dtype == np.float32 and a.flags['C_CONTIGUOUS']
        ext[i] = a.ctypes.data
    rc = L.orc_run_program(ops, ctypes.c_int(len(ops)), tens, ctypes.c_int(len(tens)), w.ctypes.data_as(ctypes.c_void_p),
                           ws.ctypes.data_as(ctypes.c_void_p), ext, ctypes.c_int(len(ext_arrays)))
    assert rc == 0
    out = {}
    for t in want_views:
        b = t.buf
        full = ws[b.offset:b.offset + b.n * b.h * b.w * b.c].reshape(b.n, b.h, b.w, b.c)
        out[t] = full[..., t.coff:t.coff + t.c].copy()
    return out
