"""ctypes binding of libcsm355.so (C ABI: include/csm355.h).

There is NO fallback: if the HIP library is missing or a call fails, a
CsmError is raised.  (The CPU oracle under oracle/ is test infrastructure and
is never imported from here.)
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# CSM_LIB=<path> loads another build of the same library (development: A/B of two kernel builds; never a fallback -- a missing
# file still raises)
LIB_PATH = os.environ.get("CSM_LIB") or os.path.join(_HERE, "libcsm355.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "csm355.h")
_lib = None


class CsmError(RuntimeError):
    pass


def declared_symbols():
    """function names declared in include/csm355.h"""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(csm_[a-z0-9_]+)\s*\(", txt)))


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CsmError("libcsm355.so not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C cartoonsegmentation_amd/csrc`" % LIB_PATH)
        if os.environ.get("CSM_LIB"):
            import sys
            print("libcsm355: CSM_LIB overrides the in-tree library: %s" % LIB_PATH, file=sys.stderr)
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.csm_last_error.restype = ctypes.c_char_p
        _lib.csm_build_info.restype = ctypes.c_char_p
        _lib.csm_warp_frame_scratch_floats.restype = ctypes.c_size_t
        _lib.csm_fill_disocclusion_scratch_bytes.restype = ctypes.c_size_t
        _lib.csm_nms_scratch_bytes.restype = ctypes.c_size_t
        _lib.csm_autozoom_scratch_floats.restype = ctypes.c_size_t
        _lib.csm_autozoom_band_scratch_bytes.restype = ctypes.c_size_t
        _lib.csm_warp_tile_scratch_bytes.restype = ctypes.c_size_t
        _lib.csm_warp_tile_header_bytes.restype = ctypes.c_size_t
        _lib.csm_warp_frames_scratch_bytes.restype = ctypes.c_size_t
        _lib.csm_percentile_scratch_bytes.restype = ctypes.c_size_t
        _lib.csm_bokeh_depth_scratch_bytes.restype = ctypes.c_size_t
        _lib.csm_kenburns_frame_scratch_bytes.restype = ctypes.c_size_t
        _lib.csm_det_decode_scratch_bytes.restype = ctypes.c_size_t
        _lib.csm_mean_std_scratch_bytes.restype = ctypes.c_size_t
        if os.environ.get("CSM_TUNER_OPTIONS"):          # measurement aid (A/B of launch forms, include/csm355.h csm_debug_conv_tuner_options); speed only
            # the value is the WHOLE bitmask (default 1 = mixed-tile launches on; bit 1 = N-grouped tile order off): "2" also clears bit 0
            import sys
            try:
                opt = int(os.environ["CSM_TUNER_OPTIONS"], 0)
            except ValueError:
                raise CsmError("CSM_TUNER_OPTIONS=%r is not an integer bitmask (default 1; see csm_debug_conv_tuner_options in "
                               "include/csm355.h)" % os.environ["CSM_TUNER_OPTIONS"])
            print("libcsm355: CSM_TUNER_OPTIONS=%d replaces the default tuner options (1)" % opt, file=sys.stderr)
            _lib.csm_debug_conv_tuner_options(opt)
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = load().csm_last_error().decode()
        raise CsmError("libcsm355 %s failed (status %d): %s" % (what, rc, msg))


def ptr(t):
    """raw device pointer of a torch tensor (or None)"""
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def stream_ptr(device=None):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


i32, i64, f32, f64 = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double
