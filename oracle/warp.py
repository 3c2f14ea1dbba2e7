"""ctypes front-end of oracle/warp_oracle.c (TEST INFRASTRUCTURE ONLY).

numpy in / numpy out; every function cites the reference in warp_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_F = np.float32
c_fp = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    so = os.path.join(_HERE, "liboracle_warp.so")
    src = os.path.join(_HERE, "warp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle_warp.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f(a):
    return np.ascontiguousarray(a, dtype=_F)


def update_zee(pts, H, W, focal, baseline):
    pts = _f(pts); B, _, N = pts.shape
    zee = np.full((B, 1, H, W), 1000000.0, _F)
    lib().orc_pointrender_update_zee(ctypes.c_int(B), ctypes.c_int64(N), ctypes.c_int(H), ctypes.c_int(W),
                                     ctypes.c_double(focal), ctypes.c_double(baseline), _p(pts), _p(zee))
    return zee


def degrid(zee, mode):
    zee = _f(zee).copy(); B, _, H, W = zee.shape
    lib().orc_pointrender_degrid(ctypes.c_int(B), ctypes.c_int(H), ctypes.c_int(W), _p(zee), ctypes.c_int(mode))
    return zee


def update_output(pts, data1, zee, focal, baseline):
    pts, data1, zee = _f(pts), _f(data1), _f(zee)
    B, C1, N = data1.shape; _, _, H, W = zee.shape
    out = np.zeros((B, C1, H, W), _F)
    lib().orc_pointrender_update_output(ctypes.c_int(B), ctypes.c_int64(N), ctypes.c_int(C1), ctypes.c_int(H),
                                        ctypes.c_int(W), ctypes.c_double(focal), ctypes.c_double(baseline),
                                        _p(pts), _p(data1), _p(zee), _p(out))
    return out


def render_pointcloud(pts, data, W, H, focal, baseline, degrid_mode=1, return_zee=False):
    pts, data = _f(pts), _f(data)
    B, C, N = data.shape
    render = np.empty((B, C, H, W), _F); existing = np.empty((B, 1, H, W), _F)
    zee = np.empty((B, 1, H, W), _F)
    lib().orc_render_pointcloud(ctypes.c_int(B), ctypes.c_int(C), ctypes.c_int64(N), ctypes.c_int(H), ctypes.c_int(W),
                                ctypes.c_double(focal), ctypes.c_double(baseline), _p(pts), _p(data),
                                ctypes.c_int(degrid_mode), _p(render), _p(existing), _p(zee))
    return (render, existing, zee) if return_zee else (render, existing)


def fill_disocclusion(inp, depth):
    inp, depth = _f(inp), _f(depth)
    B, C, H, W = inp.shape
    out = inp.copy()
    lib().orc_fill_disocclusion(ctypes.c_int(B), ctypes.c_int(C), ctypes.c_int(H), ctypes.c_int(W),
                                _p(inp), _p(depth), _p(out))
    return out


def spatial_filter_laplacian(x):
    x = _f(x); B, C, H, W = x.shape
    out = np.empty_like(x)
    lib().orc_spatial_filter_laplacian(ctypes.c_int(B * C), ctypes.c_int(H), ctypes.c_int(W), _p(x), _p(out))
    return out


def spatial_filter_median5(x):
    x = _f(x); B, C, H, W = x.shape
    out = np.empty_like(x)
    lib().orc_spatial_filter_median5(ctypes.c_int(B * C), ctypes.c_int(H), ctypes.c_int(W), _p(x), _p(out))
    return out


def spatial_filter_median3(x):
    x = _f(x); B, C, H, W = x.shape
    out = np.empty_like(x)
    lib().orc_spatial_filter_median3(ctypes.c_int(B * C), ctypes.c_int(H), ctypes.c_int(W), _p(x), _p(out))
    return out


def depth_to_points(depth, focal):
    depth = _f(depth); B, _, H, W = depth.shape
    pts = np.empty((B, 3, H, W), _F)
    lib().orc_depth_to_points(ctypes.c_int(B), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_double(focal), _p(depth), _p(pts))
    return pts


def disparity_to_points(disp_in, focal, baseline):
    disp_in = _f(disp_in); H, W = disp_in.shape[-2:]
    disp = np.empty((1, 1, H, W), _F); depth = np.empty_like(disp); valid = np.empty_like(disp)
    pts = np.empty((1, 3, H, W), _F); un = np.empty_like(pts)
    lib().orc_disparity_to_points(ctypes.c_int(H), ctypes.c_int(W), ctypes.c_double(focal), ctypes.c_double(baseline),
                                  _p(disp_in), _p(disp), _p(depth), _p(valid), _p(pts), _p(un))
    return disp, depth, valid, pts, un


def process_shift(pts, shift):
    pts = _f(pts); B, _, N = pts.shape
    out = np.empty_like(pts)
    s = np.asarray(shift, _F)
    lib().orc_process_shift(ctypes.c_int(B), ctypes.c_int64(N), ctypes.c_float(s[0]), ctypes.c_float(s[1]),
                            ctypes.c_float(s[2]), _p(pts), _p(out))
    return out


def warp_frame(pts, rgbd, H, W, focal, baseline, shift, degrid_mode=1):
    pts, rgbd = _f(pts), _f(rgbd)
    N = pts.shape[2]
    s = np.asarray(shift, _F)
    filled = np.empty((1, 4, H, W), _F); existing = np.empty((1, 1, H, W), _F)
    frame = np.empty((H, W, 3), np.uint8)
    lib().orc_warp_frame(ctypes.c_int64(N), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_double(focal),
                         ctypes.c_double(baseline), ctypes.c_float(s[0]), ctypes.c_float(s[1]), ctypes.c_float(s[2]),
                         _p(pts), _p(rgbd), ctypes.c_int(degrid_mode), _p(filled), _p(existing), _p(frame))
    return filled, existing, frame


def warp_frame_mt(pts, rgbd, H, W, focal, baseline, shift):
    """all-cores OpenMP variant of warp_frame (atomics, unspecified accumulation order): bench.py's timed CPU baseline only"""
    pts, rgbd = _f(pts), _f(rgbd)
    N = pts.shape[2]
    s = np.asarray(shift, _F)
    filled = np.empty((1, 4, H, W), _F); existing = np.empty((1, 1, H, W), _F)
    frame = np.empty((H, W, 3), np.uint8)
    lib().orc_warp_frame_mt(ctypes.c_int64(N), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_double(focal),
                            ctypes.c_double(baseline), ctypes.c_float(s[0]), ctypes.c_float(s[1]), ctypes.c_float(s[2]),
                            _p(pts), _p(rgbd), _p(filled), _p(existing), _p(frame))
    return filled, existing, frame


def shift_vector(settings, common):
    """host scalar part of process_shift (common.py:60-72), python floats like the reference"""
    cd = common['objDepthrange'][0] + (settings['fltDepthTo'] - settings['fltDepthFrom'])
    fu, fv = common['objDepthrange'][2][0], common['objDepthrange'][2][1]
    tu, tv = fu + settings['fltShiftU'], fv + settings['fltShiftV']
    w2, h2, f = common['intWidth'] / 2.0, common['intHeight'] / 2.0, common['fltFocal']
    fx, fy = ((fu - w2) * cd) / f, ((fv - h2) * cd) / f
    tx, ty = ((tu - w2) * cd) / f, ((tv - h2) * cd) / f
    return np.array([fx - tx, fy - ty, settings['fltDepthTo'] - settings['fltDepthFrom']], np.float32)
