"""CPU oracle of the Ken Burns glue (TEST INFRASTRUCTURE ONLY): LeReS depth path, depth adjustment, point cloud
set-up, autozoom search and the frame loop, restated in numpy over oracle/*.c -- independent of
cartoonsegmentation_amd/kenburns.py.  References: anime_3dkenburns/kenburns_effect.py:39-91, :563-633, :898-1081,
anime_3dkenburns/common.py:59-142, depth_modules/leres/__init__.py:69-147."""
import ctypes
import math

import numpy as np

from . import nets as onets, segment as oseg, warp as owarp

ci, cf = ctypes.c_int, ctypes.c_float
_p = oseg._p


def leres_depth(img, leres_prog_for, depth_est_size):
    """-> 'depth' (inverse-depth like) float32 [1,1,H,W]"""
    L = oseg.lib()
    H, W = img.shape[:2]
    r = depth_est_size / max(H, W)
    h, w = H, W
    if r < 1:
        if H > W:
            h, w = depth_est_size, max(1, int(round(W * r)))
        else:
            w, h = depth_est_size, max(1, int(round(H * r)))
    h, w = int(math.ceil(h / 32) * 32), int(math.ceil(w / 32) * 32)
    x = np.empty((1, 3, h, w), np.float32)
    L.orc_leres_input(_p(np.ascontiguousarray(img)), ci(H), ci(W), ci(h), ci(w), _p(x))
    y = np.zeros((1, 1, h, w), np.float32)
    onets.run_program(leres_prog_for(h, w), [x, y])
    q = np.empty((h, w), np.uint8)
    L.orc_leres_quantize(_p(y), ctypes.c_int64(h * w), cf(float(y.min())), cf(float(y.max())), _p(q))
    depth = np.empty((1, 1, H, W), np.float32)
    L.orc_resize_u8_to_f32(_p(q), ci(h), ci(w), ci(H), ci(W), _p(depth))
    pos = depth[depth > 0]
    if pos.size:
        depth[depth == 0] = pos.min()
    return depth, y, q


def depth_adjustment(masks_bool, disparity, use_medium=False):
    """kenburns_effect.py:39-91 without the resize round trip (disparity and image of equal size)"""
    adj = disparity.copy()
    for m in masks_bool:
        mf = m.astype(np.float32)[None, None]
        plane = adj * mf
        if float(plane.sum()) == 0:
            continue
        if use_medium:                                   # torch.median = lower median
            sel = plane > 0
            v = np.sort(adj[sel])
            adj[sel] = v[(v.size - 1) // 2]
            continue
        rows = np.nonzero(plane.sum(axis=3).reshape(-1) > 0.0)[0]
        top, bottom = int(rows[0]), int(rows[-1])
        r0 = int(round(top + (0.97 * (bottom - top))))
        adj = ((np.float32(1.0) - mf) * adj) + (mf * plane[:, :, r0:, :].max())
    return adj


def kenburns_config(img, masks_bool, leres_prog_for, depth_est_size, focal, baseline):
    disparity, _, _ = leres_depth(img, leres_prog_for, depth_est_size)
    disparity = depth_adjustment(masks_bool, disparity)
    disparity = (disparity / disparity.max() * np.float32(baseline)).astype(np.float32)
    _, depth, valid, pts, unaltered = _points_from_normalised(disparity, focal, baseline)
    crop = depth[0, 0, 128:-128, 128:-128]
    amin = int(crop.argmin()); cw = crop.shape[1]
    return dict(disparity=disparity, depth=depth, valid=valid, pts=pts.reshape(1, 3, -1), unaltered=unaltered.reshape(1, 3, -1),
                depthrange=(float(crop.min()), float(crop.max()), (amin % cw, amin // cw)))


def _points_from_normalised(disparity, focal, baseline):
    """kenburns_effect.py:929-933 on an already normalised disparity (oracle C works on the raw map; redo in numpy/C parts)"""
    H, W = disparity.shape[-2:]
    fb = np.float32(focal * baseline)
    depth = ((np.float32(1.0) / (disparity + np.float32(0.00001))) * fb).astype(np.float32)
    nd = (disparity / disparity.max()).astype(np.float32)
    lap = owarp.spatial_filter_laplacian(nd)
    valid = (np.abs(lap) < np.float32(0.03)).astype(np.float32)
    pts = owarp.depth_to_points((depth * valid).astype(np.float32), focal)
    un = owarp.depth_to_points(depth, focal)
    return None, depth, valid, pts, un


def autozoom_target(kc, rgb, W, H, focal, baseline, shift=100.0, zoom=1.25, degrid_mode=1, counts_out=None):
    """common.py:86-142 ; degrid_mode 0 = the raster in-place pass (what a sequential execution of the reference does),
    1 = the Jacobi form the HIP build uses; counts_out (list) receives the coverage count of every candidate tried"""
    lin = np.linspace(-shift, shift, 16)
    icw, ich = int(math.floor(0.97 * W)), int(math.floor(0.97 * H))
    cw, ch = icw / zoom, ich / zoom
    cu, cv = W / 2.0, H / 2.0
    d_from = kc['depthrange'][0]
    d_to = d_from * (cw / icw)
    common = {'objDepthrange': kc['depthrange'], 'intWidth': W, 'intHeight': H, 'fltFocal': focal, 'fltBaseline': baseline}
    best, bu, bv = 0.0, None, None
    for iu in range(16):
        for iv in range(16):
            su, sv = float(lin[iv]), float(lin[iu])
            if cu + su < cw / 2.0 or cu + su > W - (cw / 2.0) or cv + sv < ch / 2.0 or cv + sv > H - (ch / 2.0):
                continue
            s = owarp.shift_vector({'fltShiftU': su, 'fltShiftV': sv, 'fltDepthFrom': d_from, 'fltDepthTo': d_to}, common)
            ps = owarp.process_shift(kc['pts'], s)
            _, existing = owarp.render_pointcloud(ps, rgb, W, H, focal, baseline, degrid_mode=degrid_mode)
            c = float((existing > 0.0).astype(np.float32).sum())
            if counts_out is not None:
                counts_out.append(c)
            if best < c:
                best, bu, bv = c, su, sv
    return {'fltCenterU': cu + bu, 'fltCenterV': cv + bv, 'intCropWidth': int(round(icw / zoom)), 'intCropHeight': int(round(ich / zoom))}, \
           {'fltCenterU': cu, 'fltCenterV': cv, 'intCropWidth': icw, 'intCropHeight': ich}


def frames(kc, rgb, W, H, focal, baseline, objFrom, objTo, steps):
    """kenburns_effect.py:1015-1072 without inpainting / bokeh"""
    L = oseg.lib()
    common = {'objDepthrange': kc['depthrange'], 'intWidth': W, 'intHeight': H, 'fltFocal': focal, 'fltBaseline': baseline}
    rgbd = np.concatenate([rgb, kc['depth'].reshape(1, 1, -1)], 1)
    pw, ph = max(objFrom['intCropWidth'], objTo['intCropWidth']), max(objFrom['intCropHeight'], objTo['intCropHeight'])
    out = []
    for st in steps:
        f, t = 1.0 - st, 1.0 - (1.0 - st)
        su = ((f * objFrom['fltCenterU']) + (t * objTo['fltCenterU'])) - (W / 2.0)
        sv = ((f * objFrom['fltCenterV']) + (t * objTo['fltCenterV'])) - (H / 2.0)
        cwid = (f * objFrom['intCropWidth']) + (t * objTo['intCropWidth'])
        d_from = kc['depthrange'][0]
        d_to = d_from * (cwid / max(objFrom['intCropWidth'], objTo['intCropWidth']))
        s = owarp.shift_vector({'fltShiftU': su, 'fltShiftV': sv, 'fltDepthFrom': d_from, 'fltDepthTo': d_to}, common)
        _, _, fr = owarp.warp_frame(kc['pts'], rgbd, H, W, focal, baseline, s, degrid_mode=1)
        o = np.empty_like(fr)
        L.orc_crop_resize_u8(_p(np.ascontiguousarray(fr)), ci(H), ci(W), ci(ph), ci(pw), cf(W / 2.0), cf(H / 2.0), _p(o))
        out.append(o)
    return out


def inpaint_forward(img, disp, shift, segmasks, W, H, focal, baseline, ctx_prog, grid_prog, degrid_mode=1):
    """Inpaint.forward (anime_3dkenburns/models/pointcloud_inpainting.py:116-203) restated over the oracle programs.
    img [1,3,H,W], disp [1,1,H,W], shift [1,3,1]; returns dict like the reference."""
    f32 = np.float32
    depth = ((f32(1.0) / (disp + f32(0.0000001))) * f32(focal * baseline)).astype(f32)           # float / Tensor
    valid = (np.abs(owarp.spatial_filter_laplacian((disp / disp.max()).astype(f32))) < f32(0.03)).astype(f32)
    pts = owarp.depth_to_points((depth * valid).astype(f32), focal).reshape(1, 3, -1)
    mi, md = f32(img.mean(dtype=np.float64)), f32(disp.mean(dtype=np.float64))
    si, sd = f32(img.std(dtype=np.float64)), f32(disp.std(dtype=np.float64))
    ni = ((img - mi) / (si + f32(0.0000001))).astype(f32)
    nd = ((disp - md) / (sd + f32(0.0000001))).astype(f32)
    x = np.ascontiguousarray(np.concatenate([ni, nd], 1))
    ctx = np.zeros((1, 64, H, W), f32)
    onets.run_program(ctx_prog, [x, ctx])
    data = np.concatenate([ni, nd, ctx], 1).reshape(1, 68, -1)
    ps = (pts + shift.reshape(1, 3, 1)).astype(f32)
    render, existing = owarp.render_pointcloud(ps, data, W, H, focal, baseline, degrid_mode=degrid_mode)
    seg_r = None
    if segmasks is not None:
        s = np.concatenate([segmasks, nd], 1).reshape(1, segmasks.shape[1] + 1, -1)
        seg_r, _ = owarp.render_pointcloud(ps, s, W, H, focal, baseline, degrid_mode=degrid_mode)
    existing = (existing > 0.0).astype(f32)
    existing = (existing * owarp.spatial_filter_median5(existing)).astype(f32)
    render = (render * existing).astype(f32)
    gin = np.ascontiguousarray(np.concatenate([render, existing], 1))
    oi, od = np.zeros((1, 3, H, W), f32), np.zeros((1, 1, H, W), f32)
    onets.run_program(grid_prog, [gin, oi, od])
    image = (oi * (si + f32(0.0000001)) + mi).astype(f32)
    dsp = (od * (sd + f32(0.0000001)) + md).astype(f32)
    return dict(existing=existing, image=np.clip(image, 0.0, 1.0), disparity=np.where(dsp > 0, dsp, 0).astype(f32), segmasks=seg_r)


def gray_r_lut():
    return ((1.0 - np.linspace(0.0, 1.0, 256)) * 255).astype(np.uint8)


def _percentile(v, q):
    """np.percentile(v, q) with method 'linear' as numpy 1.26 (the reference's pinned version, conda_env.yaml:280)
    evaluates it: virtual index and lerp in float64.  (numpy >= 2 casts q to the array dtype first and differs in the
    last digits, which is why the fixture -- generated under numpy 2.2 -- is compared with a +-1 level tolerance.)"""
    s = np.sort(v.reshape(-1)); n = s.size
    vi = (n - 1) * (q / 100.0)
    lo = int(math.floor(vi)); hi = min(lo + 1, n - 1)
    a, b, t = float(s[lo]), float(s[hi]), vi - lo
    r = a + (b - a) * t if t < 0.5 else b - (b - a) * (1 - t)
    return np.float32(r)


def colorize_gray_r(value):
    """depth_modules/zoedepth/utils/misc.py:97-135 with cmap='gray_r', channel 0"""
    v = value.astype(np.float32).squeeze()
    vmin, vmax = _percentile(v, 2), _percentile(v, 85)
    x = ((v - vmin) / (vmax - vmin)).astype(np.float32) if vmin != vmax else v * np.float32(0)
    xa = (x * np.float32(256)).astype(np.float32)
    xa[xa == 256] = 255
    under, over = xa < 0, xa >= 256
    k = xa.astype(np.int64)
    k[under] = 0; k[over] = 255
    return gray_r_lut()[np.clip(k, 0, 255)]


def bokeh_blur(img, depth_u8, num_samples, lightness_factor, focal_plane, depth_factor=1):
    """utils/effects.py:143-181 (use_cuda branch); depth may be uint8 or float, focal_plane may be None"""
    L = oseg.lib()
    H, W = img.shape[:2]
    d = depth_u8.astype(np.float32)
    if focal_plane is not None:
        d = d.max() - np.abs(d - np.float32(focal_plane))
    if depth_factor != 1:
        d = np.power(d, depth_factor)
    d = d - d.min()
    d = d.astype(np.float32) / d.max()
    d = ((np.float32(1) - d) * np.float32(0.0005)).astype(np.float32)
    hi = np.power(img.astype(np.float32) / np.float32(255), np.float32(lightness_factor)).astype(np.float32)
    a, b, c = np.empty_like(hi), np.empty_like(hi), np.empty_like(hi)
    PI = math.pi
    for src, dst, (dx, dy) in ((hi, a, (0, 1)), (a, b, (math.cos(-PI / 6), math.sin(-PI / 6))), (b, c, (math.cos(-PI * 5 / 6), math.sin(-PI * 5 / 6)))):
        L.orc_bokeh_pass(_p(np.ascontiguousarray(src)), _p(np.ascontiguousarray(d)), _p(dst), ci(H), ci(W), ci(num_samples), cf(dx), cf(dy))
    bl = np.power((b + c) / np.float32(2), np.float32(1 / lightness_factor))
    return (bl * np.float32(255)).astype(np.uint8)
