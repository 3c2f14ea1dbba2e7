"""Seeded synthetic inputs: no files, no checkpoints.

numpy only -- shared by bench.py, __graft_entry__.smoke() and the tests so that the HIP
path and the oracle see identical inputs.
"""
import numpy as np


def box_blur(img, k=9):
    """separable box filter on HxWxC float array (edge-replicated)"""
    pad = k // 2
    out = img
    for axis in (0, 1):
        p = np.pad(out, [(pad, pad) if a == axis else (0, 0) for a in range(out.ndim)], mode='edge')
        c = np.cumsum(p, axis=axis, dtype=np.float64)
        c = np.concatenate([np.zeros_like(np.take(c, [0], axis=axis)), c], axis=axis)
        n = out.shape[axis]
        hi = np.take(c, np.arange(k, k + n), axis=axis)
        lo = np.take(c, np.arange(0, n), axis=axis)
        out = (hi - lo) / k
    return out


def image_u8(H=1024, W=1024, seed=1234):
    """low-pass filtered uniform noise, uint8 BGR HxWx3"""
    g = np.random.default_rng(seed)
    raw = g.integers(0, 256, (H, W, 3), dtype=np.uint8).astype(np.float64)
    sm = box_blur(raw, 9)
    sm = (sm - sm.min()) / max(sm.max() - sm.min(), 1e-9) * 255.0
    return sm.astype(np.uint8)


def disparity(H=1024, W=1024, seed=1234):
    """depth plane + 3 flat-topped foreground objects with crisp silhouettes (discs with a
    ~1.5 px wide sigmoid edge at every resolution, like segmented characters) so that camera moves open real disocclusions;
    raw disparity > 0, float32 [H,W]"""
    g = np.random.default_rng(seed + 7)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    d = 6.0 + 8.0 * yy / H
    for _ in range(3):
        cx, cy = g.uniform(0.2, 0.8) * W, g.uniform(0.2, 0.8) * H
        s, a = g.uniform(0.06, 0.14) * min(H, W), g.uniform(10.0, 24.0)
        r = np.sqrt((xx - cx) ** 2 + (yy - cy) ** 2)
        d = d + a / (1.0 + np.exp(np.clip((r - 2.0 * s) / 0.75, -60.0, 60.0)))
    return d.astype(np.float32)


def warp_scene(H=1024, W=1024, seed=1234):
    """inputs of one Ken Burns frame: image [1,3,P] in [0,1], raw disparity [1,1,H,W],
    camera (focal=W/2, baseline=40, kenburns_effect.py:234-235) and an autozoom-like shift request."""
    img = image_u8(H, W, seed).astype(np.float32) * np.float32(1.0 / 255.0)
    rgb = np.ascontiguousarray(img.transpose(2, 0, 1).reshape(1, 3, H * W))
    disp = disparity(H, W, seed)[None, None]
    return dict(H=H, W=W, rgb=rgb, disp=disp, focal=W / 2.0, baseline=40.0,
                shift_u=30.0 * W / 1024.0, shift_v=-20.0 * H / 1024.0, zoom=1.25)


def shift_request(scene, depth_min, depth_min_loc):
    """objSettings/objCommon dicts for process_shift (common.py:59-72) for the scene's camera move"""
    common = {'objDepthrange': (float(depth_min), 0.0, (int(depth_min_loc[0]), int(depth_min_loc[1]))),
              'intWidth': scene['W'], 'intHeight': scene['H'], 'fltFocal': scene['focal'],
              'fltBaseline': scene['baseline']}
    settings = {'fltShiftU': scene['shift_u'], 'fltShiftV': scene['shift_v'], 'fltDepthFrom': float(depth_min),
                'fltDepthTo': float(depth_min) / scene['zoom']}
    return settings, common
