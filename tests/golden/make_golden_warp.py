#!/usr/bin/env python3
"""Generate tests/golden/warp_*.npz from the REFERENCE's own code.

Run in the build container only:  python tests/golden/make_golden_warp.py
(needs /root/reference; the fixtures it writes are committed, this script is the
provenance record).  The reference's Python functions render_pointcloud,
fill_disocclusion, process_shift, spatial_filter, depth_to_points are imported by
path (ref_loader.py) and executed on CPU tensors; their CUDA kernel text is
captured from the reference's own preprocess_kernel() expansion and run
sequentially through cuda_on_cpu.h (g++ -ffp-contract=off).
CSM_GOLDEN_FP_CONTRACT=fast python tests/golden/make_golden_warp.py  writes the warp_*_fast.npz twins (g++ -ffp-contract=fast
-mfma): the two builds bracket NVRTC's default FMA contraction.
"""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

mu, co, cu = ref_loader.load_warp_modules()

SNAP = {}
_orig_call = ref_loader._CpuRawKernel.__call__


def _snap(ptr, n):
    return np.ctypeslib.as_array((ctypes.c_float * n).from_address(ptr)).copy()


def _hooked(self, grid, block, args):
    _orig_call(self, grid, block, args)
    want = SNAP.get("_want")
    if want and self.name in want:
        idx, n, key = want[self.name]
        SNAP[key] = _snap(args[idx], n)


ref_loader._CpuRawKernel.__call__ = _hooked


def scene(H, W, seed, extra=0.0, B=1):
    """disparity plane + gaussian bumps -> points via the reference's own
    depth_to_points, plus `extra`*P appended points (like inpainting appends)."""
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    disp = 8.0 + 6.0 * yy / H
    for _ in range(3):
        cx, cy, s, a = g.uniform(0, W), g.uniform(0, H), g.uniform(0.08, 0.2) * W, g.uniform(8, 25)
        disp = disp + a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    disp = disp.astype(np.float32)
    focal, baseline = float(W) / 2.0, 40.0
    disp_t = torch.from_numpy(disp)[None, None]
    disp_t = disp_t / disp_t.max() * baseline
    depth = (focal * baseline) / (disp_t + 0.00001)
    valid = (mu.spatial_filter(disp_t / disp_t.max(), 'laplacian').abs() < 0.03).float()
    pts = mu.depth_to_points(depth * valid, focal).view(1, 3, -1)
    rgb = torch.from_numpy(g.uniform(0, 1, (1, 3, H * W)).astype(np.float32))
    dep = depth.view(1, 1, -1)
    if extra > 0:
        n2 = int(extra * H * W)
        idx = torch.from_numpy(g.integers(0, H * W, n2))
        jitter = torch.from_numpy(g.normal(0, 1.5, (1, 3, n2)).astype(np.float32))
        jitter[:, 2] *= 4.0
        unalt = mu.depth_to_points(depth, focal).view(1, 3, -1)
        pts = torch.cat([pts, unalt[:, :, idx] + jitter], 2)
        rgb = torch.cat([rgb, rgb[:, :, idx]], 2)
        dep = torch.cat([dep, dep[:, :, idx]], 2)
    if B > 1:
        pts = torch.cat([pts] + [pts.flip(2) * (1.0 + 0.01 * b) for b in range(1, B)], 0)
        rgb = torch.cat([rgb] + [rgb.roll(7 * b, 2) for b in range(1, B)], 0)
        dep = torch.cat([dep] + [dep.flip(2) for b in range(1, B)], 0)
    return dict(disp=disp, disp_n=disp_t, depth=depth, valid=valid, pts=pts.contiguous(),
                rgb=rgb.contiguous(), dep=dep.contiguous(), focal=focal, baseline=baseline)


def render_case(name, H, W, seed, C, extra, shift, B=1):
    s = scene(H, W, seed, extra, B)
    focal, baseline = s['focal'], s['baseline']
    N = s['pts'].shape[2]
    if C == 3:
        data = s['rgb']
    elif C == 4:
        data = torch.cat([s['rgb'], s['dep']], 1)
    else:
        g = np.random.default_rng(seed + 99)
        data = torch.from_numpy(g.normal(0, 1, (B, C, N)).astype(np.float32))
    data = data.contiguous()
    # process_shift through the reference (common.py:59-84)
    dmin = float(s['depth'][0, 0].min())
    loc = np.unravel_index(int(s['depth'][0, 0].argmin()), (H, W))
    common = {'objDepthrange': (dmin, float(s['depth'].max()), (int(loc[1]), int(loc[0])), (0, 0)),
              'intWidth': W, 'intHeight': H, 'fltFocal': focal, 'fltBaseline': baseline}
    settings = {'tenPoints': s['pts'], 'fltShiftU': shift[0], 'fltShiftV': shift[1],
                'fltDepthFrom': dmin, 'fltDepthTo': dmin * shift[2]}
    pts_shift, ten_shift = co.process_shift(settings, common)
    pts_shift = pts_shift.contiguous()
    P = H * W
    SNAP.clear()
    SNAP['_want'] = {'kernel_pointrender_updateZee': (3, B * P, 'zee_after_zee'),
                     'kernel_pointrender_updateDegrid': (3, B * P, 'zee_after_degrid'),
                     'kernel_pointrender_updateOutput': (4, B * (C + 1) * P, 'accum')}
    render, existing = mu.render_pointcloud(pts_shift, data, W, H, focal, baseline)
    out = dict(H=H, W=W, C=C, B=B, N=N, focal=focal, baseline=baseline,
               pts=s['pts'].numpy(), shift=ten_shift.numpy().reshape(3), pts_shift=pts_shift.numpy(),
               data=data.numpy(), zee_after_zee=SNAP['zee_after_zee'].reshape(B, 1, H, W),
               zee_after_degrid_inplace=SNAP['zee_after_degrid'].reshape(B, 1, H, W),
               render=render.numpy(), existing=existing.numpy(),
               shift_settings=np.array([shift[0], shift[1], dmin, dmin * shift[2], loc[1], loc[0]], np.float64))
    if C <= 4:
        out['accum'] = SNAP['accum'].reshape(B, C + 1, H, W)
    SNAP.clear()
    if C == 4:
        # frame loop glue (kenburns_effect.py:1037-1040)
        dm = render[:, 3:4] * (existing > 0.0).float()
        filled = co.fill_disocclusion(render, dm)
        frame = (filled[0, 0:3].numpy().transpose(1, 2, 0) * 255.0).clip(0.0, 255.0).astype(np.uint8)
        out.update(fill_depth=dm.numpy(), filled=filled.numpy(), frame=frame)
    if ref_loader.FP_CONTRACT == 'fast':
        # the other end of the FMA-contraction bracket (NVRTC contracts by default): only what the decisions are
        # checked against -- z-buffers, coverage, final images -- under <name>_fast.npz
        keep = ('zee_after_zee', 'zee_after_degrid_inplace', 'existing', 'render', 'filled', 'frame', 'fill_depth')
        out = {k: v for k, v in out.items() if k in keep}
        name = name + '_fast'
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    holes = float((existing == 0).float().mean())
    print(name, 'N', N, 'holes %.3f' % holes, 'bytes', os.path.getsize(os.path.join(HERE, name + '.npz')))


def pointwise_case(name, H, W, seed):
    s = scene(H, W, seed)
    lap = mu.spatial_filter(s['disp_n'] / s['disp_n'].max(), 'laplacian')
    unalt = mu.depth_to_points(s['depth'], s['focal'])
    np.savez_compressed(os.path.join(HERE, name + '.npz'), H=H, W=W, focal=s['focal'], baseline=s['baseline'],
                        disp_raw=s['disp'], disp=s['disp_n'].numpy(), depth=s['depth'].numpy(), lap=lap.numpy(),
                        valid=s['valid'].numpy(), pts=s['pts'].numpy(), unaltered=unalt.numpy())
    print(name, 'ok')


def discfill_case(name, H, W, seed):
    """synthetic large holes incl. a hole touching the border and an all-hole row"""
    g = np.random.default_rng(seed)
    img = torch.from_numpy(g.uniform(0, 1, (2, 4, H, W)).astype(np.float32))
    depth = torch.from_numpy(g.uniform(1, 50, (2, 1, H, W)).astype(np.float32))
    for b in range(2):
        for _ in range(6):
            x0, y0 = g.integers(0, W - 4), g.integers(0, H - 4)
            w, h = g.integers(2, W // 3), g.integers(2, H // 3)
            depth[b, 0, y0:y0 + h, x0:x0 + w] = 0.0
    depth[0, 0, :, 0:3] = 0.0
    depth[1, 0, H // 2] = 0.0
    depth[1, 0, 5, 7] = -1.0
    out = co.fill_disocclusion(img, depth)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), img=img.numpy(), depth=depth.numpy(), out=out.numpy())
    print(name, 'ok')


if __name__ == '__main__':
    torch.manual_seed(0)
    render_case('warp_a_64x48_c4', 48, 64, 11, 4, 0.0, (6.0, -4.0, 0.8))
    render_case('warp_b_96x96_c3_extra', 96, 96, 12, 3, 0.5, (-9.0, 7.0, 0.85))
    render_case('warp_c_40x32_c68', 32, 40, 13, 68, 0.25, (3.0, 2.0, 0.9))
    render_case('warp_d_56x40_c4_b2', 40, 56, 14, 4, 0.3, (-5.0, -5.0, 0.75), B=2)
    render_case('warp_e_80x64_c4_big_shift', 64, 80, 15, 4, 0.0, (30.0, 20.0, 0.6))
    if ref_loader.FP_CONTRACT != 'fast':
        pointwise_case('pointwise_72x56', 56, 72, 21)
        discfill_case('discfill_48x40', 40, 48, 31)
