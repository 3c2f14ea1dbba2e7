#!/usr/bin/env python3
"""time ops.process_autozoom (batched coverage kernels) at a given frame size; prints ms per search and per candidate"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cartoonsegmentation_amd import ops, synth  # noqa: E402

H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sc = synth.warp_scene(H, W, 7)
dev = torch.device('cuda')
disp = torch.from_numpy(sc['disp']).to(dev)
disp = disp / disp.max() * sc['baseline']
depth, valid, pts, _ = ops.disparity_to_points(disp, sc['focal'], sc['baseline'])
b = min(128, H // 8)
crop = depth[0, 0, b:-b, b:-b]
dmin = float(crop.min()); loc = int(crop.argmin())
common = {'objDepthrange': (dmin, float(crop.max()), (loc % crop.shape[1], loc // crop.shape[1]), (0, 0)), 'intWidth': W, 'intHeight': H,
          'fltFocal': sc['focal'], 'fltBaseline': sc['baseline'], 'tenRawPoints': pts.view(1, 3, -1).contiguous()}
objFrom = {'fltCenterU': W / 2.0, 'fltCenterV': H / 2.0, 'intCropWidth': int(np.floor(0.97 * W)), 'intCropHeight': int(np.floor(0.97 * H))}
settings = {'fltShift': 100.0 * W / 1024.0, 'fltZoom': 1.25, 'objFrom': objFrom}
for chunk in (0, 32, 8):
    if chunk:
        os.environ['CSM_AUTOZOOM_CHUNK'] = str(chunk)
        os.environ['CSM_AUTOZOOM_PATH'] = 'planes'
    to, cands, counts = ops.process_autozoom(settings, common, return_counts=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        to = ops.process_autozoom(settings, common)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print("autozoom %dx%d chunk %2d (0 = LDS band path): %d candidates, %.2f ms per search (%.1f us per candidate), best coverage %.4f"
          % (W, H, chunk, len(cands), ms, ms * 1e3 / max(len(cands), 1), max(counts) / (H * W)), flush=True)
