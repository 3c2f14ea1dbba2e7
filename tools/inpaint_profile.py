#!/usr/bin/env python3
"""kernel-level view of the two inpaint passes of a shipped-yaml video (1024x1024): run under
rocprofv3 --kernel-trace --output-format csv; the script prints the wall time of the inpaint section and, from ROCTX-free
timestamps it records itself, nothing else -- the per-kernel sums come from the trace (tools: see profiles/README.md)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CSM_SYNTHETIC_WEIGHTS"] = "1"
from cartoonsegmentation_amd import synth  # noqa: E402
from cartoonsegmentation_amd.kenburns import KenBurnsConfig, KenBurnsPipeline  # noqa: E402

size = 1024
cfg = KenBurnsConfig(det_ckpt='synthetic', depth_est='leres', depth_est_size=640, max_size=size, refine_crf=False, depth_field=False,
                     focal=size / 2.0, mask_refine_kwargs={'refine_method': 'refinenet_isnet', 'refine_size': 720})
pipe = KenBurnsPipeline(cfg)
pipe.max_instances = 2
img = torch.from_numpy(synth.image_u8(size, size, 1234)).cuda()
objFrom = {'fltCenterU': size / 2.0, 'fltCenterV': size / 2.0, 'intCropWidth': int(np.floor(0.97 * size)), 'intCropHeight': int(np.floor(0.97 * size))}
for rep in range(3):
    kc = pipe.generate_kenburns_config(img)
    objTo = pipe.process_autozoom({'fltShift': 100.0, 'fltZoom': 1.25, 'objFrom': objFrom}, kc)
    settings = {'fltSteps': [0.0, 1.0], 'objFrom': objFrom, 'objTo': objTo, 'boolInpaint': True}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    marker = torch.zeros(7, device='cuda') + rep                  # a recognisable tiny kernel in the trace: start of the inpaint section
    pipe.process_kenburns(settings, kc, True, False, to_numpy=False)        # 2 inpaint passes + 2 frames
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("rep %d: 2 x inpaint + 2 frames %.2f ms" % (rep, (t1 - t0) * 1e3), flush=True)
