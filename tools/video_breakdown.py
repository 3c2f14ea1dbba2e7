#!/usr/bin/env python3
"""stage times of one shipped-yaml video (1024x1024): config, autozoom, 2 x inpaint, 75 x (warp, colorize, bokeh, crop)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CSM_SYNTHETIC_WEIGHTS"] = "1"
from cartoonsegmentation_amd import ops, synth  # noqa: E402
from cartoonsegmentation_amd.kenburns import KenBurnsConfig, KenBurnsPipeline  # noqa: E402

size = 1024
cfg = KenBurnsConfig(det_ckpt='synthetic', depth_est='leres', depth_est_size=640, max_size=size, refine_crf=False, depth_field=True,
                     focal=size / 2.0, mask_refine_kwargs={'refine_method': 'refinenet_isnet', 'refine_size': 720})
pipe = KenBurnsPipeline(cfg)
pipe.max_instances = 2
img = torch.from_numpy(synth.image_u8(size, size, 1234)).cuda()


def sync():
    torch.cuda.synchronize(); return time.perf_counter()


for rep, ns in enumerate((1, 1, 3)):            # rep 0 builds / tunes; then the serial frame loop and the 3-streams loop (the default)
    pipe.frame_streams = ns
    t0 = sync()
    kc = pipe.generate_kenburns_config(img); t1 = sync()
    objFrom = {'fltCenterU': size / 2.0, 'fltCenterV': size / 2.0, 'intCropWidth': int(np.floor(0.97 * size)), 'intCropHeight': int(np.floor(0.97 * size))}
    objTo = pipe.process_autozoom({'fltShift': 100.0, 'fltZoom': 1.25, 'objFrom': objFrom}, kc); t2 = sync()
    settings = {'fltSteps': np.linspace(0.0, 1.0, 75).tolist(), 'objFrom': objFrom, 'objTo': objTo, 'boolInpaint': True}
    kc.depth_field = False
    pipe.process_kenburns(settings, kc, True, False, to_numpy=False); t3 = sync()          # inpaint + 75 plain frames
    pipe.process_kenburns(settings, kc, False, False, to_numpy=False); t4 = sync()         # 75 plain frames (cloud already inpainted)
    kc.depth_field = True
    pipe.process_kenburns(settings, kc, False, False, to_numpy=False); t5 = sync()         # 75 frames with bokeh
    print("rep %d (frame_streams %d): config %.1f ms, autozoom %.1f ms, 2 x inpaint %.1f ms, 75 plain frames %.1f ms (%.0f us each), 75 bokeh frames %.1f ms "
          "(%.0f us each), N after inpaint %d" % (rep, ns, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2 - (t4 - t3)) * 1e3, (t4 - t3) * 1e3,
                                                    (t4 - t3) / 75 * 1e6, (t5 - t4) * 1e3, (t5 - t4) / 75 * 1e6, kc['tenInpaPoints'].shape[2]), flush=True)
