#!/usr/bin/env python3
"""thin-output conv layers (ISNet RSU inner layers, Inpaint / Disparity GridNet rows): tuned tile vs the tiles in CFGS"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cartoonsegmentation_amd.program import Program
from cartoonsegmentation_amd.runtime import CompiledProgram
LAYERS = [(16, 360, 360, 32, 64), (16, 360, 360, 64, 32), (16, 360, 360, 64, 16), (16, 360, 360, 32, 32), (16, 360, 360, 16, 32), (16, 360, 360, 16, 16),
          (1, 1024, 1024, 32, 32), (1, 512, 512, 64, 64), (1, 256, 256, 96, 96), (1, 1024, 1024, 72, 32), (16, 180, 180, 32, 32), (16, 180, 180, 16, 16)]
CFGS = [int(c) for c in os.environ.get('CFGS', '12 23 26 6 18 24').split()]
from cartoonsegmentation_amd import _lib
L = _lib.load()
print("cfgs", CFGS)
for (n, h, w, cin, cout) in LAYERS:
    p = Program("l")
    cin_t = (cin + 3) // 4 * 4
    x = p.buffer(n, h, w, cin_t); x.buf.first = 0
    W = (np.random.default_rng(0).standard_normal((cout, cin_t, 3, 3)) * 0.05).astype(np.float32)
    p.conv(x, W, np.zeros(cout, np.float32), pad=1, act='relu')
    p.plan()
    cp = CompiledProgram(p, 'cuda'); cp.workspace.normal_(); cp.run(); ref = cp.workspace.clone()
    per, same = [], []
    for cfg in [-1] + CFGS:
        L.csm_debug_force_conv_cfg(cfg); cp.run()
        per.append(min(cp.profile()[0] for _ in range(5))); same.append(bool(torch.equal(cp.workspace, ref)))
    L.csm_debug_force_conv_cfg(-1)
    fl = p.flops
    print("%-26s tuned T%-2d %7.1f us %6.1f TF/s | %s" % ("%dx%dx%dx%d->%d" % (n, h, w, cin, cout), cp.ops[0].tile - 1, per[0] * 1e3, fl / per[0] / 1e9,
          " ".join("%5.1f%s" % (fl / t / 1e9, "" if ok else "!") for t, ok in zip(per[1:], same[1:]))), flush=True)
