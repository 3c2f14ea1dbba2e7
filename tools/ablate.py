#!/usr/bin/env python3
"""ablation of k_conv_mfma phases on one layer: args n h w cin cout k"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cartoonsegmentation_amd.program import Program
from cartoonsegmentation_amd.runtime import CompiledProgram
from cartoonsegmentation_amd import _lib
n, h, w, cin, cout, k = [int(v) for v in sys.argv[1:7]]
p = Program("l"); p.split_k = False; x = p.buffer(n, h, w, cin)
W = (np.random.default_rng(0).standard_normal((cout, cin, k, k)) * 0.05).astype(np.float32)
p.conv(x, W, np.zeros(cout, np.float32), pad=k // 2, act='relu'); p.plan()
cp = CompiledProgram(p, 'cuda'); cp.workspace.normal_()
L = _lib.load()
names = {0: 'full', 1: 'no-gload', 2: 'no-mfma', 4: 'no-lstore', 5: 'no-gload,no-lstore', 8: 'no-epilogue', 3: 'no-gload,no-mfma', 7: 'only loop+barrier', 13: 'mfma+ldsread only'}
for cfg in (2, 3):
    for dbg in (0, 1, 2, 4, 5, 13, 8, 3, 7):
        L.csm_debug_force_conv_cfg(cfg | (dbg << 8)); cp.run()
        ms = min(cp.profile()[0] for _ in range(4))
        print("cfg %d %-22s %8.1f us  %6.1f TF/s-equiv" % (cfg, names[dbg], ms * 1e3, p.flops / ms / 1e9))
L.csm_debug_force_conv_cfg(-1)
