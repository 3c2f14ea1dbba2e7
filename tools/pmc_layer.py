#!/usr/bin/env python3
"""run ONE conv layer many times (for rocprofv3 --pmc): args: n h w cin cout k [cfg [groups]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cartoonsegmentation_amd.program import Program
from cartoonsegmentation_amd.runtime import CompiledProgram
from cartoonsegmentation_amd import _lib
n, h, w, cin, cout, k = [int(v) for v in sys.argv[1:7]]
cfg = int(sys.argv[7]) if len(sys.argv) > 7 else -1
groups = int(sys.argv[8]) if len(sys.argv) > 8 else 1
p = Program("l"); x = p.buffer(n, h, w, cin)
W = (np.random.default_rng(0).standard_normal((cout, cin // groups, k, k)) * 0.05).astype(np.float32)
p.conv(x, W, np.zeros(cout, np.float32), pad=k // 2, groups=groups, act='relu'); p.plan()
cp = CompiledProgram(p, 'cuda'); cp.workspace.normal_()
_lib.load().csm_debug_force_conv_cfg(cfg)
for _ in range(10):
    cp.run()
torch.cuda.synchronize()
print("ok ksplit", p.ops[0]['ksplit'])
