#!/usr/bin/env python3
"""time one conv layer under every tile configuration: args n h w cin cout k [stride dil groups]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cartoonsegmentation_amd.program import Program
from cartoonsegmentation_amd.runtime import CompiledProgram
from cartoonsegmentation_amd import _lib
v = [int(x) for x in sys.argv[1:]]
n, h, w, cin, cout, k = v[:6]
s, d, g = (v[6:] + [1, 1, 1])[:3] if len(v) > 6 else (1, 1, 1)
p = Program("l"); x = p.buffer(n, h, w, cin)
W = (np.random.default_rng(0).standard_normal((cout, cin // g, k, k)) * 0.05).astype(np.float32)
p.conv(x, W, np.zeros(cout, np.float32), stride=s, pad=d * (k // 2), dil=d, groups=g, act='relu'); p.plan()
os.environ["CSM_AUTOTUNE"] = "0"
cp = CompiledProgram(p, 'cuda'); cp.workspace.normal_()
if os.environ.get('ZERO') == '1':      # all-zero operands: same instruction stream, minimum switching power (DVFS check)
    cp.workspace.zero_(); cp.weights.zero_()
L = _lib.load()
names = ["128x128_4w", "128x64", "64x64", "128x128_8w", "128x32", "64x16", "D64x64", "D128x64", "D128x128", "D128x128_8w", "D256x128_8w", "D64x128", "D128x32", "NARROW", "D96x128", "D160x128", "D224x128", "D192x128", "P64x64", "P128x64", "P64x128", "P128x128", "P256x128", "P128x32", "P64x64_w8", "P128x128_w8", "P128x32_w8", "P128x128_8w"] + ["s%d" % i for i in range(28, 38)] + ["Q64x64", "Q128x64", "Q64x128", "Q128x128_8w", "Q128x32", "R128x32", "R64x64", "R128x64", "R128x32_w8", "R128x128_8w", "R64x128", "R64x64_w8"]
print("ksplit", p.ops[0]['ksplit'], "GFLOP %.2f" % (p.flops / 1e9))
for cfg in range(len(names)):
    L.csm_debug_force_conv_cfg(cfg); cp.run()
    ms = min(cp.profile()[0] for _ in range(4))
    print("cfg %2d %-12s %9.1f us  %6.1f TF/s" % (cfg, names[cfg], ms * 1e3, p.flops / ms / 1e9))
