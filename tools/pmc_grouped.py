#!/usr/bin/env python3
"""one grouped 3x3 layer on k_conv_grouped, a few launches, for rocprofv3 --pmc passes (tools/gpu/pmc_grouped.sh).
usage: pmc_grouped.py n h w channels_per_group groups"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grouped_bench import build
from cartoonsegmentation_amd.runtime import CompiledProgram

n, h, w, cg, groups = [int(v) for v in sys.argv[1:6]]
os.environ["CSM_AUTOTUNE"] = "0"
p = build(True, n, h, w, cg, groups)
cp = CompiledProgram(p, 'cuda')
c = cg * groups
x = torch.randn(n, c, h, w, device='cuda'); y = torch.empty(n, c, h, w, device='cuda')
for _ in range(5):
    cp.run(x, y)
torch.cuda.synchronize()
