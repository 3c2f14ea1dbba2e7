#!/usr/bin/env python3
"""k_conv_grouped (vector pipe, csrc/grouped.hip) against the block-diagonal matrix-pipe form of the same grouped 3x3 layer: bit
equality and time per layer on the LeReS (ResNeXt101 32x8d) conv2 shapes.  usage: grouped_bench.py   (CSM_GROUPED_PX=4|5 forces a tile)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cartoonsegmentation_amd import program as P
from cartoonsegmentation_amd.runtime import CompiledProgram


def build(valu, n, h, w, cg, groups):
    old = (P.Program.grouped_valu, P.GROUPED_VALU_MAX_CG)
    P.Program.grouped_valu, P.GROUPED_VALU_MAX_CG = valu, 32
    rng = np.random.default_rng(0)
    c = cg * groups
    p = P.Program('g')
    x_ext = p.ext_nchw(n, c, h, w); y_ext = p.ext_nchw(n, c, h, w)
    x = p.to_nhwc(x_ext)
    wt = (rng.standard_normal((c, cg, 3, 3)) / np.sqrt(9 * cg)).astype(np.float32)
    b = (rng.standard_normal(c) * 0.1).astype(np.float32)
    y = p.conv(x, wt, b, pad=1, groups=groups, act='relu')
    p.to_nchw(y, y_ext)
    P.Program.grouped_valu, P.GROUPED_VALU_MAX_CG = old
    return p


def main():
    for (n, h, w, cg, groups) in [(8, 160, 160, 8, 32), (8, 80, 80, 16, 32), (8, 40, 40, 32, 32), (1, 160, 160, 8, 32), (1, 80, 80, 16, 32),
                                  (1, 40, 40, 32, 32), (2, 256, 256, 8, 32), (2, 128, 128, 16, 32), (2, 64, 64, 32, 32)]:
        res, outs = {}, {}
        c = cg * groups
        x = torch.randn(n, c, h, w, device='cuda')
        for valu in (False, True):
            p = build(valu, n, h, w, cg, groups)
            os.environ["CSM_AUTOTUNE"] = "0" if valu else "1"
            cp = CompiledProgram(p, 'cuda')
            y = torch.empty(n, c, h, w, device='cuda')
            cp.run(x, y); cp.run(x, y)
            ci = [i for i, o in enumerate(p.ops) if o['kind'] == 1][0]
            res[valu] = min(cp.profile(x, y)[ci] for _ in range(7))
            outs[valu] = y.clone()
        fl = 2.0 * n * h * w * c * cg * 9
        byts = 2.0 * n * h * w * c * 4
        print("%2dx%3dx%3d  %4d ch, %2d per group  matrix %7.1f us %6.1f TF/s | vector %7.1f us %6.1f TF/s natural (%.2f of the packed-fp32 peak), %5.2f TB/s in+out  x%.2f  %s" % (
            n, h, w, c, cg, res[False] * 1e3, fl / res[False] / 1e9, res[True] * 1e3, fl / res[True] / 1e9, fl / res[True] / 1e9 / 157.3,
            byts / res[True] / 1e9, res[False] / res[True], "same bits" if torch.equal(outs[False], outs[True]) else "BITS DIFFER"), flush=True)


if __name__ == '__main__':
    main()
