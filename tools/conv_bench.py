#!/usr/bin/env python3
"""micro-benchmark of representative conv layers of the frame workload (multiplicity-weighted total)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cartoonsegmentation_amd.program import Program
from cartoonsegmentation_amd.runtime import CompiledProgram

# (mult, n, h, w, cin, cout, k, stride, dil, groups)   multiplicities ~ per-frame counts from tools/layer_profile.py
LAYERS = [
    (2, 2, 360, 360, 64, 64, 3, 1, 1, 1), (1, 2, 360, 360, 128, 64, 3, 1, 1, 1), (2, 2, 180, 180, 64, 128, 3, 1, 1, 1),
    (2, 2, 90, 90, 128, 256, 3, 1, 1, 1), (2, 2, 45, 45, 256, 512, 3, 1, 1, 1), (3, 2, 23, 23, 512, 512, 3, 1, 1, 1),
    (6, 2, 23, 23, 512, 256, 3, 1, 2, 1), (2, 2, 12, 12, 512, 512, 3, 1, 1, 1), (8, 2, 360, 360, 32, 32, 3, 1, 1, 1),
    (8, 2, 90, 90, 32, 32, 3, 1, 1, 1), (10, 2, 360, 360, 16, 16, 3, 1, 1, 1),
    (45, 1, 40, 40, 1024, 1024, 1, 1, 1, 1), (6, 1, 160, 160, 256, 256, 3, 1, 1, 1), (1, 1, 320, 320, 256, 128, 3, 1, 1, 1),
    (22, 1, 40, 40, 1024, 1024, 3, 1, 1, 32), (1, 1, 20, 20, 2048, 512, 3, 1, 1, 1), (5, 1, 80, 80, 256, 256, 3, 1, 1, 1),
    (7, 1, 80, 80, 512, 512, 1, 1, 1, 1), (5, 1, 40, 40, 256, 256, 3, 1, 1, 1), (5, 1, 20, 20, 2048, 2048, 1, 1, 1, 1),
    (5, 1, 160, 160, 256, 256, 1, 1, 1, 1), (3, 1, 160, 160, 256, 256, 3, 1, 1, 32),
    (18, 1, 40, 40, 256, 256, 3, 1, 1, 1), (11, 1, 80, 80, 256, 256, 3, 1, 1, 1), (6, 1, 20, 20, 512, 512, 3, 1, 1, 1),
    (6, 1, 20, 20, 256, 256, 3, 1, 1, 1), (9, 1, 80, 80, 128, 128, 3, 1, 1, 1), (12, 1, 40, 40, 256, 256, 1, 1, 1, 1),
    (9, 1, 80, 80, 128, 128, 1, 1, 1, 1), (6, 1, 20, 20, 1024, 512, 1, 1, 1, 1), (1, 1, 320, 320, 3, 32, 3, 2, 1, 1),
]


SWEEP = '--sweep' in sys.argv


def main():
    tot_ms = tot_fl = best_ms = 0.0
    rows = []
    for (mult, n, h, w, cin, cout, k, s, d, g) in LAYERS:
        p = Program("l")
        cin_t = (cin + 3) // 4 * 4
        x = p.buffer(n, h * s if s > 1 else h, w * s if s > 1 else w, cin_t)
        x.buf.first = 0
        W = (np.random.default_rng(0).standard_normal((cout, cin_t // g, k, k)) * 0.05).astype(np.float32)
        y = p.conv(x, W, np.zeros(cout, np.float32), stride=s, pad=d * (k // 2), dil=d, groups=g, act='relu')
        p.plan()
        cp = CompiledProgram(p, 'cuda')
        cp.workspace.normal_()
        cp.run()
        from cartoonsegmentation_amd import _lib
        L = _lib.load()
        per = []
        for cfg in ([-1] + (list(range(28)) if SWEEP else [])):
            L.csm_debug_force_conv_cfg(cfg)
            cp.run()
            per.append(min(cp.profile()[0] for _ in range(4)))
        L.csm_debug_force_conv_cfg(-1)
        ms = per[0]
        fl = p.flops
        rows.append((mult * ms, mult, ms, fl, "%dx%dx%dx%d->%d k%d s%d d%d g%d S%d" % (n, y.h, y.w, cin, cout, k, s, d, g, p.ops[0]['ksplit']), per))
        tot_ms += mult * ms; tot_fl += mult * fl
        best_ms += mult * min(per)
    for tm, mult, ms, fl, desc, per in sorted(rows, key=lambda r: -r[0]):
        print("%7.3f ms = x%-2d %7.1f us  %6.1f TF/s  %-36s %s" % (tm, mult, ms * 1e3, fl / ms / 1e9, desc,
              " ".join("%5.0f" % (fl / t / 1e9) for t in per[1:])))
    print("TOTAL %.3f ms  %.1f GFLOP  %.1f TF/s   (best-of-sweep total %.3f ms)" % (tot_ms, tot_fl / 1e9, tot_fl / tot_ms / 1e9, best_ms))


if __name__ == '__main__':
    main()
