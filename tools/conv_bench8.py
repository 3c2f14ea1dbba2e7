#!/usr/bin/env python3
"""conv layers of the frame workload at the bench's batch (8 frames; ISNet 16 instances), multiplicity-weighted.
CFGS="6 28 31" sweeps those tile configurations next to the tuned one; DBG=16 adds phase/experiment bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cartoonsegmentation_amd.program import Program  # (CSM_LIB=<path> selects another build of the library)
from cartoonsegmentation_amd.runtime import CompiledProgram

# (mult per step, n, h, w, cin, cout, k, stride, dil, groups)
LAYERS = [
    (6, 8, 160, 160, 256, 256, 3, 1, 1, 1), (45, 8, 40, 40, 1024, 1024, 1, 1, 1, 1), (1, 8, 320, 320, 256, 128, 3, 1, 1, 1),
    (5, 8, 80, 80, 256, 256, 3, 1, 1, 1), (22, 8, 40, 40, 1024, 1024, 3, 1, 1, 32), (7, 8, 80, 80, 512, 512, 1, 1, 1, 1),
    (5, 8, 160, 160, 256, 256, 1, 1, 1, 1), (5, 8, 20, 20, 2048, 2048, 1, 1, 1, 1), (3, 8, 160, 160, 256, 256, 3, 1, 1, 32),
    (3, 8, 80, 80, 512, 512, 3, 1, 1, 32), (5, 8, 40, 40, 256, 256, 3, 1, 1, 1),
    (2, 16, 360, 360, 64, 64, 3, 1, 1, 1), (1, 16, 360, 360, 128, 64, 3, 1, 1, 1), (2, 16, 180, 180, 64, 128, 3, 1, 1, 1),
    (2, 16, 90, 90, 128, 256, 3, 1, 1, 1), (1, 16, 180, 180, 256, 64, 3, 1, 1, 1), (2, 16, 45, 45, 256, 512, 3, 1, 1, 1),
    (2, 16, 23, 23, 512, 512, 3, 1, 1, 1), (1, 16, 360, 360, 32, 64, 3, 1, 1, 1),
    (9, 8, 80, 80, 256, 256, 3, 1, 1, 1), (16, 8, 40, 40, 256, 256, 3, 1, 1, 1), (9, 8, 80, 80, 128, 128, 3, 1, 1, 1),
    (6, 8, 20, 20, 512, 512, 3, 1, 1, 1), (12, 8, 40, 40, 256, 256, 1, 1, 1, 1), (9, 8, 80, 80, 128, 128, 1, 1, 1, 1),
    (3, 8, 160, 160, 64, 64, 3, 1, 1, 1),
    # 26..: the GridNet of the point-cloud inpainting at 1024^2 (per pass)
    (11, 1, 1024, 1024, 32, 32, 3, 1, 1, 1), (2, 1, 1024, 1024, 64, 32, 3, 1, 1, 1), (10, 1, 512, 512, 64, 64, 3, 1, 1, 1),
    (1, 16, 360, 360, 64, 32, 3, 1, 1, 1),
    # 30..: ISNet's narrow layers (side outputs, the 16-channel stage) and LeReS's last conv
    (1, 16, 360, 360, 64, 1, 3, 1, 1, 1), (1, 16, 180, 180, 64, 1, 3, 1, 1, 1), (1, 16, 360, 360, 64, 16, 3, 1, 1, 1), (1, 8, 320, 320, 128, 1, 3, 1, 1, 1),
]
CFGS = [int(c) for c in os.environ.get('CFGS', '').split()]
DBG = int(os.environ.get('DBG', '0'))
ONLY = [int(c) for c in os.environ.get('ONLY', '').split()]
SCALE_N = int(os.environ.get('SCALE_N', '1'))          # 8: the single-frame step's layers (n = 1; ISNet 2 instances)
SERIAL = int(os.environ.get('SERIAL', '-1'))            # split-K execution of the forced configurations: 0 parallel, 1 serial


def main():
    from cartoonsegmentation_amd import _lib
    L = _lib.load()
    if os.environ.get('TUNER_OPTIONS'):
        L.csm_debug_conv_tuner_options(int(os.environ['TUNER_OPTIONS']))
    tot = best = 0.0
    print("cfgs:", CFGS, "dbg:", DBG)
    for li, (mult, n, h, w, cin, cout, k, s, d, g) in enumerate(LAYERS):
        if ONLY and li not in ONLY:
            continue
        n = max(1, n // SCALE_N)
        p = Program("l")
        x = p.buffer(n, h, w, cin)
        x.buf.first = 0
        W = (np.random.default_rng(0).standard_normal((cout, cin // g, k, k)) * 0.05).astype(np.float32)
        y = p.conv(x, W, np.zeros(cout, np.float32), stride=s, pad=d * (k // 2), dil=d, groups=g, act='relu')
        p.plan()
        cp = CompiledProgram(p, 'cuda')
        cp.workspace.normal_()
        cp.run()
        ref = cp.workspace.clone()
        per, same = [], []
        for cfg in [-1] + CFGS:
            L.csm_debug_force_conv_cfg(cfg if cfg < 0 else cfg | (DBG << 8))
            L.csm_debug_force_splitk_serial(SERIAL if cfg >= 0 else -1)
            cp.run()
            per.append(min(cp.profile()[0] for _ in range(5)))
            same.append(bool(torch.equal(cp.workspace, ref)))
        L.csm_debug_force_conv_cfg(-1)
        L.csm_debug_force_splitk_serial(-1)
        fl = p.flops
        tot += mult * per[0]; best += mult * min(per)
        print("%2d x%-2d %8.1f us %6.1f TF/s T%-2d %-40s | %s" % (li, mult, per[0] * 1e3, fl / per[0] / 1e9, cp.ops[0].tile - 1,
              "%dx%dx%dx%d->%d k%d g%d S%d" % (n, h, w, cin, cout, k, g, p.ops[0]['ksplit']),
              " ".join("%5.1f%s" % (fl / t / 1e9, "" if ok else "!") for t, ok in zip(per[1:], same[1:]))), flush=True)
    print("TOTAL tuned %.3f ms, best-of-all %.3f ms" % (tot, best))


if __name__ == '__main__':
    main()
