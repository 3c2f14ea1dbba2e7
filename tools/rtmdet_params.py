#!/usr/bin/env python3
"""Per-module learnable-parameter table of the lowered RTMDet-Ins (nets/rtmdet.py), reconciled with mmdet's published counts.

mmdet 3.3.0 is not under /root/reference, so the detector is restated ("parity unpinned").  What CAN be checked
without mmdet: the number of learnable parameters of every published RTMDet-Ins size (mmdet model zoo, configs/rtmdet/README.md:
tiny 5.6 M, s 10.18 M, m 27.58 M, l 57.37 M, x 102.7 M -- all with the 80-class COCO head).  Five independent numbers, one
architecture rule: any structural mistake (a missing conv, a wrongly shared tower, a wrong expand ratio) shows up as a mismatch
in at least one of them.

Usage: python tools/rtmdet_params.py [--table l]
"""
import argparse
import os
import sys
from collections import OrderedDict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cartoonsegmentation_amd.nets.rtmdet import RTMDetConfig, build_rtmdet  # noqa: E402

LEARNABLE = {'conv_w', 'conv_b', 'bn_gamma', 'bn_beta', 'prelu'}
SIZES = {   # name: (deepen, widen, published M params of rtmdet-ins_<name>, 80 classes)
    'tiny': (0.167, 0.375, 5.6), 's': (0.33, 0.5, 10.18), 'm': (0.67, 0.75, 27.58), 'l': (1.0, 1.0, 57.37), 'x': (1.33, 1.25, 102.7)}


class CountingWeights:
    """records every distinct parameter name the builder asks for (a shared conv is asked for once per alias, counted once)"""

    def __init__(self):
        self.params = OrderedDict()

    def get(self, name, shape, kind):
        if kind in LEARNABLE:
            self.params[name] = int(np.prod(shape))
        if kind == 'bn_var':
            return np.ones(tuple(shape), np.float32)
        return np.zeros(tuple(shape), np.float32)


def count(size, num_classes=80, **over):
    d, w, _ = SIZES[size]
    cfg = RTMDetConfig(deepen_factor=d, widen_factor=w, num_classes=num_classes, feat_channels=int(256 * w), **over)
    ws = CountingWeights()
    build_rtmdet(ws, 1, 64, 64, cfg)
    return ws.params


def table(params, depth=2):
    out = OrderedDict()
    for k, v in params.items():
        key = '.'.join(k.split('.')[:depth])
        out[key] = out.get(key, 0) + v
    return out


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--table', default=None, help='print the per-module table of this size')
    a = ap.parse_args()
    print('%-5s %12s %10s %10s' % ('size', 'params', 'M', 'published'))
    for s, (_, _, pub) in SIZES.items():
        n = sum(count(s).values())
        print('%-5s %12d %10.3f %10.2f' % (s, n, n / 1e6, pub))
    n1 = sum(count('l', num_classes=1).values())
    print('l with the checkpoint\'s single class: %d (%.3f M)' % (n1, n1 / 1e6))
    if a.table:
        for k, v in table(count(a.table), 3 if a.table else 2).items():
            print('  %-44s %10d' % (k, v))
