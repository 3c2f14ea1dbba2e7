"""CPU: the build's lowering (program.py + nets/*) executed by the oracle interpreter
(oracle/nets_oracle.c) vs fixtures from the reference's own torch modules."""
import os

import numpy as np
import pytest

from cartoonsegmentation_amd.weights import SynthWeights
from oracle import nets as onets

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def rel_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.mark.parametrize("tag", ["64x64", "90x74"])
def test_isnet_vs_reference_module(tag):
    from cartoonsegmentation_amd.nets import build_isnet
    g = dict(np.load(os.path.join(GOLDEN, "net_isnet_%s.npz" % tag)))
    n, c, h, w = g['x'].shape
    prog = build_isnet(SynthWeights('isnet.'), n, h, w)
    y = np.zeros((n, 1, h, w), np.float32)
    onets.run_program(prog, [np.ascontiguousarray(g['x']), y])
    # BN folding + summation order differ from torch's kernels: fp32 roundoff level
    assert rel_err(y, g['d1']) < 2e-4, rel_err(y, g['d1'])
    thr = np.log(0.3 / 0.7)                       # sigmoid(x) > 0.3  (mask_thr, animeinsseg/__init__.py:662)
    assert ((y > thr) != (g['d1'] > thr)).mean() < 1e-3
