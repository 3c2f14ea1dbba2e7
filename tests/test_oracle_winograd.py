"""CPU checks of the Winograd F(2x2, 3x3) contract (include/csm355.h "Winograd contract"; oracle/nets_oracle.c::orc_conv_wino;
host packing in cartoonsegmentation_amd/program.py).  The GPU side (k_conv_wino == oracle, bit for bit) is tests/test_gpu_winograd.py."""
import ctypes

import numpy as np
import pytest

from cartoonsegmentation_amd import program as P
from oracle import nets as onets


def _layer(wino, n, h, w, cin, cout, act='relu', res_mode=0, seed=1):
    rng = np.random.default_rng(seed)
    old = (P.Program.winograd, P.WINO_MIN_PIXELS, P.Program.winograd4)
    P.Program.winograd, P.WINO_MIN_PIXELS, P.Program.winograd4 = wino, 0, False       # (F(2x2) here; F(4x4): test_oracle_winograd4.py)
    try:
        p = P.Program('t')
        x_ext = p.ext_nchw(n, cin, h, w)
        y_ext = p.ext_nchw(n, cout, h, w)
        x = p.to_nhwc(x_ext)
        wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
        b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        res = None
        if res_mode:
            res = p.to_nhwc(p.ext_nchw(n, cout, h, w))
        y = p.conv(x, wt, b, pad=1, act=act, res=res, res_mode=res_mode)
        p.to_nchw(y, y_ext)
    finally:
        P.Program.winograd, P.WINO_MIN_PIXELS, P.Program.winograd4 = old
    return p


@pytest.mark.parametrize("n,h,w,cin,cout,act,res_mode", [
    (1, 8, 32, 32, 64, None, 0),
    (2, 13, 37, 64, 64, 'silu', 2),            # odd height and width: the last tile row / column is half outside
    (1, 45, 45, 96, 128, 'relu', 1),
    (1, 7, 5, 256, 64, 'relu', 0),             # a map smaller than one block tile
    (2, 1, 1, 32, 64, 'relu', 0),              # one pixel: a single Winograd tile, three quarters of it outside
    (1, 2, 67, 64, 128, None, 0),
])
def test_winograd_oracle_equals_direct_oracle_to_rounding(n, h, w, cin, cout, act, res_mode):
    """the two arithmetics compute the same convolution: they differ by fp32 rounding only (transform constants 0, +-1, +-1/2)"""
    rng = np.random.default_rng(2)
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    r = rng.standard_normal((n, cout, h, w)).astype(np.float32)
    outs = []
    for wino in (False, True):
        p = _layer(wino, n, h, w, cin, cout, act, res_mode)
        assert any(o['flags'] & P.CONV_FLAG_WINOGRAD for o in p.ops) == wino
        assert all(o['ksplit'] == 1 for o in p.ops if o['flags'] & P.CONV_FLAG_WINOGRAD)
        y = np.zeros((n, cout, h, w), np.float32)
        onets.run_program(p, [x, y] + ([r] if res_mode else []))
        outs.append(y)
    assert np.isfinite(outs[1]).all()
    assert np.abs(outs[0] - outs[1]).max() <= 2e-6 * max(1.0, np.abs(outs[0]).max())


def test_host_weight_transform_is_bitwise_the_oracles():
    """U = G g G^T: the product packs its own copy (numpy float64, elementwise, fixed order), the oracle derives it from the natural
    weights in C -- same expression, same IEEE operations, so the fp32 panels are identical; and the packed device layout is the
    documented [co / 64][cb][q][f][h][co % 64][4] image of U"""
    rng = np.random.default_rng(3)
    cout, cin = 128, 96
    w = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)
    w[0, 0] = [[1e-30, 1.0, -1.0], [3.0, 1e8, 1.0], [-1e8, 2.0, 0.5]]              # cancellation / wide dynamic range
    U = np.zeros((cout, cin, 16), np.float32)
    onets.lib().orc_wino_transform_weights(w.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(cout), ctypes.c_int(cin),
                                           U.ctypes.data_as(ctypes.c_void_p))
    Up = P.wino_transform(w)
    assert Up.shape == (16, cout, cin) and np.array_equal(U.transpose(2, 0, 1), Up)
    # identity checks of the transform itself: centre-tap delta -> G e G^T, and sum_f over the interpolation points
    d = np.zeros((1, 1, 3, 3), np.float32); d[0, 0, 1, 1] = 1.0
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
    assert np.array_equal(P.wino_transform(d).reshape(4, 4), (G @ d[0, 0] @ G.T).astype(np.float32))
    packed = P.pack_wino_weights(w).reshape(cout // 64, cin // 32, 4, 16, 2, 64, 4)
    for (nt, cb, q, f, h, co, e) in [(0, 0, 0, 0, 0, 0, 0), (1, 2, 3, 15, 1, 63, 3), (0, 1, 2, 7, 1, 5, 2)]:
        assert packed[nt, cb, q, f, h, co, e] == Up[f, nt * 64 + co, cb * 32 + q * 8 + h * 4 + e]


def test_winograd_rule_is_per_sample_and_switchable():
    """the lowering decides per layer from ONE sample's shape (never the batch): same flag at batch 1 and 8; ineligible layers keep the
    direct chain; Program.winograd = False (CSM_CONV_EXACT_DIRECT=1) lowers everything to the round-1..4 arithmetic"""
    def flags(n, h, w, cin, cout, k=3, stride=1, pad=1, dil=1, groups=1):
        p = P.Program('r')
        x = p.buffer(n, h, w, cin)
        p.conv(x, np.zeros((cout, cin // groups, k, k), np.float32), None, stride=stride, pad=pad, dil=dil, groups=groups)
        return p.ops[-1]['flags'], p.ops[-1]['ksplit']
    assert P.WINO_ENABLE and P.Program.winograd
    old4, P.Program.winograd4 = P.Program.winograd4, False
    big = int(np.ceil(np.sqrt(P.WINO_MIN_PIXELS)))
    assert flags(1, big, big, 64, 64) == (P.CONV_FLAG_WINOGRAD, 1) and flags(8, big, big, 64, 64) == (P.CONV_FLAG_WINOGRAD, 1)
    small = max(2, big // 2 - 1)
    assert flags(1, small, small, 64, 64)[0] == 0 and flags(8, small, small, 64, 64)[0] == 0
    assert flags(1, big, big, 64, 32)[0] == 0                       # cout % 64
    assert flags(1, big, big, 48, 64)[0] == 0                       # cin % 32
    assert flags(1, big, big, 64, 64, stride=2)[0] == 0 and flags(1, big, big, 64, 64, pad=2, dil=2)[0] == 0
    assert flags(1, big, big, 64, 64, groups=2)[0] == 0 and flags(1, big, big, 64, 64, k=1, pad=0)[0] == 0
    P.Program.winograd = False
    try:
        assert flags(1, big, big, 64, 64)[0] == 0
    finally:
        P.Program.winograd = P.WINO_ENABLE
        P.Program.winograd4 = old4
