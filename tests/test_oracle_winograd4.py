"""CPU checks of the Winograd F(4x4, 3x3) contract (include/csm355.h "Winograd F(4x4) contract"; oracle/nets_oracle.c::orc_conv_wino4;
host packing in cartoonsegmentation_amd/program.py).  The GPU side (k_conv_wino4 == oracle, bit for bit) is tests/test_gpu_winograd4.py."""
import ctypes

import numpy as np
import pytest

from cartoonsegmentation_amd import program as P
from oracle import nets as onets


class forced:
    """lower every eligible 3x3 layer to: 'direct', 'f2' (F(2x2)) or 'f4' (F(4x4)) whatever the per-sample size rules say"""
    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.old = (P.Program.winograd, P.Program.winograd4, P.WINO_MIN_PIXELS, P.WINO4_MIN_PIXELS)
        P.Program.winograd, P.Program.winograd4 = self.mode != 'direct', self.mode == 'f4'
        P.WINO_MIN_PIXELS = P.WINO4_MIN_PIXELS = 0

    def __exit__(self, *a):
        P.Program.winograd, P.Program.winograd4, P.WINO_MIN_PIXELS, P.WINO4_MIN_PIXELS = self.old


def layer(mode, n, h, w, cin, cout, act='relu', res_mode=0, seed=1):
    rng = np.random.default_rng(seed)
    with forced(mode):
        p = P.Program('t')
        x_ext = p.ext_nchw(n, cin, h, w)
        y_ext = p.ext_nchw(n, cout, h, w)
        x = p.to_nhwc(x_ext)
        wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
        b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        res = p.to_nhwc(p.ext_nchw(n, cout, h, w)) if res_mode else None
        y = p.conv(x, wt, b, pad=1, act=act, res=res, res_mode=res_mode)
        p.to_nchw(y, y_ext)
    return p


@pytest.mark.parametrize("n,h,w,cin,cout,act,res_mode", [
    (1, 16, 32, 32, 64, None, 0),
    (2, 13, 37, 64, 64, 'silu', 2),            # odd height and width: the last tile row / column is partly outside
    (1, 45, 45, 96, 128, 'relu', 1),
    (1, 7, 5, 256, 64, 'relu', 0),             # a map smaller than one block tile
    (2, 1, 1, 32, 64, 'relu', 0),              # one pixel: a single Winograd tile, fifteen sixteenths of it outside
    (1, 2, 67, 64, 128, None, 0),
])
def test_winograd4_oracle_equals_direct_oracle_to_rounding(n, h, w, cin, cout, act, res_mode):
    """the three arithmetics compute the same convolution; F(4x4) differs from the direct chain by a few times the direct chain's own
    fp32 error (transform entries up to 8 and 1/24: measured 4-5x), far inside north_star's 1e-3"""
    rng = np.random.default_rng(2)
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    r = rng.standard_normal((n, cout, h, w)).astype(np.float32)
    outs = {}
    for mode in ('direct', 'f2', 'f4'):
        p = layer(mode, n, h, w, cin, cout, act, res_mode)
        fl = [o['flags'] for o in p.ops if o['kind'] == P.OP_CONV]
        assert fl == [{'direct': 0, 'f2': P.CONV_FLAG_WINOGRAD, 'f4': P.CONV_FLAG_WINOGRAD4}[mode]]
        assert all(o['ksplit'] == 1 for o in p.ops if o['flags'])
        y = np.zeros((n, cout, h, w), np.float32)
        onets.run_program(p, [x, y] + ([r] if res_mode else []))
        outs[mode] = y
    assert np.isfinite(outs['f4']).all()
    scale = max(1.0, np.abs(outs['direct']).max())
    assert np.abs(outs['direct'] - outs['f4']).max() <= 2e-5 * scale
    assert np.abs(outs['direct'] - outs['f2']).max() <= 2e-6 * scale


def test_host_weight_transform4_is_bitwise_the_oracles():
    """U = G g G^T (36 frequencies): numpy float64 elementwise == the oracle's C doubles (IEEE division by 6 and 24 on both sides), and
    the packed device layout is the documented one"""
    rng = np.random.default_rng(3)
    cout, cin = 128, 96
    w = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)
    w[0, 0] = [[1e-30, 1.0, -1.0], [3.0, 1e8, 1.0], [-1e8, 2.0, 0.5]]              # cancellation / wide dynamic range
    U = np.zeros((cout, cin, 36), np.float32)
    onets.lib().orc_wino4_transform_weights(w.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(cout), ctypes.c_int(cin),
                                            U.ctypes.data_as(ctypes.c_void_p))
    Up = P.wino4_transform(w)
    assert Up.shape == (36, cout, cin) and np.array_equal(U.transpose(2, 0, 1), Up)
    G = np.array([[.25, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]])
    d = rng.standard_normal((1, 1, 3, 3)).astype(np.float32)
    assert np.allclose(P.wino4_transform(d).reshape(6, 6), G @ d[0, 0].astype(np.float64) @ G.T, rtol=1e-6, atol=1e-7)
    nsteps = cin // 4
    packed = P.pack_wino4_weights(w).reshape(cout // 64, nsteps, 12, 3, 2, 32, 2, 2)         # nt, s, wave, p, lh, li, jj, t
    for (nt, s, wave, pp, lh, li, jj, t) in [(0, 0, 0, 0, 0, 0, 0, 0), (1, 23, 11, 2, 1, 31, 1, 1), (0, 7, 8, 1, 0, 5, 1, 0), (1, 10, 3, 2, 1, 17, 0, 1)]:
        i, nh = wave % 6, wave // 6
        assert packed[nt, s, wave, pp, lh, li, jj, t] == Up[6 * i + 2 * pp + jj, 64 * nt + 32 * nh + li, 8 * (s >> 1) + 4 * lh + 2 * (s & 1) + t]


def test_the_transform_matrices_are_a_convolution():
    """A^T [(G g) * (B^T d)] = the 1-D correlation of d with g: checks the three 1-D transforms of the contract against each other in
    exact rational arithmetic (integers scaled by 24)"""
    from fractions import Fraction as F
    BT = [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]]
    G = [[F(1, 4), 0, 0], [F(-1, 6), F(-1, 6), F(-1, 6)], [F(-1, 6), F(1, 6), F(-1, 6)], [F(1, 24), F(1, 12), F(1, 6)],
         [F(1, 24), F(-1, 12), F(1, 6)], [0, 0, 1]]
    AT = [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]
    rng = np.random.default_rng(0)
    d = [F(int(v)) for v in rng.integers(-9, 9, 6)]
    g = [F(int(v)) for v in rng.integers(-9, 9, 3)]
    v = [sum(BT[i][k] * d[k] for k in range(6)) for i in range(6)]
    u = [sum(G[i][k] * g[k] for k in range(3)) for i in range(6)]
    y = [sum(AT[a][i] * u[i] * v[i] for i in range(6)) for a in range(4)]
    assert y == [sum(d[a + k] * g[k] for k in range(3)) for a in range(4)]


def test_winograd4_rule_is_per_sample_and_switchable():
    def flags(n, h, w, cin, cout):
        p = P.Program('r')
        x = p.buffer(n, h, w, cin)
        p.conv(x, np.zeros((cout, cin, 3, 3), np.float32), None, pad=1)
        return p.ops[-1]['flags'], p.ops[-1]['ksplit']
    old = (P.Program.winograd, P.Program.winograd4)
    P.Program.winograd = P.Program.winograd4 = True
    try:
        big = int(np.ceil(np.sqrt(max(P.WINO4_MIN_PIXELS, P.WINO_MIN_PIXELS))))
        assert flags(1, big, big, 64, 64) == (P.CONV_FLAG_WINOGRAD4, 1) and flags(8, big, big, 64, 64) == (P.CONV_FLAG_WINOGRAD4, 1)
        # smaller maps (round 6b): F(4x4) down to 23 x 23 when cin <= 512; below that, or with a long K loop, the direct chain
        assert flags(1, 40, 40, 256, 256) == (P.CONV_FLAG_WINOGRAD4, 1) and flags(8, 40, 40, 256, 256) == (P.CONV_FLAG_WINOGRAD4, 1)
        assert flags(1, 23, 23, 512, 512)[0] == P.CONV_FLAG_WINOGRAD4
        assert flags(1, 40, 40, 1024, 256)[0] == 0 and flags(1, 22, 22, 256, 256)[0] == 0 and flags(1, 12, 12, 512, 512)[0] == 0
        assert flags(1, 45, 45, 64, 64)[0] == 0 and flags(1, 45, 45, 128, 64)[0] == 0            # narrow layers on small maps: direct
        assert flags(1, big, big, 64, 32)[0] == 0 and flags(1, big, big, 48, 64)[0] == 0
        P.Program.winograd4 = False
        assert flags(1, big, big, 64, 64) == (P.CONV_FLAG_WINOGRAD, 1)
        P.Program.winograd, P.Program.winograd4 = False, True
        assert flags(1, big, big, 64, 64)[0] == 0
    finally:
        P.Program.winograd, P.Program.winograd4 = old
