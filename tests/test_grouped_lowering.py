"""CPU checks of the vector-pipe grouped-convolution lowering (program.py::pack_grouped_weights / grouped_valu_eligible; the kernel is
csrc/grouped.hip, its GPU parity tests/test_gpu_grouped.py): the weight image is the documented permutation of the natural weights, the
rule is per layer shape (never the batch), and the oracle -- which reads the NATURAL weights and groups -- is indifferent to the flag."""
import numpy as np

from cartoonsegmentation_amd import program as P
from oracle import nets as onets


def test_grouped_weight_image_is_the_documented_permutation():
    rng = np.random.default_rng(0)
    for cg, g in ((8, 4), (16, 2), (32, 3)):
        w = rng.standard_normal((cg * g, cg, 3, 3)).astype(np.float32)
        pk = P.pack_grouped_weights(w, g)
        assert pk.dtype == np.float32 and pk.size == w.size and sorted(pk.tolist()) == sorted(w.reshape(-1).tolist())
        pk = pk.reshape(g, cg // 8, 9, cg // 8, 2, 8, 4)
        for _ in range(300):
            gi, o, tap, kb, h, i, t = [int(rng.integers(0, s)) for s in pk.shape]
            assert pk[gi, o, tap, kb, h, i, t] == w[gi * cg + 8 * o + 4 * h + t, 8 * kb + 4 * (i & 1) + (i >> 1), tap // 3, tap % 3]
        # chain order inside an 8-block: 0,4,1,5,2,6,3,7 (include/csm355.h, the direct contract)
        assert [4 * (i & 1) + (i >> 1) for i in range(8)] == [0, 4, 1, 5, 2, 6, 3, 7]


def _flags(n, h, w, cg, groups, k=3, stride=1, pad=1, dil=1, cout=None):
    p = P.Program('r')
    x = p.buffer(n, h, w, cg * groups)
    cout = cg * groups if cout is None else cout
    p.conv(x, np.zeros((cout, cg, k, k), np.float32), None, stride=stride, pad=pad, dil=dil, groups=groups)
    o = p.ops[-1]
    return o['flags'], o['groups'], o['cin_g'], o['cout_g']


def test_grouped_rule_is_per_layer_shape_and_switchable():
    assert P.GROUPED_VALU and P.Program.grouped_valu and P.GROUPED_VALU_MAX_CG == 16
    for n in (1, 8):                                                   # never the batch
        assert _flags(n, 160, 160, 8, 32) == (P.CONV_FLAG_GROUPED, 32, 8, 8)      # REAL groups travel with the flag
        assert _flags(n, 80, 80, 16, 32) == (P.CONV_FLAG_GROUPED, 32, 16, 16)
        assert _flags(n, 40, 40, 32, 32) == (0, 32, 32, 32)            # 32 per group: matrix pipe (one group = one 32-wide super-group)
    assert _flags(1, 20, 20, 64, 32)[0] == 0                           # 64 per group
    assert _flags(1, 80, 80, 16, 32, stride=2)[0] == 0 and _flags(1, 80, 80, 16, 32, pad=2, dil=2)[0] == 0
    assert _flags(1, 80, 80, 8, 4, k=1, pad=0)[0] == 0
    assert _flags(1, 80, 80, 8, 3)[0] == 0                             # 24 channels: not a multiple of the 32-channel slab
    assert _flags(1, 80, 80, 8, 4, cout=64)[0] == 0                    # cout_g != cin_g
    assert _flags(1, 80, 80, 8, 4) == (P.CONV_FLAG_GROUPED, 4, 8, 8)
    assert _flags(1, 1500, 1500, 8, 32)[0] == 0                        # one sample's view >= 2 GiB: beyond the kernel's 32-bit byte offsets
    assert _flags(4, 1400, 1400, 8, 32)[0] == P.CONV_FLAG_GROUPED      # (the bound is per sample, not per batch)
    old = P.Program.grouped_valu
    P.Program.grouped_valu = False
    try:
        assert _flags(1, 160, 160, 8, 32) == (0, 8, 32, 32)            # the block-diagonal super-group form (4 groups of 8 per 32)
    finally:
        P.Program.grouped_valu = old


def test_oracle_is_indifferent_to_the_grouped_flag():
    rng = np.random.default_rng(1)
    n, h, w, cg, g = 2, 9, 11, 8, 4
    c = cg * g
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = (rng.standard_normal((c, cg, 3, 3)) / 8).astype(np.float32)
    b = rng.standard_normal(c).astype(np.float32)
    outs = []
    for valu in (True, False):
        old = P.Program.grouped_valu
        P.Program.grouped_valu = valu
        try:
            p = P.Program('o')
            x_ext = p.ext_nchw(n, c, h, w); y_ext = p.ext_nchw(n, c, h, w)
            y = p.conv(p.to_nhwc(x_ext), wt, b, pad=1, groups=g, act='relu')
            p.to_nchw(y, y_ext)
        finally:
            P.Program.grouped_valu = old
        assert bool(p.ops[1]['flags'] & P.CONV_FLAG_GROUPED) == valu
        yo = np.zeros((n, c, h, w), np.float32)
        onets.run_program(p, [x, yo])
        outs.append(yo)
    assert np.array_equal(outs[0], outs[1])
    # and both are the plain grouped convolution (float64 reference, fp32 rounding tolerance)
    ref = np.zeros((n, c, h, w))
    xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (1, 1), (1, 1)))
    for gi in range(g):
        for ky in range(3):
            for kx in range(3):
                ref[:, gi * cg:(gi + 1) * cg] += np.einsum('oc,nchw->nohw', wt[gi * cg:(gi + 1) * cg, :, ky, kx].astype(np.float64),
                                                            xp[:, gi * cg:(gi + 1) * cg, ky:ky + h, kx:kx + w])
    ref = np.maximum(ref + b.reshape(1, -1, 1, 1), 0.0)
    assert np.abs(outs[0] - ref).max() <= 2e-5
