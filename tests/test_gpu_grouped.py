"""GPU parity of the vector-pipe grouped 3x3 kernel (csrc/grouped.hip::k_conv_grouped, csm_op.flags bit 4): the DIRECT fmaf chain of
include/csm355.h from its own weight image -- bit-exact against the oracle AND against the block-diagonal matrix-pipe form of the same
layer (the flag is a speed choice of the lowering, never a change of bits).
Reference layers: depth_modules/leres/leres/Resnext_torch.py:70-117 (conv2 of every ResNeXt bottleneck: 32 groups of 8 / 16 / 32 / 64)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from cartoonsegmentation_amd import program as P  # noqa: E402
from cartoonsegmentation_amd.program import Program  # noqa: E402
from test_gpu_nets import rnd, run_both  # noqa: E402


def _layer(valu, n, h, w, cg, groups, act, res_mode, sliced=False, tag=''):
    old = (Program.grouped_valu, P.GROUPED_VALU_MAX_CG)
    Program.grouped_valu, P.GROUPED_VALU_MAX_CG = valu, 32          # (the default rule stops at 16 channels per group; the kernel covers 32)
    try:
        c = cg * groups
        p = Program("grouped")
        x_ext = p.ext_nchw(n, c, h, w)
        y_ext = p.ext_nchw(n, c, h, w)
        x = p.to_nhwc(x_ext)
        out = None
        if sliced:                        # input and output are channel slices of wider buffers (ld > c)
            wide = p.buffer(n, h, w, c + 64)
            p.copy(x, wide.slice(32, 32 + c))
            x = wide.slice(32, 32 + c)
            out = p.buffer(n, h, w, c + 32).slice(32, 32 + c)
        W = rnd('gw%s%d%d' % (tag, cg, groups), (c, cg, 3, 3), 1.0 / np.sqrt(cg * 9))
        b = rnd('gb%s%d' % (tag, c), (c,), 0.1)
        slope = rnd('gs%s%d' % (tag, c), (c,), 0.3) if act == 'prelu' else None
        res = None
        ext_in = [rnd('gx%s%d%d%d' % (tag, n, h, w), (n, c, h, w))]
        if res_mode:
            r_ext = p.ext_nchw(n, c, h, w)
            res = p.to_nhwc(r_ext)
            ext_in.append(rnd('gr%s%d%d' % (tag, h, w), (n, c, h, w)))
            y_ext.buf.ext, r_ext.buf.ext = 2, 1
        y = p.conv(x, W, b, pad=1, groups=groups, act=act, slope=slope, res=res, res_mode=res_mode, out=out)
        p.to_nchw(y, y_ext)
    finally:
        Program.grouped_valu, P.GROUPED_VALU_MAX_CG = old
    flagged = [o for o in p.ops if o['kind'] == 1 and o['flags'] & P.CONV_FLAG_GROUPED]
    assert len(flagged) == (1 if valu else 0)
    return p, ext_in, (n, c, h, w)


CASES = [
    # n, h, w, channels per group, groups, act, res_mode, sliced
    (1, 16, 16, 8, 32, 'relu', 0, False),
    (2, 37, 45, 8, 4, 'relu', 0, False),            # ragged in both directions, one slab
    (1, 8, 40, 8, 8, None, 0, False),               # exactly one 8 x 40 tile
    (1, 1, 1, 16, 2, 'relu', 0, False),             # one pixel: every tap but the centre is padding
    (3, 9, 33, 16, 4, 'silu', 2, False),            # general epilogue: residual after the activation
    (1, 23, 83, 16, 6, 'prelu', 1, False),          # residual before a PReLU with per-channel slopes; 96 channels = 3 slabs
    (2, 20, 20, 32, 3, 'relu', 0, False),
    (1, 41, 7, 32, 2, 'hsigmoid', 0, True),         # channel slices of wider buffers on both sides
    (1, 64, 96, 8, 12, 'relu', 0, True),            # tile width 32 (96 = 3 x 32 pads less than 3 x 40)
    (2, 40, 40, 32, 32, 'relu', 0, False),          # LeReS layer3's conv2 (batch 2)
]


@pytest.mark.parametrize("case", CASES, ids=[str(i) for i in range(len(CASES))])
def test_grouped_vector_kernel_is_bit_exact_and_equals_the_matrix_form(case):
    n, h, w, cg, groups, act, res_mode, sliced = case
    outs = []
    for valu in (True, False):
        p, ext_in, shp = _layer(valu, n, h, w, cg, groups, act, res_mode, sliced, tag=str(case))
        (yo,), (yd,) = run_both(p, ext_in, [shp])
        assert np.isfinite(yd).all()
        assert np.array_equal(yo, yd), "valu=%s: max abs diff %g" % (valu, np.abs(yo - yd).max())
        outs.append(yd)
    assert np.array_equal(outs[0], outs[1])


def test_grouped_vector_kernel_both_tile_widths(monkeypatch):
    """the 32- and the 40-pixel tile of k_conv_grouped (CSM_GROUPED_PX is read once per process: run each in a child)"""
    import os, subprocess, sys
    code = ("import numpy as np, sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_grouped as T\n"
            "for cg, g in ((8, 8), (16, 4), (32, 2)):\n"
            "    p, ext_in, shp = T._layer(True, 2, 19, 75, cg, g, 'relu', 0, tag='px')\n"
            "    (yo,), (yd,) = T.run_both(p, ext_in, [shp])\n"
            "    assert np.array_equal(yo, yd), (cg, float(np.abs(yo - yd).max()))\n"
            "print('OK')\n") % (os.path.dirname(__file__), os.path.dirname(os.path.dirname(__file__)))
    for px in ('4', '5'):
        env = dict(os.environ, CSM_GROUPED_PX=px)
        r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and 'OK' in r.stdout, (px, r.stdout[-2000:], r.stderr[-2000:])


def test_grouped_vector_kernel_batch_invariance_and_repeats():
    """a sample's bits do not depend on the batch it runs in, and repeated launches agree (the scalar-load / LDS waits are hand-placed)"""
    from cartoonsegmentation_amd.runtime import CompiledProgram
    cg, groups, h, w = 8, 32, 50, 70
    c = cg * groups
    x = rnd('gbx', (4, c, h, w))
    ys = {}
    for n in (1, 4):
        p, _, shp = _layer(True, n, h, w, cg, groups, 'relu', 0, tag='inv')
        p.plan()
        cp = CompiledProgram(p, 'cuda')
        xd = torch.from_numpy(x[:n].copy()).cuda()
        first = None
        for _ in range(10):
            yd = torch.full(shp, float('nan'), device='cuda')
            cp.run(xd, yd)
            torch.cuda.synchronize()
            y = yd.cpu().numpy()
            assert first is None or np.array_equal(first, y)
            first = y
        ys[n] = first
    assert np.array_equal(ys[1][0], ys[4][0])
