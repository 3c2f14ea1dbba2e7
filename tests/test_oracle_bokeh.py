"""CPU: oracle bokeh / colorize vs fixtures produced by the reference's utils/effects.py::bokeh_blur (its CUDA text run
sequentially) and zoedepth's colorize (matplotlib)."""
import os

import numpy as np
import pytest

from oracle import kenburns as okb, segment as oseg

G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "bokeh_240x320.npz")))


def test_single_pass_bit_exact():
    import ctypes
    H, W = G['dn'].shape
    imf = (G['img'].astype(np.float32) / 255)
    out = np.empty_like(imf)
    oseg.lib().orc_bokeh_pass(oseg._p(np.ascontiguousarray(imf)), oseg._p(G['dn']), oseg._p(out), ctypes.c_int(H), ctypes.c_int(W),
                              ctypes.c_int(32), ctypes.c_float(np.cos(-np.pi / 6)), ctypes.c_float(np.sin(-np.pi / 6)))
    assert np.array_equal(out, G['one_pass'])


def test_colorize_gray_r_matches_reference_colorize():
    d = np.abs(okb.colorize_gray_r(G['depth_f']).astype(np.int32) - G['depth_u8'].astype(np.int32))
    assert d.max() <= 1 and (d == 0).mean() > 0.98       # numpy-version dependent percentile rounding (see oracle docstring)


def test_gray_r_lut_matches_matplotlib():
    mpl = pytest.importorskip("matplotlib")
    lut = mpl.colormaps['gray_r'](np.arange(256), bytes=True)[:, 0]
    assert np.array_equal(lut, okb.gray_r_lut())


@pytest.mark.parametrize("tag,fp", [("fp100", 100.0), ("fp17", 17.25)])
def test_bokeh_blur_vs_reference(tag, fp):
    out = okb.bokeh_blur(G['img'], G['depth_u8'], 32, 13, fp)
    diff = np.abs(out.astype(np.int32) - G['blur_' + tag].astype(np.int32))
    assert diff.max() <= 1 and (diff == 0).mean() > 0.999      # float32 pow implementations may differ in the last ulp
