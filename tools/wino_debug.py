#!/usr/bin/env python3
"""development aid: k_conv_wino vs the oracle on single layers, with a structured mismatch report; then timing against the tuned direct
kernels on the benchmark's own layer shapes.  usage: wino_debug.py [check] [bench]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cartoonsegmentation_amd import program as P
from cartoonsegmentation_amd.runtime import CompiledProgram
from oracle import nets as onets


def build(wino, n, h, w, cin, cout, act='relu', res_mode=0, wkind='rand', seed=0):
    P.Program.winograd = wino
    old4, P.Program.winograd4 = P.Program.winograd4, False
    old = P.WINO_MIN_PIXELS; P.WINO_MIN_PIXELS = 0
    rng = np.random.default_rng(seed)
    p = P.Program('t')
    x_ext = p.ext_nchw(n, cin, h, w)
    y_ext = p.ext_nchw(n, cout, h, w)
    x = p.to_nhwc(x_ext)
    if wkind == 'delta':
        wt = np.zeros((cout, cin, 3, 3), np.float32)
        for c in range(min(cin, cout)):
            wt[c, c, 1, 1] = 1.0
    else:
        wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32) if wkind != 'delta' else np.zeros(cout, np.float32)
    res = None
    if res_mode:
        r_ext = p.ext_nchw(n, cout, h, w)
        res = p.to_nhwc(r_ext)
    y = p.conv(x, wt, b, pad=1, act=act, res=res, res_mode=res_mode)
    p.to_nchw(y, y_ext)
    P.WINO_MIN_PIXELS = old
    P.Program.winograd = P.WINO_ENABLE
    P.Program.winograd4 = old4
    return p


def check(n, h, w, cin, cout, act='relu', res_mode=0, wkind='rand'):
    p = build(True, n, h, w, cin, cout, act, res_mode, wkind)
    assert any(o['flags'] & 4 for o in p.ops)
    rng = np.random.default_rng(5)
    xin = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    ext = [xin]
    if res_mode:
        ext.append(rng.standard_normal((n, cout, h, w)).astype(np.float32))
    # ext order: x, y, (res) -- follow the program's ext slots
    yo = np.zeros((n, cout, h, w), np.float32)
    exts_o = [xin, yo] + ext[1:]
    onets.run_program(p, exts_o)
    os.environ["CSM_AUTOTUNE"] = "0"
    cp = CompiledProgram(p, 'cuda')
    yd = torch.full((n, cout, h, w), float('nan'), device='cuda')
    cp.run(torch.from_numpy(xin).cuda(), yd, *[torch.from_numpy(e).cuda() for e in ext[1:]])
    torch.cuda.synchronize()
    yd = yd.cpu().numpy()
    bad = yd != yo
    bad |= np.isnan(yd)
    tag = "n%d %dx%d %d->%d act=%s res=%d w=%s" % (n, h, w, cin, cout, act, res_mode, wkind)
    if not bad.any():
        print("OK   ", tag)
        return True
    d = np.abs(yd - yo)
    print("FAIL ", tag, "mismatch %.4f%%  max|d| %.3g  nan %d  scale %.3g" % (100 * bad.mean(), np.nanmax(d), np.isnan(yd).sum(), np.abs(yo).max()))
    # structure
    by_y = bad.mean(axis=(0, 1, 3)); by_x = bad.mean(axis=(0, 1, 2)); by_c = bad.mean(axis=(0, 2, 3)); by_n = bad.mean(axis=(1, 2, 3))
    print("   by sample", np.round(by_n, 3))
    print("   by y%8 ", np.round([by_y[i::8].mean() for i in range(min(8, h))], 3))
    print("   by x%32", np.round([by_x[i::32].mean() for i in range(min(32, w))], 2))
    print("   by c%64", np.round([by_c[i::64].mean() for i in range(64)], 1))
    idx = np.argwhere(bad)[:6]
    for i in idx:
        print("   at", tuple(i), "hip", yd[tuple(i)], "oracle", yo[tuple(i)])
    return False


def time_layer(n, h, w, cin, cout, reps=5):
    res = {}
    for wino in (False, True):
        p = build(wino, n, h, w, cin, cout)
        os.environ["CSM_AUTOTUNE"] = "1" if not wino else "0"
        cp = CompiledProgram(p, 'cuda')
        x = torch.randn(n, cin, h, w, device='cuda'); y = torch.empty(n, cout, h, w, device='cuda')
        cp.run(x, y); cp.run(x, y)
        ci = [i for i, o in enumerate(p.ops) if o['kind'] == 1][0]
        ms = min(cp.profile(x, y)[ci] for _ in range(reps))
        res[wino] = ms
    fl = 2.0 * n * h * w * cin * cout * 9
    print("%2dx%3dx%3d %4d->%4d  direct %8.1f us %6.1f TF/s | winograd %8.1f us  %6.1f TF/s direct-equivalent (%5.1f executed)  x%.2f" % (
        n, h, w, cin, cout, res[False] * 1e3, fl / res[False] / 1e9, res[True] * 1e3, fl / res[True] / 1e9, fl / 2.25 / res[True] / 1e9, res[False] / res[True]), flush=True)


if __name__ == '__main__':
    what = sys.argv[1:] or ['check', 'bench']
    if 'check' in what:
        ok = True
        ok &= check(1, 8, 32, 32, 64, act=None, wkind='delta')
        ok &= check(1, 8, 32, 32, 64, act=None)
        ok &= check(1, 16, 64, 64, 64)
        ok &= check(2, 13, 37, 64, 128, act='silu', res_mode=2)
        ok &= check(1, 45, 45, 96, 64, act='relu', res_mode=1)
        ok &= check(3, 23, 70, 256, 256)
        print("ALL OK" if ok else "SOME FAILED")
    if 'bench' in what:
        for shp in [(8, 160, 160, 256, 256), (8, 320, 320, 256, 128), (16, 360, 360, 64, 64), (16, 360, 360, 128, 64), (16, 360, 360, 32, 64),
                    (16, 180, 180, 64, 128), (16, 180, 180, 256, 64), (8, 80, 80, 256, 256), (16, 90, 90, 128, 256), (16, 90, 90, 512, 128),
                    (8, 80, 80, 128, 128), (8, 40, 40, 256, 256), (16, 45, 45, 256, 512), (1, 160, 160, 256, 256), (2, 360, 360, 64, 64), (1, 80, 80, 256, 256)]:
            time_layer(*shp)


def variants(shapes, vs):
    """timing of the development instantiations of k_conv_wino (libcsm355_dev.so, -DCSM_WINO_DEV; CSM_WINO_VARIANT: ablation bits 1 no DMA,
    2 no barrier, 4 no LDS reads, 8 no transform VALU; 100 + option bits)"""
    for shp in shapes:
        n, h, w, cin, cout = shp
        p = build(True, n, h, w, cin, cout)
        os.environ["CSM_AUTOTUNE"] = "0"
        cp = CompiledProgram(p, 'cuda')
        x = torch.randn(n, cin, h, w, device='cuda'); y = torch.empty(n, cout, h, w, device='cuda')
        ci = [i for i, o in enumerate(p.ops) if o['kind'] == 1][0]
        fl = 2.0 * n * h * w * cin * cout * 9
        for v in vs:
            os.environ["CSM_WINO_VARIANT"] = str(v)
            cp.run(x, y); cp.run(x, y)
            ms = min(cp.profile(x, y)[ci] for _ in range(5))
            print("%2dx%3dx%3d %4d->%4d  variant %3d  %8.1f us  executed %6.1f TF/s (%.3f of the fp32 MFMA peak)" % (n, h, w, cin, cout, v, ms * 1e3, fl / 2.25 / ms / 1e9, fl / 2.25 / ms / 1e9 / 157.3), flush=True)
        os.environ["CSM_WINO_VARIANT"] = "0"


if __name__ == '__main__' and 'variants' in sys.argv[1:]:
    variants([(8, 160, 160, 256, 256), (16, 360, 360, 64, 64)], [int(v) for v in os.environ.get('WINO_VARIANTS', '0,200,201,202,204,208,215').split(',')])
