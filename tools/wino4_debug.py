#!/usr/bin/env python3
"""development aid: k_conv_wino4 (F(4x4, 3x3)) vs its oracle on single layers with a structured mismatch report, then timing against
F(2x2) and the tuned direct kernels on the benchmark's layer shapes.  usage: wino4_debug.py [check] [bench]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from cartoonsegmentation_amd import program as P
from cartoonsegmentation_amd.runtime import CompiledProgram
from oracle import nets as onets
from test_oracle_winograd4 import forced


def build(mode, n, h, w, cin, cout, act='relu', res_mode=0, wkind='rand', seed=0):
    rng = np.random.default_rng(seed)
    with forced(mode):
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
        res = p.to_nhwc(p.ext_nchw(n, cout, h, w)) if res_mode else None
        y = p.conv(x, wt, b, pad=1, act=act, res=res, res_mode=res_mode)
        p.to_nchw(y, y_ext)
    return p


def check(n, h, w, cin, cout, act='relu', res_mode=0, wkind='rand', xkind='rand'):
    p = build('f4', n, h, w, cin, cout, act, res_mode, wkind)
    assert any(o['flags'] & 8 for o in p.ops)
    rng = np.random.default_rng(5)
    xin = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    if xkind == 'ramp':
        xin = (np.arange(h * w, dtype=np.float32).reshape(1, 1, h, w) + 1000.0 * np.arange(cin, dtype=np.float32).reshape(1, cin, 1, 1)) * np.ones((n, 1, 1, 1), np.float32)
    ext = [xin]
    if res_mode:
        ext.append(rng.standard_normal((n, cout, h, w)).astype(np.float32))
    yo = np.zeros((n, cout, h, w), np.float32)
    onets.run_program(p, [xin, yo] + ext[1:])
    os.environ["CSM_AUTOTUNE"] = "0"
    cp = CompiledProgram(p, 'cuda')
    yd = torch.full((n, cout, h, w), float('nan'), device='cuda')
    cp.run(torch.from_numpy(xin).cuda(), yd, *[torch.from_numpy(e).cuda() for e in ext[1:]])
    torch.cuda.synchronize()
    yd = yd.cpu().numpy()
    bad = (yd != yo) | np.isnan(yd)
    tag = "n%d %dx%d %d->%d act=%s res=%d w=%s x=%s" % (n, h, w, cin, cout, act, res_mode, wkind, xkind)
    if not bad.any():
        print("OK   ", tag, flush=True)
        return True
    d = np.abs(yd - yo)
    print("FAIL ", tag, "mismatch %.4f%%  max|d| %.3g  nan %d  scale %.3g" % (100 * bad.mean(), np.nanmax(d), np.isnan(yd).sum(), np.abs(yo).max()))
    by_y = bad.mean(axis=(0, 1, 3)); by_x = bad.mean(axis=(0, 1, 2)); by_c = bad.mean(axis=(0, 2, 3)); by_n = bad.mean(axis=(1, 2, 3))
    print("   by sample", np.round(by_n, 3))
    print("   by y%16", np.round([by_y[i::16].mean() for i in range(min(16, h))], 2))
    print("   by x%32", np.round([by_x[i::32].mean() for i in range(min(32, w))], 2))
    print("   by c%64", np.round([by_c[i::64].mean() for i in range(64)], 1))
    for i in np.argwhere(bad)[:6]:
        print("   at", tuple(i), "hip", yd[tuple(i)], "oracle", yo[tuple(i)])
    sys.stdout.flush()
    return False


def time_layer(n, h, w, cin, cout, reps=5, modes=('direct', 'f2', 'f4')):
    res = {}
    for mode in modes:
        p = build(mode, n, h, w, cin, cout)
        os.environ["CSM_AUTOTUNE"] = "1" if mode == 'direct' else "0"
        cp = CompiledProgram(p, 'cuda')
        x = torch.randn(n, cin, h, w, device='cuda'); y = torch.empty(n, cout, h, w, device='cuda')
        cp.run(x, y); cp.run(x, y)
        ci = [i for i, o in enumerate(p.ops) if o['kind'] == 1][0]
        res[mode] = min(cp.profile(x, y)[ci] for _ in range(reps))
    fl = 2.0 * n * h * w * cin * cout * 9
    ex4 = 2.0 * n * ((h + 3) // 4) * ((w + 3) // 4) * 36 * cin * cout
    print("%2dx%3dx%3d %4d->%4d  " % (n, h, w, cin, cout) + "  ".join("%s %8.1f us" % (m, res[m] * 1e3) for m in modes) +
          "  | f4: %6.1f TF/s direct-equivalent, %5.1f executed; x%.2f vs f2, x%.2f vs direct" % (
              fl / res['f4'] / 1e9, ex4 / res['f4'] / 1e9, res.get('f2', 0) / res['f4'], res.get('direct', 0) / res['f4']), flush=True)


def variants(shapes, vs):
    """timing of the development instantiations of k_conv_wino4 (libcsm355_dev.so = make dev; CSM_WINO4_VARIANT: ablation bits 1 no DMA,
    2 no transform, 4 no MFMA, 8 no epilogue stores)"""
    for shp in shapes:
        n, h, w, cin, cout = shp
        p = build('f4', n, h, w, cin, cout)
        os.environ["CSM_AUTOTUNE"] = "0"
        cp = CompiledProgram(p, 'cuda')
        x = torch.randn(n, cin, h, w, device='cuda'); y = torch.empty(n, cout, h, w, device='cuda')
        ci = [i for i, o in enumerate(p.ops) if o['kind'] == 1][0]
        ex4 = 2.0 * n * ((h + 3) // 4) * ((w + 3) // 4) * 36 * cin * cout
        for v in vs:
            os.environ["CSM_WINO4_VARIANT"] = str(v)
            cp.run(x, y); cp.run(x, y)
            ms = min(cp.profile(x, y)[ci] for _ in range(5))
            print("%2dx%3dx%3d %4d->%4d  variant %3d  %8.1f us  executed %6.1f TF/s (%.3f of the fp32 MFMA peak)" % (n, h, w, cin, cout, v, ms * 1e3, ex4 / ms / 1e9, ex4 / ms / 1e9 / 157.3), flush=True)
        os.environ["CSM_WINO4_VARIANT"] = "0"


if __name__ == '__main__':
    what = sys.argv[1:] or ['check', 'bench']
    if 'variants' in what:
        variants([(8, 160, 160, 256, 256), (16, 360, 360, 64, 64)], [int(v) for v in os.environ.get('WINO4_VARIANTS', '0,1,2,4,8,3,11,7,15').split(',')])
    if 'check' in what:
        ok = True
        ok &= check(1, 16, 32, 32, 64, act=None, wkind='delta', xkind='ramp')
        ok &= check(1, 16, 32, 32, 64, act=None, wkind='delta')
        ok &= check(1, 16, 32, 32, 64, act=None)
        ok &= check(1, 32, 64, 64, 64)
        ok &= check(2, 13, 37, 64, 128, act='silu', res_mode=2)
        ok &= check(1, 45, 45, 96, 64, act='relu', res_mode=1)
        ok &= check(3, 23, 70, 256, 256)
        print("ALL OK" if ok else "SOME FAILED", flush=True)
    if 'forms' in what:           # one process per form: CSM_WINO4_FORM=6|3|2|1 python tools/wino4_debug.py forms
        for shp in [(1, 80, 80, 256, 256), (1, 80, 80, 128, 128), (1, 160, 160, 256, 256), (2, 360, 360, 64, 64), (2, 90, 90, 128, 256), (2, 180, 180, 64, 128),
                    (1, 40, 40, 256, 256), (2, 45, 45, 256, 512), (2, 23, 23, 512, 512), (8, 40, 40, 256, 256), (16, 23, 23, 512, 512), (16, 45, 45, 256, 512),
                    (8, 20, 20, 512, 512), (16, 45, 45, 1024, 256), (8, 80, 80, 256, 256), (8, 160, 160, 256, 256)]:
            time_layer(*shp, modes=('direct', 'f2', 'f4'))
    if 'bench' in what:
        for shp in [(8, 160, 160, 256, 256), (8, 320, 320, 256, 128), (16, 360, 360, 64, 64), (16, 360, 360, 128, 64), (16, 360, 360, 32, 64),
                    (16, 180, 180, 64, 128), (16, 180, 180, 256, 64), (8, 80, 80, 256, 256), (16, 90, 90, 128, 256), (16, 90, 90, 512, 128),
                    (8, 80, 80, 128, 128), (1, 160, 160, 256, 256), (2, 360, 360, 64, 64), (1, 80, 80, 256, 256)]:
            time_layer(*shp)
