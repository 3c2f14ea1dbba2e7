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


class _arith:
    """lower with the shipped per-sample rule ('rule': at fixture sizes every layer stays on the direct chain), or with a Winograd
    arithmetic forced on EVERY eligible 3x3 layer ('winograd' = F(2x2), 'winograd4' = F(4x4)) -- the reference-module fixtures must hold
    under all three"""
    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        from cartoonsegmentation_amd import program as P
        self.P, self.old = P, (P.Program.winograd, P.WINO_MIN_PIXELS, P.Program.winograd4, P.WINO4_MIN_PIXELS)
        if self.mode == 'winograd':
            P.Program.winograd, P.WINO_MIN_PIXELS, P.Program.winograd4 = True, 0, False
        elif self.mode == 'winograd4':
            P.Program.winograd, P.WINO_MIN_PIXELS, P.Program.winograd4, P.WINO4_MIN_PIXELS = True, 0, True, 0

    def __exit__(self, *a):
        self.P.Program.winograd, self.P.WINO_MIN_PIXELS, self.P.Program.winograd4, self.P.WINO4_MIN_PIXELS = self.old


def _n_wino(prog, arith='winograd'):
    bit = 8 if arith == 'winograd4' else 4
    return sum(1 for o in prog.ops if o['kind'] == 1 and o['flags'] & bit)


@pytest.mark.parametrize("arith", ["rule", "winograd", "winograd4"])
@pytest.mark.parametrize("tag", ["64x64", "90x74"])
def test_isnet_vs_reference_module(tag, arith):
    from cartoonsegmentation_amd.nets import build_isnet
    g = dict(np.load(os.path.join(GOLDEN, "net_isnet_%s.npz" % tag)))
    n, c, h, w = g['x'].shape
    with _arith(arith):
        prog = build_isnet(SynthWeights('isnet.'), n, h, w)
    assert (_n_wino(prog, arith) >= 40) == (arith != 'rule')
    y = np.zeros((n, 1, h, w), np.float32)
    onets.run_program(prog, [np.ascontiguousarray(g['x']), y])
    # BN folding + summation order differ from torch's kernels: fp32 roundoff level
    assert rel_err(y, g['d1']) < 2e-4, rel_err(y, g['d1'])
    thr = np.log(0.3 / 0.7)                       # sigmoid(x) > 0.3  (mask_thr, animeinsseg/__init__.py:662)
    flips = int(((y > thr) != (g['d1'] > thr)).sum())
    print("isnet %s %s: rel err %.3g, mask flips %d of %d" % (tag, arith, rel_err(y, g['d1']), flips, y.size))
    assert flips / y.size < 1e-3


@pytest.mark.parametrize("arith", ["rule", "winograd", "winograd4"])
@pytest.mark.parametrize("tag", ["64x64", "96x64"])
def test_leres_vs_reference_module(tag, arith):
    from cartoonsegmentation_amd.nets import build_leres
    g = dict(np.load(os.path.join(GOLDEN, "net_leres_%s.npz" % tag)))
    n, c, h, w = g['x'].shape
    with _arith(arith):
        prog = build_leres(SynthWeights('leres.'), n, h, w)
    assert (_n_wino(prog, arith) >= 15) == (arith != 'rule')
    y = np.zeros((n, 1, h, w), np.float32)
    onets.run_program(prog, [np.ascontiguousarray(g['x']), y])
    print("leres %s %s: rel err %.3g" % (tag, arith, rel_err(y, g['y'])))
    assert rel_err(y, g['y']) < 1e-4, rel_err(y, g['y'])


def test_refine_vs_reference_module():
    import torch
    from cartoonsegmentation_amd.nets import build_refine
    g = dict(np.load(os.path.join(GOLDEN, "net_refine_48x64.npz")))
    ti, td = torch.from_numpy(g['img']), torch.from_numpy(g['dsp'])
    mi, md = ti.mean([1, 2, 3], True), td.mean([1, 2, 3], True)
    si, sd = ti.std([1, 2, 3], False, True), td.std([1, 2, 3], False, True)
    xi, xd = ((ti - mi) / (si + 1e-7)).numpy(), ((td - md) / (sd + 1e-7)).numpy()
    H, W, h, w = 48, 64, 12, 16
    prog = build_refine(SynthWeights('refine.'), H, W, h, w)
    out = np.zeros((1, 1, H, W), np.float32)
    onets.run_program(prog, [xi, xd, out])
    r = out * (sd.numpy() + 1e-7) + md.numpy()
    r = np.where(r > 0, r, 0).astype(np.float32)
    assert rel_err(r, g['y']) < 1e-4


def test_inpaint_forward_vs_reference():
    """whole Inpaint.forward: context conv -> C=68 splat -> median-5 -> GridNet (reference fixture runs the
    reference's own CUDA text sequentially)"""
    from cartoonsegmentation_amd.nets import build_inpaint_context, build_inpaint_grid
    from oracle import kenburns as okb
    g = dict(np.load(os.path.join(GOLDEN, "net_inpaint_32x40.npz")))
    H, W = 32, 40
    ws = SynthWeights('inpaint.')
    ctx, grid = build_inpaint_context(ws, H, W), build_inpaint_grid(ws, H, W)
    # degrid_mode 0 = the in-place pass the fixture was produced with (sequential execution of the reference text)
    o = okb.inpaint_forward(g['img'], g['disp'], g['shift'], g['seg'], W, H, W / 2.0, 40.0, ctx, grid, degrid_mode=0)
    assert np.array_equal(o['existing'], g['existing'])
    assert np.array_equal(o['segmasks'], g['segmasks'])
    assert np.abs(o['image'] - g['image']).max() < 1e-4
    assert np.abs(o['disparity'] - g['disparity']).max() / g['disparity'].max() < 1e-4
    # Jacobi degrid (the HIP build's deterministic semantics): same coverage except a handful of pixels
    o1 = okb.inpaint_forward(g['img'], g['disp'], g['shift'], g['seg'], W, H, W / 2.0, 40.0, ctx, grid, degrid_mode=1)
    assert (o1['existing'] == g['existing']).mean() > 0.99


@pytest.mark.parametrize("tag,h,w", [("96x64", 96, 64), ("64x128", 64, 128), ("72x88", 72, 88)])
def test_disparity_estimator_vs_reference_modules(tag, h, w):
    """`depth_est: default`: Semantics (VGG19-BN slices, ceil-mode pools, flip + normalise) and the 6 x 4 Disparity GridNet
    (disparity_estimation.py:80-193) lowered to layer programs vs the reference's own modules; 96x64 takes the odd-height crop, 72x88 odd heights and widths (the [0,-1] crops of :172-173)"""
    from cartoonsegmentation_amd.nets import build_disparity, build_semantics
    g = dict(np.load(os.path.join(GOLDEN, "net_disparity_%s.npz" % tag)))
    sem = np.zeros_like(g['sem'])
    onets.run_program(build_semantics(SynthWeights('semantics.'), h, w), [g['x'], sem])
    assert rel_err(sem, g['sem']) < 1e-5
    d = np.zeros_like(g['disp'])
    onets.run_program(build_disparity(SynthWeights('disparity.'), h, w), [g['x'], g['sem'], d])
    assert rel_err(d, g['disp']) < 1e-5 and d.min() >= 0


def test_fast_conv_is_bit_identical_to_the_reference_loop(monkeypatch):
    """oracle/nets_oracle.c::orc_conv_fast (chain-ordered weights, four chains in flight: what bench.py's cpu_baseline times) vs the
    plain loop nest orc_conv: every output bit, on groups / stride / dilation / ragged channel counts / split-K / 7x7 / residuals"""
    from cartoonsegmentation_amd.program import Program
    rng = np.random.default_rng(9)
    cases = [dict(cin=40, cout=37, k=3, stride=1, pad=1, dil=1, groups=1, hw=(19, 23)),
             dict(cin=64, cout=64, k=3, stride=2, pad=1, dil=1, groups=8, hw=(20, 20)),
             dict(cin=96, cout=130, k=1, stride=1, pad=0, dil=1, groups=1, hw=(9, 11)),
             dict(cin=32, cout=16, k=3, stride=1, pad=4, dil=4, groups=1, hw=(17, 17)),
             dict(cin=4, cout=20, k=7, stride=2, pad=3, dil=1, groups=1, hw=(30, 26)),
             dict(cin=512, cout=24, k=3, stride=1, pad=1, dil=1, groups=1, hw=(6, 6))]          # small map, long K -> split-K
    for c in cases:
        p = Program("t")
        H, W = c['hw']
        x_ext = p.ext_nchw(1, c['cin'], H, W)
        x = p.to_nhwc(x_ext)
        w = rng.normal(0, 0.2, (c['cout'], c['cin'] // c['groups'], c['k'], c['k'])).astype(np.float32)
        b = rng.normal(0, 0.1, c['cout']).astype(np.float32)
        y = p.conv(x, w, b, stride=c['stride'], pad=c['pad'], dil=c['dil'], groups=c['groups'], act='silu')
        y2 = p.conv(y, rng.normal(0, 0.2, (c['cout'], c['cout'], 1, 1)).astype(np.float32), None, act='relu', res=y, res_mode=2)
        out_ext = p.ext_nchw(1, c['cout'], y2.h, y2.w)
        p.to_nchw(y2, out_ext)
        p.plan()
        xin = rng.normal(0, 1, (1, c['cin'], H, W)).astype(np.float32)
        o_fast, o_ref = np.zeros((1, c['cout'], y2.h, y2.w), np.float32), np.zeros((1, c['cout'], y2.h, y2.w), np.float32)
        monkeypatch.delenv("ORC_CONV_REFERENCE", raising=False)
        onets.run_program(p, [xin, o_fast])
        monkeypatch.setenv("ORC_CONV_REFERENCE", "1")
        onets.run_program(p, [xin, o_ref])
        assert np.array_equal(o_fast, o_ref), c
        assert np.isfinite(o_ref).all() and np.abs(o_ref).max() > 0


def _zoe_inputs(g):
    f = lambda k: np.ascontiguousarray(g[k].astype(np.float32))
    return [np.ascontiguousarray(f('rel')[:, None]), f('out_conv'), f('btlnck'), f('r4'), f('r3'), f('r2'), f('r1')]


def test_zoedepth_head_vs_reference_text():
    """everything of ZoeDepth.forward after the MiDaS core (zoedepth_v1.py:124-202, shipped ZoeD_M12_N configuration) vs a fixture
    produced by executing the reference's class + layers on seeded core features; the fixture pinned a reference quirk: the
    attractors run with the jit default alpha = 300, not the configured 1000"""
    from cartoonsegmentation_amd.nets import build_zoe_head
    g = dict(np.load(os.path.join(GOLDEN, "net_zoehead_64x96.npz")))
    H, W = 64, 96
    prog = build_zoe_head(SynthWeights('zoe.'), 1, H, W, [(H >> s, W >> s) for s in (5, 4, 3, 2, 1)])
    out = np.zeros((1, 1, H, W), np.float32)
    onets.run_program(prog, _zoe_inputs(g) + [out])
    assert rel_err(out, g['metric_depth']) < 1e-4, rel_err(out, g['metric_depth'])
    assert out.min() > 0


def test_torch_cpu_interpreter_agrees_with_the_c_oracle():
    """oracle/nets_torch.py (the torch-CPU / oneDNN execution that bench.py times as `cpu_baseline`) runs the same lowered
    programs as the fmaf-chain C interpreter: outputs agree at fp32 round-off, so the timed baseline computes the real thing"""
    from cartoonsegmentation_amd.nets import build_isnet, build_leres, build_rtmdet
    from oracle import nets_torch
    rng = np.random.default_rng(3)
    p = build_isnet(SynthWeights('isnet.'), 2, 64, 64)
    x = rng.normal(0, 1, (2, 4, 64, 64)).astype(np.float32)
    ya, yb = np.zeros((2, 1, 64, 64), np.float32), np.zeros((2, 1, 64, 64), np.float32)
    onets.run_program(p, [x, ya]); nets_torch.run_program(p, [x, yb])
    assert rel_err(yb, ya) < 2e-4
    p = build_leres(SynthWeights('leres.'), 1, 64, 96)
    x = rng.normal(0, 1, (1, 3, 64, 96)).astype(np.float32)
    ya, yb = np.zeros((1, 1, 64, 96), np.float32), np.zeros((1, 1, 64, 96), np.float32)
    onets.run_program(p, [x, ya]); nets_torch.run_program(p, [x, yb])
    assert rel_err(yb, ya) < 1e-4
    rp, cfg = build_rtmdet(SynthWeights('rtmdet.'), 1, 64, 64)
    x = rng.normal(0, 1, (1, 3, 64, 64)).astype(np.float32)
    want = rp.cls + rp.reg + rp.kern + [rp.mask_feat]
    va, vb = onets.run_program(rp.prog, [x], want_views=want), nets_torch.run_program(rp.prog, [x], want_views=want)
    for t in want:
        assert rel_err(vb[t], va[t]) < 1e-4


def test_gridnet_with_fused_upsample_prelu_in_both_interpreters():
    """the Inpaint GridNet (Upsample + PReLU lowered to ONE bilinear op with an activation, residual epilogues, stand-alone PReLUs)
    through the C oracle and through the torch-CPU interpreter -- two independent executions of the same program"""
    from cartoonsegmentation_amd.nets.inpaint import build_inpaint_grid
    from oracle import nets_torch
    rng = np.random.default_rng(8)
    p = build_inpaint_grid(SynthWeights('inpaint.'), 32, 48)
    assert any(o['kind'] == 4 and o['act'] for o in p.ops), "the Upsample blocks are expected to carry their PReLU in the bilinear op"
    x = rng.normal(0, 1, (1, 69, 32, 48)).astype(np.float32)
    ia, da = np.zeros((1, 3, 32, 48), np.float32), np.zeros((1, 1, 32, 48), np.float32)
    ib, db = np.zeros_like(ia), np.zeros_like(da)
    onets.run_program(p, [x, ia, da]); nets_torch.run_program(p, [x, ib, db])
    assert np.abs(ia).max() > 0 and rel_err(ib, ia) < 1e-4 and rel_err(db, da) < 1e-4


def test_zoe_infer_chain_on_the_oracle_matches_the_reference_fixture():
    """CPU statement of `depth_est: 'zoe'` around the stand-in core (tests/golden/zoe_stub_core.py): reflect padding + flip +
    PrepForMidas in torch, the metric-bins head through the oracle interpreter, bicubic resize back + crop + flip average -- against
    the fixture made by the reference's own DepthModel / ZoeDepth / MidasCore classes.  Pins the host-side rules the HIP path mirrors
    (pad sizes, Resize.get_size, feature order, TTA) without a GPU."""
    import sys
    import torch
    import torch.nn.functional as F
    sys.path.insert(0, GOLDEN)
    import zoe_stub_core as stub
    from cartoonsegmentation_amd.nets import build_zoe_head
    from cartoonsegmentation_amd.zoedepth import midas_size
    g = np.load(os.path.join(GOLDEN, "zoe_infer_70x110.npz"))
    x = torch.from_numpy(g['img'])
    H, W = x.shape[2:]
    ph, pw = int(np.sqrt(H / 2) * 3), int(np.sqrt(W / 2) * 3)
    outs = []
    for flip in (0, 1):
        xi = torch.flip(x, dims=[3]) if flip else x
        xpd = F.pad(xi, [pw, pw, ph, ph], mode='reflect')
        nw, nh = midas_size(xpd.shape[3], xpd.shape[2], int(g['net'][1]), int(g['net'][0]))
        xp = (F.interpolate(xpd, (nh, nw), mode='bilinear', align_corners=True) - 0.5) / 0.5
        assert float((xp - torch.from_numpy(g['prep%d' % flip])).abs().max()) <= 1e-6
        rel, feats = stub.core(xp)
        oc, btl, blocks = feats[0], feats[1], feats[2:]
        prog = build_zoe_head(SynthWeights('zoe.'), 1, nh, nw, [tuple(btl.shape[2:])] + [tuple(b.shape[2:]) for b in blocks])
        out = np.zeros((1, 1, nh, nw), np.float32)
        ext = [rel.reshape(1, 1, nh, nw).numpy(), oc.numpy(), btl.numpy()] + [b.numpy() for b in blocks]
        onets.run_program(prog, [np.ascontiguousarray(a) for a in ext] + [out])
        d = F.interpolate(torch.from_numpy(out), size=tuple(xpd.shape[2:]), mode='bicubic', align_corners=False)[:, :, ph:-ph, pw:-pw]
        outs.append(torch.flip(d, dims=[3]) if flip else d)
    depth = ((outs[0] + outs[1]) / 2).numpy()
    assert rel_err(depth, g['depth']) < 1e-5
