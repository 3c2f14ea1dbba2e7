"""GPU parity of the layer-program executor (csm_run_program, HIP/MFMA kernels) vs the CPU oracle
interpreter: BIT-EXACT -- both follow the fmaf-chain contract of include/csm355.h."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from cartoonsegmentation_amd.program import Program  # noqa: E402
from cartoonsegmentation_amd.weights import SynthWeights, hash_uniform  # noqa: E402
from oracle import nets as onets  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def run_both(prog, ext_in, out_shapes):
    from cartoonsegmentation_amd.runtime import CompiledProgram
    assert torch.cuda.is_available()
    prog.plan()
    outs_o = [np.zeros(s, np.float32) for s in out_shapes]
    onets.run_program(prog, [np.ascontiguousarray(a) for a in ext_in] + outs_o)
    cp = CompiledProgram(prog, 'cuda')
    outs_d = [torch.full(s, float('nan'), device='cuda') for s in out_shapes]
    cp.run(*[torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in ext_in], *outs_d)
    torch.cuda.synchronize()
    return outs_o, [t.cpu().numpy() for t in outs_d]


def rnd(name, shape, scale=1.0):
    return (hash_uniform(name, int(np.prod(shape))) * scale).astype(np.float32).reshape(shape)


CONV_CASES = [
    # n, cin, h, w, cout, k, stride, pad, dil, groups, act, res_mode
    (1, 64, 37, 29, 64, 3, 1, 1, 1, 1, 'relu', 0),
    (2, 32, 19, 23, 96, 3, 1, 1, 1, 1, 'silu', 2),
    (1, 256, 20, 20, 512, 1, 1, 0, 1, 1, 'relu', 1),
    (1, 3, 64, 48, 64, 7, 2, 3, 1, 1, 'relu', 0),
    (1, 4, 33, 31, 64, 3, 2, 1, 1, 1, None, 0),
    (1, 128, 23, 23, 128, 3, 1, 2, 2, 1, 'relu', 0),
    (1, 64, 23, 23, 64, 3, 1, 8, 8, 1, 'relu', 0),
    (1, 256, 16, 16, 256, 3, 1, 1, 1, 32, 'relu', 0),
    (1, 512, 16, 16, 512, 3, 2, 1, 1, 32, 'relu', 0),
    (1, 1024, 8, 8, 1024, 3, 1, 1, 1, 32, 'relu', 0),
    (1, 2048, 6, 6, 2048, 3, 1, 1, 1, 32, 'relu', 0),
    (1, 64, 40, 40, 1, 3, 1, 1, 1, 1, None, 0),
    (1, 256, 20, 20, 169, 1, 1, 0, 1, 1, None, 0),
    (1, 256, 10, 10, 8, 1, 1, 0, 1, 1, 'prelu', 0),
    (2, 512, 1, 1, 512, 1, 1, 0, 1, 1, 'hsigmoid', 0),
    (1, 40, 17, 17, 48, 3, 1, 1, 1, 1, 'sigmoid', 0),
    (1, 16, 90, 90, 16, 3, 1, 1, 1, 1, 'relu', 0),
    (1, 768, 12, 12, 256, 1, 1, 0, 1, 1, None, 0),
    (2, 64, 70, 66, 1, 3, 1, 1, 1, 1, None, 0),            # narrow-output kernel (cout <= 4)
    (1, 128, 33, 47, 1, 3, 1, 1, 1, 1, 'sigmoid', 0),
    (1, 64, 20, 20, 3, 3, 1, 2, 2, 1, 'relu', 0),
    (1, 36, 21, 21, 2, 3, 1, 1, 1, 1, None, 0),
    (1, 96, 30, 30, 4, 1, 1, 0, 1, 1, 'relu', 1),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[str(i) for i in range(len(CONV_CASES))])
def test_conv_bit_exact(case):
    n, cin, h, w, cout, k, stride, pad, dil, groups, act, res_mode = case
    p = Program("conv")
    x_ext = p.ext_nchw(n, cin, h, w)
    ho = (h + 2 * pad - dil * (k - 1) - 1) // stride + 1
    wo = (w + 2 * pad - dil * (k - 1) - 1) // stride + 1
    y_ext = p.ext_nchw(n, cout, ho, wo)
    x = p.to_nhwc(x_ext)
    W = rnd('w%s' % (case,), (cout, cin // groups, k, k), 1.0 / np.sqrt(cin // groups * k * k))
    if groups == 1 and x.c != cin:
        pass    # program pads weights with zero channels
    b = rnd('b%s' % (case,), (cout,), 0.1)
    slope = rnd('s%s' % (case,), (cout,), 0.3) if act == 'prelu' else None
    res = None
    ext_in = [rnd('x%s' % (case,), (n, cin, h, w))]
    if res_mode:
        r_ext = p.ext_nchw(n, cout, ho, wo)
        res = p.to_nhwc(r_ext)
        ext_in.append(rnd('r%s' % (case,), (n, cout, ho, wo)))
        # keep ext order: x, r, y
        y_ext.buf.ext, r_ext.buf.ext = 2, 1
    y = p.conv(x, W, b, stride=stride, pad=pad, dil=dil, groups=groups, act=act, slope=slope, res=res, res_mode=res_mode)
    p.to_nchw(y, y_ext)
    (yo,), (yd,) = run_both(p, ext_in, [(n, cout, ho, wo)])
    assert np.isfinite(yd).all()
    assert np.array_equal(yo, yd), "max abs diff %g" % np.abs(yo - yd).max()


def test_concat_slices_and_misc_ops():
    """virtual concat (channel slices), dwconv, pools, resizes, gavgpool, scale, add -- bit exact"""
    p = Program("misc")
    n, h, w = 2, 21, 18
    x_ext = p.ext_nchw(n, 32, h, w)
    x = p.to_nhwc(x_ext)
    cat = p.buffer(n, h, w, 96)
    a = p.conv(x, rnd('wa', (32, 32, 3, 3), 0.06), rnd('ba', (32,), 0.1), pad=1, act='silu', out=cat.slice(32, 64))
    p.dwconv(a, rnd('wd', (32, 1, 5, 5), 0.2), rnd('bd', (32,), 0.1), pad=2, act='silu', out=cat.slice(0, 32))
    mp = p.maxpool(a, 5, 1, 2)
    p.copy(mp, cat.slice(64, 96))
    y1 = p.conv(cat, rnd('wc', (64, 96, 1, 1), 0.1), rnd('bc', (64,), 0.1), act='relu')
    s = p.conv(p.gavgpool(y1), rnd('wf', (64, 64, 1, 1), 0.2), rnd('bf', (64,), 0.1), act='hsigmoid')
    y2 = p.scale(y1, s)
    d = p.maxpool(y2, 2, 2, ceil_mode=True)
    d3 = p.maxpool(y2, 3, 2, 1)
    u = p.bilinear(d, (h, w), align_corners=False)
    u2 = p.bilinear(d3, (d3.h * 2, d3.w * 2), align_corners=True)
    u3 = p.nearest(d, 2)
    z = p.add(u, y2, act='relu')
    outs = [z, u2, u3]
    exts = [p.ext_nchw(t.n, t.c, t.h, t.w) for t in outs]
    for t, e in zip(outs, exts):
        p.to_nchw(t, e)
    ro, rd = run_both(p, [rnd('xin', (n, 32, h, w))], [(t.n, t.c, t.h, t.w) for t in outs])
    for o, d_ in zip(ro, rd):
        assert np.isfinite(d_).all() and np.array_equal(o, d_), np.abs(o - d_).max()


@pytest.mark.parametrize("tag", ["64x64", "90x74"])
def test_isnet_hip_vs_oracle_and_reference(tag):
    from cartoonsegmentation_amd.nets import build_isnet
    g = dict(np.load(os.path.join(GOLDEN, "net_isnet_%s.npz" % tag)))
    n, c, h, w = g['x'].shape
    prog = build_isnet(SynthWeights('isnet.'), n, h, w)
    (yo,), (yd,) = run_both(prog, [g['x']], [(n, 1, h, w)])
    assert np.array_equal(yo, yd), "max abs diff %g" % np.abs(yo - yd).max()
    assert np.abs(yd - g['d1']).max() / np.abs(g['d1']).max() < 2e-4
    thr = np.log(0.3 / 0.7)
    assert np.array_equal(yo > thr, yd > thr)                      # bit-exact masks after threshold vs the oracle
    assert ((yd > thr) != (g['d1'] > thr)).mean() < 1e-3           # IoU-level agreement with torch's own kernels


FULL_SIZE_LAYERS = [
    # n, h, w, cin, cout, k, stride, dil, groups -- BASELINE-size layers of the three nets (too large for the CPU oracle in a test)
    (1, 160, 160, 256, 256, 3, 1, 1, 1),     # LeReS decoder 3x3
    (2, 40, 40, 1024, 1024, 1, 1, 1, 1),     # ResNeXt 1x1
    (1, 40, 40, 1024, 1024, 3, 1, 1, 32),    # ResNeXt grouped 3x3
    (2, 90, 90, 128, 256, 3, 1, 1, 1),       # ISNet
    (1, 45, 45, 256, 256, 3, 1, 4, 1),       # ISNet RSU4F dilated
    (1, 160, 160, 128, 256, 3, 2, 1, 1),     # CSPNeXt stride-2 stage conv
    (1, 333, 517, 32, 32, 3, 1, 1, 1),       # GridNet-like 32 -> 32 on a ragged frame: several tiles per persistent block, M tail
    (4, 80, 80, 32, 96, 1, 1, 1, 1),         # one-chunk 1x1 (every chunk is a tile's last), ragged N
    (2, 120, 104, 64, 96, 3, 1, 1, 1),       # 64 input channels (two channel blocks), ragged 16 x 16 tiles and N: the weights-stationary kernel
]


@pytest.mark.parametrize("layer", FULL_SIZE_LAYERS, ids=[str(i) for i in range(len(FULL_SIZE_LAYERS))])
def test_full_size_all_tile_configurations_agree_bitwise(layer):
    """size-independent property at BASELINE sizes: every tile configuration of the two independent conv kernels (register-staged
    k_conv_mfma, LDS-DMA k_conv_dma) must produce identical bits -- they share nothing but the fmaf-chain contract -- and one
    sampled output row block is checked against the chain evaluated in numpy float32."""
    from cartoonsegmentation_amd import _lib
    from cartoonsegmentation_amd.runtime import CompiledProgram
    n, h, w, cin, cout, k, stride, dil, groups = layer
    p = Program("full")
    p.winograd = False          # this test is about the DIRECT kernels' tile configurations (the Winograd layers have one kernel: test_gpu_winograd.py)
    x = p.buffer(n, h, w, cin)
    W = rnd('fw%s' % (layer,), (cout, cin // groups, k, k), 1.0 / np.sqrt(cin // groups * k * k))
    b = rnd('fb%s' % (layer,), (cout,), 0.1)
    y = p.conv(x, W, b, stride=stride, pad=dil * (k // 2), dil=dil, groups=groups, act=None)
    y.buf.keep = True
    x.buf.first = 0
    p.plan()
    os.environ["CSM_AUTOTUNE"] = "0"
    try:
        cp = CompiledProgram(p, 'cuda')
    finally:
        os.environ.pop("CSM_AUTOTUNE", None)
    xin = torch.from_numpy(rnd('fx%s' % (layer,), (n, h, w, cin))).cuda()
    xb = x.buf
    cp.workspace[xb.offset:xb.offset + xin.numel()] = xin.reshape(-1)
    L = _lib.load()
    ref, names = None, {}
    try:
        for cfg in list(range(28)) + list(range(38, 53)):       # 38..49: the persistent-block kernels (fall back when K is split / not a 3x3)
            L.csm_debug_force_conv_cfg(cfg)
            cp.run()
            out = cp.read_view(y).cpu().numpy()
            assert np.isfinite(out).all()
            if ref is None:
                ref = out
            assert np.array_equal(ref, out), "tile configuration %d differs from configuration 0 (max %g)" % (cfg, np.abs(ref - out).max())
    finally:
        L.csm_debug_force_conv_cfg(-1)
    if groups == 1:
        # the contract (include/csm355.h) evaluated in numpy for 3 output pixels x all channels: chunks = (32-channel block outer, tap
        # row-major inner); run s of `ksplit` owns chunks [s*T/S, (s+1)*T/S), starts at the bias (run 0) or 0 and is one fmaf chain with the
        # 8-channel blocks in the order 0,4,1,5,2,6,3,7; runs are added ((p0+p1)+p2)...  fmaf = exact product, one rounding.
        xn = xin.cpu().numpy()
        ho, wo = ref.shape[1], ref.shape[2]
        perm = [0, 4, 1, 5, 2, 6, 3, 7]
        S, ncb = p.ops[0]['ksplit'], (cin + 31) // 32
        T = k * k * ncb
        for (oy, ox) in ((0, 0), (ho // 2, wo // 3), (ho - 1, wo - 1)):
            parts = []
            for srun in range(S):
                acc = b.astype(np.float32).copy() if srun == 0 else np.zeros(cout, np.float32)
                for chunk in range(srun * T // S, (srun + 1) * T // S):
                    cb, tap = divmod(chunk, k * k)
                    kh, kw = divmod(tap, k)
                    iy, ix = oy * stride - dil * (k // 2) + kh * dil, ox * stride - dil * (k // 2) + kw * dil
                    if not (0 <= iy < h and 0 <= ix < w):
                        continue
                    for c8 in range(cb * 32, min(cb * 32 + 32, cin), 8):
                        for j in perm:
                            c = c8 + j
                            prod = xn[0, iy, ix, c].astype(np.float64) * W[:, c, kh, kw].astype(np.float64)
                            acc = (acc.astype(np.float64) + prod).astype(np.float32)
                parts.append(acc)
            tot = parts[0]
            for q in parts[1:]:
                tot = (tot + q).astype(np.float32)
            assert np.array_equal(tot, ref[0, oy, ox]), "fmaf chain mismatch at (%d,%d): %g" % (oy, ox, np.abs(tot - ref[0, oy, ox]).max())


# LDS-DMA tile configurations (include/csm355.h `tile`): plain 6-12 / 14-17 / 28-37, 3x3 patch 18-27 / 36, persistent 38-50, weights-stationary 51-52
DMA_CFGS = [c for c in range(6, 53) if c != 13]


def test_dma_kernels_repeated_runs_are_bitwise_stable():
    """stress test for the barrier / LDS hazard class (a stage refilled by DMA while a fragment read of it is still in flight shows as a
    few wrong values in 10^7, once in many runs): every LDS-DMA tile configuration x the eight BASELINE-size layers x both split-K
    executions, 20 repetitions each, every repetition compared bitwise with the first run of configuration 6 ON THE DEVICE (one host
    read per layer).  tools/check_isa_barriers.py is the static half of this check."""
    from cartoonsegmentation_amd import _lib
    from cartoonsegmentation_amd.runtime import CompiledProgram
    L = _lib.load()
    reps = int(os.environ.get("CSM_STRESS_REPS", "20"))
    try:
        for layer in FULL_SIZE_LAYERS:
            n, h, w, cin, cout, k, stride, dil, groups = layer
            p = Program("stress")
            p.winograd = False  # (direct LDS-DMA kernels; k_conv_wino's repeated-run test is in test_gpu_winograd.py)
            x = p.buffer(n, h, w, cin)
            W = rnd('sw%s' % (layer,), (cout, cin // groups, k, k), 1.0 / np.sqrt(cin // groups * k * k))
            y = p.conv(x, W, rnd('sb%s' % (layer,), (cout,), 0.1), stride=stride, pad=dil * (k // 2), dil=dil, groups=groups, act='relu')
            y.buf.keep = True
            x.buf.first = 0
            p.plan()
            os.environ["CSM_AUTOTUNE"] = "0"
            try:
                cp = CompiledProgram(p, 'cuda')
            finally:
                os.environ.pop("CSM_AUTOTUNE", None)
            xin = torch.from_numpy(rnd('sx%s' % (layer,), (n, h, w, cin))).cuda()
            cp.workspace[x.buf.offset:x.buf.offset + xin.numel()] = xin.reshape(-1)
            L.csm_debug_force_conv_cfg(6)
            L.csm_debug_force_splitk_serial(0)
            cp.run()
            ref = cp.read_view(y).clone()
            bad = torch.zeros(len(DMA_CFGS) * 2, dtype=torch.int32, device='cuda')
            modes = (0, 1) if p.ops[0]['ksplit'] > 1 else (0,)
            for ci, cfg in enumerate(DMA_CFGS):
                L.csm_debug_force_conv_cfg(cfg)
                for ser in modes:
                    L.csm_debug_force_splitk_serial(ser)
                    for _ in range(reps):
                        cp.run()
                        bad[2 * ci + ser] += (cp.read_view(y) != ref).sum().to(torch.int32)
            bad = bad.cpu().numpy()
            wrong = {(DMA_CFGS[i // 2], i % 2): int(v) for i, v in enumerate(bad) if v}
            assert not wrong, "layer %s: wrong values per (tile configuration, serial split-K) over %d runs: %s" % (layer, reps, wrong)
    finally:
        L.csm_debug_force_conv_cfg(-1)
        L.csm_debug_force_splitk_serial(-1)


def test_program_run_is_hipgraph_capturable():
    """CompiledProgram.run enqueues only kernels on torch's current stream (no allocation, no sync after the first, tuning, call):
    capture a small net into a hipGraph, replay it on new input, compare with the eager run bit for bit."""
    from cartoonsegmentation_amd.runtime import CompiledProgram
    p = Program("graph")
    x_ext = p.ext_nchw(1, 32, 24, 40); y_ext = p.ext_nchw(1, 64, 24, 40)
    x = p.to_nhwc(x_ext)
    t = p.conv(x, rnd('gw1', (64, 32, 3, 3), 0.1), rnd('gb1', (64,), 0.1), pad=1, act='silu')
    t = p.conv(t, rnd('gw2', (64, 64, 1, 1), 0.1), rnd('gb2', (64,), 0.1), act='relu')
    t = p.conv(t, rnd('gw3', (64, 64, 3, 3), 0.05), rnd('gb3', (64,), 0.1), pad=1, act=None, res=t, res_mode=1)
    p.to_nchw(t, y_ext)
    p.plan()
    cp = CompiledProgram(p, 'cuda')
    xin = torch.from_numpy(rnd('gx', (1, 32, 24, 40))).cuda()
    y = torch.empty((1, 64, 24, 40), device='cuda')
    cp.run(xin, y)                                  # first call tunes the tiles (not capturable), later calls only launch
    torch.cuda.synchronize()
    eager = y.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        cp.run(xin, y)                              # warm the side stream
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            cp.run(xin, y)
    y.zero_()
    g.replay(); torch.cuda.synchronize()
    assert torch.equal(y, eager)
    xin.copy_(torch.from_numpy(rnd('gx2', (1, 32, 24, 40))).cuda())      # new input in the same buffer -> replay computes it
    g.replay(); torch.cuda.synchronize()
    y2 = y.clone()
    cp.run(xin, y); torch.cuda.synchronize()
    assert torch.equal(y, y2) and not torch.equal(y2, eager)


@pytest.mark.parametrize("tag,h,w", [("96x64", 96, 64), ("64x128", 64, 128), ("72x88", 72, 88)])
def test_disparity_estimator_vs_reference_modules(tag, h, w):
    """`depth_est: default` -- Semantics (VGG19-BN slices) and the Disparity GridNet on the HIP engine against the reference's own
    modules (tests/golden/make_golden_nets.py disparity; 96x64 takes the odd-height crop of disparity_estimation.py:172)"""
    from cartoonsegmentation_amd.nets import build_disparity, build_semantics
    from cartoonsegmentation_amd.runtime import CompiledProgram
    from cartoonsegmentation_amd.weights import SynthWeights
    g = dict(np.load(os.path.join(GOLDEN, "net_disparity_%s.npz" % tag)))
    dev = torch.device('cuda')
    x = torch.from_numpy(g['x']).to(dev)
    sem = torch.empty(g['sem'].shape, device=dev)
    CompiledProgram(build_semantics(SynthWeights('semantics.'), h, w), dev).run(x, sem)
    assert np.abs(sem.cpu().numpy() - g['sem']).max() <= 1e-4 * np.abs(g['sem']).max()
    d = torch.empty(g['disp'].shape, device=dev)
    CompiledProgram(build_disparity(SynthWeights('disparity.'), h, w), dev).run(x, torch.from_numpy(g['sem']).to(dev), d)
    assert np.abs(d.cpu().numpy() - g['disp']).max() <= 1e-4 * np.abs(g['disp']).max()
    assert float(d.min()) >= 0.0


def test_zoedepth_head_vs_reference_text():
    """ZoeDepth metric-bins head on the HIP engine (1x1 convs with fused softplus / GELU, CSM_OP_ATTRACTOR, CSM_OP_LOGBINOM) vs the
    fixture from the reference's own class + layers (north_star tolerance 1e-3 relative; measured ~1e-5)"""
    from cartoonsegmentation_amd.nets import build_zoe_head
    from cartoonsegmentation_amd.runtime import CompiledProgram
    from cartoonsegmentation_amd.weights import SynthWeights
    g = dict(np.load(os.path.join(GOLDEN, "net_zoehead_64x96.npz")))
    H, W = 64, 96
    dev = torch.device('cuda')
    f = lambda k: torch.from_numpy(np.ascontiguousarray(g[k].astype(np.float32))).to(dev)
    ext = [f('rel')[:, None].contiguous(), f('out_conv'), f('btlnck'), f('r4'), f('r3'), f('r2'), f('r1')]
    out = torch.empty((1, 1, H, W), device=dev)
    CompiledProgram(build_zoe_head(SynthWeights('zoe.'), 1, H, W, [(H >> s, W >> s) for s in (5, 4, 3, 2, 1)]), dev).run(*ext, out)
    y = out.cpu().numpy()
    assert np.abs(y - g['metric_depth']).max() <= 1e-4 * np.abs(g['metric_depth']).max()


def test_zoedepth_infer_chain_vs_reference_classes():
    """`depth_est: 'zoe'` around a plugged core (tests/golden/zoe_stub_core.py, the stand-in the fixture was made with): padding +
    flip + PrepForMidas (what the core receives for both TTA passes -- here the two samples of one run), the metric depth of DepthModel.infer and the disparity of
    KenBurnsPipeline._depth_est_zoe, against the reference's own classes / text (tests/golden/make_golden_nets.py zoe_infer).
    north_star tolerance for fp32 depth: 1e-3 relative; measured ~1e-5"""
    import sys
    sys.path.insert(0, GOLDEN)
    import zoe_stub_core as stub
    from cartoonsegmentation_amd import _lib
    from cartoonsegmentation_amd.zoedepth import ZoeDepth, depth_to_disparity, midas_size
    g = dict(np.load(os.path.join(GOLDEN, "zoe_infer_70x110.npz")))
    dev = torch.device('cuda')
    seen = []

    def core(xp):
        seen.append(xp.clone())
        return stub.core(xp)
    z = ZoeDepth(SynthWeights('zoe.'), core=core, img_size=tuple(int(v) for v in g['net']), keep_aspect_ratio=True, device=dev)
    x = torch.from_numpy(g['img']).to(dev)
    depth = z.infer(x, pad_input=True, with_flip_aug=True)
    assert len(seen) == 1 and tuple(seen[0].shape) == (2, 3, 96, 128)         # the plain and the mirrored pass: samples 0 and 1 of ONE core run
    seen = [seen[0][0:1], seen[0][1:2]]
    assert tuple(seen[0].shape) == g['prep0'].shape == (1, 3, 96, 128) and midas_size(154, 104, 128, 96) == (128, 96)
    assert np.abs(seen[0].cpu().numpy() - g['prep0']).max() <= 2e-5 and np.abs(seen[1].cpu().numpy() - g['prep1']).max() <= 2e-5
    d = depth.cpu().numpy()
    assert d.shape == g['depth'].shape and np.abs(d - g['depth']).max() <= 1e-4 * np.abs(g['depth']).max()
    disp = depth_to_disparity(depth, float(g['focal']), float(g['baseline'])).cpu().numpy()
    assert np.abs(disp - g['disparity']).max() <= 1e-3 * np.abs(g['disparity']).max()
    # single pass / no padding variants run and differ from the TTA result; zeros, nan and inf follow the reference's rules
    d1 = z.infer(x, pad_input=False, with_flip_aug=False)
    assert d1.shape == depth.shape and torch.isfinite(d1).all() and not torch.equal(d1, depth)
    t = torch.tensor([[0.0, 2.0, float('nan'), float('inf'), -1e-5]], device=dev).view(1, 1, 1, 5)
    o = depth_to_disparity(t.clone(), 55.0, 40.0).view(-1).cpu().numpy()
    assert o[0] == o[1] and o[2] == 0.0 and o[3] == 0.0 and o[4] == 0.0 and abs(o[1] - 2200.0 / 2.00001) < 0.01
    # set_core(None) restores the built-in MiDaS DPT-BEiT-L program (round 4; tests/test_gpu_dpt_beit.py runs it)
    from cartoonsegmentation_amd.zoedepth import DPTBeitCore
    z.set_core(None)
    assert isinstance(z.core, DPTBeitCore)


def test_zoe_depth_estimation_is_wired_into_the_pipeline():
    """set_depth_estimation('zoe') (kenburns_effect.py:541-544) around a plugged stand-in core: generate_kenburns_config runs end to end"""
    import sys
    sys.path.insert(0, GOLDEN)
    import zoe_stub_core as stub
    os.environ["CSM_SYNTHETIC_WEIGHTS"] = "1"
    from anime_3dkenburns import KenBurnsConfig, KenBurnsPipeline
    from cartoonsegmentation_amd import _lib, synth
    cfg = KenBurnsConfig(det_ckpt='synthetic', depth_est='zoe', max_size=512, refine_crf=False, focal=176.0, num_frame=2,
                         mask_refine_kwargs={'refine_method': 'none'})
    pipe = KenBurnsPipeline(cfg)
    pipe.max_instances = 2
    pipe.animeinsseg.set_detect_size(96)
    img = synth.image_u8(320, 352, 91)
    pipe.set_zoe_core(stub.core)             # (without a plug the built-in DPT-BEiT-L core runs: tests/test_gpu_dpt_beit.py)
    kc = pipe.generate_kenburns_config(img)
    assert kc['tenRawDisparity'].shape == (1, 1, 320, 352) and torch.isfinite(kc['tenRawPoints']).all()
    frames = pipe.autozoom(kc, inpaint=False)
    assert len(frames) == 2 and frames[0].shape == (320, 352, 3)
    kcs = pipe.generate_kenburns_configs([img, synth.image_u8(320, 352, 92)])        # batched API dispatches on the selected estimator
    assert torch.equal(kcs[0]['tenRawPoints'], kc['tenRawPoints'])


def test_zoe_tail_group_is_padded_to_the_resident_batch():
    """the DPT-BEiT core keeps one program per (2 B, prepared size) and a new one costs a host re-pack of 345 M parameters.
    A tail group (fewer frames than the group before it, same frame size) is padded to the resident batch with copies of its last
    frame: the core sees the SAME batch again, and the kept frames' results are those of the frames by themselves"""
    import sys
    sys.path.insert(0, GOLDEN)
    import zoe_stub_core as stub
    os.environ["CSM_SYNTHETIC_WEIGHTS"] = "1"
    from anime_3dkenburns import KenBurnsConfig, KenBurnsPipeline
    from cartoonsegmentation_amd import synth
    cfg = KenBurnsConfig(det_ckpt='synthetic', depth_est='zoe', max_size=512, refine_crf=False, focal=176.0, num_frame=2,
                         mask_refine_kwargs={'refine_method': 'none'})
    pipe = KenBurnsPipeline(cfg)
    seen = []

    def core(x, *a, **k):
        seen.append(int(x.shape[0]))
        return stub.core(x, *a, **k)
    pipe.set_zoe_core(core)
    frames = [torch.from_numpy(synth.image_u8(160, 192, 300 + i)).cuda() for i in range(5)]
    alone = [pipe._depth_est_zoe_batch([f])[0].clone() for f in frames]
    pipe._zoe_group = None
    seen.clear()
    first = pipe._depth_est_zoe_batch(frames[:3])
    tail = pipe._depth_est_zoe_batch(frames[3:])
    torch.cuda.synchronize()
    assert len(set(seen)) == 1, seen                       # the core ran the same batch (3 frames + their mirrored passes) both times
    for got, ref in zip(first + tail, alone):
        assert torch.equal(got, ref)
    other = pipe._depth_est_zoe_batch([torch.from_numpy(synth.image_u8(128, 160, 9)).cuda()] * 2)     # another frame size: no padding
    assert len(other) == 2


@pytest.mark.parametrize("n,h,w,c,act", [(1, 8, 16, 32, 'silu'), (2, 37, 45, 64, 'silu'), (1, 80, 80, 128, 'relu'), (3, 5, 3, 32, None), (1, 20, 20, 512, 'silu')])
def test_dwconv5_bit_exact(n, h, w, c, act):
    """RTMDet's 5 x 5 depthwise layers (k_dwconv_lds) at full / ragged tiles: bias, then taps in (ky, kx) order, bit-exact against the oracle"""
    p = Program("dw5")
    x_ext = p.ext_nchw(n, c, h, w)
    y_ext = p.ext_nchw(n, c, h, w)
    y = p.dwconv(p.to_nhwc(x_ext), rnd('dw5w%d' % c, (c, 1, 5, 5), 0.2), rnd('dw5b%d' % c, (c,), 0.1), pad=2, act=act)
    p.to_nchw(y, y_ext)
    (yo,), (yd,) = run_both(p, [rnd('dw5x%d%d%d' % (n, h, w), (n, c, h, w))], [(n, c, h, w)])
    assert np.isfinite(yd).all() and np.array_equal(yo, yd)


@pytest.mark.parametrize("n,h,w,c,ho,wo,align", [(2, 23, 23, 32, 45, 45, False), (1, 45, 45, 64, 90, 90, False), (2, 12, 12, 32, 23, 23, True),
                                                  (1, 20, 20, 256, 80, 80, False), (1, 17, 31, 8, 40, 50, True), (3, 9, 9, 4, 9, 9, False)])
def test_bilinear_row_kernel_bit_exact(n, h, w, c, ho, wo, align):
    """k_bilinear_rows (output row = block coordinate, one magic-number division per word) against the oracle's resize"""
    p = Program("bil")
    x_ext = p.ext_nchw(n, c, h, w)
    y_ext = p.ext_nchw(n, c, ho, wo)
    y = p.bilinear(p.to_nhwc(x_ext), (ho, wo), align_corners=align)
    p.to_nchw(y, y_ext)
    (yo,), (yd,) = run_both(p, [rnd('bilx%d%d%d%d' % (n, h, w, c), (n, c, h, w))], [(n, c, ho, wo)])
    assert np.isfinite(yd).all() and np.array_equal(yo, yd)
