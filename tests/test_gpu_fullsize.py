"""GPU parity at BASELINE sizes: the sizes AnimeInsSeg.infer / _depth_est_leres actually run at
(animeinsseg/__init__.py:187,395-399,638-665; anime_3dkenburns/kenburns_effect.py:563-581) -- RTMDet-Ins-L @640, ISNet @720,
LeReS @640, 1024 x 1024 frames -- HIP against the CPU oracle, BIT-EXACT, plus batch invariance (n = 8 programs give every sample the
bits of the n = 1 program) and the non-square / det-1024 cases.  The oracle needs a few seconds per net on the GPU box's host cores.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from cartoonsegmentation_amd import synth  # noqa: E402
from cartoonsegmentation_amd.weights import SynthWeights  # noqa: E402
from oracle import nets as onets  # noqa: E402


def _seeded(shape, seed):
    return np.random.default_rng(seed).normal(0, 1, shape).astype(np.float32)


def _run_hip(prog, ext):
    from cartoonsegmentation_amd.runtime import CompiledProgram
    cp = CompiledProgram(prog, 'cuda')
    dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in ext]
    cp.run(*dev)
    torch.cuda.synchronize()
    return cp, dev


_ORACLE_CACHE = {}


def _rtmdet_oracle(S, x):
    """oracle head outputs of RTMDet-Ins-L @S for input x (cached: several tests look at the same run)"""
    from cartoonsegmentation_amd.nets import build_rtmdet
    key = (S, x.tobytes()[:64], float(x.sum()))
    if key not in _ORACLE_CACHE:
        rp, _ = build_rtmdet(SynthWeights('rtmdet.'), 1, S, S)
        want = rp.cls + rp.reg + rp.kern + [rp.mask_feat]
        v = onets.run_program(rp.prog, [x], want_views=want)
        _ORACLE_CACHE[key] = [v[t] for t in want]
    return _ORACLE_CACHE[key]


def test_rtmdet_640_bit_exact_and_batch_invariant():
    """RTMDet-Ins-L @640: n = 1 HIP == oracle on every head output; the n = 8 program (the bench's batch) gives sample k the bits of
    the n = 1 program run on that sample, whichever way the device executes split-K"""
    from cartoonsegmentation_amd import _lib
    from cartoonsegmentation_amd.nets import build_rtmdet
    S = 640
    xs = _seeded((8, 3, S, S), 11)
    rp1, _ = build_rtmdet(SynthWeights('rtmdet.'), 1, S, S)
    assert any(o['ksplit'] > 1 for o in rp1.prog.ops), "the 20x20 / 40x40 layers are expected to use split-K at batch 1"
    want1 = rp1.cls + rp1.reg + rp1.kern + [rp1.mask_feat]
    ref = _rtmdet_oracle(S, xs[:1])
    cp1, _ = _run_hip(rp1.prog, [xs[:1]])
    one = [cp1.read_view(t).cpu().numpy() for t in want1]
    for a, b in zip(ref, one):
        assert np.isfinite(b).all() and np.array_equal(a, b), np.abs(a - b).max()
    rp8, _ = build_rtmdet(SynthWeights('rtmdet.'), 8, S, S)
    assert [o['ksplit'] for o in rp8.prog.ops] == [o['ksplit'] for o in rp1.prog.ops]          # per-sample rule
    want8 = rp8.cls + rp8.reg + rp8.kern + [rp8.mask_feat]
    L = _lib.load()
    outs = {}
    try:
        for mode in (-1, 0, 1):                                                                 # tuned, parallel, serial split-K
            L.csm_debug_force_splitk_serial(mode)
            cp8, _ = _run_hip(rp8.prog, [xs])
            outs[mode] = [cp8.read_view(t).cpu().numpy() for t in want8]
    finally:
        L.csm_debug_force_splitk_serial(-1)
    for mode in (0, 1):
        assert all(np.array_equal(a, b) for a, b in zip(outs[-1], outs[mode])), "split-K execution mode %d changes bits" % mode
    for a, b in zip(one, outs[-1]):
        assert np.array_equal(a[0], b[0]), "sample 0 of the batch-8 program differs from the batch-1 program"
    cp1.run(torch.from_numpy(np.ascontiguousarray(xs[5:6])).cuda()); torch.cuda.synchronize()
    for t, b in zip(want1, outs[-1]):
        assert np.array_equal(cp1.read_view(t).cpu().numpy()[0], b[5])


def test_isnet_720_two_instances_bit_exact():
    from cartoonsegmentation_amd.nets import build_isnet
    T = 720
    x = _seeded((2, 4, T, T), 12)
    x[:, 3] = (x[:, 3] > 0)                                    # the mask channel is binary
    prog = build_isnet(SynthWeights('isnet.'), 2, T, T)
    yo = np.zeros((2, 1, T, T), np.float32)
    onets.run_program(prog, [x, yo])
    _, dev = _run_hip(prog, [x, np.full((2, 1, T, T), np.nan, np.float32)])
    yd = dev[1].cpu().numpy()
    assert np.isfinite(yd).all() and np.array_equal(yo, yd), np.abs(yo - yd).max()
    # one sample alone (n = 1 program) reproduces its half of the pair
    p1 = build_isnet(SynthWeights('isnet.'), 1, T, T)
    _, d1 = _run_hip(p1, [x[1:2], np.full((1, 1, T, T), np.nan, np.float32)])
    assert np.array_equal(d1[1].cpu().numpy()[0], yd[1])


def test_leres_640_bit_exact():
    from cartoonsegmentation_amd.nets import build_leres
    S = 640
    x = _seeded((1, 3, S, S), 13)
    prog = build_leres(SynthWeights('leres.'), 1, S, S)
    yo = np.zeros((1, 1, S, S), np.float32)
    onets.run_program(prog, [x, yo])
    _, dev = _run_hip(prog, [x, np.full((1, 1, S, S), np.nan, np.float32)])
    yd = dev[1].cpu().numpy()
    assert np.isfinite(yd).all() and np.array_equal(yo, yd), np.abs(yo - yd).max()
    p4 = build_leres(SynthWeights('leres.'), 4, S, S)
    x4 = np.concatenate([_seeded((3, 3, S, S), 14), x])
    _, d4 = _run_hip(p4, [x4, np.full((4, 1, S, S), np.nan, np.float32)])
    assert np.array_equal(d4[1].cpu().numpy()[3], yd[0])       # batch invariance


# frame (H, W), detector size, ISNet size: the benchmark's frame; a landscape HD frame; an odd small frame; 1080p through det 1024
INFER_CASES = [(1024, 1024, 640, 720), (720, 1280, 640, 720), (333, 517, 640, 720), (1080, 1920, 1024, 720)]


@pytest.mark.parametrize("H,W,S,T", INFER_CASES, ids=["%dx%d-det%d" % (h, w, s) for h, w, s, _ in INFER_CASES])
def test_infer_full_size_matches_oracle(H, W, S, T):
    """AnimeInsSeg.infer at the sizes the reference runs it (det 640 / 1024, refine 720) vs the oracle pipeline: instances, boxes,
    scores and masks after threshold identical"""
    from animeinsseg import AnimeInsSeg
    from cartoonsegmentation_amd.nets import build_isnet, build_rtmdet
    from oracle import segment as oseg
    img = synth.image_u8(H, W, 5)
    net = AnimeInsSeg('synthetic', default_det_size=S, refine_kwargs={'refine_method': 'refinenet_isnet', 'refine_size': T})
    inst = net.infer(img, pred_score_thr=0.3, max_instances=2, output_type='numpy')
    rp, cfg = build_rtmdet(SynthWeights('rtmdet.'), 1, S, S)
    cfg.max_per_img = 2
    d = oseg.detect(img, rp, cfg, S, pred_score_thr=0.3)
    assert d['n'] == len(inst) and d['n'] > 0
    assert np.array_equal(d['scores'], inst.scores) and np.array_equal(d['bboxes'], inst.bboxes)
    progs = {}

    def isnet_for(b):
        if b not in progs:
            progs[b] = build_isnet(SynthWeights('isnet.'), b, T, T)
        return progs[b]
    refined = oseg.refine(img, d['masks'], isnet_for, T, 0.3)
    assert inst.masks.shape == (d['n'], H, W) and np.array_equal(refined.astype(bool), inst.masks)


def test_conv_splitk_modes_and_batches_agree_on_single_layers():
    """single layers with ksplit > 1 (1x1 K = 1024 on a 20x20 map, 3x3 on 40x40, a 16-channel layer that takes the register-staged
    kernel): parallel == serial == oracle, and the batch-3 program equals three batch-1 runs"""
    from cartoonsegmentation_amd import _lib
    from cartoonsegmentation_amd.program import Program
    from cartoonsegmentation_amd.runtime import CompiledProgram
    from cartoonsegmentation_amd.weights import hash_uniform
    L = _lib.load()

    def rnd(name, shape, scale=1.0):
        return (hash_uniform(name, int(np.prod(shape))) * scale).astype(np.float32).reshape(shape)

    for (h, w, cin, cout, k) in ((20, 20, 1024, 1024, 1), (40, 40, 256, 256, 3), (45, 45, 16, 64, 3), (20, 20, 512, 96, 3)):
        outs = {}
        for n in (1, 3):
            p = Program("sk")
            p.winograd = False          # this test is about the DIRECT kernels' split-K executions (a 40 x 40 256 -> 256 layer lowers to Winograd F(4x4) since round 6)
            x_ext = p.ext_nchw(n, cin, h, w); y_ext = p.ext_nchw(n, cout, h, w)
            y = p.conv(p.to_nhwc(x_ext), rnd('skw%d%d' % (cin, k), (cout, cin, k, k), 1.0 / np.sqrt(cin * k * k)), rnd('skb', (cout,), 0.1),
                       pad=k // 2, act='silu')
            p.to_nchw(y, y_ext)
            p.plan()
            assert p.ops[1]['ksplit'] > 1, (h, w, cin, cout, k)
            xin = rnd('skx', (3, cin, h, w))[:n]
            yo = np.zeros((n, cout, h, w), np.float32)
            onets.run_program(p, [np.ascontiguousarray(xin), yo])
            try:
                for mode in (0, 1, -1):
                    L.csm_debug_force_splitk_serial(mode)
                    yd = torch.full((n, cout, h, w), float('nan'), device='cuda')
                    CompiledProgram(p, 'cuda').run(torch.from_numpy(np.ascontiguousarray(xin)).cuda(), yd)
                    assert np.array_equal(yd.cpu().numpy(), yo), (h, w, cin, cout, k, n, mode)
            finally:
                L.csm_debug_force_splitk_serial(-1)
            outs[n] = yo
        assert np.array_equal(outs[1][0], outs[3][0])
