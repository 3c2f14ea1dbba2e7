"""GPU parity of the fused Winograd F(4x4, 3x3) convolution (csrc/wino4.hip::k_conv_wino4) against the CPU oracle of the same arithmetic
(oracle/nets_oracle.c::orc_conv_wino4): BIT-EXACT -- every transform value is the contract's fp32 expression and every product sum one
fmaf chain on both sides (include/csm355.h "Winograd F(4x4) contract").  Shapes cover full and ragged block tiles (32 x 16 output pixels),
sizes that are not multiples of 4 (partly-outside Winograd tiles), 1-8 channel blocks, both residual modes, batches, channel-slice
views, whole nets with every eligible layer forced to F(4x4)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from cartoonsegmentation_amd import program as P  # noqa: E402
from oracle import nets as onets  # noqa: E402
from test_oracle_winograd4 import forced  # noqa: E402


def _run_both(prog, ext_in, out_shapes):
    from cartoonsegmentation_amd.runtime import CompiledProgram
    outs_o = [np.zeros(s, np.float32) for s in out_shapes]
    onets.run_program(prog, [np.ascontiguousarray(a) for a in ext_in[:1]] + outs_o + [np.ascontiguousarray(a) for a in ext_in[1:]])
    cp = CompiledProgram(prog, 'cuda')
    outs_d = [torch.full(s, float('nan'), device='cuda') for s in out_shapes]
    cp.run(torch.from_numpy(np.ascontiguousarray(ext_in[0])).cuda(), *outs_d, *[torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in ext_in[1:]])
    torch.cuda.synchronize()
    return outs_o, [t.cpu().numpy() for t in outs_d]


def _one_layer(rng, n, h, w, cin, cout, act, res_mode, bias=True):
    with forced('f4'):
        p = P.Program("wino4")
        x_ext = p.ext_nchw(n, cin, h, w)
        y_ext = p.ext_nchw(n, cout, h, w)
        x = p.to_nhwc(x_ext)
        wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
        b = (rng.standard_normal(cout) * 0.1).astype(np.float32) if bias else None
        slope = (rng.uniform(0.05, 0.3, cout)).astype(np.float32) if act == 'prelu' else None
        res = p.to_nhwc(p.ext_nchw(n, cout, h, w)) if res_mode else None
        y = p.conv(x, wt, b, pad=1, act=act, slope=slope, res=res, res_mode=res_mode)
        p.to_nchw(y, y_ext)
    assert [o['flags'] for o in p.ops if o['kind'] == P.OP_CONV] == [P.CONV_FLAG_WINOGRAD4]
    ext_in = [rng.standard_normal((n, cin, h, w)).astype(np.float32)]
    if res_mode:
        ext_in.append(rng.standard_normal((n, cout, h, w)).astype(np.float32))
    (yo,), (yd,) = _run_both(p, ext_in, [(n, cout, h, w)])
    return yo, yd


CASES = [
    # n, h, w, cin, cout, act, res_mode
    (1, 16, 32, 32, 64, None, 0),              # exactly one block tile, one channel block (two raw stages, eight steps)
    (1, 32, 64, 64, 64, 'relu', 0),
    (2, 13, 37, 64, 128, 'silu', 2),           # odd sizes, two output-channel tiles, residual after the activation
    (1, 45, 45, 96, 64, 'relu', 1),            # three channel blocks, residual before the activation
    (3, 23, 70, 256, 256, 'relu', 0),          # eight channel blocks (the LeReS decoder's width), batch 3
    (1, 5, 3, 32, 64, 'sigmoid', 0),           # smaller than a block tile in both directions
    (1, 90, 90, 128, 64, 'prelu', 0),
    (1, 80, 80, 256, 256, 'silu', 0),          # the rule's smallest map at its RTMDet width
]


@pytest.mark.parametrize("case", CASES, ids=[str(i) for i in range(len(CASES))])
def test_winograd4_conv_bit_exact(case):
    n, h, w, cin, cout, act, res_mode = case
    rng = np.random.default_rng(hash(case[:5]) & 0xffff)
    yo, yd = _one_layer(rng, n, h, w, cin, cout, act, res_mode)
    assert np.isfinite(yd).all() and np.array_equal(yd, yo)


def test_winograd4_reads_and_writes_channel_slices():
    """input = channels [32, 96) of a 128-channel buffer, output = channels [64, 128) of a 192-channel buffer whose other channels a
    direct 1x1 and an F(4x4) conv on another slice fill"""
    rng = np.random.default_rng(7)
    n, h, w = 2, 21, 40
    with forced('f4'):
        p = P.Program("slices")
        x_ext = p.ext_nchw(n, 128, h, w)
        y_ext = p.ext_nchw(n, 192, h, w)
        x = p.to_nhwc(x_ext)
        cat = p.buffer(n, h, w, 192)
        w1 = (rng.standard_normal((64, 64, 3, 3)) / 24).astype(np.float32)
        p.conv(x.slice(32, 96), w1, rng.standard_normal(64).astype(np.float32), pad=1, act='relu', out=cat.slice(64, 128))
        w2 = (rng.standard_normal((64, 128, 1, 1)) / 11).astype(np.float32)
        p.conv(x, w2, None, act='relu', out=cat.slice(0, 64))
        w3 = (rng.standard_normal((64, 32, 3, 3)) / 17).astype(np.float32)
        p.conv(x.slice(96, 128), w3, None, pad=1, out=cat.slice(128, 192))
        p.to_nchw(cat, y_ext)
    assert sum(1 for o in p.ops if o['flags'] & P.CONV_FLAG_WINOGRAD4) == 2
    (yo,), (yd,) = _run_both(p, [rng.standard_normal((n, 128, h, w)).astype(np.float32)], [(n, 192, h, w)])
    assert np.isfinite(yd).all() and np.array_equal(yd, yo)


def test_winograd4_is_batch_invariant_and_stable_over_repeated_runs():
    """a sample's bits do not depend on the batch it runs in, and twenty runs of the same launch give the same tensor (the barrier /
    LDS-DMA protocol of the step pipeline: private U slots, rotating transform groups, raw stages)"""
    from cartoonsegmentation_amd.runtime import CompiledProgram
    rng = np.random.default_rng(9)
    cin, cout, h, w = 128, 128, 50, 77
    wt = (rng.standard_normal((cout, cin, 3, 3)) / 34).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    x = torch.from_numpy(rng.standard_normal((4, cin, h, w)).astype(np.float32)).cuda()
    outs = {}
    for n in (1, 4):
        with forced('f4'):
            p = P.Program("b%d" % n)
            x_ext = p.ext_nchw(n, cin, h, w); y_ext = p.ext_nchw(n, cout, h, w)
            p.to_nchw(p.conv(p.to_nhwc(x_ext), wt, b, pad=1, act='silu'), y_ext)
        cp = CompiledProgram(p, 'cuda')
        y = torch.empty((n, cout, h, w), device='cuda')
        cp.run(x[:n].contiguous(), y)
        first = y.clone()
        for _ in range(20):
            y.fill_(float('nan'))
            cp.run(x[:n].contiguous(), y)
            assert torch.equal(y, first)
        outs[n] = first
    assert torch.equal(outs[4][:1], outs[1])


def test_nets_with_winograd4_layers_hip_equals_oracle():
    """ISNet and LeReS with EVERY eligible layer forced to F(4x4): HIP == oracle bit for bit; the F(4x4) program differs from the direct
    program only by rounding"""
    from cartoonsegmentation_amd import nets
    from cartoonsegmentation_amd.runtime import CompiledProgram
    from cartoonsegmentation_amd.weights import SynthWeights
    for name, build, shape_in, shape_out in (
            ('isnet', lambda: nets.build_isnet(SynthWeights('isnet.'), 1, 96, 128), (1, 4, 96, 128), (1, 1, 96, 128)),
            ('leres', lambda: nets.build_leres(SynthWeights('leres.'), 1, 96, 64), (1, 3, 96, 64), (1, 1, 96, 64))):
        x = np.random.default_rng(3).uniform(0, 1, shape_in).astype(np.float32)
        res = {}
        for mode in ('f4', 'direct'):
            with forced(mode):
                p = build()
            nw = sum(1 for o in p.ops if o['kind'] == P.OP_CONV and o['flags'] & P.CONV_FLAG_WINOGRAD4)
            assert (nw > 5) == (mode == 'f4'), (name, nw)
            yo = np.zeros(shape_out, np.float32)
            onets.run_program(p, [x, yo])
            cp = CompiledProgram(p, 'cuda')
            yd = torch.full(shape_out, float('nan'), device='cuda')
            cp.run(torch.from_numpy(x).cuda(), yd)
            torch.cuda.synchronize()
            assert np.array_equal(yd.cpu().numpy(), yo), (name, mode)
            res[mode] = yo
        assert np.abs(res['f4'] - res['direct']).max() <= 2e-4 * max(1e-3, np.abs(res['direct']).max()), name


def test_winograd4_batches_beyond_the_descriptor_range_are_split_by_sample():
    from cartoonsegmentation_amd.runtime import CompiledProgram
    rng = np.random.default_rng(31)
    n, h, w, cin, cout = 5, 26, 40, 64, 64
    with forced('f4'):
        p = P.Program("chunks")
        x_ext = p.ext_nchw(n, cin, h, w); r_ext = p.ext_nchw(n, cout, h, w); y_ext = p.ext_nchw(n, 128, h, w)
        x = p.to_nhwc(x_ext); r = p.to_nhwc(r_ext)
        cat = p.buffer(n, h, w, 128)
        p.conv(x, (rng.standard_normal((cout, cin, 3, 3)) / 24).astype(np.float32), rng.standard_normal(cout).astype(np.float32), pad=1,
               act='relu', res=r, res_mode=2, out=cat.slice(64, 128))
        p.conv(x, (rng.standard_normal((64, cin, 1, 1)) / 8).astype(np.float32), None, out=cat.slice(0, 64))
        p.to_nchw(cat, y_ext)
    xs = torch.from_numpy(rng.standard_normal((n, cin, h, w)).astype(np.float32)).cuda()
    rs = torch.from_numpy(rng.standard_normal((n, cout, h, w)).astype(np.float32)).cuda()
    cp = CompiledProgram(p, 'cuda')
    outs = []
    per_sample = h * w * cin * 4
    try:
        for limit in (None, 2 * per_sample + 100, per_sample):
            if limit is None:
                os.environ.pop("CSM_WINO_MAX_BYTES", None)
            else:
                os.environ["CSM_WINO_MAX_BYTES"] = str(limit)
            y = torch.full((n, 128, h, w), float('nan'), device='cuda')
            cp.run(xs, rs, y)
            torch.cuda.synchronize()
            outs.append(y)
    finally:
        os.environ.pop("CSM_WINO_MAX_BYTES", None)
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("seed", list(range(10)))
def test_winograd4_random_shapes_bit_exact(seed):
    rng = np.random.default_rng(4000 + seed)
    n = int(rng.integers(1, 4))
    h, w = (1, 1) if seed == 0 else ((2, 67) if seed == 1 else (int(rng.integers(1, 72)), int(rng.integers(1, 72))))
    cin, cout = int(rng.choice([32, 64, 96, 160])), int(rng.choice([64, 128, 192]))
    act = [None, 'relu', 'silu', 'prelu', 'sigmoid', 'hsigmoid'][int(rng.integers(0, 6))]
    res_mode = int(rng.integers(0, 3))
    yo, yd = _one_layer(rng, n, h, w, cin, cout, act, res_mode, bias=bool(rng.integers(0, 2)))
    assert np.isfinite(yd).all() and np.array_equal(yd, yo), (n, h, w, cin, cout, act, res_mode)


@pytest.mark.parametrize("form", ["1", "2", "3", "6"])
def test_every_execution_form_gives_the_oracles_bits(form):
    """the launcher's choice between the fused form (a block owns all 36 frequencies and finishes its outputs) and the row-split forms
    (6 / RPB x the blocks of 2 RPB waves + k_wino4_rowpass, for launches with fewer block tiles than CUs) is speed only: each form,
    forced through CSM_WINO4_FORM (read once per process, hence a subprocess), passes the single-layer, slice, chunking and
    random-shape tests above bit for bit"""
    import subprocess
    import sys
    e = dict(os.environ, CSM_WINO4_FORM=form)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k",
                        "winograd4_conv_bit_exact or channel_slices or random_shapes or descriptor_range"], env=e,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
