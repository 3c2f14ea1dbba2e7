"""GPU parity of the fused Winograd F(2x2, 3x3) convolution (csrc/wino.hip::k_conv_wino) against the CPU oracle of the same arithmetic
(oracle/nets_oracle.c::orc_conv_wino): BIT-EXACT -- every transform value is one fp32 operation and every product sum one fmaf chain on
both sides (include/csm355.h "Winograd contract").  Shapes cover full and ragged block tiles (32 x 8 output pixels), odd heights /
widths (half-outside Winograd tiles), several channel blocks, both residual modes, batches, and channel-slice views (free concat)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from cartoonsegmentation_amd import program as P  # noqa: E402
from oracle import nets as onets  # noqa: E402


class _forced:
    """lower with / without Winograd whatever the per-sample size rule says"""
    def __init__(self, wino, min_pixels=0):
        self.new = (wino, min_pixels)

    def __enter__(self):
        self.old = (P.Program.winograd, P.WINO_MIN_PIXELS, P.Program.winograd4)
        P.Program.winograd, P.WINO_MIN_PIXELS = self.new
        P.Program.winograd4 = False                # (this file is about F(2x2): tests/test_gpu_winograd4.py covers F(4x4))

    def __exit__(self, *a):
        P.Program.winograd, P.WINO_MIN_PIXELS, P.Program.winograd4 = self.old


def _run_both(prog, ext_in, out_shapes):
    from cartoonsegmentation_amd.runtime import CompiledProgram
    outs_o = [np.zeros(s, np.float32) for s in out_shapes]
    onets.run_program(prog, [np.ascontiguousarray(a) for a in ext_in[:1]] + outs_o + [np.ascontiguousarray(a) for a in ext_in[1:]])
    cp = CompiledProgram(prog, 'cuda')
    outs_d = [torch.full(s, float('nan'), device='cuda') for s in out_shapes]
    cp.run(torch.from_numpy(np.ascontiguousarray(ext_in[0])).cuda(), *outs_d, *[torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in ext_in[1:]])
    torch.cuda.synchronize()
    return outs_o, [t.cpu().numpy() for t in outs_d]


CASES = [
    # n, h, w, cin, cout, act, res_mode
    (1, 8, 32, 32, 64, None, 0),               # exactly one block tile, one channel block
    (1, 16, 64, 64, 64, 'relu', 0),
    (2, 13, 37, 64, 128, 'silu', 2),           # odd sizes, two output-channel tiles, residual after the activation
    (1, 45, 45, 96, 64, 'relu', 1),            # three channel blocks, residual before the activation
    (3, 23, 70, 256, 256, 'relu', 0),          # eight channel blocks (the LeReS decoder's width), batch 3
    (1, 5, 3, 32, 64, 'sigmoid', 0),           # smaller than a block tile in both directions
    (1, 90, 90, 128, 64, 'prelu', 0),
]


@pytest.mark.parametrize("case", CASES, ids=[str(i) for i in range(len(CASES))])
def test_winograd_conv_bit_exact(case):
    n, h, w, cin, cout, act, res_mode = case
    rng = np.random.default_rng(hash(case[:5]) & 0xffff)
    with _forced(True):
        p = P.Program("wino")
        x_ext = p.ext_nchw(n, cin, h, w)
        y_ext = p.ext_nchw(n, cout, h, w)
        x = p.to_nhwc(x_ext)
        wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
        b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        slope = (rng.uniform(0.05, 0.3, cout)).astype(np.float32) if act == 'prelu' else None
        res = p.to_nhwc(p.ext_nchw(n, cout, h, w)) if res_mode else None
        y = p.conv(x, wt, b, pad=1, act=act, slope=slope, res=res, res_mode=res_mode)
        p.to_nchw(y, y_ext)
    assert [o['flags'] & P.CONV_FLAG_WINOGRAD for o in p.ops if o['kind'] == P.OP_CONV] == [P.CONV_FLAG_WINOGRAD]
    ext_in = [rng.standard_normal((n, cin, h, w)).astype(np.float32)]
    if res_mode:
        ext_in.append(rng.standard_normal((n, cout, h, w)).astype(np.float32))
    (yo,), (yd,) = _run_both(p, ext_in, [(n, cout, h, w)])
    assert np.isfinite(yd).all() and np.array_equal(yd, yo)


def test_winograd_reads_and_writes_channel_slices():
    """torch.cat is free in the layer programs: a conv reads a channel slice of a wider buffer and writes into a slice of the consumer's.
    Here: input = channels [32, 96) of a 128-channel buffer, output = channels [64, 128) of a 192-channel buffer whose other channels a
    second (direct) conv fills"""
    rng = np.random.default_rng(7)
    n, h, w = 2, 21, 40
    with _forced(True):
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
    assert sum(1 for o in p.ops if o['flags'] & P.CONV_FLAG_WINOGRAD) == 2
    (yo,), (yd,) = _run_both(p, [rng.standard_normal((n, 128, h, w)).astype(np.float32)], [(n, 192, h, w)])
    assert np.isfinite(yd).all() and np.array_equal(yd, yo)


def test_winograd_is_batch_invariant_and_stable_over_repeated_runs():
    """a sample's bits do not depend on the batch it runs in (no split-K, no batch-dependent launch form), and twenty runs of the same
    launch give the same tensor (the LDS-DMA pipeline's barrier protocol: tools/check_isa_barriers.py is the static half of this)"""
    from cartoonsegmentation_amd.runtime import CompiledProgram
    rng = np.random.default_rng(9)
    cin, cout, h, w = 128, 128, 50, 77
    wt = (rng.standard_normal((cout, cin, 3, 3)) / 34).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    x = torch.from_numpy(rng.standard_normal((4, cin, h, w)).astype(np.float32)).cuda()
    outs = {}
    for n in (1, 4):
        with _forced(True):
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


def test_nets_with_winograd_layers_hip_equals_oracle():
    """whole nets lowered with the Winograd rule forced on for every eligible layer (small inputs would otherwise stay direct): ISNet and
    LeReS, HIP == oracle bit for bit; and the Winograd program differs from the direct program only by rounding"""
    from cartoonsegmentation_amd import nets
    from cartoonsegmentation_amd.runtime import CompiledProgram
    from cartoonsegmentation_amd.weights import SynthWeights
    for name, build, shape_in, shape_out in (
            ('isnet', lambda: nets.build_isnet(SynthWeights('isnet.'), 1, 96, 128), (1, 4, 96, 128), (1, 1, 96, 128)),
            ('leres', lambda: nets.build_leres(SynthWeights('leres.'), 1, 96, 64), (1, 3, 96, 64), (1, 1, 96, 64))):
        x = np.random.default_rng(3).uniform(0, 1, shape_in).astype(np.float32)
        res = {}
        for wino in (True, False):
            with _forced(wino):
                p = build()
            nw = sum(1 for o in p.ops if o['kind'] == P.OP_CONV and o['flags'] & P.CONV_FLAG_WINOGRAD)
            assert (nw > 5) == wino, (name, nw)
            yo = np.zeros(shape_out, np.float32)
            onets.run_program(p, [x, yo])
            cp = CompiledProgram(p, 'cuda')
            yd = torch.full(shape_out, float('nan'), device='cuda')
            cp.run(torch.from_numpy(x).cuda(), yd)
            torch.cuda.synchronize()
            assert np.array_equal(yd.cpu().numpy(), yo), (name, wino)
            res[wino] = yo
        assert np.abs(res[True] - res[False]).max() <= 2e-4 * max(1e-3, np.abs(res[False]).max()), name


def test_batches_beyond_the_descriptor_range_are_split_by_sample():
    """the kernels reach the activations through a 32-bit range-checked buffer descriptor: a launch takes as many samples as fit 2 GiB of
    input view and larger batches are split by sample.  Exercised with a lowered limit (CSM_WINO_MAX_BYTES): 5 samples in chunks of 2 / 1
    give the bits of the single launch, with a residual and channel-slice output"""
    from cartoonsegmentation_amd.runtime import CompiledProgram
    rng = np.random.default_rng(31)
    n, h, w, cin, cout = 5, 26, 40, 64, 64
    with _forced(True):
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
def test_winograd_random_shapes_bit_exact(seed):
    """seeded random layers: any sample count / map size (down to 1 x 1: a single half-outside tile) / channel-block count / column-tile
    count / activation / residual mode, both block-tile geometries decided by the launcher"""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 4))
    h, w = (1, 1) if seed == 0 else ((2, 67) if seed == 1 else (int(rng.integers(1, 72)), int(rng.integers(1, 72))))
    cin, cout = int(rng.choice([32, 64, 96, 160])), int(rng.choice([64, 128, 192]))
    act = [None, 'relu', 'silu', 'prelu', 'sigmoid', 'hsigmoid'][int(rng.integers(0, 6))]
    res_mode = int(rng.integers(0, 3))
    with _forced(True):
        p = P.Program("rnd")
        x_ext = p.ext_nchw(n, cin, h, w)
        y_ext = p.ext_nchw(n, cout, h, w)
        x = p.to_nhwc(x_ext)
        wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
        b = (rng.standard_normal(cout) * 0.1).astype(np.float32) if rng.integers(0, 2) else None
        slope = rng.uniform(0.05, 0.3, cout).astype(np.float32) if act == 'prelu' else None
        res = p.to_nhwc(p.ext_nchw(n, cout, h, w)) if res_mode else None
        p.to_nchw(p.conv(x, wt, b, pad=1, act=act, slope=slope, res=res, res_mode=res_mode), y_ext)
    assert [o['flags'] & P.CONV_FLAG_WINOGRAD for o in p.ops if o['kind'] == P.OP_CONV] == [P.CONV_FLAG_WINOGRAD]
    ext_in = [rng.standard_normal((n, cin, h, w)).astype(np.float32)]
    if res_mode:
        ext_in.append(rng.standard_normal((n, cout, h, w)).astype(np.float32))
    (yo,), (yd,) = _run_both(p, ext_in, [(n, cout, h, w)])
    assert np.isfinite(yd).all() and np.array_equal(yd, yo), (n, h, w, cin, cout, act, res_mode)


@pytest.mark.parametrize("env", [{"CSM_WINO_WAVES": "4"}, {"CSM_WINO_GEO": "1"}, {"CSM_WINO_GEO": "0"}],
                         ids=["four_wave_form", "geometry_16x16_forced", "geometry_32x8_forced"])
def test_every_kernel_form_gives_the_oracles_bits(env):
    """the launcher's choices are speed only: the one-wave-per-SIMD form (k_conv_wino) and each block-tile geometry of k_conv_wino8, forced
    through their environment switches (read once per process, hence a subprocess), pass the single-layer and slice tests above bit for bit"""
    import subprocess
    import sys
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k",
                        "winograd_conv_bit_exact or channel_slices or random_shapes"], env=e, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
