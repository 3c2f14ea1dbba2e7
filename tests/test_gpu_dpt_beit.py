"""GPU parity of the MiDaS DPT-BEiT core of ZoeDepth (nets/dpt_beit.py + csrc/tokens.hip) against the CPU oracle interpreter.
The linear / conv layers are bit-exact fmaf chains on both sides; LayerNorm, softmax and the attention sums are reductions whose order
differs (the oracle accumulates in double): tolerance 1e-4 relative at reduced size, 1e-3 (north_star's fp32 depth tolerance) for the
full BEiT-L at 384 x 512.  The wiring itself is pinned on the CPU (tests/test_oracle_dpt_beit.py: independent modules + HuggingFace)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from cartoonsegmentation_amd.nets import DPTBeitConfig, build_dpt_beit  # noqa: E402
from cartoonsegmentation_amd.weights import SynthWeights  # noqa: E402
from oracle import nets as onets  # noqa: E402


def _outs(n, H, W, cfg, fill):
    gh, gw, F = H // 16, W // 16, cfg.features
    shapes = [(n, 1, H, W), (n, cfg.head_features_2, H, W), (n, F, gh // 2, gw // 2)] + [(n, F, gh << k, gw << k) for k in range(4)]
    return [np.full(s, fill, np.float32) for s in shapes]


def _compare(prog, x, cfg, tol):
    from cartoonsegmentation_amd.runtime import CompiledProgram
    n, _, H, W = x.shape
    ref = _outs(n, H, W, cfg, 0.0)
    onets.run_program(prog, [x] + ref)
    cp = CompiledProgram(prog, 'cuda')
    dev = [torch.from_numpy(a).cuda() for a in _outs(n, H, W, cfg, np.nan)]
    cp.run(torch.from_numpy(x).cuda(), *dev)
    torch.cuda.synchronize()
    for name, r, d in zip(('rel', 'out_conv', 'l4_rn', 'r4', 'r3', 'r2', 'r1'), ref, dev):
        d = d.cpu().numpy()
        assert np.isfinite(d).all(), name
        err = np.abs(d - r).max() / np.abs(r).max()
        assert err < tol, (name, err)
    return cp


SMALL = dict(embed=64, depth=3, heads=2, base_grid=(4, 4), hooks=(0, 1, 2, 2), features=32, neck=(32, 32, 64, 64))


@pytest.mark.parametrize("n,H,W,kw", [
    (2, 64, 96, dict(SMALL, hooks=(0, 1, 2, 1))),                     # ragged token count (25 tokens), batch 2, a block hooked twice
    (1, 160, 128, dict(SMALL, heads=1, readout='ignore')),            # head dimension 64, Slice readout
    (1, 96, 96, dict(SMALL, embed=128, heads=1)),                     # head dimension 128 (no key-range split in the P V phase)
    (1, 576, 576, dict(SMALL, depth=1, hooks=(0, 0, 0, 0))),          # 1297 tokens: the 16-row query tile of the attention kernel
])
def test_dpt_beit_small_hip_vs_oracle(n, H, W, kw):
    cfg = DPTBeitConfig(**kw)
    prog = build_dpt_beit(SynthWeights('dptbeit_small.'), n, H, W, cfg)
    x = np.random.default_rng(n * H + W).normal(0, 1, (n, 3, H, W)).astype(np.float32)
    _compare(prog, x, cfg, 1e-4)


def test_attention_bias_window_and_table_gather_agree():
    """the attention kernel reads the relative position bias of a tile pair from an LDS window of the table (DMA) when the window fits, else
    from the table in global memory: both paths on the same program (a 36 x 36 token grid: several key tiles per half, class-token row /
    column, padded last tiles) give the same tensors, and the gather path by itself meets the oracle tolerance"""
    from cartoonsegmentation_amd import _lib
    from cartoonsegmentation_amd.runtime import CompiledProgram
    L = _lib.load()
    cfg = DPTBeitConfig(**dict(SMALL, depth=2, hooks=(0, 1, 1, 1)))
    n, H, W = 2, 576, 576
    prog = build_dpt_beit(SynthWeights('dptbeit_small.'), n, H, W, cfg)
    x = torch.from_numpy(np.random.default_rng(5).normal(0, 1, (n, 3, H, W)).astype(np.float32)).cuda()
    cp = CompiledProgram(prog, 'cuda')
    res = []
    try:
        for opt in (0, 1):
            L.csm_debug_attention_options(opt)
            dev = [torch.from_numpy(a).cuda() for a in _outs(n, H, W, cfg, np.nan)]
            cp.run(x, *dev)
            torch.cuda.synchronize()
            res.append(dev)
    finally:
        L.csm_debug_attention_options(0)
    for a, b in zip(*res):
        assert torch.isfinite(a).all() and float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())
    try:
        L.csm_debug_attention_options(1)
        xs = np.random.default_rng(6).normal(0, 1, (1, 3, 64, 96)).astype(np.float32)
        _compare(build_dpt_beit(SynthWeights('dptbeit_small.'), 1, 64, 96, DPTBeitConfig(**SMALL)), xs, DPTBeitConfig(**SMALL), 1e-4)
    finally:
        L.csm_debug_attention_options(0)


def test_dpt_beit_large_384x512_hip_vs_oracle():
    """the network of BASELINE configs[2] itself: BEiT-L/16 (24 blocks, 1024 wide, 16 heads) + DPT decoder on a 384 x 512 prepared input
    (24 x 32 + 1 = 769 tokens: the relative-position table is re-sampled from the 24 x 24 pre-training window)"""
    cfg = DPTBeitConfig()
    prog = build_dpt_beit(SynthWeights('zoe.core.core.'), 1, 384, 512, cfg)
    assert 5.0e11 < prog.flops < 1.2e12
    x = np.random.default_rng(11).normal(0, 1, (1, 3, 384, 512)).astype(np.float32)
    _compare(prog, x, cfg, 1e-3)


def test_zoedepth_runs_on_its_builtin_core():
    """ZoeDepth.forward_prepared = built-in DPT-BEiT core -> metric-bins head, against the two programs on the oracle (reduced core width)"""
    from cartoonsegmentation_amd.nets import build_zoe_head
    from cartoonsegmentation_amd.zoedepth import DPTBeitCore, PrefixedWeights, ZoeDepth
    ws = SynthWeights('zoe_small.')
    cfg = DPTBeitConfig(embed=128, depth=4, heads=2, base_grid=(6, 6), hooks=(0, 1, 2, 3), features=256, neck=(64, 64, 128, 128))
    core = DPTBeitCore(PrefixedWeights(ws, 'core.core.'), cfg)
    z = ZoeDepth(ws, core=core, img_size=(96, 128), device='cuda')
    xp = np.random.default_rng(5).normal(0, 1, (1, 3, 96, 128)).astype(np.float32)
    got = z.forward_prepared(torch.from_numpy(xp).cuda()).cpu().numpy()
    ref = _outs(1, 96, 128, cfg, 0.0)
    onets.run_program(build_dpt_beit(PrefixedWeights(ws, 'core.core.'), 1, 96, 128, cfg), [xp] + ref)
    sizes = [tuple(r.shape[2:]) for r in ref[2:]]
    out = np.zeros((1, 1, 96, 128), np.float32)
    onets.run_program(build_zoe_head(ws, 1, 96, 128, sizes), ref + [out])
    assert np.isfinite(got).all() and np.abs(got - out).max() <= 1e-3 * np.abs(out).max()
    # and DepthModel.infer (padding, flip TTA, resize back) runs end to end on it
    img = torch.rand(1, 3, 70, 110, device='cuda')
    d = z.infer(img)
    assert d.shape == (1, 1, 70, 110) and torch.isfinite(d).all() and float(d.min()) > 0


def test_core_program_cache_is_bounded():
    """DPTBeitCore keeps the `max_programs` most recently used (batch, height, width) programs: a third shape evicts the least recently
    used one (its packed weights and workspace are released), and coming back to an evicted shape rebuilds it with identical results"""
    from cartoonsegmentation_amd.zoedepth import DPTBeitCore
    cfg = DPTBeitConfig(**SMALL)
    core = DPTBeitCore(SynthWeights('dptbeit_small.'), cfg, max_programs=2)
    rng = np.random.default_rng(21)
    x1 = torch.from_numpy(rng.normal(0, 1, (1, 3, 64, 96)).astype(np.float32)).cuda()
    first = core(x1)[0].clone()
    core(torch.from_numpy(rng.normal(0, 1, (1, 3, 96, 64)).astype(np.float32)).cuda())
    assert list(core._progs) == [(1, 64, 96), (1, 96, 64)] and core.evictions == 0
    core(x1)                                                             # touch: (1, 96, 64) becomes the least recently used
    core(torch.from_numpy(rng.normal(0, 1, (2, 3, 64, 64)).astype(np.float32)).cuda())
    assert list(core._progs) == [(1, 64, 96), (2, 64, 64)] and core.evictions == 1
    core(torch.from_numpy(rng.normal(0, 1, (1, 3, 96, 64)).astype(np.float32)).cuda())
    assert (1, 64, 96) not in core._progs and core.evictions == 2
    again = core(x1)[0]
    torch.cuda.synchronize()
    assert torch.equal(first, again) and len(core._progs) == 2
