"""MiDaS DPT-BEiT core of ZoeDepth (the reference loads it through torch.hub, base_models/midas.py:341) -- CPU checks of the
build's three statements against each other and against an EXTERNAL implementation:

  1. the lowered layer program (nets/dpt_beit.py) executed by the C oracle interpreter and by the torch-CPU executor,
  2. the independent nn.Module restatement with MiDaS / timm attribute names (oracle/dpt_beit_torch.py), whose state_dict feeds (1),
  3. HuggingFace transformers' DPTForDepthEstimation over a BeitBackbone (installed in this image; unrelated code base), the weights of
     (2) mapped name by name.

(2) == (3) pins the restatement against an implementation this build did not write; (1) == (2) pins the lowering -- qkv packing, folded
scales, the re-sampled relative-position table and its arithmetic index, ProjectReadout, ConvTranspose-as-GEMM, the fusion order.
All at a reduced width / depth with an input whose token grid differs from the pre-training window (the table re-sampling runs).
Against MiDaS' own code the core stays [EXT, unpinned]: it is not under /root/reference."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from cartoonsegmentation_amd.nets import DPTBeitConfig, build_dpt_beit, resample_rel_table  # noqa: E402
from cartoonsegmentation_amd.weights import StateDictWeights  # noqa: E402
from oracle import nets as onets, nets_torch  # noqa: E402
from oracle.dpt_beit_torch import DPTBeit, fill_deterministic, gen_relative_position_index  # noqa: E402

KW = dict(embed=64, depth=4, heads=2, base_grid=(4, 4), hooks=(0, 1, 2, 3), features=32, neck=(32, 32, 64, 64))
NAMES = ('rel', 'out_conv', 'l4_rn', 'r4', 'r3', 'r2', 'r1')


def _module_outputs(m, x):
    with torch.no_grad():
        rel, f = m(torch.from_numpy(x))
    return [rel.unsqueeze(1).numpy()] + [f[k].numpy() for k in NAMES[1:]]


def _program_outputs(prog, x, cfg, runner):
    n, _, H, W = x.shape
    gh, gw = H // 16, W // 16
    F = cfg.features
    outs = [np.zeros((n, 1, H, W), np.float32), np.zeros((n, cfg.head_features_2, H, W), np.float32), np.zeros((n, F, gh // 2, gw // 2), np.float32)]
    outs += [np.zeros((n, F, gh << k, gw << k), np.float32) for k in range(4)]
    runner(prog, [x] + outs)
    return outs


@pytest.mark.parametrize("H,W,readout", [(64, 96, 'project'), (96, 64, 'project'), (64, 64, 'ignore')])
def test_lowered_program_matches_the_independent_modules(H, W, readout):
    kw = dict(KW, readout=readout)
    m = fill_deterministic(DPTBeit(**kw), seed=3).eval()
    x = np.random.default_rng(1).normal(0, 1, (2, 3, H, W)).astype(np.float32)
    ref = _module_outputs(m, x)
    cfg = DPTBeitConfig(**kw)
    prog = build_dpt_beit(StateDictWeights(m.state_dict()), 2, H, W, cfg)
    for runner in (onets.run_program, nets_torch.run_program):
        got = _program_outputs(prog, x, cfg, runner)
        for name, a, b in zip(NAMES, got, ref):
            assert a.shape == b.shape, (name, a.shape, b.shape)
            err = np.abs(a - b).max() / np.abs(b).max()
            assert err < 2e-5, (runner.__module__, name, err)
    # the check has teeth: swapping two hooked blocks (what a wrong hook order would do) moves the outputs by orders of magnitude more
    cfg_bad = DPTBeitConfig(**dict(kw, hooks=(1, 0, 2, 3)))
    bad = _program_outputs(build_dpt_beit(StateDictWeights(m.state_dict()), 2, H, W, cfg_bad), x, cfg_bad, onets.run_program)
    assert np.abs(bad[0] - ref[0]).max() / np.abs(ref[0]).max() > 1e-2


def test_relative_position_table_resampling_and_index():
    """host re-sampling == MiDaS's F.interpolate route on a NON-square pre-training window (its reshape(1, old_width, old_height, -1) axis
    order matters there), identity at the native window; arithmetic index of the kernels == timm's gen_relative_position_index"""
    heads, base = 3, (5, 3)
    T0 = (2 * base[0] - 1) * (2 * base[1] - 1) + 3
    table = np.random.default_rng(2).normal(0, 1, (T0, heads)).astype(np.float32)
    sq = np.random.default_rng(3).normal(0, 1, (7 * 7 + 3, heads)).astype(np.float32)
    assert np.array_equal(resample_rel_table(sq, (4, 4), (4, 4)), sq)          # native square window: bilinear at scale 1 is the identity
    for grid in ((4, 6), (7, 2), (5, 4), (5, 3)):      # (5, 3) = the native NON-square window: MiDaS's axis order still re-samples it
        old_h, old_w, new_h, new_w = 2 * base[0] - 1, 2 * base[1] - 1, 2 * grid[0] - 1, 2 * grid[1] - 1
        t = torch.from_numpy(table)
        sub = t[:T0 - 3].reshape(1, old_w, old_h, -1).permute(0, 3, 1, 2)
        ref = torch.nn.functional.interpolate(sub, size=(new_h, new_w), mode="bilinear").permute(0, 2, 3, 1).reshape(new_h * new_w, -1)
        ref = torch.cat([ref, t[T0 - 3:]]).numpy()
        got = resample_rel_table(table, base, grid)
        assert got.shape == ref.shape and np.abs(got - ref).max() < 4e-6          # float32 lerp order
        gh, gw = grid
        idx = gen_relative_position_index(grid).numpy()
        T = new_h * new_w + 3
        for i in (0, 1, gw, gh * gw):
            for j in (0, 1, gw + 1, gh * gw):
                if i == 0:
                    want = T - 1 if j == 0 else T - 3
                elif j == 0:
                    want = T - 2
                else:
                    yi, xi, yj, xj = (i - 1) // gw, (i - 1) % gw, (j - 1) // gw, (j - 1) % gw
                    want = (yi - yj + gh - 1) * (2 * gw - 1) + (xi - xj + gw - 1)
                assert idx[i, j] == want


def _to_hf(sd, depth, E):
    """state_dict of oracle/dpt_beit_torch.DPTBeit -> the names of transformers' DPTForDepthEstimation(backbone = BeitBackbone)"""
    out = {}
    out['backbone.beit.embeddings.cls_token'] = sd['pretrained.model.cls_token']
    for s in ('weight', 'bias'):
        out['backbone.beit.embeddings.patch_embeddings.projection.' + s] = sd['pretrained.model.patch_embed.proj.' + s]
    for i in range(depth):
        a, b = 'pretrained.model.blocks.%d.' % i, 'backbone.beit.layers.%d.' % i
        out[b + 'lambda_1'], out[b + 'lambda_2'] = sd[a + 'gamma_1'], sd[a + 'gamma_2']
        qkv = sd[a + 'attn.qkv.weight']
        out[b + 'attention.q_proj.weight'], out[b + 'attention.k_proj.weight'], out[b + 'attention.v_proj.weight'] = qkv[:E], qkv[E:2 * E], qkv[2 * E:]
        out[b + 'attention.q_proj.bias'], out[b + 'attention.v_proj.bias'] = sd[a + 'attn.q_bias'], sd[a + 'attn.v_bias']
        out[b + 'relative_position_bias.relative_position_bias_table'] = sd[a + 'attn.relative_position_bias_table']
        for s in ('weight', 'bias'):
            out[b + 'attention.o_proj.' + s] = sd[a + 'attn.proj.' + s]
            out[b + 'layernorm_before.' + s], out[b + 'layernorm_after.' + s] = sd[a + 'norm1.' + s], sd[a + 'norm2.' + s]
            out[b + 'mlp.fc1.' + s], out[b + 'mlp.fc2.' + s] = sd[a + 'mlp.fc1.' + s], sd[a + 'mlp.fc2.' + s]
    for k in range(4):
        a = 'pretrained.act_postprocess%d.' % (k + 1)
        for s in ('weight', 'bias'):
            out['neck.reassemble_stage.readout_projects.%d.0.%s' % (k, s)] = sd[a + '0.project.0.' + s]
            out['neck.reassemble_stage.layers.%d.projection.%s' % (k, s)] = sd[a + '3.' + s]
            if k != 2:
                out['neck.reassemble_stage.layers.%d.resize.%s' % (k, s)] = sd[a + '4.' + s]
        out['neck.convs.%d.weight' % k] = sd['scratch.layer%d_rn.weight' % (k + 1)]
        a, b = 'scratch.refinenet%d.' % (k + 1), 'neck.fusion_stage.layers.%d.' % (3 - k)      # HF fuses from the last stage: layers.0 = refinenet4
        for s in ('weight', 'bias'):
            out[b + 'projection.' + s] = sd[a + 'out_conv.' + s]
            for u in (1, 2):
                for c in (1, 2):
                    out[b + 'residual_layer%d.convolution%d.%s' % (u, c, s)] = sd[a + 'resConfUnit%d.conv%d.%s' % (u, c, s)]
    for j in (0, 2, 4):
        for s in ('weight', 'bias'):
            out['head.head.%d.%s' % (j, s)] = sd['scratch.output_conv.%d.%s' % (j, s)]
    return out


def test_independent_modules_match_huggingface_dpt_beit():
    tf = pytest.importorskip("transformers")
    try:
        from transformers import BeitConfig, DPTConfig, DPTForDepthEstimation
    except Exception as e:                                   # pragma: no cover
        pytest.skip("transformers without DPT / BEiT: %r" % (e,))
    E, depth, heads = KW['embed'], KW['depth'], KW['heads']
    bc = BeitConfig(image_size=16 * KW['base_grid'][0], patch_size=16, hidden_size=E, num_hidden_layers=depth, num_attention_heads=heads,
                    intermediate_size=4 * E, use_relative_position_bias=True, use_shared_relative_position_bias=False,
                    use_absolute_position_embeddings=False, use_mask_token=False, layer_scale_init_value=0.1, layer_norm_eps=1e-6,
                    hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, drop_path_rate=0.0,
                    out_features=["stage%d" % (h + 1) for h in KW['hooks']], reshape_hidden_states=False, add_fpn=False)
    cfg = DPTConfig(backbone_config=bc, neck_hidden_sizes=list(KW['neck']), fusion_hidden_size=KW['features'], readout_type="project",
                    reassemble_factors=[4, 2, 1, 0.5], is_hybrid=False, use_batch_norm_in_fusion_residual=False, add_projection=False,
                    head_in_index=-1, hidden_act="gelu")
    hf = DPTForDepthEstimation(cfg).eval()
    m = fill_deterministic(DPTBeit(**KW), seed=5).eval()
    mapped = _to_hf(m.state_dict(), depth, E)
    missing, unexpected = hf.load_state_dict(mapped, strict=False)
    # refinenet4 has no second input: MiDaS (and HF) still carry its resConfUnit1 weights; nothing else may be missing
    assert not unexpected and all('relative_position_index' in k for k in missing), (missing, unexpected)
    for H, W in ((64, 96), (96, 64), (64, 64)):
        x = torch.from_numpy(np.random.default_rng(7).normal(0, 1, (2, 3, H, W)).astype(np.float32))
        with torch.no_grad():
            rel, _ = m(x)
            ref = hf(pixel_values=x).predicted_depth
        assert rel.shape == ref.shape
        err = float((rel - ref).abs().max() / ref.abs().max())
        assert err < 2e-5, (H, W, err)
